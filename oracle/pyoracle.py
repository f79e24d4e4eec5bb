"""ctypes bindings to the CPU oracle (and, when built, to the reference's own C code).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "libqmann_oracle.so"
REF_SO = HERE / "_ref" / "libqmann_ref.so"

QO_MAX_HOP = 8
SM_CUDA, SM_CPU_POW2, SM_CPU_EXP_PLAN = 0, 1, 2

_f32p = C.POINTER(C.c_float)


def _fp(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


class QoModel(C.Structure):
    _fields_ = [
        ("n_hop", C.c_uint), ("dim_emb", C.c_uint), ("dim_input", C.c_uint),
        ("attention_mode", C.c_uint), ("num_bit", C.c_uint), ("softmax_variant", C.c_int),
        ("f_fixed", C.c_bool), ("en_lin_map", C.c_bool),
        ("iwl", C.c_uint * QO_MAX_HOP), ("frac", C.c_uint * QO_MAX_HOP),
        ("iwl_w", C.c_uint * QO_MAX_HOP), ("frac_w", C.c_uint * QO_MAX_HOP),
        ("iwl_att", C.c_uint * QO_MAX_HOP), ("frac_att", C.c_uint * QO_MAX_HOP),
        ("iwl_bin", C.c_uint), ("frac_bin", C.c_uint),
        ("w_q", _f32p),
        ("w_a", _f32p * QO_MAX_HOP), ("w_c", _f32p * QO_MAX_HOP), ("w_h", _f32p * QO_MAX_HOP),
        ("w_ans", _f32p),
        ("f_shift_based", C.c_bool), ("en_sc_att", C.c_bool), ("sc_att", C.c_float * QO_MAX_HOP),
        ("en_non_lin", C.c_bool),
    ]


class QoTaps(C.Structure):
    _fields_ = [(n, _f32p) for n in
                ("u0", "keys", "vals", "scores", "probs", "o", "lu", "u", "logits", "out_probs")]


def load_lazy(path) -> C.CDLL:
    """Open one of oracle/_ref's shared objects with RTLD_LAZY.  They are the reference's objects + glue and nothing
    else: layer.o's cuda_* imports stay undefined (no stand-ins are linked), which is fine as long as no such function is
    ever CALLED (en_gpu_model = false everywhere here).  ctypes.CDLL always adds RTLD_NOW, so the handle comes from
    dlopen itself.  RTLD_LOCAL: nothing else in the process binds to these symbols, and they cannot bind to a product
    library the process may have loaded privately -- a stray cuda_* call ends the process with a symbol lookup error."""
    import os
    libc = C.CDLL(None)
    libc.dlopen.restype = C.c_void_p
    libc.dlopen.argtypes = [C.c_char_p, C.c_int]
    libc.dlerror.restype = C.c_char_p
    h = libc.dlopen(str(path).encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
    if not h:
        raise OSError(f"dlopen({path}, RTLD_LAZY): {(libc.dlerror() or b'?').decode()}")
    return C.CDLL(str(path), handle=h)


class Oracle:
    def __init__(self, path: Path = ORACLE_SO):
        if not path.exists():
            raise FileNotFoundError(f"{path} missing: run `make -C oracle oracle`")
        L = self.L = C.CDLL(str(path))
        u, f, i32 = C.c_uint, C.c_float, C.c_int32
        L.qo_float2fixed.restype = i32; L.qo_float2fixed.argtypes = [f, u, u]
        L.qo_fixed2float.restype = f; L.qo_fixed2float.argtypes = [i32, u]
        L.qo_quant.restype = f; L.qo_quant.argtypes = [f, u, u]
        L.qo_fixed_mul.restype = f; L.qo_fixed_mul.argtypes = [f, f, u, u, u, u]
        L.qo_fixed_add.restype = f; L.qo_fixed_add.argtypes = [f, f, u, u, u, u]
        L.qo_code8.restype = C.c_int; L.qo_code8.argtypes = [f, u, u]
        L.qo_hamming_similarity.restype = u; L.qo_hamming_similarity.argtypes = [i32, i32, u]
        L.qo_hamming_similarity_w.restype = f; L.qo_hamming_similarity_w.argtypes = [i32, i32, u]
        L.qo_cuda_hamming_similarity.restype = f
        L.qo_cuda_hamming_similarity.argtypes = [i32, i32, u, C.c_bool]
        L.qo_dense_fwd.restype = None
        L.qo_dense_fwd.argtypes = [_f32p, _f32p, _f32p, u, u, C.c_char_p, C.c_bool, u, u, u, u]
        L.qo_dense_mat_fwd.restype = None
        L.qo_dense_mat_fwd.argtypes = [_f32p, _f32p, _f32p, u, u, u, C.c_bool, u, u]
        L.qo_dot_mat_vec_fwd.restype = None
        L.qo_dot_mat_vec_fwd.argtypes = [_f32p, _f32p, _f32p, u, u, C.c_bool, C.c_bool, u, u, u, u]
        L.qo_dot_mat_vec_fwd_appx.restype = None
        L.qo_dot_mat_vec_fwd_appx.argtypes = [_f32p, _f32p, _f32p, u, u, C.c_bool, u, u, u, C.c_bool]
        L.qo_attention_hamming.restype = None
        L.qo_attention_hamming.argtypes = [_f32p, _f32p, _f32p, u, u, u, u, u, C.c_int]
        L.qo_softmax_fwd.restype = None
        L.qo_softmax_fwd.argtypes = [_f32p, _f32p, u, C.c_int, C.c_bool]
        L.qo_sum_vec_fwd.restype = None
        L.qo_sum_vec_fwd.argtypes = [_f32p, _f32p, _f32p, u, C.c_bool, u, u]
        L.qo_activation_fwd.restype = None
        L.qo_activation_fwd.argtypes = [_f32p, _f32p, u, C.c_char_p, C.c_bool, u, u]
        L.qo_argmax_hi.restype = u; L.qo_argmax_hi.argtypes = [_f32p, u]
        L.qo_cross_entropy_run.restype = u
        L.qo_cross_entropy_run.argtypes = [_f32p, _f32p, u, _f32p, C.POINTER(u), _f32p]
        b = C.c_bool
        L.qo_dot_mat_vec_bwd.restype = None
        L.qo_dot_mat_vec_bwd.argtypes = [_f32p] * 5 + [u, u, b, b, u, u]
        L.qo_dot_mat_vec_bwd_appx.restype = None
        L.qo_dot_mat_vec_bwd_appx.argtypes = [_f32p] * 5 + [u, u, b, u, u, u, b]
        L.qo_softmax_bwd.restype = None
        L.qo_softmax_bwd.argtypes = [_f32p] * 3 + [u, b]
        L.qo_dense_bwd.restype = None
        L.qo_dense_bwd.argtypes = [_f32p] * 6 + [u, u, C.c_char_p, b, u, u]
        L.qo_dense_mat_bwd.restype = None
        L.qo_dense_mat_bwd.argtypes = [_f32p] * 5 + [u, u, u, b, u, u]
        L.qo_mat_w_up.restype = f
        L.qo_mat_w_up.argtypes = [_f32p, _f32p, u, u, u, f, f, f, b, u, u]
        L.qo_dup_grad_bwd.restype = None
        L.qo_dup_grad_bwd.argtypes = [_f32p] * 3 + [u, b, u, u]
        L.qo_memn2n_forward.restype = u
        L.qo_memn2n_forward.argtypes = [C.POINTER(QoModel), _f32p, u, _f32p, C.POINTER(QoTaps)]
        L.qo_memn2n_forward_mem.restype = u
        L.qo_memn2n_forward_mem.argtypes = [C.POINTER(QoModel), _f32p, _f32p, u, _f32p, C.POINTER(QoTaps)]
        vp = C.c_void_p
        L.qo_memn2n_forward_words_batch.restype = None
        L.qo_memn2n_forward_words_batch.argtypes = [C.POINTER(QoModel), vp, u, vp, u, vp, u, u, vp, vp, vp, vp]

    # ---- scalar helpers (vectorised over numpy inputs for convenience) ----
    def quant(self, x, iwl, frac):
        x = np.asarray(x, dtype=np.float32)
        return np.array([self.L.qo_quant(float(v), iwl, frac) for v in x.ravel()],
                        dtype=np.float32).reshape(x.shape)

    def float2fixed(self, x, iwl, frac):
        x = np.asarray(x, dtype=np.float32)
        return np.array([self.L.qo_float2fixed(float(v), iwl, frac) for v in x.ravel()],
                        dtype=np.int32).reshape(x.shape)

    def code8(self, x, iwl, frac):
        x = np.asarray(x, dtype=np.float32)
        return np.array([self.L.qo_code8(float(v), iwl, frac) for v in x.ravel()],
                        dtype=np.int32).reshape(x.shape)

    # ---- operators ----
    def dense_fwd(self, w, x, f_fixed, fmt_in, fmt_w, act=b"NULL"):
        w = np.ascontiguousarray(w, np.float32); x = np.ascontiguousarray(x, np.float32)
        out = np.empty(w.shape[0], np.float32)
        self.L.qo_dense_fwd(_fp(w), _fp(x), _fp(out), w.shape[1], w.shape[0], act, f_fixed,
                            fmt_in[0], fmt_in[1], fmt_w[0], fmt_w[1])
        return out

    def dense_mat_fwd(self, w, in_mat, f_fixed, fmt):
        w = np.ascontiguousarray(w, np.float32); in_mat = np.ascontiguousarray(in_mat, np.float32)
        out = np.empty((in_mat.shape[0], w.shape[0]), np.float32)
        self.L.qo_dense_mat_fwd(_fp(w), _fp(in_mat), _fp(out), w.shape[1], w.shape[0], in_mat.shape[0],
                                f_fixed, fmt[0], fmt[1])
        return out

    def dot_mat_vec_fwd(self, mat, vec, f_trans, f_fixed, fmt_m, fmt_v):
        mat = np.ascontiguousarray(mat, np.float32); vec = np.ascontiguousarray(vec, np.float32)
        r, c = mat.shape
        out = np.empty(c if f_trans else r, np.float32)
        self.L.qo_dot_mat_vec_fwd(_fp(mat), _fp(vec), _fp(out), r, c, f_trans, f_fixed,
                                  fmt_m[0], fmt_m[1], fmt_v[0], fmt_v[1])
        return out

    def dot_mat_vec_fwd_appx(self, mat, vec, f_trans, f_fixed, iwl, frac, num_bit):
        mat = np.ascontiguousarray(mat, np.float32); vec = np.ascontiguousarray(vec, np.float32)
        r, c = mat.shape
        out = np.empty(c if f_trans else r, np.float32)
        self.L.qo_dot_mat_vec_fwd_appx(_fp(mat), _fp(vec), _fp(out), r, c, f_fixed, iwl, frac, num_bit, f_trans)
        return out

    def attention_hamming(self, mat, vec, iwl, frac_code, num_bit, variant):
        mat = np.ascontiguousarray(mat, np.float32); vec = np.ascontiguousarray(vec, np.float32)
        out = np.empty(mat.shape[0], np.float32)
        self.L.qo_attention_hamming(_fp(mat), _fp(vec), _fp(out), mat.shape[0], mat.shape[1], iwl, frac_code,
                                    num_bit, variant)
        return out

    def softmax_fwd(self, x, variant=SM_CUDA, shift_based=False):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.qo_softmax_fwd(_fp(x), _fp(out), x.size, variant, shift_based)
        return out

    def sum_vec_fwd(self, a, b, f_fixed, fmt):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        out = np.empty_like(a)
        self.L.qo_sum_vec_fwd(_fp(a), _fp(b), _fp(out), a.size, f_fixed, fmt[0], fmt[1])
        return out

    def activation_fwd(self, x, act, f_fixed=False, fmt=(0, 0)):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.qo_activation_fwd(_fp(x), _fp(out), x.size, act, f_fixed, fmt[0], fmt[1])
        return out

    def argmax_hi(self, x):
        x = np.ascontiguousarray(x, np.float32)
        return int(self.L.qo_argmax_hi(_fp(x), x.size))

    def cross_entropy_run(self, h, y):
        h = np.ascontiguousarray(h, np.float32); y = np.ascontiguousarray(y, np.float32)
        cost = np.zeros(1, np.float32); cnt = C.c_uint(0); grad = np.empty_like(h)
        pred = self.L.qo_cross_entropy_run(_fp(h), _fp(y), h.size, _fp(cost), C.byref(cnt), _fp(grad))
        return int(pred), float(cost[0]), int(cnt.value), grad

    # ---- training verbs ----
    def dot_mat_vec_bwd(self, mat, vec, grad_in, f_trans, f_fixed, fmt_m, appx_bits=None):
        mat = np.ascontiguousarray(mat, np.float32); vec = np.ascontiguousarray(vec, np.float32)
        grad_in = np.ascontiguousarray(grad_in, np.float32)
        r, c = mat.shape
        gm = np.zeros((r, c), np.float32); gv = np.zeros(r if f_trans else c, np.float32)
        if appx_bits is None:
            self.L.qo_dot_mat_vec_bwd(_fp(mat), _fp(vec), _fp(grad_in), _fp(gm), _fp(gv), r, c, f_trans, f_fixed,
                                      fmt_m[0], fmt_m[1])
        else:
            self.L.qo_dot_mat_vec_bwd_appx(_fp(mat), _fp(vec), _fp(grad_in), _fp(gm), _fp(gv), r, c, f_fixed,
                                           fmt_m[0], fmt_m[1], appx_bits, f_trans)
        return gm, gv

    def softmax_bwd(self, out_vec, grad_in, shift_based=False):
        out_vec = np.ascontiguousarray(out_vec, np.float32); grad_in = np.ascontiguousarray(grad_in, np.float32)
        g = np.empty_like(out_vec)
        self.L.qo_softmax_bwd(_fp(out_vec), _fp(grad_in), _fp(g), out_vec.size, shift_based)
        return g

    def dense_bwd(self, w, w_del, x, out, grad_in, f_fixed, fmt_w, act=b"NULL"):
        w = np.ascontiguousarray(w, np.float32); w_del = np.array(w_del, np.float32, copy=True)
        x = np.ascontiguousarray(x, np.float32); out = np.ascontiguousarray(out, np.float32)
        gi = np.array(grad_in, np.float32, copy=True); go = np.zeros(w.shape[1], np.float32)
        self.L.qo_dense_bwd(_fp(w), _fp(w_del), _fp(x), _fp(out), _fp(gi), _fp(go), w.shape[1], w.shape[0], act,
                            f_fixed, fmt_w[0], fmt_w[1])
        return w_del, go, gi

    def dense_mat_bwd(self, in_mat, w, w_del, grad_in, f_fixed, fmt):
        in_mat = np.ascontiguousarray(in_mat, np.float32); w = np.ascontiguousarray(w, np.float32)
        w_del = np.array(w_del, np.float32, copy=True); grad_in = np.ascontiguousarray(grad_in, np.float32)
        go = np.zeros_like(in_mat)
        self.L.qo_dense_mat_bwd(_fp(in_mat), _fp(w), _fp(w_del), _fp(grad_in), _fp(go), w.shape[1], w.shape[0],
                                in_mat.shape[0], f_fixed, fmt[0], fmt[1])
        return w_del, go

    def mat_w_up(self, w, w_del, batch, lr, lam, max_norm, f_fixed, fmt):
        w = np.array(w, np.float32, copy=True); w_del = np.array(w_del, np.float32, copy=True)
        norm = self.L.qo_mat_w_up(_fp(w), _fp(w_del), w.shape[1], w.shape[0], batch, lr, lam, max_norm, f_fixed,
                                  fmt[0], fmt[1])
        return w, w_del, float(norm)

    def dup_grad_bwd(self, a, b, f_fixed, fmt):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); out = np.empty_like(a)
        self.L.qo_dup_grad_bwd(_fp(a), _fp(b), _fp(out), a.size, f_fixed, fmt[0], fmt[1])
        return out

    # ---- composite ----
    def make_model(self, cfg: dict, weights: dict):
        """cfg: n_hop, dim_emb, dim_input, attention_mode, softmax_variant, f_fixed, en_lin_map,
        fmt (list of (iwl,frac) per hop), fmt_w, fmt_att, fmt_bin.  weights: w_q, w_a[h], w_c[h], w_h[h], w_ans
        (float32 arrays; kept alive on the returned object)."""
        m = QoModel()
        m.n_hop, m.dim_emb, m.dim_input = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
        m.attention_mode = cfg["attention_mode"]
        m.num_bit = cfg.get("num_bit", 8)
        m.softmax_variant = cfg.get("softmax_variant", SM_CUDA)
        m.f_fixed = cfg.get("f_fixed", True)
        m.en_lin_map = cfg.get("en_lin_map", True)
        for h in range(cfg["n_hop"]):
            m.iwl[h], m.frac[h] = cfg["fmt"][h]
            m.iwl_w[h], m.frac_w[h] = cfg["fmt_w"][h]
            m.iwl_att[h], m.frac_att[h] = cfg["fmt_att"][h]
        m.iwl_bin, m.frac_bin = cfg["fmt_bin"]
        m.f_shift_based = bool(cfg.get("softmax_shift_based", False))
        m.en_non_lin = bool(cfg.get("en_non_lin", False))
        if cfg.get("att_scale") is not None:
            m.en_sc_att = True
            for h in range(cfg["n_hop"]):
                m.sc_att[h] = float(np.float32(cfg["att_scale"][h]))
        keep = []

        def hold(a):
            a = np.ascontiguousarray(a, np.float32)
            keep.append(a)
            return _fp(a)
        if weights.get("w_q") is not None:
            m.w_q = hold(weights["w_q"])
        for h in range(cfg["n_hop"]):
            if weights.get("w_a") is not None:
                m.w_a[h] = hold(weights["w_a"][h]); m.w_c[h] = hold(weights["w_c"][h])
            if m.en_lin_map:
                m.w_h[h] = hold(weights["w_h"][h])
        m.w_ans = hold(weights["w_ans"])
        m._keep = keep
        return m

    def _taps(self, m, n_sen, want):
        H, D, V = m.n_hop, m.dim_emb, m.dim_input
        shapes = {"u0": (D,), "keys": (H, n_sen, D), "vals": (H, n_sen, D), "scores": (H, n_sen),
                  "probs": (H, n_sen), "o": (H, D), "lu": (H, D), "u": (H, D), "logits": (V,), "out_probs": (V,)}
        t = QoTaps(); arrs = {}
        for k in want:
            arrs[k] = np.zeros(shapes[k], np.float32)
            setattr(t, k, _fp(arrs[k]))
        return t, arrs

    def forward(self, m, story, question, taps=("u0", "scores", "probs", "o", "lu", "u", "logits", "out_probs")):
        story = np.ascontiguousarray(story, np.float32); question = np.ascontiguousarray(question, np.float32)
        n_sen = story.shape[0]
        t, arrs = self._taps(m, n_sen, taps)
        pred = self.L.qo_memn2n_forward(C.byref(m), _fp(story), n_sen, _fp(question), C.byref(t))
        return int(pred), arrs

    def forward_words_batch(self, m, story_words, question_words, row_off, n_threads=None):
        """every story of a set through qo_memn2n_forward on C threads.  story_words uint16 [rows][W] (the last valid entry of a
        row is its time index), question_words uint16 [n][Wq], row_off [n + 1].  Returns pred [n] uint32, u_final [n][D],
        top2_gap [n] (two largest output probabilities), near_step [n] bool (a softmax weight on a truncation step)."""
        import os
        sw = np.ascontiguousarray(story_words, np.uint16); qw = np.ascontiguousarray(question_words, np.uint16)
        ro = np.ascontiguousarray(row_off, np.uint32)
        n = qw.shape[0]
        if sw.shape[0] == 0:
            sw = np.full((1, max(sw.shape[1], 1)), 0xFFFF, np.uint16)
        pred = np.zeros(n, np.uint32); u = np.zeros((n, m.dim_emb), np.float32)
        gap = np.zeros(n, np.float32); near = np.zeros(n, np.uint8)
        nt = n_threads or min(64, max(1, len(os.sched_getaffinity(0))))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self.L.qo_memn2n_forward_words_batch(C.byref(m), ptr(sw), sw.shape[1], ptr(qw), qw.shape[1], ptr(ro), n, nt, ptr(pred), ptr(u),
                                             ptr(gap), ptr(near))
        return pred, u, gap, near.astype(bool)

    def forward_mem(self, m, keys, vals, u0, taps=("scores", "probs", "o", "lu", "u", "logits", "out_probs")):
        keys = np.ascontiguousarray(keys, np.float32); vals = np.ascontiguousarray(vals, np.float32)
        u0 = np.ascontiguousarray(u0, np.float32)
        n_sen = keys.shape[1]
        t, arrs = self._taps(m, n_sen, taps)
        pred = self.L.qo_memn2n_forward_mem(C.byref(m), _fp(keys), _fp(vals), n_sen, _fp(u0), C.byref(t))
        return int(pred), arrs


class Reference:
    """The reference's own live C code (oracle/_ref); available wherever the prebuilt .so is."""

    def __init__(self, path: Path = REF_SO):
        if not path.exists():
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        L = self.L = load_lazy(path)
        u, f = C.c_uint, C.c_float
        L.ref_float_quant.restype = f; L.ref_float_quant.argtypes = [f, u, u]
        L.ref_float2fixed.restype = C.c_int; L.ref_float2fixed.argtypes = [f, u, u]
        L.ref_fixed2float.restype = f; L.ref_fixed2float.argtypes = [C.c_int, u, u]
        L.ref_fixed_mul.restype = f; L.ref_fixed_mul.argtypes = [f, f, u, u]
        L.ref_fixed_add.restype = f; L.ref_fixed_add.argtypes = [f, f, u, u]
        L.hamming_similarity.restype = u; L.hamming_similarity.argtypes = [C.c_int, C.c_int, u]
        L.hamming_similarity_w.restype = f; L.hamming_similarity_w.argtypes = [C.c_int, C.c_int, u, C.c_bool]
        L.exp_plan.restype = f; L.exp_plan.argtypes = [f]
        L.ref_softmax_fwd.restype = None; L.ref_softmax_fwd.argtypes = [_f32p, _f32p, u, C.c_int, C.c_int]
        L.ref_sum_vec_fwd.restype = None; L.ref_sum_vec_fwd.argtypes = [_f32p, _f32p, _f32p, u, C.c_int, u, u]
        L.ref_dense_mat_fwd.restype = None
        L.ref_dense_mat_fwd.argtypes = [_f32p, _f32p, _f32p, u, u, u, C.c_int, u, u]
        L.ref_cross_entropy_run.restype = f; L.ref_cross_entropy_run.argtypes = [_f32p, _f32p, _f32p, u]
        L.ref_activation_fwd.restype = None; L.ref_activation_fwd.argtypes = [_f32p, _f32p, u, C.c_char_p]
        L.ref_babi_load.restype = C.c_int
        L.ref_babi_load.argtypes = [C.c_char_p, C.c_char_p, u, u, u, C.POINTER(u), C.POINTER(u), C.POINTER(u)]
        if hasattr(L, "ref_babi_load_pe"):
            L.ref_babi_load_pe.restype = C.c_int
            L.ref_babi_load_pe.argtypes = [C.c_char_p, C.c_char_p, u, u, u, C.POINTER(u), C.POINTER(u), C.POINTER(u), C.POINTER(u)]
        L.ref_babi_nsen.restype = u; L.ref_babi_nsen.argtypes = [u]
        L.ref_babi_get.restype = None; L.ref_babi_get.argtypes = [u, _f32p, _f32p, _f32p]

    def softmax_fwd(self, x, exp_plan=False, shift_based=False):
        x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
        self.L.ref_softmax_fwd(_fp(x), _fp(out), x.size, int(exp_plan), int(shift_based))
        return out

    def sum_vec_fwd(self, a, b, f_fixed, fmt):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); out = np.empty_like(a)
        self.L.ref_sum_vec_fwd(_fp(a), _fp(b), _fp(out), a.size, int(f_fixed), fmt[0], fmt[1])
        return out

    def dense_mat_fwd(self, w, in_mat, f_fixed, fmt):
        w = np.ascontiguousarray(w, np.float32); in_mat = np.ascontiguousarray(in_mat, np.float32)
        out = np.empty((in_mat.shape[0], w.shape[0]), np.float32)
        self.L.ref_dense_mat_fwd(_fp(w), _fp(in_mat), _fp(out), w.shape[1], w.shape[0], in_mat.shape[0],
                                 int(f_fixed), fmt[0], fmt[1])
        return out

    def cross_entropy_run(self, h, y):
        h = np.ascontiguousarray(h, np.float32); y = np.ascontiguousarray(y, np.float32); g = np.empty_like(h)
        cost = self.L.ref_cross_entropy_run(_fp(h), _fp(y), _fp(g), h.size)
        return float(cost), g

    def activation_fwd(self, x, act):
        x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
        self.L.ref_activation_fwd(_fp(x), _fp(out), x.size, act)
        return out


class RefForward:
    """oracle/ref_forward.c: the test-phase forward composed from the reference's LIVE C functions
    (dense_mat_fwd, softmax_fwd, sum_vec_fwd, hamming_similarity*) plus the port for the ops whose CPU
    bodies are dead -- bench.py's "reference+port" CPU baseline.  flags: "O0" (the reference's own
    `gcc -w`) or "O2".  The model struct is Oracle.make_model's (softmax_variant must be a CPU form)."""

    def __init__(self, flags: str = "O2"):
        path = HERE / "_ref" / f"libqmann_refcpu_{flags}.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} missing: run `make -C oracle refcpu` where /root/reference exists")
        L = self.L = load_lazy(path)
        u = C.c_uint
        pp = C.POINTER(_f32p)
        L.rf_create.restype = C.c_void_p; L.rf_create.argtypes = [C.POINTER(QoModel), u]
        L.rf_destroy.restype = None; L.rf_destroy.argtypes = [C.c_void_p]
        L.rf_forward.restype = u; L.rf_forward.argtypes = [C.c_void_p, _f32p, u, _f32p, _f32p]
        L.rf_forward_mem.restype = u; L.rf_forward_mem.argtypes = [C.c_void_p, _f32p, _f32p, u, _f32p, _f32p]
        L.rf_time.restype = C.c_double
        L.rf_time.argtypes = [C.POINTER(QoModel), u, pp, pp, pp, C.POINTER(u), u, u, C.c_double,
                              C.POINTER(C.c_ulong), C.POINTER(C.c_double), C.POINTER(u)]
        L.rf_build_flags.restype = C.c_char_p
        self.flags = L.rf_build_flags().decode()

    def forward(self, m, story, question, max_sen=None):
        story = np.ascontiguousarray(story, np.float32); question = np.ascontiguousarray(question, np.float32)
        ctx = self.L.rf_create(C.byref(m), max_sen or max(1, story.shape[0]))
        u = np.empty(m.dim_emb, np.float32)
        pred = self.L.rf_forward(ctx, _fp(story), story.shape[0], _fp(question), _fp(u))
        self.L.rf_destroy(ctx)
        return int(pred), u

    def forward_mem(self, m, keys, vals, u0):
        keys = np.ascontiguousarray(keys, np.float32); vals = np.ascontiguousarray(vals, np.float32)
        u0 = np.ascontiguousarray(u0, np.float32)
        ctx = self.L.rf_create(C.byref(m), max(1, keys.shape[1]))
        u = np.empty(m.dim_emb, np.float32)
        pred = self.L.rf_forward_mem(ctx, _fp(keys), _fp(vals), keys.shape[1], _fp(u0), _fp(u))
        self.L.rf_destroy(ctx)
        return int(pred), u

    def time(self, m, pool, n_threads: int, seconds: float):
        """pool: list of (story [n_sen][V], question [V]) or of (keys [H][S][D], vals [H][S][D], u0 [D]).
        Returns dict(qps, n, secs, preds) -- a C pthread loop over the pool for about `seconds`."""
        mem = len(pool[0]) == 3
        keep = [[np.ascontiguousarray(x, np.float32) for x in item] for item in pool]
        n = len(keep)
        arr = lambda k: (_f32p * n)(*[_fp(item[k]) for item in keep])
        a, b = arr(0), arr(1)
        u0 = arr(2) if mem else None
        ns = (C.c_uint * n)(*[(item[0].shape[1] if mem else item[0].shape[0]) for item in keep])
        preds = (C.c_uint * n)(*([0xFFFFFFFF] * n))
        done, wall = C.c_ulong(0), C.c_double(0.0)
        qps = self.L.rf_time(C.byref(m), max(ns), a, b, u0, ns, n, n_threads, seconds, C.byref(done), C.byref(wall), preds)
        return dict(qps=float(qps), n=int(done.value), secs=float(wall.value), preds=list(preds))
