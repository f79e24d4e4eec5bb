"""Host-side mirror of the reference's test-phase forward, driving the C-ABI library.

The reference wires `dense`, `dense_mat`, `dot_mat_vec`, `softmax`, `sum_vec`, `dense`,
`softmax`, `cross_entropy` layer structs per query (MemN2N/MemN2N.c:2406-2697).  `QNet`
holds the same configuration (formats per hop, attention mode, weight tables) and runs the
same stages for a whole batch through the batched entry points of include/qmann_batch.h.
torch is used for device buffers only; every computation is a call into libqmann_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import abi


def pad16(d: int) -> int:
    for p in (64, 128, 256):
        if d <= p:
            return p
    raise ValueError(f"dim_emb {d} > 256 is not supported by the batched kernels")


def to_signmag(codes):
    """two's-complement int8 codes (numpy or torch, |code| <= 127) -> sign-magnitude bytes"""
    if isinstance(codes, torch.Tensor):
        return torch.where(codes < 0, (-codes) | -128, codes).to(torch.int8)
    c = codes.astype(np.int16)
    return np.where(c < 0, (-c) | 0x80, c).astype(np.uint8).view(np.int8)


def from_signmag(b):
    """sign-magnitude bytes (numpy int8/uint8) -> integer codes (int16)"""
    u = b.view(np.uint8).astype(np.int16)
    return np.where(u & 0x80, -(u & 0x7F), u & 0x7F).astype(np.int16)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _DevArray:
    """a raw device pointer as something torch.as_tensor can alias (no copy; the owner keeps the memory alive)"""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2}


def babi_cfg(dim_input, attention_mode=2, softmax_base=0, iwl=5, n_hop=3, D=60, en_mq=True):
    """run.sh default iwl=5 (MemN2N/run.sh:6,18); EN_MQ shifts the weight formats of hops 0 and 2
    (MemN2N/MemN2N.c:748-754)."""
    frac = 7 - iwl
    fmt = [(iwl, frac)] * n_hop
    fmt_w = list(fmt)
    if en_mq and n_hop >= 3:
        fmt_w[0] = (iwl + 1, frac - 1)
        fmt_w[2] = (iwl - 1, frac + 1)
    return dict(n_hop=n_hop, dim_emb=D, dim_input=int(dim_input), attention_mode=attention_mode,
                softmax_variant=softmax_base, f_fixed=True, en_lin_map=True, fmt=fmt, fmt_w=fmt_w,
                fmt_att=list(fmt), fmt_bin=(iwl, frac))


def _weights_struct(w: dict, H: int, D: int, V: int, en_lin_map: bool):
    """abi.Weights over contiguous float32 numpy arrays (kept alive by the caller's dict)."""
    ws = abi.Weights()
    ws.n_hop, ws.dim_emb, ws.dim_input = H, D, V
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    ws.w_q, ws.w_ans = ptr(w["w_q"]), ptr(w["w_ans"])
    for h in range(H):
        ws.w_a[h], ws.w_c[h] = ptr(w["w_a"][h]), ptr(w["w_c"][h])
        if en_lin_map:
            ws.w_h[h] = ptr(w["w_h"][h])
    return ws


def save_weights(directory, weights: dict, cfg: dict, fixed: bool = True):
    """Write the reference-layout weight files (include/qmann_weights.h) for `weights` (the dict QNet takes)."""
    H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
    lin = bool(cfg.get("en_lin_map", True))
    w = {k: ([np.ascontiguousarray(m, np.float32) for m in v] if isinstance(v, (list, tuple))
             else np.ascontiguousarray(v, np.float32)) for k, v in weights.items()}
    fm = (abi.Fmt * H)(*[abi.Fmt(*f) for f in cfg["fmt_w"]]) if fixed else None
    abi.check(abi.lib.qmann_weights_save(str(directory).encode(), C.byref(_weights_struct(w, H, D, V, lin)), fm),
              "qmann_weights_save")


def load_weights(directory, cfg: dict, from_fixed: bool = False) -> dict:
    """Read the weight files back into the dict QNet takes (from_fixed: decode the *_fixed files on cfg['fmt_w'])."""
    H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
    lin = bool(cfg.get("en_lin_map", True))
    w = {"w_q": np.empty((D, V), np.float32), "w_ans": np.empty((V, D), np.float32),
         "w_a": [np.empty((D, V), np.float32) for _ in range(H)], "w_c": [np.empty((D, V), np.float32) for _ in range(H)],
         "w_h": [np.empty((D, D), np.float32) for _ in range(H)] if lin else None}
    fm = (abi.Fmt * H)(*[abi.Fmt(*f) for f in cfg["fmt_w"]])
    abi.check(abi.lib.qmann_weights_load(str(directory).encode(), C.byref(_weights_struct(w, H, D, V, lin)),
                                         1 if from_fixed else 0, fm), "qmann_weights_load")
    return w


def _net_from_cfg(cfg: dict):
    """abi.Net (qmann_net) with the dimensions, modes and Q-formats of `cfg`; lin_map pointers left NULL."""
    n = abi.Net()
    n.n_hop, n.dim_emb, n.dim_emb_pad, n.dim_input = cfg["n_hop"], cfg["dim_emb"], pad16(cfg["dim_emb"]), cfg["dim_input"]
    n.attention_mode = cfg["attention_mode"]
    n.softmax_base = cfg.get("softmax_variant", 0)      # 0 e^x, 1 2^x, 2 exp_plan
    n.softmax_shift_based = 1 if cfg.get("softmax_shift_based") else 0
    if cfg.get("att_scale") is not None:                # EN_SC_ATT: one learnt scalar per hop
        n.en_att_scale = 1
        for h in range(cfg["n_hop"]):
            n.att_scale[h] = float(np.float32(cfg["att_scale"][h]))
    n.en_non_linearity = 1 if cfg.get("en_non_lin") else 0
    if cfg.get("en_pe"):                                # EN_PE: position weights on the question's word slots
        n.en_pe, n.pe_dim_word = 1, int(cfg["pe_dim_word"])
    n.en_lin_map = 1 if cfg.get("en_lin_map", True) else 0
    n.num_bit = cfg.get("num_bit", 8)
    for h in range(cfg["n_hop"]):
        n.act[h] = abi.Fmt(*cfg["fmt"][h])
        n.w[h] = abi.Fmt(*cfg["fmt_w"][h])
        n.att[h] = abi.Fmt(*cfg["fmt_att"][h])
    n.bin = abi.Fmt(*cfg["fmt_bin"])
    return n


@dataclass
class HopTaps:
    score_codes: torch.Tensor
    scores: torch.Tensor
    probs: torch.Tensor
    o: torch.Tensor
    u: torch.Tensor


def _zero_cost_and_match(dev):
    """the two accumulators of the answer layer (float cost, int32 match count): one 8-byte allocation, ONE fill on the stream
    (two torch.zeros are two 6-us fill kernels in front of a 0.6-ms forward)"""
    acc = torch.zeros(2, dtype=torch.int32, device=dev)
    return acc[:1].view(torch.float32), acc[1:]


class QNet:
    """cfg: the dict shape used by oracle/pyoracle.py (n_hop, dim_emb, dim_input, attention_mode,
    softmax_variant (0 e^x / 1 2^x), en_lin_map, fmt, fmt_w, fmt_att, fmt_bin).
    weights: float32 numpy arrays w_q [D][V], w_a[h]/w_c[h] [D][V], w_h[h] [D][D], w_ans [V][D];
    w_q / w_a / w_c may be absent for nets that start from ready-made memories."""

    def __init__(self, cfg: dict, weights: dict, device="cuda:0", stream=None):
        self.cfg = cfg
        self.dev = torch.device(device)
        self.H, self.D, self.V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
        self.Dp = pad16(self.D)
        self.stream = stream
        n = self.net = _net_from_cfg(cfg)

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.dev)
        self.w_ans = up(weights["w_ans"])
        self.w_q = up(weights["w_q"]) if weights.get("w_q") is not None else None
        self.w_a = [up(w) for w in weights["w_a"]] if weights.get("w_a") is not None else None
        self.w_c = [up(w) for w in weights["w_c"]] if weights.get("w_c") is not None else None
        self.lin_map_i8 = []
        if n.en_lin_map:
            for h in range(self.H):
                wf = up(weights["w_h"][h])
                self.lin_map_i8.append(self.quantize_i8(wf, cfg["fmt_w"][h], abi.CODE_SIGNMAG))
                n.lin_map[h] = self.lin_map_i8[h].data_ptr()

    @classmethod
    def from_model(cls, cfg: dict, hm: "HostModel", stream=None):
        """A QNet that computes from the parameters of a library model object (qmann_model_net): the linear-map codes and
        the answer matrix are the ones inside `hm`'s parameter blob -- e.g. a replica built from a broadcast blob."""
        self = cls.__new__(cls)
        self.cfg, self.dev, self.stream = cfg, hm.dev, stream
        self.H, self.D, self.V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
        self.Dp = pad16(self.D)
        self.net, wa = hm.net_and_w_ans()
        self._owner = hm                                   # keeps the blob alive
        self.w_ans = torch.as_tensor(_DevArray(wa, (self.V, self.D), "<f4"), device=self.dev)
        self.w_q = self.w_a = self.w_c = None
        self.lin_map_i8 = []
        return self

    # ---- helpers -------------------------------------------------------------------------
    def _s(self):
        return C.c_void_p(self.stream) if self.stream else None

    def quantize_i8(self, x: torch.Tensor, fmt, layout=abi.CODE_TWOS) -> torch.Tensor:
        """float [rows][cols] -> int8 codes [rows][Dp-padded cols] of Q(fmt)(x)."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        rows = x.numel() // x.shape[-1]
        cols = x.shape[-1]
        pitch = pad16(cols) if cols <= 256 else cols
        out = torch.empty((*x.shape[:-1], pitch), dtype=torch.int8, device=x.device)
        abi.check(abi.lib.qmann_quantize_i8(_ptr(x), _ptr(out), rows, cols, pitch, abi.Fmt(*fmt), layout, self._s()),
                  "qmann_quantize_i8")
        return out

    # ---- stages --------------------------------------------------------------------------
    def embed(self, story: torch.Tensor, question: torch.Tensor):
        """story [rows_total][V] float, question [B][V] float -> keys, vals int8 [H][rows][Dp], u0 [B][D]."""
        rows = story.shape[0]
        keys = torch.empty((self.H, rows, self.Dp), dtype=torch.int8, device=self.dev)
        vals = torch.empty_like(keys)
        wa = (C.c_void_p * self.H)(*[w.data_ptr() for w in self.w_a])
        wc = (C.c_void_p * self.H)(*[w.data_ptr() for w in self.w_c])
        abi.check(abi.lib.qmann_embed_story(C.byref(self.net), _ptr(story), rows, wa, wc, _ptr(keys), _ptr(vals),
                                            rows * self.Dp, self._s()), "qmann_embed_story")
        u0 = torch.empty((question.shape[0], self.D), dtype=torch.float32, device=self.dev)
        abi.check(abi.lib.qmann_embed_query(C.byref(self.net), _ptr(question), _ptr(self.w_q), _ptr(u0),
                                            question.shape[0], self._s()), "qmann_embed_query")
        return keys, vals, u0

    def make_tables(self):
        """int8 [V][Dp] gather tables of the embedding matrices (word-index input path)."""
        def tab(w, fmt):
            out = torch.empty((self.V, self.Dp), dtype=torch.int8, device=self.dev)
            abi.check(abi.lib.qmann_quantize_table_i8(_ptr(w), _ptr(out), self.D, self.Dp, self.V, abi.Fmt(*fmt),
                                                      self._s()), "qmann_quantize_table_i8")
            return out
        self.t_q = tab(self.w_q, self.cfg["fmt_w"][0])
        self.t_a = [tab(self.w_a[h], self.cfg["fmt_w"][h]) for h in range(self.H)]
        self.t_c = [tab(self.w_c[h], self.cfg["fmt_w"][h]) for h in range(self.H)]

    def embed_idx(self, story_words: torch.Tensor, question_words: torch.Tensor, time_last=True):
        """story_words int16/uint16-as-int16 [rows][W], question_words [B][Wq] (0xFFFF = unused)."""
        rows, W = story_words.shape
        keys = torch.empty((self.H, rows, self.Dp), dtype=torch.int8, device=self.dev)
        vals = torch.empty_like(keys)
        ta = (C.c_void_p * self.H)(*[t.data_ptr() for t in self.t_a])
        tc = (C.c_void_p * self.H)(*[t.data_ptr() for t in self.t_c])
        abi.check(abi.lib.qmann_embed_story_idx(C.byref(self.net), _ptr(story_words), rows, W, int(time_last), ta, tc,
                                                _ptr(keys), _ptr(vals), rows * self.Dp, self._s()),
                  "qmann_embed_story_idx")
        B, Wq = question_words.shape
        u0 = torch.empty((B, self.D), dtype=torch.float32, device=self.dev)
        abi.check(abi.lib.qmann_embed_query_idx(C.byref(self.net), _ptr(question_words), Wq, _ptr(self.t_q), _ptr(u0), B,
                                                self._s()), "qmann_embed_query_idx")
        return keys, vals, u0

    def hops(self, keys, vals, row_off, max_slots, u0, taps=False, u_out=None):
        """keys/vals int8 [H][rows][Dp]; row_off uint32 (as int32 tensor) [B+1]; u0 [B][D] float."""
        B = u0.shape[0]
        rows = keys.shape[1]
        assert keys.shape == (self.H, rows, self.Dp) and keys.is_contiguous() and vals.is_contiguous()
        if u_out is None:
            u_out = torch.empty_like(u0)
        tp, tobj = None, None
        if taps:
            tobj = HopTaps(
                torch.zeros((self.H, rows), dtype=torch.int32, device=self.dev),
                torch.zeros((self.H, rows), dtype=torch.float32, device=self.dev),
                torch.zeros((self.H, rows), dtype=torch.float32, device=self.dev),
                torch.zeros((B, self.H, self.D), dtype=torch.float32, device=self.dev),
                torch.zeros((B, self.H, self.D), dtype=torch.float32, device=self.dev))
            tp = abi.Taps(tobj.score_codes.data_ptr(), tobj.scores.data_ptr(), tobj.probs.data_ptr(),
                          tobj.o.data_ptr(), tobj.u.data_ptr())
        abi.check(abi.lib.qmann_hops_i8(C.byref(self.net), _ptr(keys), _ptr(vals), rows * self.Dp, _ptr(row_off),
                                        max_slots, _ptr(u0), _ptr(u_out), C.byref(tp) if tp else None, B,
                                        self._s()), "qmann_hops_i8")
        return (u_out, tobj) if taps else u_out

    def pack_planes(self, sm_codes: torch.Tensor, num_bit: int) -> torch.Tensor:
        """sign-magnitude int8 [..., rows, Dp] -> packed bit planes int64 [..., rows, Dp/64, num_bit]."""
        assert sm_codes.dtype == torch.int8 and sm_codes.is_contiguous() and sm_codes.shape[-1] == self.Dp
        rows = sm_codes.numel() // self.Dp
        out = torch.empty((*sm_codes.shape[:-1], self.Dp // 64, num_bit), dtype=torch.int64, device=sm_codes.device)
        abi.check(abi.lib.qmann_pack_bitplanes(_ptr(sm_codes), _ptr(out), rows, self.Dp, num_bit, self._s()),
                  "qmann_pack_bitplanes")
        return out

    def hops_packed(self, key_planes, vals, row_off, max_slots, u0, taps=False):
        """Hamming V0 / V1 attention: key_planes int64 [H][rows][Dp/64][num_bit], vals int8 [H][rows][Dp]."""
        B = u0.shape[0]
        rows = vals.shape[1]
        nb = key_planes.shape[-1]
        u_out = torch.empty_like(u0)
        tp, tobj = None, None
        if taps:
            tobj = HopTaps(
                torch.zeros((self.H, rows), dtype=torch.int32, device=self.dev),
                torch.zeros((self.H, rows), dtype=torch.float32, device=self.dev),
                torch.zeros((self.H, rows), dtype=torch.float32, device=self.dev),
                torch.zeros((B, self.H, self.D), dtype=torch.float32, device=self.dev),
                torch.zeros((B, self.H, self.D), dtype=torch.float32, device=self.dev))
            tp = abi.Taps(tobj.score_codes.data_ptr(), tobj.scores.data_ptr(), tobj.probs.data_ptr(),
                          tobj.o.data_ptr(), tobj.u.data_ptr())
        abi.check(abi.lib.qmann_hops_packed(C.byref(self.net), _ptr(key_planes), rows * (self.Dp // 64) * nb * 8,
                                            _ptr(vals), rows * self.Dp, _ptr(row_off), max_slots, _ptr(u0),
                                            _ptr(u_out), C.byref(tp) if tp else None, B, self._s()),
                  "qmann_hops_packed")
        return (u_out, tobj) if taps else u_out

    def answer(self, u, answer=None, want_probs=False, serial=False):
        """serial=True: qmann_answer_f32_serial (the reference's order of additions, bit-equal logits); default: the library's
        choice (the fused bf16 form at the bAbI shapes, within 1e-5 on the probabilities)"""
        B = u.shape[0]
        pred = torch.empty(B, dtype=torch.int32, device=self.dev)
        probs = torch.empty((B, self.V), dtype=torch.float32, device=self.dev) if want_probs else None
        cost, match = _zero_cost_and_match(self.dev) if answer is not None else (None, None)
        fn = abi.lib.qmann_answer_f32_serial if serial else abi.lib.qmann_answer_f32
        abi.check(fn(C.byref(self.net), _ptr(self.w_ans), _ptr(u), _ptr(answer), _ptr(pred),
                     _ptr(probs), _ptr(cost), _ptr(match), B, self._s()), "qmann_answer_f32")
        return pred, probs, cost, match

    def answer_i8(self, u, w_ans_i8, w_fmt, answer=None, want_probs=False):
        """Answer layer on the int8 matrix cores; w_ans_i8 [V][Dp] two's-complement codes of Q(w_fmt)."""
        B = u.shape[0]
        pred = torch.empty(B, dtype=torch.int32, device=self.dev)
        logits = torch.empty((B, self.V), dtype=torch.float32, device=self.dev)
        probs = torch.empty((B, self.V), dtype=torch.float32, device=self.dev) if want_probs else None
        cost, match = _zero_cost_and_match(self.dev) if answer is not None else (None, None)
        abi.check(abi.lib.qmann_answer_i8(C.byref(self.net), _ptr(w_ans_i8), abi.Fmt(*w_fmt), _ptr(u), _ptr(logits),
                                          _ptr(answer), _ptr(pred), _ptr(probs), _ptr(cost), _ptr(match), B,
                                          self._s()), "qmann_answer_i8")
        return pred, probs, cost, match, logits

    def forward_bow(self, story, question, row_off, max_slots, answer=None, taps=False):
        keys, vals, u0 = self.embed(story, question)
        r = self.hops(keys, vals, row_off, max_slots, u0, taps=taps)
        u = r[0] if taps else r
        pred, probs, cost, match = self.answer(u, answer, want_probs=taps)
        return dict(pred=pred, probs=probs, cost=cost, match=match, u=u, u0=u0, keys=keys, vals=vals,
                    taps=r[1] if taps else None)


class HostModel:
    """include/qmann_model.h through ctypes: the library's own host-side object (C++), one call per batch.
    `weights` is the dict QNet takes (host float arrays); nothing of the orchestration happens in Python."""

    def __init__(self, cfg: dict, weights: dict, device="cuda:0", stream=None):
        self.cfg, self.dev, self.stream = cfg, torch.device(device), stream
        H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
        self.D = D
        w = {k: ([np.ascontiguousarray(m, np.float32) for m in v] if isinstance(v, (list, tuple))
                 else np.ascontiguousarray(v, np.float32)) for k, v in weights.items() if v is not None}
        self._net = _net_from_cfg(cfg)
        ws = _weights_struct(w, H, D, V, bool(cfg.get("en_lin_map", True)))
        h = C.c_void_p()
        idx = self.dev.index if self.dev.index is not None else -1
        abi.check(abi.lib.qmann_model_create_on(C.byref(h), idx, C.byref(self._net), C.byref(ws), self._s()), "qmann_model_create_on")
        self.h = h

    @classmethod
    def from_params(cls, cfg: dict, blob_ptr: int, nbytes: int, device="cuda:0", stream=None):
        """qmann_model_create_from_params: a replica from a parameter blob (device or host pointer)."""
        self = cls.__new__(cls)
        self.cfg, self.dev, self.stream = cfg, torch.device(device), stream
        self.D = cfg["dim_emb"]
        h = C.c_void_p()
        idx = self.dev.index if self.dev.index is not None else -1
        abi.check(abi.lib.qmann_model_create_from_params(C.byref(h), idx, C.c_void_p(blob_ptr), nbytes, self._s()),
                  "qmann_model_create_from_params")
        self.h = h
        return self

    def params(self):
        """(device pointer, bytes) of the model's quantised parameter blob (owned by the model)."""
        p, n = C.c_void_p(), C.c_size_t()
        abi.check(abi.lib.qmann_model_params(self.h, C.byref(p), C.byref(n)), "qmann_model_params")
        return p.value, n.value

    def params_bytes(self) -> bytes:
        """the blob copied to the host (through boundary B's own D2H verb)"""
        p, n = self.params()
        buf = np.empty((n + 3) // 4, np.float32)
        abi.lib.cuda_copy_dev2host(buf.ctypes.data_as(C.c_void_p), C.c_void_p(p), (n + 3) // 4)
        return buf.tobytes()[:n]

    def net_and_w_ans(self):
        """(abi.Net with the device lin_map pointers of this model, device pointer of its answer matrix)"""
        net, wa = abi.Net(), C.c_void_p()
        abi.check(abi.lib.qmann_model_net(self.h, C.byref(net), C.byref(wa)), "qmann_model_net")
        return net, wa.value

    def _s(self):
        return C.c_void_p(self.stream) if self.stream else None

    def close(self):
        if getattr(self, "h", None) and abi is not None:      # (module globals are gone at interpreter shutdown)
            abi.lib.qmann_model_destroy(self.h)
            self.h = None

    __del__ = close

    def _out(self, B, answer):
        pred = torch.empty(B, dtype=torch.int32, device=self.dev)
        cost, match = _zero_cost_and_match(self.dev) if answer is not None else (None, None)
        return pred, cost, match

    def last_u(self, B):
        """copy of the final hop state of the last batch, [B][D]"""
        p = abi.lib.qmann_model_last_u(self.h)
        host = np.empty((B, self.D), np.float32)
        abi.lib.cuda_copy_dev2host(host.ctypes.data_as(C.c_void_p), C.c_void_p(p), B * self.D)   # boundary B's own D2H verb
        return torch.from_numpy(host).to(self.dev)

    def forward_words(self, story_words, question_words, row_off, max_slots, answer=None):
        B = question_words.shape[0]
        pred, cost, match = self._out(B, answer)
        abi.check(abi.lib.qmann_model_forward_words(self.h, _ptr(story_words), story_words.shape[0], story_words.shape[1],
                                                    _ptr(question_words), question_words.shape[1], _ptr(row_off),
                                                    max_slots, B, _ptr(answer), _ptr(pred), _ptr(cost), _ptr(match),
                                                    self._s()), "qmann_model_forward_words")
        return pred, cost, match

    def forward_bow(self, story, question, row_off, max_slots, answer=None):
        B = question.shape[0]
        pred, cost, match = self._out(B, answer)
        abi.check(abi.lib.qmann_model_forward_bow(self.h, _ptr(story), story.shape[0], _ptr(question), _ptr(row_off),
                                                  max_slots, B, _ptr(answer), _ptr(pred), _ptr(cost), _ptr(match),
                                                  self._s()), "qmann_model_forward_bow")
        return pred, cost, match
