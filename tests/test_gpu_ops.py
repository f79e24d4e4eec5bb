"""GPU parity of boundary B (the op-at-a-time `cuda_*` entry points) against the CPU oracle.

Every call goes through the C-ABI of libqmann_hip.so with raw device pointers, the way the
reference's layer.c drives it.  Fixed-point results must be bit-exact; float-mode sums and
softmax carry the tolerance stated in each test (north_star: 1e-5).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg
from pyoracle import SM_CPU_POW2, SM_CUDA

pytestmark = pytest.mark.gpu

FORMATS = [(5, 2), (6, 1), (4, 3), (2, 5), (0, 7)]


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    load_pkg()
    import qmann_amd.abi as abi

    class Env:
        pass
    e = Env()
    e.torch, e.abi, e.lib = torch, abi, abi.lib
    e.dev = torch.device("cuda:0")

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(e.dev)

    def ptr(t):
        return C.c_void_p(t.data_ptr())
    e.up, e.ptr = up, ptr
    e.empty = lambda *s: torch.empty(s, dtype=torch.float32, device=e.dev)
    return e


def grid_vals(rng, shape, fmt, sigma_codes=40):
    iwl, frac = fmt
    m = (1 << (iwl + frac)) - 1
    k = np.clip(np.rint(rng.normal(0, sigma_codes, shape)), -m, m)
    return (k / (1 << frac)).astype(np.float32)


@pytest.mark.parametrize("fmt_w,fmt_in", [((5, 2), (5, 2)), ((6, 1), (5, 2)), ((4, 3), (5, 2)), ((2, 5), (2, 5))])
@pytest.mark.parametrize("dims", [(30, 60), (60, 60), (256, 60), (128, 128), (1, 7)])
def test_dense_fwd_fixed_bit_exact(env, oracle, fmt_w, fmt_in, dims):
    dim_in, dim_out = dims
    rng = np.random.default_rng(hash((fmt_w, fmt_in, dims)) % 2**32)
    w = rng.normal(0, 2.0, (dim_out, dim_in)).astype(np.float32)       # off-grid on purpose
    x = rng.normal(0, 2.0, dim_in).astype(np.float32)
    dw, dx, do = env.up(w), env.up(x), env.empty(dim_out)
    env.lib.cuda_dense_fwd(env.ptr(dw), None, env.ptr(dx), env.ptr(do), None, dim_in, dim_out, b"NULL", True,
                           fmt_in[0], fmt_in[1], fmt_w[0], fmt_w[1], 3, False)
    np.testing.assert_array_equal(do.cpu().numpy(), oracle.dense_fwd(w, x, True, fmt_in, fmt_w))


def test_dense_fwd_float_and_relu(env, oracle):
    rng = np.random.default_rng(7)
    w = rng.normal(0, 0.1, (30, 60)).astype(np.float32)
    x = grid_vals(rng, 60, (5, 2))
    dw, dx, do = env.up(w), env.up(x), env.empty(30)
    env.lib.cuda_dense_fwd(env.ptr(dw), None, env.ptr(dx), env.ptr(do), None, 60, 30, b"NULL", False, 8, 7, 8, 7, 3,
                           False)
    # float mode: wave-shuffle sum order differs from the reference's serial loop -> 1e-5 relative
    np.testing.assert_allclose(do.cpu().numpy(), oracle.dense_fwd(w, x, False, (8, 7), (8, 7)), rtol=1e-5,
                               atol=1e-6)
    env.lib.cuda_dense_fwd(env.ptr(dw), None, env.ptr(dx), env.ptr(do), None, 60, 30, b"RELU", True, 5, 2, 5, 2, 3,
                           False)
    np.testing.assert_array_equal(do.cpu().numpy(), oracle.dense_fwd(w, x, True, (5, 2), (5, 2), b"RELU"))


def test_dense_mat_fwd_matches_reference_golden(env, gold):
    """Straight against the reference's own dense_mat_fwd outputs (tests/golden/ref_dense_mat.npz)."""
    g = gold("ref_dense_mat.npz")
    for k, (n_sen, dim_input, iwl, frac) in enumerate(g["cases"]):
        X, W = g[f"X{k}"], g[f"W{k}"]
        dX, dW, dO = env.up(X), env.up(W), env.empty(X.shape[0], W.shape[0])
        env.lib.cuda_dense_mat_fwd(env.ptr(dW), None, env.ptr(dX), env.ptr(dO), None, X.shape[1], W.shape[0],
                                   X.shape[0], True, int(iwl), int(frac), 3, False)
        np.testing.assert_array_equal(dO.cpu().numpy(), g[f"fixed{k}"], err_msg=f"case {k}")
        env.lib.cuda_dense_mat_fwd(env.ptr(dW), None, env.ptr(dX), env.ptr(dO), None, X.shape[1], W.shape[0],
                                   X.shape[0], False, int(iwl), int(frac), 3, False)
        np.testing.assert_allclose(dO.cpu().numpy(), g[f"float{k}"], rtol=1e-5, atol=1e-5, err_msg=f"case {k}")


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("shape", [(1, 60), (10, 60), (50, 60), (64, 256), (1000, 128), (10000, 128)])
def test_dot_mat_vec_scores_and_readout_fixed(env, oracle, fmt, shape):
    r, c = shape
    rng = np.random.default_rng(hash((fmt, shape)) % 2**32)
    M = grid_vals(rng, (r, c), fmt, 30)
    u = grid_vals(rng, c, fmt, 30)
    dM, du, ds = env.up(M), env.up(u), env.empty(r)
    env.lib.cuda_dot_mat_vec_fwd(env.ptr(dM), env.ptr(du), env.ptr(ds), None, r, c, False, True, fmt[0], fmt[1],
                                 fmt[0], fmt[1], 3, False)
    np.testing.assert_array_equal(ds.cpu().numpy(), oracle.dot_mat_vec_fwd(M, u, False, True, fmt, fmt))
    p = oracle.softmax_fwd(rng.normal(0, 3, r).astype(np.float32), SM_CUDA)
    dp, do = env.up(p), env.empty(c)
    env.lib.cuda_dot_mat_vec_fwd(env.ptr(dM), env.ptr(dp), env.ptr(do), None, r, c, True, True, fmt[0], fmt[1],
                                 fmt[0], fmt[1], 3, False)
    np.testing.assert_array_equal(do.cpu().numpy(), oracle.dot_mat_vec_fwd(M, p, True, True, fmt, fmt))


def test_dot_mat_vec_float_mode(env, oracle):
    rng = np.random.default_rng(11)
    M = grid_vals(rng, (50, 60), (5, 2))
    u = grid_vals(rng, 60, (5, 2))
    dM, du, ds = env.up(M), env.up(u), env.empty(50)
    env.lib.cuda_dot_mat_vec_fwd(env.ptr(dM), env.ptr(du), env.ptr(ds), None, 50, 60, False, False, 5, 2, 5, 2, 3,
                                 False)
    # operands sit on a grid, so even the float dot product is exact
    np.testing.assert_array_equal(ds.cpu().numpy(), oracle.dot_mat_vec_fwd(M, u, False, False, (5, 2), (5, 2)))


@pytest.mark.parametrize("iwl", [5, 2, 6])
@pytest.mark.parametrize("shape", [(10, 60), (50, 256), (777, 128)])
def test_hamming_appx_attention_bit_exact(env, oracle, iwl, shape):
    """CUDA mode-3 attention (lib/layer_cuda.cu:355-541): integer arithmetic scaled by 2^-10."""
    r, c = shape
    rng = np.random.default_rng(hash((iwl, shape)) % 2**32)
    frac = 7 - iwl
    M = grid_vals(rng, (r, c), (iwl, frac), 50)
    u = grid_vals(rng, c, (iwl, frac), 50)
    # sprinkle off-grid values, saturating values and negative underflows ("minus zero" words)
    M.ravel()[::7] += np.float32(0.013)
    M.ravel()[::11] = np.float32(2.0 ** iwl + 3.0)
    M.ravel()[::13] = np.float32(-1e-4)
    u[::5] = np.float32(-(2.0 ** iwl) - 1.0)
    dM, du, ds = env.up(M), env.up(u), env.empty(r)
    env.lib.cuda_dot_mat_vec_fwd_appx(env.ptr(dM), env.ptr(du), env.ptr(ds), None, None, r, c, True, iwl, frac, 3,
                                      1 + iwl + frac, False, False)
    np.testing.assert_array_equal(ds.cpu().numpy(),
                                  oracle.dot_mat_vec_fwd_appx(M, u, False, True, iwl, frac, 1 + iwl + frac))


@pytest.mark.parametrize("dim", [1, 2, 10, 50, 256, 1024, 10000])
def test_softmax_any_dim(env, oracle, dim):
    rng = np.random.default_rng(dim)
    x = (rng.integers(-127, 128, dim) / 4.0).astype(np.float32)
    dx, do, dm = env.up(x), env.empty(dim), env.empty(1)
    env.lib.qmann_abi_set_softmax_base(0)
    env.lib.cuda_softmax_fwd(env.ptr(do), env.ptr(dx), None, None, env.ptr(dm), dim, False, False)
    got = do.cpu().numpy()
    assert dm.cpu().numpy()[0] == x.max()
    np.testing.assert_allclose(got, oracle.softmax_fwd(x, SM_CUDA), rtol=1e-5, atol=1e-7)   # north_star tolerance
    env.lib.qmann_abi_set_softmax_base(1)
    env.lib.cuda_softmax_fwd(env.ptr(do), env.ptr(dx), None, None, env.ptr(dm), dim, False, False)
    np.testing.assert_allclose(do.cpu().numpy(), oracle.softmax_fwd(x, SM_CPU_POW2), rtol=1e-5, atol=1e-7)
    env.lib.qmann_abi_set_softmax_base(0)


def test_softmax_cpu_base_against_reference_golden(env, gold):
    g = gold("ref_softmax.npz")
    env.lib.qmann_abi_set_softmax_base(1)
    try:
        for k in range(7):
            x = g[f"x{k}"]
            dx, do, dm = env.up(x), env.empty(x.size), env.empty(1)
            env.lib.cuda_softmax_fwd(env.ptr(do), env.ptr(dx), None, None, env.ptr(dm), x.size, False, False)
            np.testing.assert_allclose(do.cpu().numpy(), g[f"pow2_{k}"], rtol=1e-5, atol=1e-7)
    finally:
        env.lib.qmann_abi_set_softmax_base(0)


def test_sum_vec_matches_reference_golden(env, gold):
    g = gold("ref_sum_vec.npz")
    a, b = g["a"], g["b"]
    da, db, do = env.up(a), env.up(b), env.empty(a.size)
    for iwl, frac in [(0, 7), (2, 5), (4, 3), (5, 2), (6, 1)]:
        env.lib.cuda_sum_vec_fwd(env.ptr(da), env.ptr(db), env.ptr(do), a.size, True, iwl, frac, 3, False)
        np.testing.assert_array_equal(do.cpu().numpy(), g[f"q{iwl}_{frac}"])
    env.lib.cuda_sum_vec_fwd(env.ptr(da), env.ptr(db), env.ptr(do), a.size, False, 5, 2, 3, False)
    np.testing.assert_array_equal(do.cpu().numpy(), g["float"])


def test_cross_entropy_run_and_loads(env, oracle):
    torch, lib = env.torch, env.lib
    rng = np.random.default_rng(3)
    dim = 30
    slots = [C.c_void_p() for _ in range(8)]
    lib.cuda_cross_entropy_constructor(*[C.byref(s) for s in slots], dim)
    lib.cuda_cross_entropy_init(*slots[:6], slots[7], dim)
    want_cost, want_cnt = np.float32(0), 0
    for t in range(20):
        h = rng.random(dim).astype(np.float32)
        if t % 3 == 0:
            h[5] = h[17] = h.max() + 0.1           # tie: the highest index must win
        h /= h.sum()
        y = np.zeros(dim, np.float32); y[17 if t % 2 else rng.integers(0, dim)] = 1.0
        dh, dy = env.up(h), env.up(y)
        lib.cuda_cross_entropy_run(*slots[:7], None, env.ptr(dh), env.ptr(dy), None, None, slots[7], None, dim, 3)
        pred = np.zeros(1, np.uint32)
        lib.cuda_copy_dev2host(pred.ctypes.data_as(C.c_void_p), slots[6], 1)
        opred, ocost, ocnt, ograd = oracle.cross_entropy_run(h, y)
        assert int(pred[0]) == opred
        g = np.zeros(dim, np.float32)
        lib.cuda_copy_dev2host(g.ctypes.data_as(C.c_void_p), slots[7], dim)
        np.testing.assert_array_equal(g, ograd)
        want_cost = np.float32(want_cost + np.float32(ocost)); want_cnt += ocnt
    c3 = (C.c_float * 3)(); m3 = (C.c_uint * 3)()
    lib.cuda_cross_entropy_cost_load(*slots[:3], C.byref(c3, 0), C.byref(c3, 4), C.byref(c3, 8))
    lib.cuda_cross_entropy_m_cnt_load(*slots[3:6], C.byref(m3, 0), C.byref(m3, 4), C.byref(m3, 8))
    assert m3[2] == want_cnt and m3[0] == 0 and m3[1] == 0
    assert c3[2] == pytest.approx(float(want_cost), rel=1e-6)
    lib.cuda_cross_entropy_m_cnt_load(*slots[3:6], C.byref(m3, 0), C.byref(m3, 4), C.byref(m3, 8))
    assert m3[2] == 0                                   # the load resets the accumulators
    lib.cuda_cross_entropy_destructor(*slots)


def test_copy_accum_set_value(env):
    rng = np.random.default_rng(5)
    src = rng.normal(0, 1, (7, 13)).astype(np.float32)       # [row][col]
    ds, dd = env.up(src), env.empty(13, 7)
    env.lib.cuda_copy_mat(env.ptr(ds), env.ptr(dd), 13, 7, True)
    np.testing.assert_array_equal(dd.cpu().numpy(), src.T)
    d2 = env.up(np.ones((7, 13), np.float32))
    env.lib.cuda_accum_mat(env.ptr(ds), env.ptr(d2), 13, 7, False)
    np.testing.assert_array_equal(d2.cpu().numpy(), src + 1.0)
    d3 = env.up(np.ones(100, np.float32))
    env.lib.cuda_set_value(env.ptr(d3), 0.0, 100, 0, 10)
    want = np.ones(100, np.float32); want[::10] = 0
    np.testing.assert_array_equal(d3.cpu().numpy(), want)


def test_one_query_through_the_layer_sequence(env, oracle, gold):
    """Drive the op-level ABI in the reference's own order (MemN2N/MemN2N.c:2626-2697) for bAbI
    queries and compare every stage with the oracle's composite forward."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_golden", ROOT / "oracle" / "gen_golden.py")
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    b = gold("babi_qa1_test64.npz")
    V = int(b["dim_input"]); D = 60; H = 3
    story, ques, n_sen = b["story"].astype(np.float32), b["question"].astype(np.float32), b["n_sen"]
    lib, up, ptr, empty = env.lib, env.up, env.ptr, env.empty
    for mode in (2, 3):
        cfg = gg.babi_cfg(V, mode, 0)
        wts = gg.seeded_weights(1234, H, D, V, 1.0)
        m = oracle.make_model(cfg, wts)
        dwq, dwans = up(wts["w_q"]), up(wts["w_ans"])
        dwa = [up(w) for w in wts["w_a"]]; dwc = [up(w) for w in wts["w_c"]]; dwh = [up(w) for w in wts["w_h"]]
        off = 0
        for i in range(12):
            ns = int(n_sen[i])
            X, q = story[off:off + ns], ques[i]; off += ns
            opred, t = oracle.forward(m, X, q)
            dX, dq = up(X), up(q)
            u = empty(D)
            fw0 = cfg["fmt_w"][0]
            lib.cuda_dense_fwd(ptr(dwq), None, ptr(dq), ptr(u), None, V, D, b"NULL", True, *fw0, *fw0, 3, False)
            np.testing.assert_array_equal(u.cpu().numpy(), t["u0"])
            for h in range(H):
                fw, fa, fm, fb = cfg["fmt_w"][h], cfg["fmt"][h], cfg["fmt_att"][h], cfg["fmt_bin"]
                Mk, Mc = empty(ns, D), empty(ns, D)
                lib.cuda_dense_mat_fwd(ptr(dwa[h]), None, ptr(dX), ptr(Mk), None, V, D, ns, True, *fw, 3, False)
                lib.cuda_dense_mat_fwd(ptr(dwc[h]), None, ptr(dX), ptr(Mc), None, V, D, ns, True, *fw, 3, False)
                s, p, o, lu, un = empty(ns), empty(ns), empty(D), empty(D), empty(D)
                if mode == 2:
                    lib.cuda_dot_mat_vec_fwd(ptr(Mk), ptr(u), ptr(s), None, ns, D, False, True, *fm, *fb, 3, False)
                else:
                    lib.cuda_dot_mat_vec_fwd_appx(ptr(Mk), ptr(u), ptr(s), None, None, ns, D, True, *fm, 3,
                                                  1 + fm[0] + fm[1], False, False)
                np.testing.assert_array_equal(s.cpu().numpy(), t["scores"][h], err_msg=f"scores q{i} h{h} m{mode}")
                lib.cuda_softmax_fwd(ptr(p), ptr(s), None, None, ptr(empty(1)), ns, False, False)
                np.testing.assert_allclose(p.cpu().numpy(), t["probs"][h], rtol=1e-5, atol=1e-7)
                if mode == 2:
                    lib.cuda_dot_mat_vec_fwd(ptr(Mc), ptr(p), ptr(o), None, ns, D, True, True, *fa, *fa, 3, False)
                else:
                    lib.cuda_dot_mat_vec_fwd_appx(ptr(Mc), ptr(p), ptr(o), None, None, ns, D, True, *fa, 3,
                                                  1 + fa[0] + fa[1], True, False)
                lib.cuda_dense_fwd(ptr(dwh[h]), None, ptr(u), ptr(lu), None, D, D, b"NULL", True, *fb, *fw, 3, False)
                lib.cuda_sum_vec_fwd(ptr(lu), ptr(o), ptr(un), D, True, *fa, 3, False)
                np.testing.assert_array_equal(o.cpu().numpy(), t["o"][h])
                np.testing.assert_array_equal(lu.cpu().numpy(), t["lu"][h])
                np.testing.assert_array_equal(un.cpu().numpy(), t["u"][h])
                u = un
            a, ph = empty(V), empty(V)
            lib.cuda_dense_fwd(ptr(dwans), None, ptr(u), ptr(a), None, D, V, b"NULL", False, 8, 7, 8, 7, 3, False)
            np.testing.assert_allclose(a.cpu().numpy(), t["logits"], rtol=1e-5, atol=1e-5)
            lib.cuda_softmax_fwd(ptr(ph), ptr(a), None, None, ptr(empty(1)), V, False, False)
            np.testing.assert_allclose(ph.cpu().numpy(), t["out_probs"], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# random shapes / formats through the per-op verbs (word lengths 1..7, the binarising (0,0) operand format,
# off-grid and saturating inputs)
# ---------------------------------------------------------------------------------------------
def rand_fmt(rng, lo=1, hi=7):
    wl = int(rng.integers(lo, hi + 1)); iwl = int(rng.integers(0, wl + 1))
    return (iwl, wl - iwl)


def wild(rng, shape, fmt):
    """values around the format's range: on-grid, off-grid, beyond the limits, tiny negatives"""
    lim = 2.0 ** fmt[0]
    x = rng.normal(0, 0.6 * lim + 0.05, shape).astype(np.float32)
    flat = x.reshape(-1)
    flat[::7] = np.float32(lim * 1.5); flat[::11] = np.float32(-lim * 2.0); flat[::13] = np.float32(-1e-5)
    return x


@pytest.mark.parametrize("seed", range(30))
def test_forward_verbs_random_shapes_and_formats(env, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    # dense_fwd: operand format may be the binarising (0,0)
    dim_in, dim_out = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    fw, fi = rand_fmt(rng), (rand_fmt(rng) if seed % 5 else (0, 0))
    w, x = wild(rng, (dim_out, dim_in), fw), wild(rng, dim_in, fi if sum(fi) else (2, 5))
    dw, dx, do = env.up(w), env.up(x), env.empty(dim_out)
    env.lib.cuda_dense_fwd(env.ptr(dw), None, env.ptr(dx), env.ptr(do), None, dim_in, dim_out, b"NULL", True,
                           fi[0], fi[1], fw[0], fw[1], 3, False)
    np.testing.assert_array_equal(do.cpu().numpy(), oracle.dense_fwd(w, x, True, fi, fw), err_msg="dense_fwd")
    # dense_mat_fwd
    n, f = int(rng.integers(1, 70)), rand_fmt(rng)
    X = wild(rng, (n, dim_in), f)
    dX, dO = env.up(X), env.empty(n, dim_out)
    env.lib.cuda_dense_mat_fwd(env.ptr(dw), None, env.ptr(dX), env.ptr(dO), None, dim_in, dim_out, n, True, f[0], f[1], 3, False)
    np.testing.assert_array_equal(dO.cpu().numpy(), oracle.dense_mat_fwd(w, X, True, f), err_msg="dense_mat_fwd")
    # dot_mat_vec_fwd: scores (matrix and vector formats differ) and the transposed read-out
    r, c = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    fm, fv = rand_fmt(rng), (rand_fmt(rng) if seed % 4 else (0, 0))
    M, u = wild(rng, (r, c), fm), wild(rng, c, fv if sum(fv) else (2, 5))
    dM, du, ds = env.up(M), env.up(u), env.empty(r)
    env.lib.cuda_dot_mat_vec_fwd(env.ptr(dM), env.ptr(du), env.ptr(ds), None, r, c, False, True, fm[0], fm[1], fv[0], fv[1], 3, False)
    np.testing.assert_array_equal(ds.cpu().numpy(), oracle.dot_mat_vec_fwd(M, u, False, True, fm, fv), err_msg="scores")
    p = oracle.softmax_fwd(rng.normal(0, 3, r).astype(np.float32), SM_CUDA)
    dp, dq = env.up(p), env.empty(c)
    env.lib.cuda_dot_mat_vec_fwd(env.ptr(dM), env.ptr(dp), env.ptr(dq), None, r, c, True, True, fm[0], fm[1], fm[0], fm[1], 3, False)
    np.testing.assert_array_equal(dq.cpu().numpy(), oracle.dot_mat_vec_fwd(M, p, True, True, fm, fm), err_msg="read-out")
    # sum_vec_fwd
    a_, b_ = wild(rng, c, fm), wild(rng, c, fm)
    da, db, dc = env.up(a_), env.up(b_), env.empty(c)
    env.lib.cuda_sum_vec_fwd(env.ptr(da), env.ptr(db), env.ptr(dc), c, True, fm[0], fm[1], 3, False)
    np.testing.assert_array_equal(dc.cpu().numpy(), oracle.sum_vec_fwd(a_, b_, True, fm), err_msg="sum_vec")
    # approximate attention: word length 8 formats, n compared bits
    ia = int(rng.integers(1, 7))
    Ma, ua = wild(rng, (r, c), (ia, 7 - ia)), wild(rng, c, (ia, 7 - ia))
    dMa, dua, dsa = env.up(Ma), env.up(ua), env.empty(r)
    env.lib.cuda_dot_mat_vec_fwd_appx(env.ptr(dMa), env.ptr(dua), env.ptr(dsa), None, None, r, c, True, ia, 7 - ia, 3, 8, False, False)
    np.testing.assert_array_equal(dsa.cpu().numpy(), oracle.dot_mat_vec_fwd_appx(Ma, ua, False, True, ia, 7 - ia, 8), err_msg="appx")


# ---------------------------------------------------------------------------------------------
# the deferred queue behind the forward verbs (include/qmann_abi.h "Deferred execution"), driven verb by verb
# ---------------------------------------------------------------------------------------------
def _host_loop(env, cfg, wts, story, ques, ans_onehot, n_sen, nq, defer, stray=None, dev_wts=None):
    """MemN2N.c's test loop (:2378-2702) in ctypes: per query the 31 verbs in the reference's order on FIXED layer buffers
    (as the host's structs hold them), accumulators fetched once at the end.  `stray`: a query index after which an
    unrelated verb (a vector sum on scratch) is issued -- the pattern breaks there and must fall back to the verbs."""
    lib, up, ptr, empty, torch = env.lib, env.up, env.ptr, env.empty, env.torch
    V, D, H = cfg["dim_input"], cfg["dim_emb"], cfg["n_hop"]
    lib.qmann_abi_set_defer(defer)
    if dev_wts is None:
        dev_wts = {k: ([up(w) for w in v] if isinstance(v, list) else up(v)) for k, v in wts.items()}
    dwq, dwans, dwa, dwc, dwh = dev_wts["w_q"], dev_wts["w_ans"], dev_wts["w_a"], dev_wts["w_c"], dev_wts["w_h"]
    offs = np.concatenate([[0], np.cumsum(n_sen[:nq])]).astype(np.int64)
    dm, dq, da = up(story[:offs[nq]]), up(ques[:nq]), up(ans_onehot[:nq])
    S = int(n_sen[:nq].max())
    u0 = empty(D); Mk = [empty(S, D) for _ in range(H)]; Mc = [empty(S, D) for _ in range(H)]
    s = [empty(S) for _ in range(H)]; p = [empty(S) for _ in range(H)]; o = [empty(D) for _ in range(H)]
    lu = [empty(D) for _ in range(H)]; sv = [empty(D) for _ in range(H)]
    a, ph, mxs, grad = empty(V), empty(V), empty(1), empty(V)
    cost = torch.zeros(3, device=env.dev); cnt = torch.zeros(3, dtype=torch.int32, device=env.dev)
    pred = torch.zeros(1, dtype=torch.int32, device=env.dev)
    junk = [empty(D) for _ in range(3)]
    fptr = lambda t, i: C.c_void_p(t.data_ptr() + 4 * i)
    for i in range(nq):
        ns = int(n_sen[i])
        X = C.c_void_p(dm.data_ptr() + 4 * V * int(offs[i])); q = C.c_void_p(dq.data_ptr() + 4 * V * i)
        y = C.c_void_p(da.data_ptr() + 4 * V * i)
        fw0 = cfg["fmt_w"][0]
        lib.cuda_dense_fwd(ptr(dwq), None, q, ptr(u0), None, V, D, b"NULL", True, *fw0, *fw0, 3, False)
        u = u0
        for h in range(H):
            fw, fa, fm, fb = cfg["fmt_w"][h], cfg["fmt"][h], cfg["fmt_att"][h], cfg["fmt_bin"]
            lib.cuda_dense_mat_fwd(ptr(dwa[h]), None, X, ptr(Mk[h]), None, V, D, ns, True, *fw, 3, False)
            lib.cuda_dense_mat_fwd(ptr(dwc[h]), None, X, ptr(Mc[h]), None, V, D, ns, True, *fw, 3, False)
            lib.cuda_dot_mat_vec_fwd(ptr(Mk[h]), ptr(u), ptr(s[h]), None, ns, D, False, True, *fm, *fb, 3, False)
            lib.cuda_softmax_fwd(ptr(p[h]), ptr(s[h]), None, None, ptr(mxs), ns, False, False)
            lib.cuda_dot_mat_vec_fwd(ptr(Mc[h]), ptr(p[h]), ptr(o[h]), None, ns, D, True, True, *fa, *fa, 3, False)
            lib.cuda_dense_fwd(ptr(dwh[h]), None, ptr(u), ptr(lu[h]), None, D, D, b"NULL", True, *fb, *fw, 3, False)
            lib.cuda_sum_vec_fwd(ptr(lu[h]), ptr(o[h]), ptr(sv[h]), D, True, *fa, 3, False)
            u = sv[h]
        lib.cuda_dense_fwd(ptr(dwans), None, ptr(u), ptr(a), None, D, V, b"NULL", False, 8, 7, 8, 7, 3, False)
        lib.cuda_softmax_fwd(ptr(ph), ptr(a), None, None, ptr(mxs), V, False, False)
        lib.cuda_cross_entropy_run(fptr(cost, 0), fptr(cost, 1), fptr(cost, 2), fptr(cnt, 0), fptr(cnt, 1), fptr(cnt, 2), ptr(pred),
                                   None, ptr(ph), y, None, None, ptr(grad), None, V, 3)
        if stray is not None and i == stray:
            lib.cuda_sum_vec_fwd(ptr(junk[0]), ptr(junk[1]), ptr(junk[2]), D, False, 0, 0, 3, False)
    hc = (C.c_float * 3)(); hm = (C.c_uint * 3)()
    lib.cuda_cross_entropy_cost_load(fptr(cost, 0), fptr(cost, 1), fptr(cost, 2), hc, C.byref(hc, 4), C.byref(hc, 8))
    lib.cuda_cross_entropy_m_cnt_load(fptr(cnt, 0), fptr(cnt, 1), fptr(cnt, 2), hm, C.byref(hm, 4), C.byref(hm, 8))
    st = env.abi.defer_stats()
    lib.qmann_abi_set_defer(0)
    return dict(match=int(hm[2]), cost=float(hc[2]), last_u=sv[H - 1].cpu().numpy().copy(), last_p=ph.cpu().numpy().copy(),
                last_pred=int(pred.item()), last_scores=s[0].cpu().numpy()[:int(n_sen[nq - 1])].copy(), stats=st)


def test_deferred_queue_equals_the_verbs(env, gold):
    """the same host loop with the queue off, on and in verify mode: match count equal, cost within the float tolerance, and
    after the loop every layer buffer holds the LAST query's values as the serial loop leaves them.  With a stray verb in the
    middle the run splits in two batched pieces and the stray verb executes in order between them."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_golden", ROOT / "oracle" / "gen_golden.py")
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    b = gold("babi_qa1_test64.npz")
    V, D, H, nq = int(b["dim_input"]), 60, 3, 40
    cfg = gg.babi_cfg(V, 2, 0)
    wts = gg.seeded_weights(4321, H, D, V, 1.0)
    args = (env, cfg, wts, b["story"].astype(np.float32), b["question"].astype(np.float32), b["answer"].astype(np.float32),
            b["n_sen"].astype(np.int64), nq)
    s0 = env.abi.defer_stats()
    off = _host_loop(*args, defer=0)
    assert off["stats"]["queries_batched"] == s0["queries_batched"]
    on = _host_loop(*args, defer=1)
    assert on["stats"]["queries_batched"] - off["stats"]["queries_batched"] == nq and on["stats"]["batches"] - off["stats"]["batches"] == 1
    ver = _host_loop(*args, defer=2)
    assert ver["stats"]["verify_mismatch"] == off["stats"]["verify_mismatch"]
    split = _host_loop(*args, defer=1, stray=17)
    assert split["stats"]["batches"] - ver["stats"]["batches"] == 2
    for r in (on, ver, split):
        assert r["match"] == off["match"] and r["last_pred"] == off["last_pred"]
        assert r["cost"] == pytest.approx(off["cost"], rel=1e-5)
        np.testing.assert_array_equal(r["last_u"], off["last_u"])
        np.testing.assert_array_equal(r["last_scores"], off["last_scores"])
        np.testing.assert_array_equal(r["last_p"], off["last_p"])


def test_invalidate_forgets_the_cached_model_when_the_host_rewrites_weights(env, gold):
    """The batched model behind the queue is cached on the weight POINTERS.  A host that overwrites the weight values in place
    with its own copies (here: torch) between two forward phases calls qmann_abi_invalidate_model(), which drops the cache: the
    second phase computes with the new values (= the verbs with the queue off) and a new model was built.  qmann_abi_flush()
    alone is a read barrier and keeps the model: a phase after a plain flush builds nothing."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_golden", ROOT / "oracle" / "gen_golden.py")
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    b = gold("babi_qa1_test64.npz")
    V, D, H, nq = int(b["dim_input"]), 60, 3, 32
    cfg = gg.babi_cfg(V, 2, 0)
    w1, w2 = gg.seeded_weights(11, H, D, V, 1.0), gg.seeded_weights(12, H, D, V, 1.0)
    dev = {k: ([env.up(w) for w in v] if isinstance(v, list) else env.up(v)) for k, v in w1.items()}
    args = (env, cfg, None, b["story"].astype(np.float32), b["question"].astype(np.float32), b["answer"].astype(np.float32),
            b["n_sen"].astype(np.int64), nq)
    first = _host_loop(*args, defer=1, dev_wts=dev)
    env.lib.qmann_abi_flush()                                           # drain only: the cached model survives
    again = _host_loop(*args, defer=1, dev_wts=dev)
    assert again["stats"]["models_built"] == first["stats"]["models_built"] and again["match"] == first["match"]
    first = again
    for k, v in w2.items():                                             # the host's own writes: same pointers, new values
        for t, w in zip(dev[k] if isinstance(v, list) else [dev[k]], v if isinstance(v, list) else [v]):
            t.copy_(env.torch.from_numpy(np.ascontiguousarray(w, np.float32)))
    env.torch.cuda.synchronize()
    env.lib.qmann_abi_invalidate_model()
    second = _host_loop(*args, defer=1, dev_wts=dev)
    plain = _host_loop(*args, defer=0, dev_wts=dev)
    assert second["stats"]["models_built"] == first["stats"]["models_built"] + 1
    assert second["stats"]["queries_batched"] - first["stats"]["queries_batched"] == nq
    assert second["match"] == plain["match"] and second["last_pred"] == plain["last_pred"]
    np.testing.assert_array_equal(second["last_u"], plain["last_u"])
    assert not np.array_equal(second["last_u"], first["last_u"])         # (the two weight sets do give different states)


def test_deferred_verbs_become_visible_at_a_flush(env):
    """queue on: a forward verb's result is not there until a synchronising verb or qmann_abi_flush() -- the contract of
    include/qmann_abi.h; a lone verb (no pattern) is simply executed at the flush"""
    lib, up, ptr, empty = env.lib, env.up, env.ptr, env.empty
    a, b, out = up(np.arange(8, dtype=np.float32)), up(np.ones(8, np.float32)), env.torch.zeros(8, device=env.dev)
    lib.qmann_abi_set_defer(1)
    lib.cuda_sum_vec_fwd(ptr(a), ptr(b), ptr(out), 8, False, 0, 0, 3, False)
    env.torch.cuda.synchronize()
    assert float(out.sum()) == 0.0                       # recorded, not launched
    lib.qmann_abi_flush()
    env.torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np.arange(8) + 1.0)
    out.zero_()
    lib.cuda_sum_vec_fwd(ptr(a), ptr(b), ptr(out), 8, False, 0, 0, 3, False)
    host = np.zeros(8, np.float32)
    lib.cuda_copy_dev2host(host.ctypes.data_as(C.c_void_p), ptr(out), 8)        # a synchronising verb drains the queue first
    np.testing.assert_array_equal(host, np.arange(8) + 1.0)
    lib.qmann_abi_set_defer(0)
