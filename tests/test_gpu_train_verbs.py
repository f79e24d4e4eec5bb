"""GPU parity of the backward / weight-update verbs of boundary B against the
oracle's restatement.  Every output element is reduced serially in the reference's order on both
sides, so results are compared bit for bit (the one exception is the gradient norm, which the
reference itself accumulates with a float atomic in arbitrary row order)."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    load_pkg()
    import qmann_amd.abi as abi

    class Env:
        pass
    e = Env()
    e.torch, e.lib = torch, abi.lib
    e.dev = torch.device("cuda:0")
    e.up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(e.dev)
    e.ptr = lambda t: C.c_void_p(t.data_ptr())
    e.zeros = lambda *s: torch.zeros(s, dtype=torch.float32, device=e.dev)
    return e


def rnd(rng, shape, s=1.0):
    return rng.normal(0, s, shape).astype(np.float32)


@pytest.mark.parametrize("f_trans", [False, True])
@pytest.mark.parametrize("f_fixed", [True, False])
@pytest.mark.parametrize("shape", [(10, 60), (50, 60), (1, 7)])
def test_dot_mat_vec_bwd(env, oracle, f_trans, f_fixed, shape):
    r, c = shape
    rng = np.random.default_rng(r * 100 + c + f_trans)
    M, v = rnd(rng, (r, c), 2.0), rnd(rng, r if f_trans else c, 2.0)
    gi = rnd(rng, c if f_trans else r, 0.3)
    dM, dv, dg = env.up(M), env.up(v), env.up(gi)
    gm, gv = env.zeros(r, c), env.zeros(r if f_trans else c)
    env.lib.cuda_dot_mat_vec_bwd(env.ptr(dM), env.ptr(dv), env.ptr(dg), env.ptr(gm), env.ptr(gv), None, r, c, f_trans,
                                 f_fixed, 5, 2, 5, 2, 3, False)
    om, ov = oracle.dot_mat_vec_bwd(M, v, gi, f_trans, f_fixed, (5, 2))
    np.testing.assert_array_equal(gm.cpu().numpy(), om)
    np.testing.assert_array_equal(gv.cpu().numpy(), ov)


@pytest.mark.parametrize("iwl", [5, 2])
def test_dot_mat_vec_bwd_appx_surrogate_gradients(env, oracle, iwl):
    rng = np.random.default_rng(iwl)
    r, c = 23, 60
    M, v, gi = rnd(rng, (r, c), 4.0), rnd(rng, c, 4.0), rnd(rng, r, 0.3)
    M.ravel()[::9] = np.float32(2.0 ** iwl + 1.0); v[::11] = np.float32(-1e-4)
    dM, dv, dg = env.up(M), env.up(v), env.up(gi)
    gm, gv = env.zeros(r, c), env.zeros(c)
    env.lib.cuda_dot_mat_vec_bwd_appx(env.ptr(dM), env.ptr(dv), env.ptr(dg), env.ptr(gm), env.ptr(gv), None, None, r, c,
                                      True, iwl, 7 - iwl, 3, 8, False, False, 1)
    om, ov = oracle.dot_mat_vec_bwd(M, v, gi, False, True, (iwl, 7 - iwl), appx_bits=8)
    np.testing.assert_array_equal(gm.cpu().numpy(), om)
    np.testing.assert_array_equal(gv.cpu().numpy(), ov)


@pytest.mark.parametrize("dim", [1, 10, 50, 300])
def test_softmax_bwd(env, oracle, dim):
    rng = np.random.default_rng(dim)
    y = rng.random(dim).astype(np.float32); y /= y.sum()
    gi = rnd(rng, dim)
    dy, dg, go = env.up(y), env.up(gi), env.zeros(dim)
    for sb in (False, True):
        env.lib.cuda_softmax_bwd(env.ptr(dg), env.ptr(dy), env.ptr(go), None, dim, sb, False)
        np.testing.assert_array_equal(go.cpu().numpy(), oracle.softmax_bwd(y, gi, sb))


@pytest.mark.parametrize("dims", [(30, 60), (60, 60), (60, 30)])
def test_dense_bwd_and_w_up(env, oracle, dims):
    dim_in, dim_out = dims
    rng = np.random.default_rng(dim_in + 7 * dim_out)
    w, x, gi = rnd(rng, (dim_out, dim_in)), rnd(rng, dim_in), rnd(rng, dim_out, 0.5)
    w_del0 = rnd(rng, (dim_out, dim_in), 0.2)
    out = rnd(rng, dim_out)
    dw, dwd, dx, dout, dgi, dgo = env.up(w), env.up(w_del0), env.up(x), env.up(out), env.up(gi), env.zeros(dim_in)
    env.lib.cuda_dense_bwd(env.ptr(dw), env.ptr(dwd), None, None, env.ptr(dx), env.ptr(dout), env.ptr(dgi), env.ptr(dgo),
                           None, dim_in, dim_out, b"NULL", True, 5, 2, 5, 2, 3, False)
    owd, ogo, _ = oracle.dense_bwd(w, w_del0, x, out, gi, True, (5, 2))
    np.testing.assert_array_equal(dwd.cpu().numpy(), owd)
    np.testing.assert_array_equal(dgo.cpu().numpy(), ogo)
    # weight update, with and without clipping, fixed and float
    for max_norm in (40.0, 0.5):
        for f_fixed in (True, False):
            dw2, dwd2, dn = env.up(w), env.up(owd), env.zeros(1)
            lr, lam, mx = (C.c_float(0.3), C.c_float(0.0), C.c_float(max_norm))
            env.lib.cuda_dense_w_up(env.ptr(dw2), env.ptr(dwd2), None, None, env.ptr(dn), None, dim_in, dim_out, 32,
                                    C.byref(lr), C.byref(lam), C.byref(mx), f_fixed, 2, 5, 3, False)
            ow, owd2, onorm = oracle.mat_w_up(w, owd, 32, 0.3, 0.0, max_norm, f_fixed, (2, 5))
            gnorm = float(dn.cpu().numpy()[0])
            assert gnorm == pytest.approx(onorm, rel=1e-6)              # float atomic: row order is arbitrary
            if abs(gnorm - onorm) == 0.0 or gnorm <= max_norm:
                np.testing.assert_array_equal(dw2.cpu().numpy(), ow)
            else:
                np.testing.assert_allclose(dw2.cpu().numpy(), ow, rtol=1e-6, atol=2.0 ** -5)
            assert not dwd2.cpu().numpy().any()


def test_dense_mat_bwd(env, oracle):
    rng = np.random.default_rng(3)
    dim_in, dim_out, dim_len = 30, 60, 9
    X, w, gi = rnd(rng, (dim_len, dim_in)), rnd(rng, (dim_out, dim_in)), rnd(rng, (dim_len, dim_out), 0.3)
    w_del0 = rnd(rng, (dim_out, dim_in), 0.2)
    for f_fixed in (True, False):
        dX, dw, dwd, dgi, dgo = env.up(X), env.up(w), env.up(w_del0), env.up(gi), env.zeros(dim_len, dim_in)
        env.lib.cuda_dense_mat_bwd(env.ptr(dX), env.ptr(dw), env.ptr(dwd), None, None, env.ptr(dgi), env.ptr(dgo), None,
                                   dim_in, dim_out, dim_len, f_fixed, 5, 2, 3, False)
        owd, ogo = oracle.dense_mat_bwd(X, w, w_del0, gi, f_fixed, (5, 2))
        np.testing.assert_array_equal(dwd.cpu().numpy(), owd)
        np.testing.assert_array_equal(dgo.cpu().numpy(), ogo)


def test_dup_grad_and_sum_vec_bwd(env, oracle):
    rng = np.random.default_rng(4)
    a, b = rnd(rng, 60, 0.4), rnd(rng, 60, 0.4)
    da, db, do = env.up(a), env.up(b), env.zeros(60)
    env.lib.cuda_dup_grad_bwd(env.ptr(do), env.ptr(da), env.ptr(db), None, 60, True, 5, 2, 3)
    np.testing.assert_array_equal(do.cpu().numpy(), oracle.dup_grad_bwd(a, b, True, (5, 2)))
    env.lib.cuda_sum_vec_bwd(env.ptr(do), env.ptr(da), None, None, 60)
    np.testing.assert_array_equal(do.cpu().numpy(), a)


@pytest.mark.parametrize("seed", range(20))
def test_backward_verbs_random_shapes_and_formats(env, oracle, seed):
    """dot_mat_vec_bwd (both orientations), dense_bwd, dense_mat_bwd with random shapes and word lengths 2..8
    (the gradient format is Q(1.wl-2), so the word length matters)."""
    rng = np.random.default_rng(9500 + seed)
    wl = int(rng.integers(2, 8)); iwl = int(rng.integers(0, wl + 1)); fmt = (iwl, wl - iwl)
    r, c = int(rng.integers(1, 90)), int(rng.integers(1, 130))
    for f_trans in (False, True):
        M, v = rnd(rng, (r, c), 2.0), rnd(rng, r if f_trans else c, 2.0)
        gi = rnd(rng, c if f_trans else r, 0.3)
        dM, dv, dg = env.up(M), env.up(v), env.up(gi)
        gm, gv = env.zeros(r, c), env.zeros(r if f_trans else c)
        env.lib.cuda_dot_mat_vec_bwd(env.ptr(dM), env.ptr(dv), env.ptr(dg), env.ptr(gm), env.ptr(gv), None, r, c, f_trans,
                                     True, fmt[0], fmt[1], fmt[0], fmt[1], 3, False)
        om, ov = oracle.dot_mat_vec_bwd(M, v, gi, f_trans, True, fmt)
        np.testing.assert_array_equal(gm.cpu().numpy(), om, err_msg=f"grad mat trans={f_trans}")
        np.testing.assert_array_equal(gv.cpu().numpy(), ov, err_msg=f"grad vec trans={f_trans}")
    dim_in, dim_out = int(rng.integers(1, 120)), int(rng.integers(1, 120))
    w, x, gi = rnd(rng, (dim_out, dim_in)), rnd(rng, dim_in), rnd(rng, dim_out, 0.5)
    w_del0, out = rnd(rng, (dim_out, dim_in), 0.2), rnd(rng, dim_out)
    dw, dwd, dx, dout, dgi, dgo = env.up(w), env.up(w_del0), env.up(x), env.up(out), env.up(gi), env.zeros(dim_in)
    env.lib.cuda_dense_bwd(env.ptr(dw), env.ptr(dwd), None, None, env.ptr(dx), env.ptr(dout), env.ptr(dgi), env.ptr(dgo),
                           None, dim_in, dim_out, b"NULL", True, fmt[0], fmt[1], fmt[0], fmt[1], 3, False)
    owd, ogo, _ = oracle.dense_bwd(w, w_del0, x, out, gi, True, fmt)
    np.testing.assert_array_equal(dwd.cpu().numpy(), owd, err_msg="dense w_del")
    np.testing.assert_array_equal(dgo.cpu().numpy(), ogo, err_msg="dense grad_out")
