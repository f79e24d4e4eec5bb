"""Edge cases and caller errors of the batched C-ABI: empty batches, stories without sentences, a single query, bounds that
the kernels cannot serve, null pointers -- a caller error must come back as a QMANN_E* code with nothing launched, never as
a fault or an exit.  (The reference has no such tests: its host sizes everything from define.h; these are the cases a
library call has to survive.)"""
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
    import qmann_amd.model as model

    class Env:
        pass
    e = Env()
    e.torch, e.model, e.abi, e.dev = torch, model, abi, torch.device("cuda:0")
    return e


def weights(seed, H, D, V):
    rng = np.random.default_rng(seed)
    return {"w_q": rng.normal(0, 1.0, (D, V)).astype(np.float32),
            "w_a": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
            "w_c": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
            "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
            "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


MODES = [(1, 8), (2, 8), (3, 8), (10, 4), (11, 4)]


@pytest.mark.parametrize("mode,nb", MODES)
def test_stories_without_sentences(env, mode, nb):
    """every story empty: o = 0 in every hop, the state is driven by the linear map alone -- equal to the general chain of
    kernels, in every attention mode (short-memory and streaming kernels alike see S = 0)"""
    torch, model = env.torch, env.model
    V, D, B = 30, 60, 7
    cfg = model.babi_cfg(V, attention_mode=mode, D=D, en_mq=False)
    cfg["num_bit"] = nb
    wts = weights(mode, 3, D, V)
    qw = np.full((B, 8), 0xFFFF, np.uint16)
    qw[:, 0] = np.arange(B) % V
    sw = np.full((1, 8), 0xFFFF, np.uint16)                          # (one dummy row: rows_total = 0 below)
    row_off = torch.zeros(B + 1, dtype=torch.int32, device=env.dev)
    hm = model.HostModel(cfg, wts)
    d_qw = torch.from_numpy(qw.view(np.int16)).to(env.dev)
    d_sw = torch.from_numpy(sw.view(np.int16)).to(env.dev)[:0]
    for max_slots in (1, 50, 300):                                   # lean / small / streaming kernels
        pred, _, _ = hm.forward_words(d_sw, d_qw, row_off, max_slots)
        torch.cuda.synchronize()
        u = hm.last_u(B).cpu().numpy()
        assert np.isfinite(u).all()
        if max_slots == 1:
            u_first, p_first = u, pred.cpu().numpy()
        else:
            np.testing.assert_array_equal(u, u_first)
            np.testing.assert_array_equal(pred.cpu().numpy(), p_first)
    hm.close()
    # the same through the op-by-op chain
    net = model.QNet(cfg, wts)
    ques = np.zeros((B, V), np.float32)
    ques[np.arange(B), qw[:, 0]] = 1.0
    keys, vals, u0 = net.embed(torch.zeros((1, V), device=env.dev), torch.from_numpy(ques).to(env.dev))
    uc = net.hops(keys, vals, row_off, 1, u0, taps=True)[0]
    torch.cuda.synchronize()
    np.testing.assert_array_equal(u_first, uc.cpu().numpy())


def test_empty_batch_and_single_query(env):
    torch, model = env.torch, env.model
    V, D = 30, 60
    cfg = model.babi_cfg(V, attention_mode=2, D=D)
    hm = model.HostModel(cfg, weights(1, 3, D, V))
    sw = torch.from_numpy(np.array([[3, 4, 20, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF]], np.uint16).view(np.int16)).to(env.dev)
    qw = torch.from_numpy(np.array([[3, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF]], np.uint16).view(np.int16)).to(env.dev)
    ro = torch.tensor([0, 1], dtype=torch.int32, device=env.dev)
    pred, _, _ = hm.forward_words(sw, qw, ro, 1)
    torch.cuda.synchronize()
    assert 0 <= int(pred[0]) < V
    pred0, _, _ = hm.forward_words(sw[:0], qw[:0], ro[:1], 1)       # n_query = 0: returns at once
    assert pred0.numel() == 0
    hm.close()


def test_caller_errors_come_back_as_codes(env):
    torch, model, abi = env.torch, env.model, env.abi
    V, D, B, S = 30, 60, 4, 5
    cfg = model.babi_cfg(V, attention_mode=2, D=D)
    net = model.QNet(cfg, weights(2, 3, D, V))
    rows = B * S
    keys = torch.zeros((3, rows, 64), dtype=torch.int8, device=env.dev)
    u0 = torch.zeros((B, D), device=env.dev)
    u1 = torch.zeros_like(u0)
    ro = (torch.arange(B + 1, dtype=torch.int32, device=env.dev) * S)
    p = lambda t: C.c_void_p(t.data_ptr())
    call = lambda net_, k, v, stride, r, ms, a, b, n: abi.lib.qmann_hops_i8(C.byref(net_), k, v, stride, r, ms, a, b, None, n, None)
    ok = call(net.net, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), B)
    assert ok == 0
    assert call(net.net, None, p(keys), rows * 64, p(ro), S, p(u0), p(u1), B) == abi.QMANN_EINVAL          # null keys
    assert call(net.net, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), 1 << 24) == abi.QMANN_ERANGE  # a launch holds < 2^32 threads
    assert call(net.net, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), 0) == 0                        # empty batch
    assert call(net.net, p(keys), p(keys), rows * 64, p(ro), 200000, p(u0), p(u1), B) == abi.QMANN_ERANGE   # scores beyond the LDS
    bad = abi.Net.from_buffer_copy(bytes(net.net))
    bad.dim_emb_pad = 96
    assert call(bad, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), B) == abi.QMANN_EUNSUPPORTED
    bad = abi.Net.from_buffer_copy(bytes(net.net))
    bad.n_hop = 0
    assert call(bad, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), B) == abi.QMANN_EINVAL
    bad = abi.Net.from_buffer_copy(bytes(net.net))
    bad.dim_emb = 65
    assert call(bad, p(keys), p(keys), rows * 64, p(ro), S, p(u0), p(u1), B) == abi.QMANN_EINVAL
    # taps need distinct hop planes
    taps = abi.Taps(0, 0, 0, u0.data_ptr(), 0)
    rc = abi.lib.qmann_hops_i8(C.byref(net.net), p(keys), p(keys), 0, p(ro), S, p(u0), p(u1), C.byref(taps), B, None)
    assert rc == abi.QMANN_EINVAL
    # qmann_pack_bitplanes reads 16 bytes per lane and stores 8: a misaligned array is refused, not faulted on
    planes = torch.zeros((rows, 1, 8), dtype=torch.int64, device=env.dev)
    pack = lambda k, pl, nb: abi.lib.qmann_pack_bitplanes(k, pl, rows, 64, nb, None)
    assert pack(p(keys), p(planes), 8) == 0
    assert pack(C.c_void_p(keys.data_ptr() + 1), p(planes), 8) == abi.QMANN_EINVAL
    assert pack(p(keys), C.c_void_p(planes.data_ptr() + 4), 8) == abi.QMANN_EINVAL
    assert pack(p(keys), p(planes), 9) == abi.QMANN_EINVAL and pack(None, p(planes), 8) == abi.QMANN_EINVAL
    torch.cuda.synchronize()


def test_workspace_regrows_between_batches(env):
    """the model object keeps a workspace that grows on demand: a small batch, a larger one, a small one again, on a
    side stream -- every query's result equals the one-shot run's"""
    torch, model = env.torch, env.model
    rng = np.random.default_rng(12)
    V, D, B = 40, 60, 600
    cfg = model.babi_cfg(V, attention_mode=2, D=D)
    wts = weights(3, 3, D, V)
    n_sen = rng.integers(0, 12, B)
    rows = int(n_sen.sum())
    sw = np.full((rows, 8), 0xFFFF, np.uint16)
    for r in range(rows):
        n = int(rng.integers(1, 7))
        sw[r, :n] = rng.integers(0, V - 10, n)
        sw[r, n] = V - 10 + int(rng.integers(0, 10))                 # time entry
    qw = np.full((B, 8), 0xFFFF, np.uint16)
    qw[:, :2] = rng.integers(0, V - 10, (B, 2))
    off = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int64)
    dev = env.dev

    def run(hm, q0, q1):
        ro = torch.from_numpy((off[q0:q1 + 1] - off[q0]).astype(np.int32)).to(dev)
        s = torch.from_numpy(np.ascontiguousarray(sw[off[q0]:off[q1]]).view(np.int16)).to(dev)
        q = torch.from_numpy(np.ascontiguousarray(qw[q0:q1]).view(np.int16)).to(dev)
        pred, _, _ = hm.forward_words(s, q, ro, 16)
        torch.cuda.synchronize()
        return pred.cpu().numpy(), hm.last_u(q1 - q0).cpu().numpy()

    hm = model.HostModel(cfg, wts)
    p_all, u_all = run(hm, 0, B)
    hm.close()
    side = torch.cuda.Stream()
    hm = model.HostModel(cfg, wts, stream=side.cuda_stream)
    for q0, q1 in ((0, 5), (5, 400), (400, 403), (403, B)):
        p, u = run(hm, q0, q1)
        np.testing.assert_array_equal(p, p_all[q0:q1])
        np.testing.assert_array_equal(u, u_all[q0:q1])
    hm.close()


def test_stale_hip_error_of_the_host_is_not_blamed_on_the_next_call(env):
    """A HIP error the host application left behind on this thread (here: hipSetDevice on a device that does not exist)
    must not make the next, valid qmann_* call return QMANN_EHIP: the launch checks consume errors (hipGetLastError) and a
    batched entry point starts from a clean state (csrc/rt.h)."""
    torch, abi = env.torch, env.abi
    hip = C.CDLL("libamdhip64.so")
    src = torch.linspace(-3, 3, 64 * 8, device=env.dev).reshape(8, 64).contiguous()
    dst = torch.empty((8, 64), dtype=torch.int8, device=env.dev)

    def call():
        return abi.lib.qmann_quantize_i8(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 8, 64, 64, abi.Fmt(5, 2), 1, None)
    assert call() == 0
    want = dst.cpu().numpy().copy()
    assert hip.hipSetDevice(1234) != 0                               # leaves a "last error" behind
    assert call() == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dst.cpu().numpy(), want)


def test_failed_workspace_allocation_returns_ehip_and_the_model_recovers(env):
    """qmann_model: a workspace that cannot be allocated (4 G story rows) comes back as QMANN_EHIP -- not as a later
    QMANN_EINVAL from a null pointer --, no capacity is recorded for it, and the next valid call on the same object
    allocates afresh and gives the usual result."""
    torch, model, abi = env.torch, env.model, env.abi
    V, D, B = 30, 60, 5
    cfg = model.babi_cfg(V, attention_mode=2, D=D)
    hm = model.HostModel(cfg, weights(3, 3, D, V))
    rng = np.random.default_rng(5)
    sw = np.full((B * 4, 8), 0xFFFF, np.uint16); sw[:, :3] = rng.integers(1, 20, (B * 4, 3)); sw[:, 3] = 20 + np.arange(B * 4) % 4
    qw = np.full((B, 8), 0xFFFF, np.uint16); qw[:, :2] = rng.integers(1, 20, (B, 2))
    d_sw = torch.from_numpy(sw.view(np.int16)).to(env.dev); d_qw = torch.from_numpy(qw.view(np.int16)).to(env.dev)
    ro = torch.arange(0, B * 4 + 1, 4, dtype=torch.int32, device=env.dev)
    pred0, _, _ = hm.forward_words(d_sw, d_qw, ro, 4)
    torch.cuda.synchronize()
    pred = torch.empty(B, dtype=torch.int32, device=env.dev)
    rc = abi.lib.qmann_model_forward_words(hm.h, C.c_void_p(d_sw.data_ptr()), 0xFFFFFFF0, 8, C.c_void_p(d_qw.data_ptr()), 8,
                                           C.c_void_p(ro.data_ptr()), 4, B, None, C.c_void_p(pred.data_ptr()), None, None, None)
    assert rc == -5, rc                                              # QMANN_EHIP
    pred1, _, _ = hm.forward_words(d_sw, d_qw, ro, 4)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pred1.cpu().numpy(), pred0.cpu().numpy())
    hm.close()
