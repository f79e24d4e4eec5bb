"""The float answer layer on the bf16 matrix cores (csrc/batch_io.hip::k_answer_mfma: the default of qmann_answer_f32 at the bAbI
shapes) against the serial-order kernels (QMANN_ANSWER_EXACT: bit-equal to the reference's loop, lib/layer_cuda.cu:70-80) and
against a float64 softmax of the exact logits.  north_star grants the float softmax 1e-5; the criteria are those of
test_gpu_batch.py::run_case: probabilities within rtol 1e-5 / atol 1e-7, predictions equal wherever the top-2 gap exceeds 1e-6."""
import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    load_pkg()
    import qmann_amd.model as model

    class Env:
        pass
    e = Env()
    e.torch, e.model, e.dev = torch, model, torch.device("cuda:0")
    return e


def make(env, D, V, B, seed, base=0, sigma_u=20.0, sigma_w=0.1, frac=2):
    rng = np.random.default_rng(seed)
    fmt = [(7 - frac, frac)] * 3
    cfg = dict(n_hop=3, dim_emb=D, dim_input=V, attention_mode=2, softmax_variant=base, f_fixed=True, en_lin_map=True,
               fmt=fmt, fmt_w=list(fmt), fmt_att=list(fmt), fmt_bin=fmt[0])
    wts = {"w_h": [np.zeros((D, D), np.float32) for _ in range(3)], "w_ans": rng.normal(0, sigma_w, (V, D)).astype(np.float32)}
    net = env.model.QNet(cfg, wts, device="cuda:0")
    u = (np.clip(np.rint(rng.normal(0, sigma_u, (B, D))), -127, 127) / (1 << frac)).astype(np.float32)   # on the activation grid
    y = rng.integers(0, V, B).astype(np.int32)
    y[::7] = V + 3                                               # labels outside the dictionary count for nothing
    return net, wts["w_ans"], u, y


def run(env, monkeypatch, net, u, y, exact):
    torch = env.torch
    if exact:
        monkeypatch.setenv("QMANN_ANSWER_EXACT", "1")
    else:
        monkeypatch.delenv("QMANN_ANSWER_EXACT", raising=False)
    env.model.abi.lib.qmann_tuning_reload()
    du = torch.from_numpy(u).to(env.dev); dy = torch.from_numpy(y).to(env.dev)
    pred, probs, cost, match = net.answer(du, dy, want_probs=True)
    pred2, _, cost2, match2 = net.answer(du, dy, want_probs=False)              # the production call: no probabilities
    torch.cuda.synchronize()
    assert torch.equal(pred, pred2) and int(match.item()) == int(match2.item())
    assert float(cost2.item()) == pytest.approx(float(cost.item()), rel=1e-5, abs=1e-6)
    monkeypatch.delenv("QMANN_ANSWER_EXACT", raising=False)
    env.model.abi.lib.qmann_tuning_reload()
    return pred.cpu().numpy(), probs.cpu().numpy(), float(cost.item()), int(match.item())


def exact_softmax(w, u, base):
    l = u.astype(np.float64) @ w.astype(np.float64).T
    z = l - l.max(1, keepdims=True)
    e = np.exp(z) if base == 0 else np.exp2(z)
    return e / e.sum(1, keepdims=True), l


@pytest.mark.parametrize("V", [1, 2, 15, 16, 17, 30, 31, 32, 33, 48, 64, 80, 100, 129, 192, 238, 240, 255, 256])
@pytest.mark.parametrize("D", [60, 64, 17])
def test_fused_answer_layer_equals_the_serial_one_within_tolerance(env, monkeypatch, V, D):
    B = 1003
    for base in (0, 1):
        net, w, u, y = make(env, D, V, B, seed=V * 7 + D + base, base=base)
        pf, qf, cf, mf = run(env, monkeypatch, net, u, y, exact=False)
        pe, qe, ce, me = run(env, monkeypatch, net, u, y, exact=True)
        np.testing.assert_allclose(qf, qe, rtol=1e-5, atol=1e-7)
        ref, _ = exact_softmax(w, u, base)
        np.testing.assert_allclose(qf, ref, rtol=1e-5, atol=1e-7)            # (the fused form is the closer of the two to exact arithmetic)
        top2 = np.sort(qe, axis=1)[:, -2:] if V > 1 else np.stack([np.zeros(B), np.ones(B)], 1)
        clear = (top2[:, 1] - top2[:, 0]) > 1e-6
        assert np.array_equal(pf[clear], pe[clear])
        assert clear.mean() > 0.9 or V == 1
        assert cf == pytest.approx(ce, rel=2e-5, abs=1e-5)
        valid = y < V
        assert mf == int((pf[valid] == y[valid]).sum()) and me == int((pe[valid] == y[valid]).sum())
        assert abs(mf - me) <= int((~clear).sum())
        np.testing.assert_allclose(qf.sum(1), 1.0, rtol=1e-5)


def test_fused_answer_layer_ties_go_to_the_highest_index(env, monkeypatch):
    """equal logits <=> equal probabilities: duplicated answer rows tie exactly; lib/layer_cuda.cu:1918-1939 takes the highest index"""
    D, V, B = 60, 238, 500
    net, w, u, y = make(env, D, V, B, seed=5)
    w2 = w.copy()
    w2[200] = w2[7]; w2[237] = w2[100]; w2[16] = w2[15]
    wts = {"w_h": [np.zeros((D, D), np.float32) for _ in range(3)], "w_ans": w2}
    cfg = dict(n_hop=3, dim_emb=D, dim_input=V, attention_mode=2, softmax_variant=0, f_fixed=True, en_lin_map=True,
               fmt=[(5, 2)] * 3, fmt_w=[(5, 2)] * 3, fmt_att=[(5, 2)] * 3, fmt_bin=(5, 2))
    net = env.model.QNet(cfg, wts, device="cuda:0")
    pf, qf, _, _ = run(env, monkeypatch, net, u, y, exact=False)
    pe, qe, _, _ = run(env, monkeypatch, net, u, y, exact=True)
    assert not np.isin(pf, [7, 100, 15]).any() and not np.isin(pe, [7, 100, 15]).any()
    assert np.isin(pf, [200, 237, 16]).sum() > 0                              # (the duplicated rows do win sometimes)
    top2 = np.sort(qe, axis=1)[:, -3:]
    clear = (top2[:, 2] - top2[:, 0]) > 1e-6                                  # the best and the third differ: only the exact tie remains
    assert np.array_equal(pf[clear], pe[clear])


@pytest.mark.parametrize("sigma_w,sigma_u", [(0.3, 40.0), (1.0, 60.0)])
def test_fused_answer_layer_large_logits_stay_within_1e_5_absolute(env, monkeypatch, sigma_w, sigma_u):
    """logits of magnitude 50 .. 300: a unit in the last place of a logit is 4e-6 .. 3e-5, so two correct float evaluations of the
    same softmax differ by more than 1e-5 RELATIVE; both stay within 1e-5 ABSOLUTE of the float64 softmax of the exact logits
    (north_star's tolerance on the probabilities), and the fused form is the closer one."""
    D, V, B = 60, 238, 1000
    net, w, u, y = make(env, D, V, B, seed=9, sigma_u=sigma_u, sigma_w=sigma_w)
    pf, qf, _, _ = run(env, monkeypatch, net, u, y, exact=False)
    pe, qe, _, _ = run(env, monkeypatch, net, u, y, exact=True)
    ref, l = exact_softmax(w, u, 0)
    assert np.abs(l).max() > 50
    err_f, err_e = np.abs(qf - ref).max(), np.abs(qe - ref).max()
    print(f"max |logit| {np.abs(l).max():.0f}: fused {err_f:.2e}, serial {err_e:.2e} absolute error of the probabilities")
    assert err_f <= 1e-5 and err_f <= err_e * 1.5 + 1e-7
    top2 = np.sort(ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    assert np.array_equal(pf[clear], pe[clear])


def test_fused_answer_layer_many_queries(env, monkeypatch):
    net, w, u, y = make(env, 60, 30, 200003, seed=11)
    pf, qf, cf, mf = run(env, monkeypatch, net, u, y, exact=False)
    pe, qe, ce, me = run(env, monkeypatch, net, u, y, exact=True)
    np.testing.assert_allclose(qf, qe, rtol=1e-5, atol=1e-7)
    top2 = np.sort(qe, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-6
    assert np.array_equal(pf[clear], pe[clear]) and abs(mf - me) <= int((~clear).sum())
    assert cf == pytest.approx(ce, rel=1e-4)
