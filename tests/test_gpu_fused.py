"""The one-kernel forward for bAbI-sized queries (csrc/fwd_lean.hip: word indices -> embedding in LDS -> hops -> answer)
against the staged pipeline of the same library (embed kernels -> hop kernel -> answer kernel, each checked against the
oracle elsewhere): predictions, the final hop state, match counts and cost must agree -- bit for bit where integers or
per-query floats are concerned.  Plus the FULL real-data sets of BASELINE configs 2 and 3 (1 000 qa1 test stories, the
20 000-story joint set; fixtures made by the reference's sample.c) with a spread sample checked against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLD, load_pkg

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


def weights(seed, H, D, V, sigma=1.0):
    rng = np.random.default_rng(seed)
    return {"w_q": rng.normal(0, sigma, (D, V)).astype(np.float32),
            "w_a": [rng.normal(0, sigma, (D, V)).astype(np.float32) for _ in range(H)],
            "w_c": [rng.normal(0, sigma, (D, V)).astype(np.float32) for _ in range(H)],
            "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
            "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


def run_both(env, cfg, wts, sw, qw, n_sen, ans=None, max_slots=None):
    """fused and staged forward of the same batch through the host model; returns the fused outputs"""
    torch, model = env.torch, env.model
    B = len(n_sen)
    row_off = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int32)
    d_sw = torch.from_numpy(np.ascontiguousarray(sw).view(np.int16)).to(env.dev)
    d_qw = torch.from_numpy(np.ascontiguousarray(qw).view(np.int16)).to(env.dev)
    d_ro = torch.from_numpy(row_off).to(env.dev)
    d_ans = torch.from_numpy(ans.astype(np.int32)).to(env.dev) if ans is not None else None
    ms = int(max_slots if max_slots is not None else max(int(n_sen.max()), 1))
    hm = model.HostModel(cfg, wts, device="cuda:0")
    out = {}
    for name in ("fused", "staged"):
        if name == "staged":
            os.environ["QMANN_NO_FUSED"] = "1"
        try:
            pred, cost, match = hm.forward_words(d_sw, d_qw, d_ro, ms, d_ans)
            torch.cuda.synchronize()
            out[name] = (pred.cpu().numpy(), hm.last_u(B).cpu().numpy(),
                         None if cost is None else float(cost.item()), None if match is None else int(match.item()))
        finally:
            os.environ.pop("QMANN_NO_FUSED", None)
    hm.close()
    pf, uf, cf, mf = out["fused"]
    ps, us, cs, ms_ = out["staged"]
    bad = np.flatnonzero((uf != us).any(1))
    assert bad.size == 0, f"final hop state differs for {bad.size} of {B} queries, first {bad[:5]} (slots {n_sen[bad[:5]]})"
    assert np.array_equal(pf, ps)
    if ans is not None:
        assert mf == ms_
        assert cf == pytest.approx(cs, rel=1e-4, abs=1e-3)          # one float atomic per wavefront: the order of the adds differs
    assert np.abs(us).sum() > 0
    return pf, uf


def words16(a8, W):
    out = np.full((a8.shape[0], W), 0xFFFF, np.uint16)
    out[:, :a8.shape[1]] = np.where(a8 == 0xFF, 0xFFFF, a8.astype(np.uint16))
    return out


def words_to_bow(words, V, with_time):
    out = np.zeros((words.shape[0], V), np.float32)
    for r, row in enumerate(words):
        ent = [int(w) for w in row if w != 0xFFFF]
        if with_time and ent:
            t = ent.pop()
            for w in ent:
                out[r, w] += 1.0
            out[r, t] = 1.0
        else:
            for w in ent:
                out[r, w] += 1.0
    return out


def oracle_sample(oracle, cfg, wts, sw, qw, n_sen, pred, u, pick):
    """spread sample against the oracle (its own forward from bag-of-words rows); a final state may differ only where a
    softmax weight of the oracle sits within 1e-5 of a truncation step (test_gpu_batch.py explains)"""
    m = oracle.make_model(cfg, wts)
    V, H = cfg["dim_input"], cfg["n_hop"]
    offs = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int64)
    excused = 0
    for i in pick:
        st = words_to_bow(sw[offs[i]:offs[i + 1]], V, True)
        qu = words_to_bow(qw[i:i + 1], V, False)[0]
        op, t = oracle.forward(m, st, qu, taps=("u", "probs", "out_probs"))
        if np.array_equal(u[i], t["u"][H - 1]):
            top2 = np.sort(t["out_probs"])[-2:]
            if top2[1] - top2[0] > 1e-6:
                assert int(pred[i]) == op, f"prediction of story {i}"
            continue
        near = False
        for h in range(H):
            x = t["probs"][h].astype(np.float64) * (1 << cfg["fmt"][h][1])
            k = np.rint(x)
            near |= bool(((np.abs(x - k) <= 1e-5 * np.maximum(1.0, np.abs(x))) & (k > 0)).any())
        assert near, f"story {i}: final state differs from the oracle's"
        excused += 1
    assert excused <= max(2, len(pick) // 8), f"{excused} of {len(pick)} stories hit the p-on-a-step exclusion"


@pytest.mark.parametrize("mode,nb", [(2, 8), (3, 8), (10, 8), (11, 8)])
def test_full_qa1_test_set(env, oracle, mode, nb):
    """BASELINE config 2 on all 1 000 qa1 test stories (config 2 proper is mode 2; the other score modes ride along)"""
    g = np.load(GOLD / "babi_qa1_test1000_words.npz")
    sw, qw = words16(g["story_words"], 8), words16(g["question_words"], 8)
    n_sen, ans = g["n_sen"].astype(np.int64), g["answer"].astype(np.int64)
    V = int(g["dim_input"])
    cfg = env.model.babi_cfg(V, attention_mode=mode, en_mq=(mode == 2))
    cfg["num_bit"] = nb
    wts = weights(11, 3, 60, V)
    pred, u = run_both(env, cfg, wts, sw, qw, n_sen, ans)
    oracle_sample(oracle, cfg, wts, sw, qw, n_sen, pred, u, list(range(0, 1000, 25)))


@pytest.mark.parametrize("mode,nb", [(11, 8), (10, 8), (3, 8), (2, 8)])
def test_full_joint_test_set(env, oracle, mode, nb):
    """BASELINE config 3 on the whole 20-task joint test set (20 000 stories, 2..64 sentences, 238 inputs)"""
    g = np.load(GOLD / "babi_joint20_test20000_words.npz")
    sw, qw = words16(g["story_words"], 16), words16(g["question_words"], 16)
    n_sen, ans = g["n_sen"].astype(np.int64), g["answer"].astype(np.int64)
    ans = np.where(ans == 0xFF, 0xFFFF, ans)                     # no label (the word is not in the dictionary)
    V = int(g["dim_input"])
    cfg = env.model.babi_cfg(V, attention_mode=mode, en_mq=False)
    cfg["num_bit"] = nb
    wts = weights(12, 3, 60, V)
    pred, u = run_both(env, cfg, wts, sw, qw, n_sen, ans)
    oracle_sample(oracle, cfg, wts, sw, qw, n_sen, pred, u, list(range(7, 20000, 400)))


def random_stories(rng, B, V, dd, W, S_list, dup_every=3):
    n_sen = np.array([S_list[i % len(S_list)] for i in range(B)], np.int64)
    rng.shuffle(n_sen)
    rows = int(n_sen.sum())
    sw = np.full((max(rows, 1), W), 0xFFFF, np.uint16)
    for r in range(rows):
        n = int(rng.integers(0, W))                              # 0 .. W-1 words, then the time entry
        ws = rng.integers(0, dd if r % dup_every else min(dd, 4), n)        # every few rows: a tiny vocabulary -> repeated words
        t = dd + int(rng.integers(0, V - dd))
        ent = list(ws) + [t]
        if r % 17 == 5:
            ent[0] = V + 3                                       # out-of-range word: ignored
        if r % 29 == 7 and len(ent) > 1:
            ent[-2] = t                                          # a word slot equal to the time entry
        sw[r, :len(ent)] = ent
    qw = np.full((B, 8), 0xFFFF, np.uint16)
    for q in range(B):
        ws = rng.integers(0, dd if q % 4 else 3, int(rng.integers(0, 8)))
        qw[q, :len(ws)] = ws
    return sw, qw, n_sen


@pytest.mark.parametrize("mode,nb", [(2, 8), (3, 8), (10, 2), (11, 4)])
@pytest.mark.parametrize("V,D,W", [(30, 60, 8), (238, 60, 16), (70, 64, 11), (500, 20, 5)])
def test_random_word_lists(env, mode, nb, V, D, W):
    """ragged stories (0..64 sentences), repeated words, out-of-range words, empty rows; small and large dictionaries
    (tables in LDS or in L2), answer layer inside the kernel or as its own launch"""
    rng = np.random.default_rng(V * 7 + D + mode)
    sw, qw, n_sen = random_stories(rng, 400, V, V - 12, W, [0, 1, 2, 3, 4, 5, 9, 16, 17, 40, 64])
    cfg = env.model.babi_cfg(V, attention_mode=mode, D=D, en_mq=(mode == 2))
    cfg["num_bit"] = nb
    ans = rng.integers(0, V, 400)
    run_both(env, cfg, weights(V + mode, 3, D, V, 1.5), sw, qw, n_sen, ans)


OPTIONS = {
    "pow2": dict(softmax_variant=1), "exp_plan": dict(softmax_variant=2), "relu": dict(en_non_lin=True),
    "no_lin_map": dict(en_lin_map=False), "binary": dict(fmt_bin=(0, 0)),
    "scale": dict(att_scale=[-0.5, 0.25, -0.125]), "fractional_weights": dict(fmt_w=[(0, 7), (0, 6), (1, 6)]),
    "short_words": dict(fmt=[(3, 2)] * 3, fmt_att=[(2, 4)] * 3, fmt_w=[(2, 3), (4, 1), (1, 2)], fmt_bin=(3, 1)),
}


@pytest.mark.parametrize("opt", sorted(OPTIONS))
def test_options(env, opt):
    rng = np.random.default_rng(5)
    V, D = 40, 60
    sw, qw, n_sen = random_stories(rng, 200, V, V - 10, 8, [1, 2, 6, 10, 33])
    cfg = env.model.babi_cfg(V, attention_mode=2, D=D, en_mq=False)
    cfg.update(OPTIONS[opt])
    run_both(env, cfg, weights(77, 3, D, V, 1.5), sw, qw, n_sen, rng.integers(0, V, 200))


def test_many_queries_persistent_grid(env):
    rng = np.random.default_rng(9)
    V, D = 30, 60
    sw, qw, n_sen = random_stories(rng, 50000, V, 20, 8, [2, 4, 6, 8, 10], dup_every=50)
    cfg = env.model.babi_cfg(V, attention_mode=2, D=D)
    run_both(env, cfg, weights(3, 3, D, V), sw, qw, n_sen, rng.integers(0, V, 50000))
