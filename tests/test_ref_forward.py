"""oracle/ref_forward.c (bench.py's "reference+port" CPU baseline): the forward composed from the reference's
LIVE C functions + the port must equal the oracle's own composite with the CPU softmax form, bit for bit, at
both flag sets (the reference's `gcc -w` and `-O2`).  Needs oracle/_ref (built where /root/reference exists)."""
import numpy as np
import pytest

from pyoracle import HERE, Oracle, RefForward, SM_CPU_POW2, SM_CPU_EXP_PLAN

pytestmark = pytest.mark.skipif(not (HERE / "_ref" / "libqmann_refcpu_O2.so").exists(),
                                reason="oracle/_ref not built (needs /root/reference)")


def _cfg(mode, V, D=60, H=3, variant=SM_CPU_POW2, nb=8, **kw):
    fmt = [(5, 2)] * H
    c = dict(n_hop=H, dim_emb=D, dim_input=V, attention_mode=mode, softmax_variant=variant, f_fixed=True, en_lin_map=True,
             fmt=fmt, fmt_w=[(6, 1), (5, 2), (4, 3)][:H], fmt_att=list(fmt), fmt_bin=(5, 2), num_bit=nb)
    c.update(kw)
    return c


def _weights(rng, H, D, V):
    return {"w_q": rng.normal(0, 1, (D, V)).astype(np.float32),
            "w_a": [rng.normal(0, 1, (D, V)).astype(np.float32) for _ in range(H)],
            "w_c": [rng.normal(0, 1, (D, V)).astype(np.float32) for _ in range(H)],
            "w_h": [rng.normal(0, 1, (D, D)).astype(np.float32) for _ in range(H)],
            "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


def _bow(rng, n_sen, V, n_dict):
    s = np.zeros((n_sen, V), np.float32)
    for r in range(n_sen):
        for w in rng.integers(1, n_dict, rng.integers(1, 7)):
            s[r, w] += 1.0
        s[r, n_dict + n_sen - 1 - r] = 1.0
    q = np.zeros(V, np.float32)
    q[rng.integers(1, n_dict, 3)] = 1.0
    return s, q


@pytest.mark.parametrize("flags", ["O0", "O2"])
@pytest.mark.parametrize("mode,variant,extra", [
    (2, SM_CPU_POW2, {}), (1, SM_CPU_POW2, {}), (3, SM_CPU_POW2, {}), (10, SM_CPU_POW2, {}), (11, SM_CPU_POW2, dict(nb=4)),
    (2, SM_CPU_EXP_PLAN, {}), (2, SM_CPU_POW2, dict(softmax_shift_based=True)), (2, SM_CPU_POW2, dict(en_non_lin=True)),
])
def test_bow_forward_equals_oracle(flags, mode, variant, extra):
    rng = np.random.default_rng(7 + mode)
    V, n_dict = 30, 20
    extra = dict(extra)
    cfg = _cfg(mode, V, variant=variant, nb=extra.pop("nb", 8), **extra)
    w = _weights(rng, 3, 60, V)
    ora, rf = Oracle(), RefForward(flags)
    m = ora.make_model(cfg, w)
    for n_sen in (1, 2, 7, 10):
        s, q = _bow(rng, n_sen, V, n_dict)
        p0, t = ora.forward(m, s, q, taps=("u",))
        p1, u1 = rf.forward(m, s, q)
        assert p0 == p1
        assert np.array_equal(t["u"][-1], u1)


@pytest.mark.parametrize("flags", ["O0", "O2"])
def test_mem_forward_equals_oracle_and_timer(flags):
    rng = np.random.default_rng(11)
    H, D, V, S = 3, 128, 64, 300
    cfg = _cfg(2, V, D=D)
    w = _weights(rng, H, D, V)
    ora, rf = Oracle(), RefForward(flags)
    m = ora.make_model(cfg, w)
    pool = []
    for _ in range(3):
        keys = (np.clip(np.rint(rng.normal(0, 3.5, (H, S, D))), -127, 127) / 4).astype(np.float32)
        vals = (np.clip(np.rint(rng.normal(0, 30, (H, S, D))), -127, 127) / 4).astype(np.float32)
        u0 = (np.clip(np.rint(rng.normal(0, 3.5, D)), -127, 127) / 4).astype(np.float32)
        pool.append((keys, vals, u0))
    want = []
    for keys, vals, u0 in pool:
        p0, t = ora.forward_mem(m, keys, vals, u0, taps=("u",))
        p1, u1 = rf.forward_mem(m, keys, vals, u0)
        assert p0 == p1 and np.array_equal(t["u"][-1], u1)
        want.append(p0)
    r = rf.time(m, pool, n_threads=2, seconds=0.3)
    assert r["n"] >= 3 and r["qps"] > 0 and r["preds"] == want
    assert ("-O2" in rf.flags) == (flags == "O2")
