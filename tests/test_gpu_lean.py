"""The instruction-lean hop kernel for bAbI-sized memories (csrc/hops_lean.h: <= 64 slots, 64-byte rows, no taps)
against the general one-wavefront kernel (csrc/hops_small.h, itself checked against the oracle in test_gpu_batch.py):
the hop outputs must be identical bit for bit, for every score mode, format combination and option the reference has.
A request with taps always takes the general kernel, one without takes the lean kernel when it applies."""
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


def both_paths(env, cfg, B, S_list, seed, sigma_k=30.0, sigma_u=20.0, sigma_h=1.0, require_nonzero=True):
    torch, model = env.torch, env.model
    H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
    rng = np.random.default_rng(seed)
    wts = {"w_h": [rng.normal(0, sigma_h, (D, D)).astype(np.float32) for _ in range(H)],
           "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    Dp = net.Dp
    assert Dp == 64
    n_slots = np.array([S_list[i % len(S_list)] for i in range(B)], np.int64)
    rng.shuffle(n_slots)
    row_off = np.concatenate([[0], np.cumsum(n_slots)]).astype(np.int32)
    R = max(int(row_off[-1]), 1)
    keys = np.zeros((H, R, Dp), np.int8); vals = np.zeros((H, R, Dp), np.int8)
    keys[:, :, :D] = np.clip(np.rint(rng.normal(0, sigma_k, (H, R, D))), -127, 127)
    vals[:, :, :D] = np.clip(np.rint(rng.normal(0, sigma_k, (H, R, D))), -127, 127)
    keys[:, 1::5, ::4] = 0
    for h in range(H):
        mk = (1 << sum(cfg["fmt_att"][h])) - 1; mv = (1 << sum(cfg["fmt"][h])) - 1
        keys[h] = np.clip(keys[h], -mk, mk); vals[h] = np.clip(vals[h], -mv, mv)
    w0 = cfg["fmt_w"][0]
    m0 = (1 << sum(w0)) - 1
    u0 = (np.clip(np.rint(rng.normal(0, sigma_u, (B, D))), -m0, m0) / (1 << w0[1])).astype(np.float32)
    sk = model.to_signmag(keys)
    sk[:, ::11, 3] = np.int8(-128)                      # "minus zero" bytes (0x80) are legal memory codes
    dk = torch.from_numpy(sk).to(env.dev); dv = torch.from_numpy(model.to_signmag(vals)).to(env.dev)
    dro = torch.from_numpy(row_off).to(env.dev); du0 = torch.from_numpy(u0).to(env.dev)
    ms = int(n_slots.max())
    u_gen, _ = net.hops(dk, dv, dro, ms, du0, taps=True)        # taps: general kernel
    u_lean = net.hops(dk, dv, dro, ms, du0)                     # no taps: lean kernel
    torch.cuda.synchronize()
    a, b = u_lean.cpu().numpy(), u_gen.cpu().numpy()
    bad = np.flatnonzero((a != b).any(1))
    assert bad.size == 0, f"{bad.size} of {B} queries differ, first {bad[:5]}, slots {n_slots[bad[:5]]}"
    assert not require_nonzero or np.abs(b).sum() > 0             # (a test case that computes nothing tests nothing)
    return net


def cfg_of(mode, D=60, H=3, iwl=5, base=0, nb=8, **kw):
    frac = 7 - iwl
    fmt = [(iwl, frac)] * H
    c = dict(n_hop=H, dim_emb=D, dim_input=40, attention_mode=mode, softmax_variant=base, f_fixed=True, en_lin_map=True,
             fmt=fmt, fmt_w=list(fmt), fmt_att=list(fmt), fmt_bin=(iwl, frac), num_bit=nb)
    c.update(kw)
    return c


@pytest.mark.parametrize("mode,nb", [(2, 8), (3, 8), (10, 8), (10, 2), (10, 1), (11, 8), (11, 4), (11, 1)])
@pytest.mark.parametrize("D", [60, 64, 17])
def test_lean_equals_general_all_modes(env, mode, nb, D):
    both_paths(env, cfg_of(mode, D=D, nb=nb), B=300, S_list=[0, 1, 2, 3, 9, 10, 16, 17, 33, 50, 63, 64], seed=100 + mode * 7 + D)


@pytest.mark.parametrize("sparse", ["0", "1"])
@pytest.mark.parametrize("mode,iwl", [(2, 5), (2, 2), (2, 6), (3, 5), (11, 3), (10, 5)])
def test_lean_value_tile_and_sparse_read_out_agree_with_the_general_kernel(env, monkeypatch, mode, iwl, sparse):
    """the lean kernel either copies a story's whole value tile to LDS or fetches only the rows whose weight code Q(p) is not
    zero (the launcher chooses by mean story length and 2^frac; QMANN_LEAN_SPARSE forces it): both against the general kernel,
    with formats of 1, 2, 4 and 5 fraction bits (up to 2, 4, 16, 32 surviving rows: several fetch rounds) and flat scores
    (small sigma_k: many rows share the weight)"""
    monkeypatch.setenv("QMANN_LEAN_SPARSE", sparse)
    env.model.abi.lib.qmann_tuning_reload()                 # (the switches are read once per process; conftest re-reads them after the test)
    for sk in (30.0, 1.5):
        both_paths(env, cfg_of(mode, iwl=iwl), B=200, S_list=[0, 1, 2, 5, 16, 17, 40, 50, 64], seed=40 + mode + iwl, sigma_k=sk)


@pytest.mark.parametrize("mode,nb", [(3, 8), (10, 8), (11, 4)])
@pytest.mark.parametrize("iwl", [5, 3])
def test_lean_equals_general_hamming_under_mixed_quantisation(env, mode, nb, iwl):
    """EN_MQ weight formats: mode 3 runs its three lane sums (kHamCoarse, kHamSame, kHamFine: ham_common.h), one per hop"""
    frac = 7 - iwl
    c = cfg_of(mode, iwl=iwl, nb=nb, fmt_w=[(iwl + 1, frac - 1), (iwl, frac), (iwl - 1, frac + 1)])
    both_paths(env, c, B=300, S_list=[0, 1, 2, 9, 10, 17, 33, 50, 64], seed=900 + mode + iwl, sigma_u=40.0)


@pytest.mark.parametrize("iwl", [2, 3, 6])
@pytest.mark.parametrize("mode", [2, 3, 11])
def test_lean_other_word_splits(env, mode, iwl):
    both_paths(env, cfg_of(mode, iwl=iwl), B=120, S_list=[1, 5, 31, 64], seed=200 + iwl)


@pytest.mark.parametrize("mode", [2, 11])
@pytest.mark.parametrize("bin_fmt", [(5, 2), (6, 1), (7, 0), (0, 0), (4, 2), (2, 2)])
def test_lean_linear_map_clamp_corrections(env, mode, bin_fmt):
    """large linear-map codes against large operands: many per-product clamps, for operand formats with 0, 1 and 2
    fractional bits and the binarised operand"""
    cfg = cfg_of(mode, nb=4)
    cfg["fmt_bin"] = bin_fmt
    both_paths(env, cfg, B=200, S_list=[1, 4, 20, 64], seed=700 + bin_fmt[0] * 8 + bin_fmt[1], sigma_u=80.0, sigma_h=12.0)
    cfg["fmt_w"] = [(2, 5) if mode == 2 else (5, 2), (4, 3), (3, 3)]   # (the byte forms need u0's grid inside the attention grid)
    both_paths(env, cfg, B=100, S_list=[3, 30], seed=750 + bin_fmt[0], sigma_u=80.0, sigma_h=1.5)


def test_lean_persistent_grid_many_queries(env):
    """more queries than resident wavefronts: every wavefront walks several queries"""
    both_paths(env, cfg_of(2), B=60000, S_list=[2, 6, 10], seed=300)
    both_paths(env, cfg_of(11), B=30000, S_list=[7, 50], seed=301)


VARIANTS = {
    "pow2": dict(softmax_variant=1), "exp_plan": dict(softmax_variant=2),
    "pow2_shift": dict(softmax_variant=1, softmax_shift_based=True),
    "exp_shift_scaled": dict(softmax_variant=0, softmax_shift_based=True, att_scale=[0.02, 0.015, 0.03]),
    "scale_negative": dict(softmax_variant=0, att_scale=[-0.5, 0.25, -0.125]),
    "relu": dict(en_non_lin=True), "no_lin_map": dict(en_lin_map=False),
    "binary": dict(fmt_bin=(0, 0)), "binary_relu": dict(fmt_bin=(0, 0), en_non_lin=True),
    "en_mq": dict(fmt_w=[(6, 1), (5, 2), (4, 3)]), "one_hop": dict(n_hop=1), "five_hops": dict(n_hop=5),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("mode", [2, 3, 10])
def test_lean_options(env, mode, variant):
    extra = dict(VARIANTS[variant])
    H = extra.pop("n_hop", 3)
    if mode != 2 and variant == "exp_shift_scaled":
        extra["att_scale"] = [0.002, 0.001, 0.0015]
    cfg = cfg_of(mode, H=H)
    cfg.update(extra)
    if "att_scale" in cfg:
        cfg["att_scale"] = (cfg["att_scale"] * 2)[:H]
    both_paths(env, cfg, B=96, S_list=[1, 2, 9, 50, 64], seed=400 + mode)


@pytest.mark.parametrize("seed", range(16))
def test_lean_random_formats_fixed(env, seed):
    rng = np.random.default_rng(9000 + seed)

    def fmt(lo=2, hi=7):
        wl = int(rng.integers(lo, hi + 1)); iwl = int(rng.integers(0, wl + 1))
        return (iwl, wl - iwl)
    H = int(rng.integers(1, 5))
    cfg = dict(n_hop=H, dim_emb=int(rng.choice([20, 60, 64])), dim_input=40, attention_mode=2,
               softmax_variant=int(rng.integers(0, 3)), f_fixed=True, en_lin_map=bool(rng.integers(0, 4)),
               fmt=[fmt() for _ in range(H)], fmt_w=[fmt() for _ in range(H)], fmt_att=[fmt() for _ in range(H)],
               fmt_bin=fmt(1, 7), en_non_lin=bool(rng.integers(0, 2)))
    both_paths(env, cfg, B=64, S_list=[1, 3, 9, 33, 64], seed=9100 + seed, sigma_k=25.0)


@pytest.mark.parametrize("seed", range(12))
def test_lean_random_formats_hamming(env, seed):
    rng = np.random.default_rng(9500 + seed)
    ia = int(rng.integers(1, 7)); att = (ia, 7 - ia)

    def inside():
        i = int(rng.integers(0, ia + 1)); f = int(rng.integers(0, 7 - ia + 1))
        if i + f < 2:
            i, f = min(ia, 1), max(1, min(7 - ia, 1))
        return (i, f)

    def free():
        wl = int(rng.integers(2, 8)); i = int(rng.integers(0, wl + 1))
        return (i, wl - i)
    H = 3
    cfg = cfg_of([3, 10, 11][seed % 3], nb=int(rng.choice([1, 2, 4, 8])))
    cfg.update(fmt=[inside() for _ in range(H)], fmt_w=[inside()] + [free() for _ in range(H - 1)], fmt_att=[att] * H,
               fmt_bin=free(), en_lin_map=bool(rng.integers(0, 5)))
    both_paths(env, cfg, B=64, S_list=[1, 5, 33, 64], seed=9600 + seed, sigma_k=40.0, sigma_u=40.0)


def test_soak_find_pow2_total_on_a_float_rounding_boundary(env):
    """found by tools/soak.py (case 12750221): 2^x base, one dominant slot and a runner-up exactly 2^-24 below it.  Two
    double totals summed in different orders then round to different floats, Q(p) differs by a code and the two kernels
    disagreed; both now add these bases the way the reference's CPU softmax does (a float total in slot order)."""
    cfg = {'n_hop': 3, 'dim_emb': 20, 'dim_input': 40, 'attention_mode': 2, 'softmax_variant': 1, 'f_fixed': True, 'en_lin_map': False,
           'fmt': [(0, 7), (1, 2), (3, 0)], 'fmt_w': [(4, 2), (2, 2), (3, 0)], 'fmt_att': [(0, 4), (6, 0), (3, 3)], 'fmt_bin': (6, 1),
           'en_non_lin': True}
    both_paths(env, cfg, B=184, S_list=[39, 41, 20, 5, 49, 8], seed=12750221, sigma_k=27.262290262548344,
               sigma_u=68.24222274885791, sigma_h=5.570029277416961)


@pytest.mark.parametrize("taps", [False, True])
def test_pow2_total_is_the_reference_float_sum(env, oracle, taps):
    """scores 63 and 39 on a Q6.0 grid, base 2^x: the terms are 1 and 2^-24.  The reference's float total is 1 (the small term
    is rounded away), p = 1 exactly and Q1.2(p) = 4; a double total would give p = 1 - 2^-23 and the code 3.  Both
    short-memory kernels must give the oracle's hop output (lean: no taps, general: taps)."""
    torch, model = env.torch, env.model
    D, V = 8, 10
    cfg = dict(n_hop=1, dim_emb=D, dim_input=V, attention_mode=2, softmax_variant=1, f_fixed=True, en_lin_map=False,
               fmt=[(1, 2)], fmt_w=[(6, 1)], fmt_att=[(6, 0)], fmt_bin=(6, 1))
    wts = {"w_h": [np.zeros((D, D), np.float32)], "w_ans": np.zeros((V, D), np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    keys = np.zeros((1, 3, 64), np.int8); vals = np.zeros((1, 3, 64), np.int8)
    keys[0, :, 0] = [63, 39, -5]
    vals[0, 0, :D] = [7, -7, 5, 3, -1, 6, -6, 2]
    vals[0, 1, :D] = [-7, 7, 1, 1, 1, 1, 1, 1]
    u0 = np.zeros((1, D), np.float32); u0[0, 0] = 1.0
    ro = torch.tensor([0, 3], dtype=torch.int32, device=env.dev)
    dk = torch.from_numpy(model.to_signmag(keys)).to(env.dev); dv = torch.from_numpy(model.to_signmag(vals)).to(env.dev)
    out = net.hops(dk, dv, ro, 3, torch.from_numpy(u0).to(env.dev), taps=taps)
    u = (out[0] if taps else out).cpu().numpy()[0]
    m = oracle.make_model(cfg, {**wts, "w_q": np.zeros((D, V), np.float32), "w_a": [np.zeros((D, V), np.float32)],
                                "w_c": [np.zeros((D, V), np.float32)]})
    _, t = oracle.forward_mem(m, keys[:, :, :D].astype(np.float32), vals[:, :, :D].astype(np.float32) / 4.0, u0[0])
    assert t["probs"][0][0] == 1.0 and list(t["scores"][0]) == [63.0, 39.0, -5.0]
    np.testing.assert_array_equal(u, t["u"][0])


@pytest.mark.parametrize("mode", [2, 3, 11])
@pytest.mark.parametrize("max_slots", [10, 16, 33])
def test_story_longer_than_the_bound_is_cut_not_spilled(env, mode, max_slots):
    """qmann_batch.h: a story longer than max_slots is CUT to max_slots.  The lean kernel sizes a wavefront's value tile
    by round16(max_slots); a longer story must neither write past the tile (its neighbours in the workgroup would read
    garbage) nor change any in-bound query.  Every query must equal the same batch with the long stories shortened by
    hand, in the lean kernel (no taps) and the general one (taps)."""
    torch, model = env.torch, env.model
    cfg = cfg_of(mode)
    H, D, V, B = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"], 400
    rng = np.random.default_rng(4400 + mode * 100 + max_slots)
    wts = {"w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
           "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    Dp = net.Dp
    n_slots = rng.integers(0, max_slots + 1, B)
    long_q = rng.choice(B, 40, replace=False)
    n_slots[long_q] = rng.integers(max_slots + 1, 65, long_q.size)         # up to 64: the worst case for the tile
    n_slots[long_q[:4]] = [64, 63, max_slots + 1, 200]                     # (also one beyond a wavefront)
    row_off = np.concatenate([[0], np.cumsum(n_slots)]).astype(np.int64)
    R = int(row_off[-1])
    keys = np.zeros((H, R, Dp), np.int8); vals = np.zeros((H, R, Dp), np.int8)
    keys[:, :, :D] = np.clip(np.rint(rng.normal(0, 30, (H, R, D))), -127, 127)
    vals[:, :, :D] = np.clip(np.rint(rng.normal(0, 30, (H, R, D))), -127, 127)
    u0 = (np.clip(np.rint(rng.normal(0, 20, (B, D))), -127, 127) / 4.0).astype(np.float32)
    # the same batch with the long stories cut by hand
    keep = np.concatenate([np.arange(row_off[q], row_off[q] + min(n_slots[q], max_slots)) for q in range(B)]).astype(np.int64)
    cut_off = np.concatenate([[0], np.cumsum(np.minimum(n_slots, max_slots))]).astype(np.int32)
    sk, sv = model.to_signmag(keys), model.to_signmag(vals)
    dev = env.dev
    du0 = torch.from_numpy(u0).to(dev)
    full = [torch.from_numpy(x).to(dev) for x in (sk, sv)]
    cut = [torch.from_numpy(np.ascontiguousarray(x[:, keep])).to(dev) for x in (sk, sv)]
    ro_full = torch.from_numpy(row_off.astype(np.int32)).to(dev); ro_cut = torch.from_numpy(cut_off).to(dev)
    want = net.hops(cut[0], cut[1], ro_cut, max_slots, du0, taps=True)[0].cpu().numpy()
    got_lean = net.hops(full[0], full[1], ro_full, max_slots, du0).cpu().numpy()
    got_gen = net.hops(full[0], full[1], ro_full, max_slots, du0, taps=True)[0].cpu().numpy()
    for name, got in (("lean", got_lean), ("general", got_gen)):
        bad = np.flatnonzero((got != want).any(1))
        assert bad.size == 0, f"{name}: {bad.size} queries differ (long ones among them: {np.intersect1d(bad, long_q).size})"
    assert np.abs(want).sum() > 0
