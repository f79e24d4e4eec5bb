"""Four queries per wavefront for stories of at most 16 rows (csrc/hops_quad.h) and the split of a mixed batch into a short
and a long index list (k_split_by_length + k_hops_quad + k_hops_lean<..., LIST>), against the general one-wavefront kernel
(csrc/hops_small.h, itself checked against the oracle in test_gpu_batch.py) and against the lean kernel alone
(QMANN_NO_QUAD): the hop outputs must be identical bit for bit in every score mode, format combination and option."""
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


def three_paths(env, monkeypatch, cfg, B, S_list, seed, sigma_k=30.0, sigma_u=20.0, sigma_h=1.0, shuffle=True, max_slots=None, repeat=0,
                force_quad=True):
    """general kernel (taps) == production path (quad / split) == lean kernel alone (QMANN_NO_QUAD)"""
    torch, model = env.torch, env.model
    H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
    rng = np.random.default_rng(seed)
    wts = {"w_h": [rng.normal(0, sigma_h, (D, D)).astype(np.float32) for _ in range(H)],
           "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    Dp = net.Dp
    assert Dp == 64
    n_slots = np.array([S_list[i % len(S_list)] for i in range(B)], np.int64)
    if shuffle:
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
    ms = int(n_slots.max()) if max_slots is None else max_slots
    # (batches of <= 8 192 stories keep one story per wavefront by default -- csrc/hops_lean.h::launch_lean --: the quad kernel is
    # under test here at every batch size)
    monkeypatch.setenv("QMANN_QUAD_MIN_QUERIES", "0" if force_quad else "8192")
    model.abi.lib.qmann_tuning_reload()
    u_gen, _ = net.hops(dk, dv, dro, ms, du0, taps=True)        # taps: general kernel
    u_prod = net.hops(dk, dv, dro, ms, du0)                     # no taps: quad / split (where they apply)
    u_again = [net.hops(dk, dv, dro, ms, du0) for _ in range(repeat)]       # back to back, no synchronisation between the calls
    monkeypatch.setenv("QMANN_NO_QUAD", "1")
    model.abi.lib.qmann_tuning_reload()
    u_lean = net.hops(dk, dv, dro, ms, du0)                     # the lean kernel alone
    monkeypatch.delenv("QMANN_NO_QUAD")
    monkeypatch.delenv("QMANN_QUAD_MIN_QUERIES")
    model.abi.lib.qmann_tuning_reload()
    torch.cuda.synchronize()
    g = u_gen.cpu().numpy()
    for name, t in [("production", u_prod), ("lean", u_lean)] + [(f"production, call {i + 2}", t) for i, t in enumerate(u_again)]:
        a = t.cpu().numpy()
        bad = np.flatnonzero((a != g).any(1))
        assert bad.size == 0, f"{name}: {bad.size} of {B} queries differ from the general kernel, first {bad[:5]}, slots {n_slots[bad[:5]]}"
    assert np.abs(g).sum() > 0
    return net


def cfg_of(mode, D=60, H=3, iwl=5, base=0, nb=8, **kw):
    frac = 7 - iwl
    fmt = [(iwl, frac)] * H
    c = dict(n_hop=H, dim_emb=D, dim_input=40, attention_mode=mode, softmax_variant=base, f_fixed=True, en_lin_map=True,
             fmt=fmt, fmt_w=list(fmt), fmt_att=list(fmt), fmt_bin=(iwl, frac), num_bit=nb)
    c.update(kw)
    return c


MODES = [(2, 8), (3, 8), (10, 8), (10, 2), (10, 1), (11, 8), (11, 4), (11, 1)]


@pytest.mark.parametrize("mode,nb", MODES)
@pytest.mark.parametrize("D", [60, 64, 17])
def test_quad_equals_general_all_modes(env, monkeypatch, mode, nb, D):
    """every story <= 16 rows: the quad kernel alone; 301 queries leave the last wavefront's quad one query short"""
    three_paths(env, monkeypatch, cfg_of(mode, D=D, nb=nb), B=301, S_list=[0, 1, 2, 3, 4, 5, 8, 9, 10, 12, 13, 15, 16], seed=100 + mode * 7 + D)


@pytest.mark.parametrize("mode,nb", [(2, 8), (3, 8), (11, 8)])
def test_mixed_batch_is_split_by_length(env, monkeypatch, mode, nb):
    """the 20-task shape: most stories short, a few up to 64 rows -- k_split_by_length, then the quad kernel on the short list and
    the lean kernel on the long one; in story order, shuffled, all long, all short but the bound"""
    S = [2, 4, 6, 8, 10, 3, 16, 17, 5, 40, 9, 64, 1, 0, 12, 33]
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=1000, S_list=S, seed=500 + mode, shuffle=False)
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=1003, S_list=S, seed=510 + mode)
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=130, S_list=[17, 64, 33, 20], seed=520 + mode)
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=130, S_list=[2, 9, 16], seed=530 + mode, max_slots=64)


@pytest.mark.parametrize("B", [64, 8192, 8193])
def test_small_batches_keep_one_story_per_wavefront_by_default(env, monkeypatch, B):
    """the launch rule at its default threshold, on either side of it (the production call must equal the general kernel whichever form it takes)"""
    three_paths(env, monkeypatch, cfg_of(2), B=B, S_list=[1, 3, 6, 10, 16], seed=600 + B, force_quad=False)
    three_paths(env, monkeypatch, cfg_of(11, nb=4), B=B, S_list=[2, 9, 16, 30, 5, 7], seed=610 + B, force_quad=False)


def test_quad_persistent_grid_many_queries(env, monkeypatch):
    """more quads than resident wavefronts: every wavefront walks several quads (the task-1 lengths, then the joint mix)"""
    three_paths(env, monkeypatch, cfg_of(2), B=200003, S_list=[2, 4, 6, 8, 10], seed=300)
    three_paths(env, monkeypatch, cfg_of(3), B=100001, S_list=[2, 4, 6, 8, 10, 14, 3, 20, 7, 64, 5, 9], seed=301)


@pytest.mark.parametrize("mode,nb", [(2, 8), (3, 8), (10, 1), (11, 4)])
def test_split_batch_runs_its_two_kernels_side_by_side(env, monkeypatch, mode, nb):
    """>= 32 768 queries of mixed length: the short stories' kernel on the caller's stream, the long stories' on a second one
    (csrc/hops_lean.h::launch_lean, fork / join by events); several forwards back to back reuse the same two events"""
    S = [2, 4, 6, 8, 10, 14, 3, 20, 7, 64, 5, 9]
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=40001, S_list=S, seed=310 + mode, repeat=4)
    three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=33000, S_list=[3, 9, 16, 12], seed=320 + mode, max_slots=64, repeat=2)   # (nothing long: the side kernel finds an empty list)
    monkeypatch.setenv("QMANN_NO_CORUN", "1")
    env.model.abi.lib.qmann_tuning_reload()
    try:
        three_paths(env, monkeypatch, cfg_of(mode, nb=nb), B=40001, S_list=S, seed=310 + mode)
    finally:
        monkeypatch.delenv("QMANN_NO_CORUN")
        env.model.abi.lib.qmann_tuning_reload()


VARIANTS = {
    "pow2": dict(softmax_variant=1), "exp_plan": dict(softmax_variant=2),
    "exp_shift_scaled": dict(softmax_variant=0, softmax_shift_based=True, att_scale=[0.02, 0.015, 0.03]),
    "scale_negative": dict(softmax_variant=0, att_scale=[-0.5, 0.25, -0.125]),
    "relu": dict(en_non_lin=True), "no_lin_map": dict(en_lin_map=False),
    "binary": dict(fmt_bin=(0, 0)), "binary_relu": dict(fmt_bin=(0, 0), en_non_lin=True),
    "en_mq": dict(fmt_w=[(6, 1), (5, 2), (4, 3)]), "one_hop": dict(n_hop=1), "five_hops": dict(n_hop=5),
    "no_lin_map_relu": dict(en_lin_map=False, en_non_lin=True),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("mode", [2, 3, 10])
def test_quad_options(env, monkeypatch, mode, variant):
    """the options of the reference's define.h (softmax bases the quad kernel does not take fall back to the lean kernel: the
    three paths must still agree)"""
    extra = dict(VARIANTS[variant])
    H = extra.pop("n_hop", 3)
    if mode != 2 and variant == "exp_shift_scaled":
        extra["att_scale"] = [0.002, 0.001, 0.0015]
    cfg = cfg_of(mode, H=H)
    cfg.update(extra)
    if "att_scale" in cfg:
        cfg["att_scale"] = (cfg["att_scale"] * 2)[:H]
    three_paths(env, monkeypatch, cfg, B=150, S_list=[1, 2, 5, 9, 13, 16], seed=400 + mode)
    three_paths(env, monkeypatch, cfg, B=150, S_list=[1, 2, 9, 16, 30, 64], seed=450 + mode)


@pytest.mark.parametrize("iwl", [2, 3, 6, 7])
@pytest.mark.parametrize("mode", [2, 3, 11])
def test_quad_other_word_splits(env, monkeypatch, mode, iwl):
    """1, 4 and 5 fraction bits: up to 2, 16, 32 surviving rows (several fetch rounds per hop); flat scores share the weight"""
    if iwl == 7 and mode != 2:
        pytest.skip("Q7.0 attention operands: fixed-point scores only")
    for sk in (30.0, 1.5):
        three_paths(env, monkeypatch, cfg_of(mode, iwl=iwl), B=200, S_list=[1, 5, 11, 16], seed=200 + iwl, sigma_k=sk)


@pytest.mark.parametrize("mode,nb", [(3, 8), (10, 8), (11, 4)])
@pytest.mark.parametrize("iwl", [5, 3])
def test_quad_hamming_under_mixed_quantisation(env, monkeypatch, mode, nb, iwl):
    """EN_MQ weight formats: mode 3 runs its three lane sums (kHamCoarse, kHamSame, kHamFine: ham_common.h), one per hop"""
    frac = 7 - iwl
    c = cfg_of(mode, iwl=iwl, nb=nb, fmt_w=[(iwl + 1, frac - 1), (iwl, frac), (iwl - 1, frac + 1)])
    three_paths(env, monkeypatch, c, B=300, S_list=[0, 1, 2, 9, 10, 16], seed=900 + mode + iwl, sigma_u=40.0)


@pytest.mark.parametrize("mode", [2, 11])
@pytest.mark.parametrize("bin_fmt", [(5, 2), (6, 1), (7, 0), (0, 0), (4, 2), (2, 2)])
def test_quad_linear_map_clamp_corrections(env, monkeypatch, mode, bin_fmt):
    """large linear-map codes against large operands: many per-product clamps, for operand formats with 0, 1 and 2 fractional
    bits, shorter operand words and the binarised operand"""
    cfg = cfg_of(mode, nb=4)
    cfg["fmt_bin"] = bin_fmt
    three_paths(env, monkeypatch, cfg, B=200, S_list=[1, 4, 9, 16], seed=700 + bin_fmt[0] * 8 + bin_fmt[1], sigma_u=80.0, sigma_h=12.0)


@pytest.mark.parametrize("seed", range(16))
def test_quad_random_formats(env, monkeypatch, seed):
    """random formats: word length 7 everywhere (the quad kernel's domain) or not (the lean kernel's) -- the three paths agree"""
    rng = np.random.default_rng(9000 + seed)

    def fmt(lo=2, hi=7, force7=False):
        wl = 7 if force7 else int(rng.integers(lo, hi + 1)); iwl = int(rng.integers(0, wl + 1))
        return (iwl, wl - iwl)
    H = int(rng.integers(1, 5))
    f7 = bool(seed % 4)
    cfg = dict(n_hop=H, dim_emb=int(rng.choice([20, 60, 64])), dim_input=40, attention_mode=2,
               softmax_variant=0 if f7 else int(rng.integers(0, 3)), f_fixed=True, en_lin_map=bool(rng.integers(0, 4)),
               fmt=[fmt(force7=f7) for _ in range(H)], fmt_w=[fmt(force7=f7) for _ in range(H)], fmt_att=[fmt(force7=f7) for _ in range(H)],
               fmt_bin=fmt(1, 7), en_non_lin=bool(rng.integers(0, 2)))
    three_paths(env, monkeypatch, cfg, B=128, S_list=[1, 3, 9, 16, 33, 64], seed=9100 + seed, sigma_k=25.0)


@pytest.mark.parametrize("mode", [2, 3, 11])
def test_quad_against_the_oracle(env, monkeypatch, oracle, mode):
    """the quad kernel's final hop state against the oracle directly (task-1 lengths), Q5.2 with the EN_MQ weight formats"""
    torch, model = env.torch, env.model
    cfg = cfg_of(mode, fmt_w=[(6, 1), (5, 2), (4, 3)] if mode == 2 else [(5, 2)] * 3)
    H, D, V, B = 3, 60, 40, 96
    rng = np.random.default_rng(7700 + mode)
    wts = {"w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
           "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    n_slots = np.array([[2, 4, 6, 8, 10, 16, 1, 13][i % 8] for i in range(B)], np.int64)
    row_off = np.concatenate([[0], np.cumsum(n_slots)]).astype(np.int32)
    R = int(row_off[-1])
    keys = np.zeros((H, R, 64), np.int8); vals = np.zeros((H, R, 64), np.int8)
    keys[:, :, :D] = np.clip(np.rint(rng.normal(0, 30, (H, R, D))), -127, 127)
    vals[:, :, :D] = np.clip(np.rint(rng.normal(0, 30, (H, R, D))), -127, 127)
    w0 = cfg["fmt_w"][0]
    u0 = (np.clip(np.rint(rng.normal(0, 20, (B, D))), -127, 127) / (1 << w0[1])).astype(np.float32)
    dk = torch.from_numpy(model.to_signmag(keys)).to(env.dev); dv = torch.from_numpy(model.to_signmag(vals)).to(env.dev)
    u = net.hops(dk, dv, torch.from_numpy(row_off).to(env.dev), int(n_slots.max()), torch.from_numpy(u0).to(env.dev)).cpu().numpy()
    m = oracle.make_model(cfg, {**wts, "w_q": np.zeros((D, V), np.float32), "w_a": [np.zeros((D, V), np.float32)] * H,
                                "w_c": [np.zeros((D, V), np.float32)] * H})
    excused = 0
    for q in range(B):
        a, b = int(row_off[q]), int(row_off[q + 1])
        kf = np.stack([keys[h, a:b, :D].astype(np.float32) / (1 << cfg["fmt_att"][h][1]) for h in range(H)])
        vf = np.stack([vals[h, a:b, :D].astype(np.float32) / (1 << cfg["fmt"][h][1]) for h in range(H)])
        _, t = oracle.forward_mem(m, kf, vf, u0[q])
        if not np.array_equal(u[q], t["u"][H - 1]):
            near = False
            for h in range(H):
                x = t["probs"][h].astype(np.float64) * (1 << cfg["fmt"][h][1])
                k = np.rint(x)
                near |= bool(((np.abs(x - k) <= 1e-5 * np.maximum(1.0, np.abs(x))) & (k > 0)).any())
            assert near, f"query {q} ({n_slots[q]} rows) differs from the oracle"
            excused += 1
    assert excused == 0, f"{excused} queries needed the p-on-a-step excuse (observed: 0)"
