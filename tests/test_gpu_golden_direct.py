"""The reference's own outputs straight against the HIP kernels -- no oracle in between -- for the two rows `north_star`
calls bit-exact: the quantiser (lib/common.h:187-221, the FLOAT_QUANT / FLOAT2FIXED / FIXED_MUL /
FIXED_ADD macros) and the Hamming similarities (a6: lib/common.c:223-312, hamming_similarity{,_w}).  The fixtures
tests/golden/ref_quant.npz and ref_hamming.npz hold what the reference's compiled C code returned (oracle/gen_golden.py);
here they are fed to qmann_quantize_i8, the fixed-point verbs, qmann_pack_bitplanes and the popcount score kernels."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    load_pkg()
    import qmann_amd.abi as abi
    import qmann_amd.model as model

    class Env:
        pass
    e = Env()
    e.torch, e.abi, e.model, e.lib = torch, abi, model, abi.lib
    e.dev = torch.device("cuda:0")
    e.up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(e.dev)
    e.ptr = lambda t: C.c_void_p(t.data_ptr())
    return e


def test_quantize_i8_equals_the_reference_macros(env, gold):
    """every (x, iwl, frac) of ref_quant.npz: the sign-magnitude byte is the reference's FLOAT2FIXED word (sign bit + the
    magnitude bits, minus zero included), the two's-complement byte times 2^-frac is its FLOAT_QUANT value"""
    torch, abi = env.torch, env.abi
    g = gold("ref_quant.npz")
    x = g["x"]
    n = x.size
    dx = env.up(x)
    seen = 0
    for i, (iwl, frac) in enumerate(g["formats"]):
        iwl, frac = int(iwl), int(frac)
        if iwl + frac == 0 or iwl + frac > 7:
            continue                                            # (0, 0) binarises: not an int8 code
        seen += 1
        word = g["word"][i].astype(np.int64) & 0xFFFFFFFF
        assert (word & 0x7FFFFF80).max() == 0                    # the magnitude fits 7 bits
        want_sm = (((word >> 31) << 7) | (word & 0x7F)).astype(np.uint8)
        for pitch in (n, n + 38):                                # dense rows and padded rows (padding must come back zero)
            for layout, want in ((abi.CODE_SIGNMAG, want_sm),
                                 (abi.CODE_TWOS, np.rint(g["quant"][i] * (1 << frac)).astype(np.int8).view(np.uint8))):
                out = torch.full((2, pitch), 0x55, dtype=torch.int8, device=env.dev)
                src = torch.stack([dx, dx.flip(0)]).contiguous()
                abi.check(env.lib.qmann_quantize_i8(env.ptr(src), env.ptr(out), 2, n, pitch, abi.Fmt(iwl, frac), layout, None),
                          "qmann_quantize_i8")
                got = out.cpu().numpy().view(np.uint8)
                np.testing.assert_array_equal(got[0, :n], want, err_msg=f"Q{iwl}.{frac} layout {layout}")
                np.testing.assert_array_equal(got[1, :n], want[::-1], err_msg=f"Q{iwl}.{frac} layout {layout} (row 1)")
                assert not got[:, n:].any()
        # the grid value the byte stands for is the macro's float
        sm = want_sm.astype(np.int16)
        val = np.where(sm & 0x80, -(sm & 0x7F), sm & 0x7F).astype(np.float32) / np.float32(1 << frac)
        np.testing.assert_array_equal(val, g["quant"][i])
        assert (want_sm == 0x80).any(), "the fixture reaches the minus-zero word"
    assert seen >= 5


def test_fixed_add_and_mul_verbs_equal_the_reference_macros(env, gold):
    """FIXED_ADD through cuda_sum_vec_fwd and FIXED_MUL through a one-column cuda_dot_mat_vec_fwd (one product per row, the
    accumulation starts from zero), operands as the fixture holds them -- raw floats, the macros quantise them"""
    g = gold("ref_quant.npz")
    a, b = g["a"], g["b"]
    n = a.size
    da, db = env.up(a), env.up(b)
    do = env.torch.empty(n, dtype=env.torch.float32, device=env.dev)
    for i, (iwl, frac) in enumerate(g["formats"][:-1]):
        iwl, frac = int(iwl), int(frac)
        env.lib.cuda_sum_vec_fwd(env.ptr(da), env.ptr(db), env.ptr(do), n, True, iwl, frac, 3, False)
        np.testing.assert_array_equal(do.cpu().numpy(), g["add"][i], err_msg=f"FIXED_ADD Q{iwl}.{frac}")
        # rows = the a's, the one-element vector = one b: every distinct b once
        got = np.empty(n, np.float32)
        for j in range(n):
            env.lib.cuda_dot_mat_vec_fwd(env.ptr(da[j:]), env.ptr(db[j:]), env.ptr(do[j:]), None, 1, 1, False, True,
                                         iwl, frac, iwl, frac, 3, False)
        got = do.cpu().numpy()
        np.testing.assert_array_equal(got, g["mul"][i], err_msg=f"FIXED_MUL Q{iwl}.{frac}")


def _ham_inputs(g):
    """the 4 096 word pairs as 32 one-slot stories of 128 columns (packed planes need rows of at least 16 bytes: 128 columns
    at one bit per column): keys = top bytes of one word, u = the other's as values of
    the Q5.2 grid.  A top byte of 0x80 ("minus zero": sign set, the seven magnitude bits clear) on the u side is a negative
    value too small for the grid -- the reference's operand word FLOAT2FIXED(u, iwl, 31 - iwl) keeps its sign bit and its top
    seven magnitude bits are zero (lib/common.h:210), which is what the kernels' byte rule reproduces."""
    a = (g["a"].astype(np.int64) & 0xFFFFFFFF) >> 24
    b = (g["b"].astype(np.int64) & 0xFFFFFFFF) >> 24
    kb = a.astype(np.uint8).reshape(32, 128)
    ub = b.astype(np.int16).reshape(32, 128)
    u = np.where(ub & 0x80, -(ub & 0x7F), ub & 0x7F).astype(np.float32) / np.float32(4.0)
    u[ub == 0x80] = -np.float32(2.0 ** -10)
    assert (kb == 0x80).any() and (ub == 0x80).any() and ((kb == 0x80) & (ub == 0x80)).any()
    return kb, u


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("mode", [10, 11])
@pytest.mark.parametrize("num_bit", [1, 4, 8])
def test_popcount_scores_equal_the_reference_hamming_similarity(env, gold, mode, num_bit, packed):
    """V0 (mode 10) = sum over the columns of hamming_similarity, V1 (mode 11) = sum of hamming_similarity_w, for the
    fixture's word pairs and n in {1, 4, 8}: from packed bit planes (qmann_pack_bitplanes + qmann_hops_packed, __popcll) and
    straight from the int8 keys (qmann_hops_i8).  Only the top n <= 8 bits of a word enter, i.e. its top byte."""
    torch, model = env.torch, env.model
    g = gold("ref_hamming.npz")
    ni = int(np.flatnonzero(g["num_bit"] == num_bit)[0])
    kb, u = _ham_inputs(g)
    if mode == 10:
        want = g["sim"][ni].astype(np.int64).reshape(32, 128).sum(1)
        unit = 1.0
    else:
        w = g["sim_w"][ni].astype(np.float64) * (1 << num_bit)
        assert np.array_equal(w, np.rint(w))                      # weights are multiples of 2^-n
        want = np.rint(w).astype(np.int64).reshape(32, 128).sum(1)
        unit = 1.0 / (1 << num_bit)
    H, D, V = 1, 128, 40
    cfg = dict(n_hop=H, dim_emb=D, dim_input=V, attention_mode=mode, softmax_variant=0, f_fixed=True, en_lin_map=True,
               fmt=[(5, 2)], fmt_w=[(5, 2)], fmt_att=[(5, 2)], fmt_bin=(5, 2), num_bit=num_bit)
    rng = np.random.default_rng(1)
    wts = {"w_h": [rng.normal(0, 1, (D, D)).astype(np.float32)], "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    assert net.Dp == 128
    dk = torch.from_numpy(kb.view(np.int8).reshape(1, 32, 128)).to(env.dev)
    dv = torch.zeros_like(dk)
    ro = torch.arange(33, dtype=torch.int32, device=env.dev)                        # 32 stories of one slot
    du = torch.from_numpy(u).to(env.dev)
    if packed:
        planes = net.pack_planes(dk, num_bit)
        pl = planes.cpu().numpy().view(np.uint64)                                   # the planes themselves: bit i of every byte
        for i in range(num_bit):
            bits = ((kb >> (7 - i)) & 1).reshape(32, 2, 64)
            np.testing.assert_array_equal(pl[0, :, :, i], np.packbits(bits, axis=-1, bitorder="little").view(np.uint64)[..., 0])
        _, taps = net.hops_packed(planes, dv, ro, 1, du, taps=True)
    else:
        _, taps = net.hops(dk, dv, ro, 1, du, taps=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(taps.score_codes.cpu().numpy()[0].astype(np.int64), want)
    np.testing.assert_array_equal(taps.scores.cpu().numpy()[0], (want * unit).astype(np.float32))


def test_verbose_forward_verb_is_a_sync_point_and_prints_the_reference_dump(tmp_path):
    """verbose = true (lib/layer_cuda.cu:2868-2882 for the softmax): the deferred queue is drained first -- the queued
    sum_vec feeding it has run -- and the dump carries the reference's labels and `%f, ` rows.  Run as a child process:
    the dump goes to the C stdout."""
    import subprocess
    import sys
    from conftest import ROOT
    child = tmp_path / "child.py"
    child.write_text(f"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, {str(ROOT / 'tests')!r})
from conftest import load_pkg
load_pkg()
import qmann_amd.abi as abi
lib = abi.lib
lib.qmann_abi_set_defer(1)                      # as a C host has it: forward verbs are queued
dev = torch.device("cuda:0")
a = torch.tensor([0.25, 1.0, -0.5, 2.0], device=dev); b = torch.tensor([0.5, 0.25, 0.5, -1.0], device=dev)
s = torch.zeros(4, device=dev); p = torch.zeros(4, device=dev); m = torch.zeros(1, device=dev)
torch.cuda.synchronize()
ptr = lambda t: C.c_void_p(t.data_ptr())
lib.cuda_sum_vec_fwd(ptr(a), ptr(b), ptr(s), 4, True, 5, 2, 3, False)           # queued, not run
lib.cuda_softmax_fwd(ptr(p), ptr(s), None, None, ptr(m), 4, False, True)        # verbose: drains, runs, dumps
lib.cuda_sum_vec_fwd(ptr(a), ptr(b), ptr(s), 4, True, 5, 2, 3, True)
w = torch.tensor([[1.0, 0.5], [0.25, -1.0], [2.0, 2.0]], device=dev); x = torch.tensor([1.0, 2.0], device=dev); y = torch.zeros(3, device=dev)
lib.cuda_dense_fwd(ptr(w), None, ptr(x), ptr(y), None, 2, 3, b"NULL", True, 5, 2, 5, 2, 3, True)
lib.cuda_dot_mat_vec_fwd(ptr(w), ptr(x), ptr(y), None, 3, 2, False, True, 5, 2, 5, 2, 3, True)
lib.qmann_abi_flush()
""")
    r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout
    blocks = out.split("\n< ")
    names = [b.split(" >")[0] for b in blocks[1:]]
    assert names == ["cuda_softmax_fwd", "cuda_sum_vec_fwd", "cuda_dense_fwd", "cuda_dot_mat_vec_fwd"], out
    sm = blocks[1].splitlines()
    assert sm[1] == "dev_in_vec> dim: 4" and sm[2] == "0.750000, 1.250000, 0.000000, 1.000000"      # the queued sum has run
    assert sm[3] == "dev_out_vec> dim: 4"
    pr = np.array([float(v) for v in sm[4].split(", ")])
    e = np.exp(np.array([0.75, 1.25, 0.0, 1.0]) - 1.25)
    np.testing.assert_allclose(pr, e / e.sum(), atol=2e-6)
    sv = blocks[2].splitlines()
    assert sv[1] == "dev_in_vec_a> dim: 4" and sv[3] == "dev_in_vec_b> dim: 4" and sv[5] == "dev_out_vec> dim: 4"
    assert sv[6] == "0.750000, 1.250000, 0.000000, 1.000000"
    dn = blocks[3].splitlines()
    assert dn[1] == "dev_in_vec> dim: 2" and dn[3] == "dev_w_mat> dim_out: 3, dim_in: 2"
    assert dn[4:7] == ["1.000000, 0.500000", "0.250000, -1.000000", "2.000000, 2.000000"]
    assert dn[7] == "dev_out_vec> dim: 3" and dn[8] == "2.000000, -1.750000, 6.000000"
    dt = blocks[4].splitlines()
    assert dt[1] == "f_trans: false" and dt[2] == "dev_in_mat> dim_mat_r: 3, dim_mat_c: 2"
    assert dt[6] == "dev_in_vec> dim: 2" and dt[8] == "dev_out_vec> dim: 3" and dt[9] == "2.000000, -1.750000, 6.000000"
