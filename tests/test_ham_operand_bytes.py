"""CPU: the one-byte-per-operand form of the Hamming attentions' operands under mixed quantisation (EN_MQ, MemN2N.c:748-754),
derived in csrc/ham_common.h ("APPX under EN_MQ") and DESIGN.md §2, checked EXHAUSTIVELY against the oracle's 32-bit word
arithmetic (oracle/qmann_oracle.c::qo_float2fixed -- pinned by the reference's own macros in test_oracle_golden.py -- and
appx_pair, the restatement of lib/layer_cuda.cu:355-420):
  * the structure of the operand words the derivation rests on, for every value of every grid involved;
  * the byte rule of the embedding kernels (sign of the value | magnitude on the attention grid, clamp, minus zero);
  * mode 3's per-column term computed from two bytes, for every pair of operand values, in the three hop kinds.
The GPU tests (test_gpu_batch.py::test_hops_appx_under_mixed_quantisation, ...) check the kernels; this file checks the
arithmetic they implement, where no GPU is needed."""
import numpy as np
import pytest


def grid_values(fmt):
    m = (1 << (fmt[0] + fmt[1])) - 1
    codes = np.arange(-m, m + 1, dtype=np.int64)
    return codes, (codes / float(1 << fmt[1])).astype(np.float32)


def key_bytes(values, kind, wk, att):
    """the rule of csrc/batch_io.hip (ew_to_bytes / sm_byte with the minus-zero flag) and of ham_ubyte (csrc/ham_common.h)"""
    v = values.astype(np.float64)
    if kind == "fine":
        mag = np.minimum(np.floor(np.abs(v) * (1 << wk[1])), 127)
    else:
        mag = np.minimum(np.floor(np.abs(v) * (1 << att[1])), 127)
        mag[v == -float(1 << att[0])] = 0
    return mag.astype(np.int64) | np.where(v < 0, 0x80, 0)


def byte_term(kb, ub, kind):
    """one column of csrc/ham_common.h::appx_lane_sum_k, in units of 2^-10"""
    ks, us = kb >> 7, ub >> 7
    k8, um = kb & 0x7F, ub & 0x7F
    km = k8 >> 1 if kind == "fine" else k8
    same = 127 - np.abs(km - um) + ((kind == "fine") & (k8 & 1 == 1) & (km < um))
    s = km + um + ((km & um & 1) if kind == "coarse" else 0)
    carry = s >= 128
    val = 127 - (s & 127)
    larger_neg = np.where(km >= um, ks, us) == 1
    opp = np.where(carry & ~larger_neg, val, -val)
    return np.where(ks == us, same, opp)


@pytest.mark.parametrize("iwl", [1, 2, 3, 4, 5, 6])
def test_operand_words_have_the_structure_the_byte_forms_rest_on(oracle, iwl):
    I, F = iwl, 7 - iwl
    att = (I, F)
    for kind, fmt in (("same", att), ("coarse", (I + 1, F - 1)), ("fine", (I - 1, F + 1))):
        codes, vals = grid_values(fmt)
        words = oracle.float2fixed(vals, I, 31 - I).astype(np.int64) & 0xFFFFFFFF
        sign, mag = words >> 31, words & 0x7FFFFFFF
        neg_zero = vals == -float(1 << I)
        np.testing.assert_array_equal(sign[~neg_zero], (vals < 0).astype(np.int64)[~neg_zero])
        if kind == "same":
            np.testing.assert_array_equal(mag, np.abs(codes) << 24)
        elif kind == "fine":
            np.testing.assert_array_equal(mag, np.abs(codes) << 23)           # the key's own code, one bit below the compared ones
            assert np.abs(codes).max() <= 127
        else:
            inside = np.abs(vals) < float(1 << I)
            np.testing.assert_array_equal(mag[inside], (np.abs(codes[inside]) * 2) << 24)      # even codes of the attention grid
            sat = (np.abs(vals) > float(1 << I)) | (vals == float(1 << I))
            assert sat.any() and (mag[sat] == 0x7FFFFFFF).all()               # 24 low ones under a top byte of 127
            assert neg_zero.sum() == 1 and words[neg_zero][0] == 0x80000000   # exactly -2^iwl: "minus zero"
        # the byte the kernels carry is the word's top byte (fine keys: bits 30..23)
        b = key_bytes(vals, kind, fmt, att)
        top = (words >> 23) & 0x7F if kind == "fine" else (words >> 24) & 0x7F
        np.testing.assert_array_equal(b & 0x7F, top)
        np.testing.assert_array_equal(b >> 7, sign)


@pytest.mark.parametrize("iwl", [1, 3, 5, 6])
@pytest.mark.parametrize("kind", ["same", "coarse", "fine"])
def test_mode3_term_from_two_bytes_equals_the_word_arithmetic_for_every_operand_pair(oracle, iwl, kind):
    I, F = iwl, 7 - iwl
    att = (I, F)
    wk = {"same": att, "coarse": (I + 1, F - 1), "fine": (I - 1, F + 1)}[kind]
    wu = (I + 1, F - 1) if kind == "coarse" else att                         # u's grid: hop 0's weight grid, or inside the attention grid
    _, kv = grid_values(wk)
    _, uv = grid_values(wu)
    kb = key_bytes(kv, kind, wk, att)
    ub = key_bytes(uv, "coarse" if kind == "coarse" else "same", wu, att)
    for j, u in enumerate(uv):
        want = oracle.dot_mat_vec_fwd_appx(kv.reshape(-1, 1), np.array([u], np.float32), False, True, I, F, 8)
        got = byte_term(kb, np.full_like(kb, ub[j]), kind)
        # one column: |term| <= 127 units of 2^-10, far inside the final clamp at 2^iwl
        np.testing.assert_array_equal(got, np.rint(want.astype(np.float64) * 1024).astype(np.int64), err_msg=f"u = {u}")


def test_mode3_kinds_that_need_their_correction_terms(oracle):
    """the corrections are not vacuous: without them the byte form differs from the word arithmetic in exactly the cases
    the derivation names (two saturated operands of opposite sign; odd finer key below u of the same sign)"""
    I, F = 5, 2
    # coarse: +sat against -sat
    want = oracle.dot_mat_vec_fwd_appx(np.array([[40.0]], np.float32), np.array([-50.0], np.float32), False, True, I, F, 8)
    assert int(round(float(want[0]) * 1024)) == int(byte_term(np.array([0x7F]), np.array([0xFF]), "coarse")[0])
    assert int(byte_term(np.array([0x7F]), np.array([0xFF]), "same")[0]) != int(round(float(want[0]) * 1024))
    # fine: key 0.375 (code 3 on Q4.3) against u 0.75 (code 3 on Q5.2), same sign
    want = oracle.dot_mat_vec_fwd_appx(np.array([[0.375]], np.float32), np.array([0.75], np.float32), False, True, I, F, 8)
    assert int(round(float(want[0]) * 1024)) == int(byte_term(np.array([3]), np.array([3]), "fine")[0]) == 126
