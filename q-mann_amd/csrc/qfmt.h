// qfmt.h -- the Q(iwl.frac) number format of Q-MANN, shared by every kernel.
//
// Behaviour restated from lib/layer_cuda.h:207-259 (device macros) and
// lib/common.h:178-227 (host macros) of the reference: a value is
// t = trunc(x * 2^frac) toward zero, saturated symmetrically at
// +-(2^(iwl+frac) - 1); the reference carries t as a sign-magnitude word and
// immediately decodes it back to the float t / 2^frac.  For word length 8
// (iwl + frac == 7) |t| <= 127, so t fits an int8 two's-complement code and
// all products / sums of such values are exact small integers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QM_HD __host__ __device__ __forceinline__
#else
#define QM_HD static inline
#endif

// x . 2^-k, exact (a full-precision float division costs ~10 instructions on the device)
QM_HD float qm_scale_down(float x, uint32_t k) { return __builtin_ldexpf(x, -(int)k); }

struct QFmt {
    uint32_t iwl;
    uint32_t frac;
};

// two's-complement integer code of Q(x)
QM_HD int32_t qm_code(float x, uint32_t iwl, uint32_t frac)
{
    const int32_t M = (int32_t)((1u << (iwl + frac)) - 1u);
    const float scale = (float)(1 << frac);
    const float maxf = qm_scale_down((float)M, frac);
    if (x > maxf) return M;
    if (x < -maxf) return -M;
    float p = x * scale;
    // only reachable when iwl + frac == 31: the device conversion saturates
    if (p >= 2147483648.0f) return 2147483647;
    if (p <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)p;
}

// sign-magnitude word as the reference builds it (negative values that
// truncate to zero keep the sign bit: 0x80000000)
QM_HD uint32_t qm_signmag(float x, uint32_t iwl, uint32_t frac)
{
    int32_t t = qm_code(x, iwl, frac);
    if (x >= 0.0f) return (uint32_t)t & 0x7FFFFFFFu;
    return (uint32_t)(-(int64_t)t) | 0x80000000u;
}

// code of the OPERAND quantiser Qv(x): as qm_code, plus the iwl + frac == 0 case, which binarises to
// +-1 (BINARY_MODE: iwl_bin = frac_bin = 0, MemN2N/MemN2N.c:769-775; lib/layer_cuda.h:207-253)
QM_HD int32_t qm_code_or_sign(float x, uint32_t iwl, uint32_t frac)
{
    if (iwl + frac == 0) return (x >= 0.0f) ? 1 : -1;
    return qm_code(x, iwl, frac);
}

QM_HD float qm_decode(int32_t code, uint32_t frac)
{
    return qm_scale_down((float)code, frac);
}

// FLOAT_QUANT, including the iwl + frac == 0 binarisation
QM_HD float qm_quant(float x, uint32_t iwl, uint32_t frac)
{
    if (iwl + frac == 0) return (x >= 0.0f) ? 1.0f : -1.0f;
    const int32_t k = qm_code(x, iwl, frac);
    // iwl + frac == 31 only: x == -2^iwl is not below the float limit, converts to INT32_MIN, and the reference's
    // sign-magnitude word of that is "minus zero" -- CUDA_FIXED2FLOAT gives 0 (lib/layer_cuda.h:246-253)
    if (k == (int32_t)0x80000000) return 0.0f;
    return qm_decode(k, frac);
}

// FIXED_MUL: Qa(Qa(a) * Qb(b))
QM_HD float qm_fixed_mul(float a, float b, QFmt fa, QFmt fb)
{
    return qm_quant(qm_quant(a, fa.iwl, fa.frac) * qm_quant(b, fb.iwl, fb.frac), fa.iwl, fa.frac);
}

// integer form of FIXED_MUL for word-length-8 codes: ka in format a, kb in
// format b; result code in format a.  ka*kb / 2^frac_b truncated toward zero,
// saturated at +-max_a.
QM_HD int32_t qm_mul_code(int32_t ka, int32_t kb, uint32_t frac_b, int32_t max_a)
{
    int32_t p = ka * kb;
    int32_t t = (p + ((p >> 31) & ((1 << frac_b) - 1))) >> frac_b;
    return t > max_a ? max_a : (t < -max_a ? -max_a : t);
}

// ---- operands of the Hamming-family attentions under mixed quantisation (host side; the arithmetic: ham_common.h) ---------
// The reference hands these attentions the embedding outputs on the WEIGHT grid of their hop (EN_MQ: Q(iwl+1.frac-1),
// Q(iwl.frac), Q(iwl-1.frac+1) for hops 0, 1, 2; MemN2N/MemN2N.c:748-754) and re-encodes them as Q(iwl_att, 31 - iwl_att)
// words (lib/layer_cuda.cu:355-420, :2515; lib/common.c:223-312 on the same alignment).  Modes 10 / 11 compare the top bits of
// those words as they are: one byte per operand carries them for ANY grid (Q_att truncation, clamp, sign of the value, minus
// zero at -2^iwl_att).  Mode 3 first does arithmetic on the whole words: what a byte can carry is classified per hop.
enum { kHamSame = 0, kHamCoarse = 1, kHamFine = 2, kHamNone = 3 };

// src: the grid of u entering the hop (w[0] for hop 0, act[h-1] after); wk: the grid of the hop's keys (w[h]); att: the
// attention format (word length 8)
static inline int ham_hop_kind(QFmt src, QFmt wk, QFmt att)
{
    const bool u_in = src.iwl <= att.iwl && src.frac <= att.frac, k_in = wk.iwl <= att.iwl && wk.frac <= att.frac;
    if (u_in && k_in) return kHamSame;
    if (src.frac + 1 <= att.frac && wk.frac + 1 <= att.frac) return kHamCoarse;      // odd codes free: 127 marks the saturated word
    if (u_in && wk.frac == att.frac + 1 && wk.iwl + 1 <= att.iwl) return kHamFine;
    return kHamNone;
}

// the format of a hop's KEY BYTES (what the embedding kernels quantise to) and whether the minus-zero rule applies to them
// (a key of exactly -2^iwl_att: only on a wider grid).  mode: QMANN_ATT_* (3 = APPX, 10 / 11 = Hamming V0 / V1)
static inline QFmt ham_key_format(uint32_t mode, QFmt src, QFmt wk, QFmt att, bool *minus_zero)
{
    const bool ham = mode == 3u || mode == 10u || mode == 11u;
    *minus_zero = ham && wk.iwl > att.iwl;
    if (mode == 3u && ham_hop_kind(src, wk, att) == kHamFine) return wk;
    return att;
}
