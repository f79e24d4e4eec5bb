// hops_lean.h -- every hop of a query whose memory has at most 64 slots and whose rows are 64 bytes (bAbI
// width: D <= 64), one wavefront per query, written for instruction count: at these sizes the chip is bound by
// VALU issue (SQ_INSTS_VALU x 4 cycles accounts for the whole run time of hops_small.h), not by memory.
//
// Same arithmetic, stage by stage, as hops_small.h / the streaming kernels (which remain the reference
// implementation inside this library and the path for taps); what changes is where the work sits:
//   * the hop state u lives in a register (lane c owns column c), never in LDS as floats;
//   * the per-lane constants of the packed multiply (ScanConst: 16 columns per lane) are not rebuilt by every lane
//     from 16 codes (~200 VALU per hop): lane c prepares column c once (magnitude pre-shifted, sign byte) and
//     stores it into an LDS image laid out so that a lane's ue / uo / s7 words are three 16-byte loads;
//   * value rows go from global memory straight into the wavefront's LDS tile (global_load_lds_dwordx4), keys and
//     linear-map rows are requested a hop ahead into registers;
//   * the read-out and the hop update work on integer codes (no float round trips);
//   * workgroups are persistent (a wavefront walks queries q, q + stride, ...), so per-workgroup tables -- exp of
//     the 255 possible score differences per hop, the linear maps -- are built once.
// Selected by qmann_hops_i8 when no taps are requested (see lean_supported()).
#pragma once
#include "hops_small.h"

#include <limits.h>
#include <stdlib.h>

namespace {

constexpr int kLeanWaves = 8;                       // wavefronts (= queries in flight) per workgroup
constexpr int kLeanBlock = kLeanWaves * kWave;
// Wavefronts per SIMD a lean kernel is compiled for (its register budget: 128 at four, 80 at six) and its persistent grid is sized
// by.  With the buffer loads the fixed-point kernel needs 80-83 registers, so a THIRD 8-wavefront workgroup per CU fits wherever
// LDS has room for it (up to ~16 value-tile rows: bAbI task 1, the sparse variant at any length): 0.705 -> 0.652-0.665 ms at
// |mem| = 50 and +7.5 % on the task-1 forward (interleaved A/B); the Hamming forms gain 7-8 % the same way.  Where LDS holds two
// workgroups anyway (the joint set's 64-row tiles) the tighter budget only costs (-2.3 % on the mode-3 joint forward): those
// launches take the four-wave build.
constexpr int kLeanWpsWide = 4, kLeanWpsTight = 6;

// per-workgroup LDS: exp tables [n_hop][256] float, linear maps [n_hop][64][64] sign-magnitude bytes
// per-wavefront LDS slice: value tile [rows_pad][64], then the small arrays below
constexpr uint32_t kLwE = 0;            // u16 [32]  pre-shifted |u| of even columns (ScanConst::ue image)
constexpr uint32_t kLwO = 64;           // u16 [32]  odd columns (ScanConst::uo image)
constexpr uint32_t kLwS = 128;          // u8  [64]  0x80 where u < 0 (ScanConst::s7 image)
constexpr uint32_t kLwUb = 192;         // u8  [64]  sign-magnitude Q_att(u) bytes (Hamming forms)
constexpr uint32_t kLwUn = 0;           // f32 [64]  new u (row-group layout -> column layout); reuses the four images above,
                                        //            which are dead once the linear map's constants sit in registers
constexpr uint32_t kLwSc = 256;         // i16 [64]  scores, slot r
constexpr uint32_t kLwOc = 384;         // i16 [64]  read-out codes o[c]
constexpr uint32_t kLwBytes = 512;

constexpr uint32_t kLmHopBytes = 64u * 192u;       // a hop's linear map in LDS: 64 rows x {even magnitudes, odd magnitudes, signs}

struct LeanArgs {
    uint32_t rows_pad;        // value-tile rows per wavefront (max_slots rounded up to 16)
    uint32_t exp_table;       // 1: fixed-point scores, e^x base, no scale layer: exp(-d . unit) comes from a table
    uint32_t lm_in_lds;       // 1: the linear maps are staged in LDS
    const uint32_t *list;     // the queries to run, by index (the long stories of a batch whose short ones take hops_quad.h); nullptr: all
    const uint32_t *n_list;   // device word with the list's length
    const uint32_t *n_other;  // side-by-side launch (launch_lean): device word with the OTHER list's length; nullptr otherwise
};

// Two kernels side by side (launch_lean: a batch split by story length, >= 32 768 queries): each is launched with every
// workgroup it could have resident -- three per CU for the short stories' kernel, two for the long stories' -- and trims itself
// here, once both list lengths are known ON THE DEVICE: workgroups past the returned count leave at once and their room goes to
// the other kernel.  A long story costs about 2.7 short ones (4.2 against 1.55 ns per query, each kernel alone on the 20-task
// set).  Long stories' share of the work up to 0.45: one workgroup per CU for them, two for the short ones (the 20-task set:
// 9 % long, share 0.21; measured best of 3+1 / 3+3 / 2+2 / 1+1 / 2+1); below 0.05 the short kernel keeps all three.  Above
// 0.45 nobody trims: the two full grids then run mostly one after the other, as without the second stream -- no mix of
// lengths makes the pair slower than that.
__device__ __forceinline__ uint32_t corun_groups(uint32_t n_short, uint32_t n_long, uint32_t grid, bool long_side)
{
    const float cl = 2.7f * (float)n_long, share = cl / (cl + (float)n_short + 1.0f);
    if (long_side) return share <= 0.45f ? grid / 2u : grid;
    return (share >= 0.05f && share <= 0.45f) ? grid - grid / 3u : grid;
}

// Wavefront reductions on the DPP network (row shifts, then the two row broadcasts; the total lands in lane 63 and is
// handed to every lane through an SGPR).  A shuffle-based butterfly costs no vector instructions either, but each of its
// 6 steps is a dependent LDS-crossbar round trip (~120 cycles): 18 of them per hop were a fifth of a wavefront's time.
template <typename Op>
__device__ __forceinline__ int dpp_reduce_i32(int v, int identity, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xF, 0xF, false));      // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xF, 0xF, false));      // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xF, 0xF, false));      // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xF, 0xF, false));      // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xA, 0xF, false));      // row_bcast:15 -> rows 1, 3
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xC, 0xF, false));      // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    return dpp_reduce_i32(v, INT_MIN, [](int a, int b) { return a > b ? a : b; });
}
__device__ __forceinline__ float wave_max_f32(float v)
{
    float r = v;                                        // identity for lanes the shift does not reach: -inf
#define QM_DPP_MAXF(CTRL, RM) r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xFF800000, __builtin_bit_cast(int, r), CTRL, RM, 0xF, false)))
    QM_DPP_MAXF(0x111, 0xF); QM_DPP_MAXF(0x112, 0xF); QM_DPP_MAXF(0x114, 0xF); QM_DPP_MAXF(0x118, 0xF); QM_DPP_MAXF(0x142, 0xA); QM_DPP_MAXF(0x143, 0xC);
#undef QM_DPP_MAXF
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 63));
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#define QM_DPP_ADDD(CTRL, RM)                                                                                          \
    do {                                                                                                               \
        const uint64_t b_ = __builtin_bit_cast(uint64_t, v);                                                           \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b_, CTRL, RM, 0xF, false);        \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b_ >> 32), CTRL, RM, 0xF, false); \
        v += __builtin_bit_cast(double, (uint64_t)lo_ | ((uint64_t)hi_ << 32));                                        \
    } while (0)
    QM_DPP_ADDD(0x111, 0xF); QM_DPP_ADDD(0x112, 0xF); QM_DPP_ADDD(0x114, 0xF); QM_DPP_ADDD(0x118, 0xF); QM_DPP_ADDD(0x142, 0xA); QM_DPP_ADDD(0x143, 0xC);
#undef QM_DPP_ADDD
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane c publishes column c of the packed-multiply constants for operand code k (word length wl of the matrix
// format, frac fv of the vector format): see ScanConst / make_scan_const in hops_common.h
__device__ __forceinline__ void publish_scan_const(uint8_t *lw, uint32_t lane, int k, uint32_t wl, int fv)
{
    const int pre = (int)(16u - wl) - fv;
    const uint32_t a = (uint32_t)(k < 0 ? -k : k) << pre;
    const uint16_t m = (uint16_t)(a > 0xFFFFu ? 0xFFFFu : a);
    *(uint16_t *)(lw + ((lane & 1u) ? kLwO : kLwE) + (lane >> 1) * 2) = m;
    lw[kLwS + lane] = (uint8_t)(k < 0 ? 0x80u : 0u);
}
__device__ __forceinline__ uint32_t fetch_scan_const(ScanConst &c, const uint8_t *lw, uint32_t chunk, uint32_t wl)
{
    const i32x4 e = *(const i32x4 *)(lw + kLwE + chunk * 16);
    const i32x4 o = *(const i32x4 *)(lw + kLwO + chunk * 16);
    const i32x4 s = *(const i32x4 *)(lw + kLwS + chunk * 16);
#pragma unroll
    for (int d = 0; d < 4; d++) { c.ue[d] = (uint32_t)e[d]; c.uo[d] = (uint32_t)o[d]; c.s7[d] = (uint32_t)s[d]; }
    return 16u - wl;
}

// W7: every format of the launch (attention, linear-map weights, activations) has word length 7 -- the stock 8-bit configurations
template <bool W7>
__device__ __forceinline__ void publish_const(uint8_t *lw, uint32_t lane, int k, uint32_t wl, int fv)
{
    if (!W7) { publish_scan_const(lw, lane, k, wl, fv); return; }
    const uint32_t a = (uint32_t)(k < 0 ? -k : k) << (8 - fv);                  // fv <= 7
    *(uint16_t *)(lw + ((lane & 1u) ? kLwO : kLwE) + (lane >> 1) * 2) = (uint16_t)(a > 0x7FFFu ? 0x7FFFu : a);
    lw[kLwS + lane] = (uint8_t)(k < 0 ? 0x80u : 0u);
}
template <bool W7>
__device__ __forceinline__ int lane_sum_w(const i32x4 x, const ScanConst &c, uint32_t sh)
{
    return W7 ? lane_row_sum7(x, c) : lane_row_sum(x, c, sh);
}
// the same on a row that is already split into even / odd magnitudes and sign bits (the linear maps in LDS)
template <bool W7>
__device__ __forceinline__ int lane_sum_split(const i32x4 ev, const i32x4 od, const i32x4 sn, const ScanConst &c, uint32_t sh)
{
    int acc = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t tb;
        if (W7) {
            tb = __builtin_amdgcn_perm(pk_mul_sat_i16((uint32_t)od[d], c.uo[d]), pk_mul_sat_i16((uint32_t)ev[d], c.ue[d]), 0x07030501u);
        } else {
            const u16x2 te = __builtin_bit_cast(u16x2, pk_mul_sat_u16((uint32_t)ev[d], c.ue[d])) >> (unsigned short)sh;
            const u16x2 to = __builtin_bit_cast(u16x2, pk_mul_sat_u16((uint32_t)od[d], c.uo[d])) >> (unsigned short)sh;
            tb = __builtin_bit_cast(uint32_t, te) | (__builtin_bit_cast(uint32_t, to) << 8);
        }
        const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, (uint32_t)sn[d] ^ c.s7[d]);
        acc = __builtin_amdgcn_sdot4((int)tb, (int)sg, acc, false);
    }
    return acc;
}

// exp tables [n_hop][256] and linear maps [n_hop][64][64] of a (persistent) workgroup; ends WITHOUT a barrier
__device__ __forceinline__ void lean_stage_tables(const HopArgs &a, const LeanArgs &la, float *etab, uint8_t *lmap, uint32_t tid,
                                                  uint32_t nthreads)
{
    const uint32_t H = a.n_hop;
    if (la.exp_table) {
        for (uint32_t i = tid; i < H * 256u; i += nthreads) {
            const uint32_t h = i >> 8, d = i & 255u;
            // score - max on the score grid: -(d . 2^-frac_att), exact; the same call the per-slot form makes
            etab[i] = sm_exp(-qm_scale_down((float)d, a.att[h].frac), SmCfg{a.softmax_base, false, false, 1.0f});
        }
    }
    if (la.lm_in_lds) {
        // The linear maps are constant, so the part of the packed multiply that touches only the matrix bytes is done here,
        // once per workgroup: a row is kept as three 64-byte images -- |H| of the even columns and of the odd columns as
        // 16-bit lanes (the multiplier's operands) and the sign bits -- 6 instead of 9 operations per 4 columns in the hop.
        for (uint32_t h = 0; h < H; h++) {
            const uint32_t *src = (const uint32_t *)a.lin_map[h];
            for (uint32_t i = tid; i < 64u * 16u; i += nthreads) {           // dword i & 15 of row i >> 4
                const uint32_t r = i >> 4, d = i & 15u;
                const uint32_t w = r < a.D ? src[i] : 0u;
                uint32_t *dst = (uint32_t *)(lmap + h * kLmHopBytes + r * 192u);
                dst[d] = w & 0x007F007Fu;
                dst[16 + d] = (w >> 8) & 0x007F007Fu;
                dst[32 + d] = w & 0x80808080u;
            }
        }
    }
}

// does a story of S rows end in a narrow pass (see lean_hop)?
template <int MODE, bool W7>
__device__ __forceinline__ bool lean_tail_rows(uint32_t S)
{
    return MODE == kModeFixed && W7 && (S & 15u) != 0u && (S & 15u) <= 4u;
}

// The read-out weight code Q(p) = min(trunc(p . 2^frac), maxa) of a slot, p = the softmax quotient of `sm_quot`.
// For the e^x base the reference's quotient is (float)((double)e / total) (lib/layer_cuda.cu:2039): a double division per lane,
// some 25 vector instructions, to produce an integer that is 0 for almost every slot.  The code only needs to know on which side
// of the steps k . 2^-frac the quotient falls, so it is first taken from x = e . (2^frac . rcp((float)total)) -- off from
// p . 2^frac by less than x . 2^-21 <= 2^-14 (one ulp of v_rcp_f32, the rounding of total to float, the product's rounding,
// the final rounding of p itself) -- and that answer is final unless some lane's x lies within 2^-11 of an integer >= 1: then,
// and only then (a wavefront-uniform branch; a few launches in a thousand take it), every lane runs the exact division.
// Bit-identical to always dividing; ~18 instructions per hop less.
__device__ __forceinline__ int lean_weight_code(float e, double total, bool live, const SmCfg &smc, QFmt fa, int maxa)
{
    int kp;
    if (smc.base == QMANN_SOFTMAX_EXP && !smc.shift) {
        const float r2 = __builtin_ldexpf(__builtin_amdgcn_rcpf((float)total), (int)fa.frac);      // (wavefront-uniform value)
        const float x = live ? e * r2 : 0.0f;                          // (an empty story has total = 0)
        const float xr = __builtin_rintf(x);
        const bool near = xr >= 1.0f && __builtin_fabsf(x - xr) <= 4.8828125e-04f;
        if (__builtin_expect(__ballot(near) != 0, 0)) {
            const float p = live ? (float)((double)e / total) : 0.0f;
            kp = (int)__builtin_ldexpf(p, (int)fa.frac);
        } else {
            kp = (int)x;
        }
    } else {
        const float p = live ? sm_quot(e, total, smc) : 0.0f;
        kp = (int)__builtin_ldexpf(p, (int)fa.frac);
    }
    return kp > maxa ? maxa : kp;
}

// The end of a hop, shared by the one-wavefront kernels (this file and hops_mid.h): the read-out code `acc` of column `lane`
// is clamped and published, the linear map H.u runs on the pre-split rows in LDS, and u' = Qa(Qa(Hu) + Qa(o)) replaces `u`.
//   kb_code   Q_bin(u) of column `lane` (the linear map's operand)
//   csc / csh the scan constants still in registers; `reuse`: they are already those of (word length of w[h], frac_bin)
//   acc_of()  the read-out code of column `lane`; called AFTER the linear map's row sums, so that value rows requested from
//             global memory just before this function (the sparse read-out of lean_hop) arrive while those sums are taken
template <bool W7, typename AccOf>
__device__ __forceinline__ void lean_finish_hop(const HopArgs &a, uint32_t h, uint32_t lane, uint8_t *lw, const uint8_t *lmap, float &u,
                                                AccOf acc_of, int kb_code, ScanConst &csc, uint32_t &csh, bool reuse)
{
    constexpr uint32_t LPR = 4;
    const uint32_t sub = lane >> 2, chunk = lane & 3u, D = a.D;
    const QFmt fa = a.act[h], fb = a.bin, fw = a.w[h];
    const int maxa = W7 ? 127 : (1 << (fa.iwl + fa.frac)) - 1;        // (W7: every format of the launch has word length 7)
    const uint32_t wl_w = W7 ? 7u : fw.iwl + fw.frac;
    auto publish_o = [&]() {
        int acc = acc_of();
        acc = acc > maxa ? maxa : (acc < -maxa ? -maxa : acc);
        *(int16_t *)(lw + kLwOc + lane * 2) = (int16_t)acc;
        return acc;
    };

    QM_MARK("linear map + hop update");
    // ---- linear map + hop update ---------------------------------------------------------------------
    if (a.en_lin_map) {
        if (!reuse) {
            wave_sync();                                          // every lane is done with the previous image
            publish_const<W7>(lw, lane, kb_code, wl_w, (int)fb.frac);
        }
        wave_sync();
        if (!reuse) csh = fetch_scan_const(csc, lw, chunk, wl_w);
        const int maxw = (1 << wl_w) - 1;
        const uint32_t n_it = (D + 15u) / 16u;
        int keep = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if ((uint32_t)t < n_it) {                             // wavefront-uniform
                const uint8_t *hr = lmap + h * kLmHopBytes + (t * 16 + sub) * 192u + chunk * 16u;
                const int s = row_lanes_sum<LPR>(lane_sum_split<W7>(*(const i32x4 *)hr, *(const i32x4 *)(hr + 64), *(const i32x4 *)(hr + 128), csc, csh));
                if (chunk == (uint32_t)t) keep = s;
            }
        }
        // lane (sub, chunk) now holds the sum of row o = 16 . chunk + sub
        (void)publish_o();
        wave_sync();                                              // every column's read-out code is in LDS
        const uint32_t o_i = chunk * 16u + sub;
        const int kw = keep > maxw ? maxw : (keep < -maxw ? -maxw : keep);       // Qw of the row sum
        const uint32_t mag = (uint32_t)(kw < 0 ? -kw : kw);
        const uint32_t ma = fa.frac >= fw.frac ? mag << (fa.frac - fw.frac) : mag >> (fw.frac - fa.frac);
        const int lam = ma > (uint32_t)maxa ? maxa : (int)ma;                  // Qa of that value
        int un = (kw < 0 ? -lam : lam) + (int)*(const int16_t *)(lw + kLwOc + o_i * 2);
        un = un > maxa ? maxa : (un < -maxa ? -maxa : un);
        wave_sync();                                              // (kLwUn shares its bytes with the constant images)
        *(float *)(lw + kLwUn + o_i * 4) = qm_scale_down((float)un, fa.frac);
        wave_sync();
        u = (lane < D) ? *(const float *)(lw + kLwUn + lane * 4) : 0.0f;
    } else {
        int un = ((lane < D) ? qm_code(u, fa.iwl, fa.frac) : 0) + publish_o();
        un = un > maxa ? maxa : (un < -maxa ? -maxa : un);
        u = qm_scale_down((float)un, fa.frac);
    }
    wave_sync();                                                  // the next hop rewrites the images
}

// One hop of one query on one wavefront.  u: the hop state, lane c owns column c (updated in place).
//   key_of(j)         this lane's 16-byte piece (chunk = lane & 3) of key row 16 j + (lane >> 2), j = 0..3
//   before_readout()  called once the read-out weights are known, before the value tile `vt` is read
//                     (wait for / produce the tile; must leave the wavefront synchronised)
// `lw`: this wavefront's small arrays (kLwBytes); `lmap` / `etab`: the workgroup's tables.
//   vg / SPARSE       SPARSE: there is no value tile `vt`; the rows whose weight code is non-zero (at most 2^frac of
//                     them, four at the default Q5.2) are fetched from global memory `vg` (this story's rows of the hop's value
//                     plane), the first four requested BEFORE before_readout() and the linear map, which hide their round trip.
//                     At |mem| = 50 the whole-tile read doubled the kernel's HBM traffic and the kernel was bound by it.
//   tail_key          lean_tail_rows(S): this lane's dword (piece lane & 15) of key row (S & ~15) + (lane >> 4)
template <int MODE, int NB, bool W7, bool SPARSE, typename KeyOf, typename BeforeReadout>
__device__ __forceinline__ void lean_hop(const HopArgs &a, const LeanArgs &la, uint32_t h, uint32_t S, uint32_t lane, const uint8_t *vt,
                                         uint8_t *lw, const uint8_t *lmap, const float *etab, float &u, KeyOf key_of,
                                         BeforeReadout before_readout, const uint8_t *vg = nullptr, int tail_key = 0)
{
    constexpr uint32_t Dp = 64, LPR = 4;
    const uint32_t sub = lane >> 2, chunk = lane & 3u, D = a.D;
    const QFmt fa = a.act[h], fm = a.att[h], fb = a.bin, fw = a.w[h];
    const int maxa = W7 ? 127 : (1 << (fa.iwl + fa.frac)) - 1;
    const bool relu = hop_relu(a, h);
    QM_MARK("operand codes + publish");
    // ---- column c: operand codes -----------------------------------------------------------------
    const int kb_code = (lane < D) ? qm_code_or_sign(u, fb.iwl, fb.frac) : 0;        // Q_bin(u): linear map (and fixed scores)
    const uint32_t wl_m = (W7 && MODE == kModeFixed) ? 7u : fm.iwl + fm.frac, wl_w = W7 ? 7u : fw.iwl + fw.frac;
    if (MODE == kModeFixed) {
        int ka = kb_code;
        if (relu && ka < 0) ka = (fb.iwl + fb.frac == 0) ? 1 : 0;                    // see make_scan_const
        publish_const<W7>(lw, lane, ka, wl_m, (int)fb.frac);
    } else {
        const float ua = relu_if(u, relu);
        lw[kLwUb + lane] = (uint8_t)ham_ubyte(ua, fm, lane < D);
    }
    wave_sync();

    QM_MARK("scan");
    // ---- scores ------------------------------------------------------------------------------------
    ScanConst csc;
    uint32_t csh = 0;
    float unit = 1.0f;
    // A story's last 1 .. 4 rows (S mod 16 <= 4) do not get a whole pass of 16 rows x 4 lanes: they take the narrow pass below,
    // 4 rows x 16 lanes x one dword (lean_tail_rows(): fixed-point scores at word length 7, the stock configuration).  At the
    // bAbI cap of 50 rows that is 3 passes + a quarter instead of 4; a 1 .. 4-row story needs no full pass at all.
    const uint32_t S_reg = lean_tail_rows<MODE, W7>(S) ? (S & ~15u) : S;       // rows the regular passes cover
    auto scan = [&](auto lane_sum, int lim, bool wrap = false) {         // wrap: mode 3's final quantiser (appx_clamp, ham_common.h)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j * 16 < (int)S_reg) {                            // wavefront-uniform
                const int v = row_lanes_sum<LPR>(lane_sum(key_of(j)));
                if (chunk == 0) *(int16_t *)(lw + kLwSc + (j * 16 + sub) * 2) = (int16_t)(v > lim ? lim : (v < -lim ? -lim : ((wrap && v == -lim) ? 0 : v)));
            }
        }
    };
    if (MODE == kModeFixed) {
        unit = qm_scale_down(1.0f, fm.frac);
        csh = fetch_scan_const(csc, lw, chunk, wl_m);
        scan([&](const i32x4 x) { return lane_sum_w<W7>(x, csc, csh); }, (1 << wl_m) - 1);
        if (lean_tail_rows<MODE, W7>(S)) {                        // wavefront-uniform
            const uint32_t c16 = lane & 15u, row = S_reg + (lane >> 4);
            const int v = row_lanes_sum<16>(dword_sum7((uint32_t)tail_key, *(const uint32_t *)(lw + kLwE + c16 * 4),
                                                       *(const uint32_t *)(lw + kLwO + c16 * 4), *(const uint32_t *)(lw + kLwS + c16 * 4), 0));
            const int lim = (1 << wl_m) - 1;
            if (c16 == 0 && row < S) *(int16_t *)(lw + kLwSc + row * 2) = (int16_t)(v > lim ? lim : (v < -lim ? -lim : v));
        }
    } else if (mode_is_appx(MODE)) {
        unit = 1.0f / 1024.0f;
        AppxConst c;
        make_appx_const(c, lw + kLwUb, chunk * 16, D);
        const int lim = 1 << (fm.iwl + 10);
        const uint32_t kind = MODE == kModeAppxMq ? ham_kind_of(a, h) : (uint32_t)kHamSame;             // (wavefront-uniform)
        if (kind == kHamFine) scan([&](const i32x4 x) { return appx_lane_sum_k<kHamFine>(x, c); }, lim, true);
        else if (kind == kHamCoarse) scan([&](const i32x4 x) { return appx_lane_sum_k<kHamCoarse>(x, c); }, lim, true);
        else scan([&](const i32x4 x) { return appx_lane_sum(x, c); }, lim, true);
    } else {
        if (MODE == kModeV1Bytes) unit = qm_scale_down(1.0f, NB);
        HamByteConst c;
        make_hambyte_const<MODE, NB>(c, lw + kLwUb, chunk * 16, D);
        scan([&](const i32x4 x) { return hambyte_lane_sum<MODE, NB>(x, c); }, 32767);
    }
    wave_sync();

    QM_MARK("softmax + weight codes");
    // ---- softmax over slots, slot r in lane r ------------------------------------------------------
    const bool live = lane < S;
    const int code = live ? (int)*(const int16_t *)(lw + kLwSc + lane * 2) : 0;
    const SmCfg smc = sm_cfg(a, h);
    float e;
    double total;
    if (MODE == kModeFixed && la.exp_table) {
        const int mxc = wave_max_i32(live ? code : -32768);
        e = live ? etab[h * 256u + (uint32_t)(mxc - code)] : 0.0f;
    } else {
        const float xs = live ? sm_scaled((float)code * unit, smc) : -INFINITY;
        const float mx = wave_max_f32(xs);
        e = live ? sm_exp(xs - mx, smc) : 0.0f;
    }
    total = smc.base == QMANN_SOFTMAX_EXP ? wave_sum_f64((double)e)            // the CUDA kernel's double total
                                          : (double)wave_serial_sum_f32(e, S);     // the CPU softmax's float total, in slot order
    // Q(p) for 0 <= p <= 1: trunc(p . 2^frac), saturated (qm_code without the cases a probability cannot reach)
    int kp = lean_weight_code(e, total, live, smc, fa, maxa);

    QM_MARK("survivors, read-out, key prefetch");
    // ---- read-out over the rows whose weight code is non-zero (lane c owns column c) -----------------
    uint64_t m = __ballot(kp != 0);
    int rr[4], kk[4];
    uint32_t bb[4];
    int acc = 0;
    auto pick = [&]() {                                           // the next (up to) four surviving rows and their weight codes
#pragma unroll
        for (int i = 0; i < 4; i++) {
            rr[i] = m ? __builtin_ctzll(m) : -1;                  // (wavefront-uniform)
            m &= m - 1;                                           // (0 stays 0)
            kk[i] = rr[i] >= 0 ? __builtin_amdgcn_readlane(kp, rr[i] >= 0 ? rr[i] : 0) : 0;
        }
    };
    auto add = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t t = ((uint32_t)kk[i] * (bb[i] & 0x7Fu)) >> fa.frac;   // |Q(p) . v| / 2^frac toward zero
            t = t > (uint32_t)maxa ? (uint32_t)maxa : t;
            acc += (bb[i] & 0x80u) ? -(int)t : (int)t;
        }
    };
    const bool reuse = MODE == kModeFixed && wl_w == wl_m && !relu;
    if (SPARSE) {
        auto fetch = [&]() {
#pragma unroll
            for (int i = 0; i < 4; i++) bb[i] = rr[i] >= 0 ? (uint32_t)vg[(uint32_t)rr[i] * Dp + lane] : 0u;
        };
        pick();
        fetch();                                                  // requested first: older than the key prefetch below
        before_readout();
        lean_finish_hop<W7>(a, h, lane, lw, lmap, u, [&]() {
            add();
            while (m) { pick(); fetch(); add(); }                 // more than four survivors: formats with frac > 2, rarely
            return acc;
        }, kb_code, csc, csh, reuse);
        return;
    }
    before_readout();
    while (m) {                                                   // up to four surviving rows per round, their bytes read together
        pick();
#pragma unroll
        for (int i = 0; i < 4; i++) bb[i] = rr[i] >= 0 ? vt[(uint32_t)rr[i] * Dp + lane] : 0u;
        add();
    }
    lean_finish_hop<W7>(a, h, lane, lw, lmap, u, [&]() { return acc; }, kb_code, csc, csh, reuse);
}

// LIST: the queries to run come from an index list (la.list / la.n_list) -- a separate instantiation, so that the plain kernel
// keeps its registers (the six-wave build sits exactly at its 80-register budget)
template <int MODE, int NB, bool W7, bool SPARSE, int WPS, bool LIST = false>
__global__ void __launch_bounds__(kLeanBlock, WPS)
k_hops_lean(const HopArgs a, const LeanArgs la)
{
    constexpr uint32_t Dp = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));   // uniform: the query's row range lives in SGPRs, which a buffer resource needs
    const uint32_t sub = lane >> 2, chunk = lane & 3u;
    const uint32_t D = a.D, H = a.n_hop;
    float *etab = (float *)smem;                                        // [H][256]
    uint8_t *lmap = smem + (la.exp_table ? H * 1024u : 0u);             // [H][64][3][64] (lean_stage_tables)
    uint8_t *wbase = lmap + (la.lm_in_lds ? H * kLmHopBytes : 0u);
    const uint32_t wslice = la.rows_pad * 64u + kLwBytes;
    uint8_t *vt = wbase + wave * wslice;                                // value tile
    uint8_t *lw = vt + la.rows_pad * 64u;                               // small arrays
    uint32_t n_groups = gridDim.x;
    if (LIST && la.n_other) {                                           // (side by side with the short stories' kernel)
        n_groups = corun_groups(*la.n_other, *la.n_list, gridDim.x, true);
        if (blockIdx.x >= n_groups) return;
    }
    lean_stage_tables(a, la, etab, lmap, tid, kLeanBlock);
    __syncthreads();

    // (rows_total carries the query count, < 2^24: qmann_hops_i8; with an index list the items are list[0 .. *n_list))
    const uint32_t q_stride = n_groups * kLeanWaves, n_query = LIST ? *la.n_list : a.rows_total;
    // The next query's first key tile and its u0 are requested during the current query's last hop (its row offsets
    // a query earlier still), so a wavefront does not sit through a cold HBM round trip at every query start.
    uint32_t qi = blockIdx.x * kLeanWaves + wave;                       // item index; q: the query it names
    if (qi >= n_query) return;
    uint32_t q = LIST ? la.list[qi] : qi;
    uint32_t r0 = a.row_off[q], S;
    {
        // a story longer than the caller's bound is cut to the bound (qmann_batch.h): the value tile holds
        // round16(max_slots) rows per wavefront and max_slots <= 64 here (lean_supported)
        const uint32_t S_in = a.row_off[q + 1] - r0;
        S = S_in < a.max_slots ? S_in : a.max_slots;
    }
    i32x4 kq[4];
    int kt = 0;                                                         // the narrow tail pass's dword (lean_hop)
    float u_next;
    // Keys of hop h of the query whose rows start at `base` (S_ rows): BUFFER loads bounded by the story's own bytes (a raw
    // buffer resource per story and hop: base, size).  A row past the story's end reads as zeros without touching memory, so
    // the loads carry no lane predicate, no saved exec mask and no zero-initialised destination -- round 3's predicated global
    // loads kept five 64-bit lane masks per hop in scalar registers the kernel does not have (spilled to vector lanes and read
    // back, two vector-issue slots each) and cleared 17 registers per hop.  1 368 -> 1 214 vector instructions in the kernel,
    // 125 -> 83 registers, -5 % run time at |mem| = 50.  Two things this needs: (1) the resource must be wavefront-UNIFORM in
    // the compiler's eyes (hence the readfirstlane on the wavefront index above) -- otherwise every load becomes a
    // readfirstlane "waterfall" loop with a vmcnt(0) in front, which is what made the first attempt of the round 2.5 % slower
    // than the global loads; (2) a scheduling barrier behind the requests, so that they stay where they are issued (a hop ahead)
    // instead of sinking towards their first use.
    const uint32_t lane_off = sub * Dp + chunk * 16u;
    auto load_keys_of = [&](uint32_t h, uint32_t base, uint32_t S_) {
        const uint8_t *k0 = (const uint8_t *)a.keys + (size_t)h * a.key_hop_stride + (size_t)base * Dp;
        const bool tail = lean_tail_rows<MODE, W7>(S_);
        const uint32_t S_reg = tail ? (S_ & ~15u) : S_;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)k0, 0, (int)(S_reg * Dp), kRawBufferFlags);
#pragma unroll
        for (int j = 0; j < 4; j++)
            kq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane_off + (uint32_t)j * 16u * Dp), 0, kBufferNt);
        if (MODE == kModeFixed && W7) {
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)k0, 0, (int)(tail ? S_ * Dp : 0u), kRawBufferFlags);
            kt = __builtin_amdgcn_raw_buffer_load_b32(rt, (int)(S_reg * Dp + lane * 4u), 0, kBufferNt);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    load_keys_of(0, r0, S);
    u_next = (lane < D) ? a.u0[q * D + lane] : 0.0f;
    for (; qi < n_query; qi += q_stride) {
        const uint32_t qin = qi + q_stride;
        uint32_t r0n = 0, Sn = 0, qn = 0;
        if (qin < n_query) {
            qn = LIST ? la.list[qin] : qin;
            r0n = a.row_off[qn];
            const uint32_t S_in = a.row_off[qn + 1] - r0n;
            Sn = S_in < a.max_slots ? S_in : a.max_slots;
        }
        float u = u_next;
        for (uint32_t h = 0; h < H; h++) {
            // this hop's value rows: global memory -> this wavefront's LDS tile, 1 KiB (16 rows) per instruction
            const uint8_t *vg = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp;
            if (!SPARSE) {                                                // (SPARSE: no tile, see lean_hop)
                // this hop's value rows: global memory -> this wavefront's LDS tile, 1 KiB (16 rows) per instruction, through a
                // buffer resource as the keys (rows past the story's end arrive as zeros).  The instruction's immediate offset
                // would move the LDS address as well as the global one: the piece's 1 KiB goes into the lane offset.
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)vg, 0, (int)(S * Dp), kRawBufferFlags);
#define QM_TILE_PIECE(J)                                                                                                   \
                if ((J) * 16 < (int)S)                                    /* wavefront-uniform */                          \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (void __attribute__((address_space(3))) *)(vt + (J) * 1024), 16, \
                                                             (int)(lane_off + (J) * 1024u), 0, 0, 0)
                QM_TILE_PIECE(0); QM_TILE_PIECE(1); QM_TILE_PIECE(2); QM_TILE_PIECE(3);
#undef QM_TILE_PIECE
            }
            lean_hop<MODE, NB, W7, SPARSE>(a, la, h, S, lane, vt, lw, lmap, etab, u, [&](int j) { return kq[j]; }, [&]() {
                if (!SPARSE) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the value tile has landed
                    wave_sync();
                }
                // in flight during the read-out and the linear map: the next hop's keys, or the next query's first keys
                // (unconditionally: with nothing to come the resource is empty -- r0n = Sn = 0 -- and the loads bring zeros for free)
                const bool more = h + 1 < H;
                load_keys_of(more ? h + 1 : 0u, more ? r0 : r0n, more ? S : Sn);
                if (!more && qin < n_query) u_next = (lane < D) ? a.u0[(size_t)qn * D + lane] : 0.0f;
            }, vg, kt);
        }
        QM_MARK("end of query");
        if (lane < D) a.u_out[q * D + lane] = relu_if(u, a.en_non_lin != 0);
        r0 = r0n; S = Sn; q = qn;
    }
}

}  // namespace

#include "hops_quad.h"       // stories of <= 16 rows: four queries per wavefront (uses the helpers above)

namespace {

// what the lean kernel covers (everything else keeps the general kernels)
inline bool lean_supported(const HopArgs &a, uint32_t max_slots, uint32_t key_row_bytes)
{
    return a.Dp == 64 && key_row_bytes == 64 && max_slots <= (uint32_t)kWave && !a.tap_codes && !a.tap_scores && !a.tap_probs &&
           !a.tap_o && !a.tap_u && !qm_tuning().no_lean;
}

template <int MODE, int NB, bool W7, bool SPARSE>
inline void launch_lean_w(HopArgs a, uint32_t max_slots, uint32_t n_query, hipStream_t st, const uint32_t *list = nullptr,
                          const uint32_t *n_list = nullptr, const uint32_t *n_other = nullptr)
{
    LeanArgs la{};
    la.list = list; la.n_list = n_list; la.n_other = n_other;
    la.rows_pad = SPARSE ? 0u : ((max_slots ? max_slots : 1u) + 15u) & ~15u;       // (SPARSE: no value tile)
    la.exp_table = (MODE == kModeFixed && a.softmax_base == QMANN_SOFTMAX_EXP && !a.softmax_shift && !a.en_att_scale) ? 1u : 0u;
    la.lm_in_lds = a.en_lin_map ? 1u : 0u;
    a.rows_total = n_query;                                               // (the kernel has no taps: the field carries the query count)
    const size_t lds = (la.exp_table ? a.n_hop * 1024u : 0u) + (la.lm_in_lds ? a.n_hop * kLmHopBytes : 0u) +
                       (size_t)kLeanWaves * (la.rows_pad * 64u + kLwBytes);
    // Persistent workgroups: exactly as many as are resident AT ONCE (rt.h::qm_resident_groups: registers by the build's
    // __launch_bounds__, LDS, the 32-wavefront limit).  Through round 3 the grid was sized by LDS alone (three per CU at bAbI sizes
    // where registers allowed two): the late third ran alone on half-empty CUs.  For THIS kernel that cost nothing measurable (it is
    // bound by vector issue: the late workgroups run twice as fast) -- for the latency-bound embedding kernel the same mistake
    // doubled the run time.
    const uint32_t need = (n_query + kLeanWaves - 1) / kLeanWaves;
    auto go = [&](auto kernel, int wps) {
        if (lds > kLdsDefaultLimit) QM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t resident = qm_resident_groups(kLeanWaves, (unsigned)wps, lds);
        kernel<<<need < resident ? need : resident, kLeanBlock, lds, st>>>(a, la);
    };
    // the six-wave build where it buys a third workgroup per CU (every attention mode: the Hamming forms need 81-91 registers in
    // the four-wave build and all but the EN_MQ mode-3 form fit 80 without a spill; 50-slot synthetic memories in an interleaved
    // A/B: mode 3 0.91 -> 0.84 ms, weighted Hamming 0.70 -> 0.65 ms)
    const bool tight = !qm_tuning().no_tight &&
                       qm_resident_groups(kLeanWaves, kLeanWpsTight, lds) > qm_resident_groups(kLeanWaves, kLeanWpsWide, lds);
    if (list) go(k_hops_lean<MODE, NB, W7, SPARSE, kLeanWpsWide, true>, kLeanWpsWide);       // (the long stories of a split batch: few)
    else if (tight) go(k_hops_lean<MODE, NB, W7, SPARSE, kLeanWpsTight>, kLeanWpsTight);
    else go(k_hops_lean<MODE, NB, W7, SPARSE, kLeanWpsWide>, kLeanWpsWide);
}

template <int MODE, int NB>
inline void launch_lean_all(const HopArgs &a, uint32_t max_slots, uint32_t n_query, hipStream_t st, const uint32_t *list = nullptr,
                            const uint32_t *n_list = nullptr, const uint32_t *n_other = nullptr);

// Short-memory launches: stories of at most 16 rows take the four-queries-per-wavefront kernel (hops_quad.h), longer ones the
// one-wavefront-per-query kernel below.  A batch whose bound allows both (the 20-task set: up to 64 rows, 91 % of the stories
// <= 16) is split on the device into two index lists first (k_split_by_length); the two kernels then run each over its list --
// one after the other on the stream, or side by side on two streams where that pays (below).  Batches of at most
// QMANN_QUAD_MIN_QUERIES stories keep one story per wavefront.  QMANN_NO_QUAD keeps everything on the lean kernel (A/B).
template <int MODE, int NB>
inline void launch_lean(const HopArgs &a, uint32_t max_slots, uint32_t n_query, hipStream_t st)
{
    // Small batches keep one story per wavefront: the GPU holds 6 144 wavefronts at once, so up to a few thousand stories each has
    // one of its own and the batch takes ONE story's latency, where four stories per wavefront take their linear maps in turn
    // (task-1 forward, interleaved: 256 stories 34.8 -> 27.2 us, 1 024 36.8 -> 29.3, 4 096 40.4 -> 38.9, 8 192 53.7 -> 50.0,
    // 16 384 72.1 <- 75.2: the quad form from there on; 64-story serving batches replayed from a graph 39.2 -> 31.1 us)
    if (!quad_supported(a, MODE, max_slots, n_query) || n_query <= qm_tuning().quad_min_queries) {
        launch_lean_all<MODE, NB>(a, max_slots, n_query, st);
        return;
    }
    if (max_slots <= kQuadSlots) {
        launch_quad<MODE, NB, 1>(a, QuadArgs{nullptr, nullptr, n_query, nullptr, nullptr}, n_query, st);
        return;
    }
    // A batch is split only where short stories can be many: mean length (known from the plane size; tied hops carry none and
    // take the bAbI guess, stories short next to their cap) within the short form's 16 rows.  A batch of long stories -- the
    // |mem| = 50 shape of BASELINE's metric -- goes whole through the four-chunk form (QMANN_NO_QUAD_LONG: the lean kernel).
    const uint32_t mean_slots = (a.rows_total && n_query) ? a.rows_total / n_query : max_slots / 8u;
    if (mean_slots > kQuadSlots) {
        // (fixed-point scores only: the form is 2 % ahead of the lean kernel at 50 rows -- 7 % fewer vector instructions per query,
        // four wavefronts per SIMD against six -- which does not pay for ten more instantiations in the Hamming modes)
        if constexpr (MODE == kModeFixed) {
            if (!qm_tuning().no_quad_long) { launch_quad<MODE, NB, 4>(a, QuadArgs{nullptr, nullptr, n_query, nullptr, nullptr}, n_query, st); return; }
        }
        launch_lean_all<MODE, NB>(a, max_slots, n_query, st);
        return;
    }
    uint32_t *ws = nullptr;
    {
        const QmSplitReady r = qm_split_ready;                                // (prepared by the host model beside the story embedding)
        qm_split_ready = QmSplitReady{nullptr, 0, 0, nullptr};
        if (r.ws && r.row_off == a.row_off && r.n_query == n_query && r.max_slots == max_slots) ws = r.ws;
    }
    if (!ws) ws = split_lists(a.row_off, n_query, max_slots, st, st);
    if (!ws) { launch_lean_all<MODE, NB>(a, max_slots, n_query, st); return; }
    // Large batches: the two kernels SIDE BY SIDE -- the long stories' kernel on a second stream (forked and joined by events, so
    // the caller's stream sees one ordered step and a stream capture takes both branches).  Both are bound by vector issue and
    // latency, not by a shared unit: together they fill the issue slots either leaves empty alone.  How the CUs' room is divided
    // is decided on the device from the two list lengths (corun_groups above).  Kernel trace on the 20-task set (262 000
    // queries, 9 % long): 433 us for both (they end within 10 us of each other) against 370 + 98 us in sequence; forward
    // 1.033 -> 0.994 ms (interleaved A/B).  The pair pays where there is enough to overlap (tools/corun_mix.py, 6- and 30-row
    // stories, 262 144 queries, side by side against in sequence: 5 % long -0.9 %, 9 % -6.7 %, 15 % -5.0 %): it is taken when the
    // LAST split batch on this stream -- its short-story kernel stores the two counts to pinned host memory, no copy, no
    // synchronisation -- gave the long stories 0.12 .. 0.45 of the work; unknown (first batch) or outside (the same tool: 0 / 1 /
    // 25 / 35 % long: +-0.2 %): in sequence.  QMANN_NO_CORUN: always in sequence.
    uint32_t *publish = nullptr;
    if (n_query >= kQmCorunMinQueries && !qm_tuning().no_corun) {
        QmSide *sd = qm_side_stream(st);
        bool pays = false;
        if (sd) publish = (uint32_t *)sd->last_counts;
        if (sd) {
            const uint32_t n_short = sd->last_counts[0], n_long = sd->last_counts[1];
            if (n_short != 0xFFFFFFFFu && n_long != 0xFFFFFFFFu) {
                const float cl = 2.7f * (float)n_long, share = cl / (cl + (float)n_short + 1.0f);
                pays = share >= 0.12f && share <= 0.45f;
            }
        }
        if (pays) {
            QM_HIP(hipEventRecord(sd->fork, st));
            QM_HIP(hipStreamWaitEvent(sd->side, sd->fork, 0));
            launch_lean_all<MODE, NB>(a, max_slots, n_query, sd->side, ws + 2 + n_query, ws + 1, ws);
            QM_HIP(hipEventRecord(sd->join, sd->side));
            launch_quad<MODE, NB, 1>(a, QuadArgs{ws + 2, ws, n_query, ws + 1, publish}, n_query, st);
            QM_HIP(hipStreamWaitEvent(st, sd->join, 0));
            return;
        }
    }
    launch_quad<MODE, NB, 1>(a, QuadArgs{ws + 2, ws, n_query, nullptr, publish}, n_query, st);
    // (the few long stories of such a batch: the lean kernel -- the four-chunk quad form measured 3 % SLOWER on the joint forward)
    launch_lean_all<MODE, NB>(a, max_slots, n_query, st, ws + 2 + n_query, ws + 1);
}

template <int MODE, int NB>
inline void launch_lean_all(const HopArgs &a, uint32_t max_slots, uint32_t n_query, hipStream_t st, const uint32_t *list, const uint32_t *n_list,
                            const uint32_t *n_other)
{
    bool w7 = true;
    for (uint32_t h = 0; h < a.n_hop; h++)
        w7 = w7 && (MODE != kModeFixed || a.att[h].iwl + a.att[h].frac == 7) && (!a.en_lin_map || a.w[h].iwl + a.w[h].frac == 7) &&
             a.act[h].iwl + a.act[h].frac == 7;
    // Sparse read-out (only the value rows that survive Q(p) are fetched, lean_hop) where those rows -- at most 2^frac_act per
    // hop -- are a small part of the longest story; otherwise the whole value tile is copied to LDS at the start of the hop.
    // QMANN_LEAN_SPARSE=0 / 1 forces the choice (A/B).
    // Measured (A/B, 262 144 queries): 50-row stories +16 % (the whole-tile read doubled the HBM traffic of a kernel that was bound
    // by it), the real 20-task set (2 .. 64 rows, mean 9.3) -1.4 % (the tile arrives during the scan, the sparse rows only during
    // the linear map): the choice goes by the MEAN story length, which the launch knows from the plane size.
    uint32_t surv = 1;
    for (uint32_t h = 0; h < a.n_hop; h++) surv = surv > (1u << a.act[h].frac) ? surv : (1u << a.act[h].frac);
    uint32_t mean_slots = (a.rows_total && n_query) ? a.rows_total / n_query : max_slots / 8u;     // (tied hops: no plane size; real stories are short next to their cap)
    if (list) mean_slots = mean_slots > kQuadSlots + 1u ? mean_slots : kQuadSlots + 1u;          // (the long stories of a split batch: each has > 16 rows)
    bool sparse = surv * 4u <= mean_slots;
    // ... and wherever the value tiles would cost a workgroup per CU: 8 wavefronts x 64 rows x 64 bytes = 32 KB of tiles leave
    // room for two workgroups, the sparse variant (no tile) for three at the six-wave register budget.  The real 20-task set
    // (max 64 rows, mean 9.3): +4 % on the mode-3 forward, +5 % on the weighted-Hamming one (A/B, QMANN_LEAN_SPARSE=0 / 1);
    // task 1 (max 10 rows: 8 KB of tiles, three workgroups either way) keeps the tile.
    {
        const size_t fixed_lds = (a.en_lin_map ? a.n_hop * kLmHopBytes : 0u) + (size_t)kLeanWaves * kLwBytes + a.n_hop * 1024u;
        const size_t tile_lds = fixed_lds + (size_t)kLeanWaves * ((((max_slots ? max_slots : 1u) + 15u) & ~15u) * 64u);
        if (!qm_tuning().no_tight && qm_resident_groups(kLeanWaves, kLeanWpsTight, tile_lds) < qm_resident_groups(kLeanWaves, kLeanWpsTight, fixed_lds))
            sparse = true;
    }
    if (qm_tuning().lean_sparse >= 0) sparse = qm_tuning().lean_sparse == 1;
    if (w7) { if (sparse) launch_lean_w<MODE, NB, true, true>(a, max_slots, n_query, st, list, n_list, n_other); else launch_lean_w<MODE, NB, true, false>(a, max_slots, n_query, st, list, n_list, n_other); }
    else { if (sparse) launch_lean_w<MODE, NB, false, true>(a, max_slots, n_query, st, list, n_list, n_other); else launch_lean_w<MODE, NB, false, false>(a, max_slots, n_query, st, list, n_list, n_other); }
}

}  // namespace
