// hops_common.h -- pieces shared by the fused hop kernels (batch_hops.hip: fixed-point dot
// attention; batch_hops_ham.hip: Hamming-family attention): the streaming row scan, the LDS
// carve-up, block reductions, and the stages after the softmax (sparse exact read-out, linear
// map, hop update).
#pragma once
#include "qfmt.h"
#include "rt.h"
#include "../../include/qmann_batch.h"

// launch arguments of every fused hop kernel (a named type: it crosses translation units)
namespace qmann {
struct HopArgs {
    const int8_t *keys;
    const int8_t *vals;
    size_t hop_stride;       // bytes between hop planes of the value memory (and of int8 keys)
    size_t key_hop_stride;   // bytes between hop planes of the key memory (packed-code modes)
    const uint32_t *row_off;
    const float *u0;
    float *u_out;
    int32_t *tap_codes;
    float *tap_scores;
    float *tap_probs;
    float *tap_o;
    float *tap_u;
    const int8_t *lin_map[QMANN_MAX_HOP];
    uint32_t rows_total;
    uint32_t max_slots;      // the caller's bound on slots per query: LDS is sized by it, longer stories are cut to it
    uint32_t n_hop, D, Dp, softmax_base, en_lin_map;
    QFmt act[QMANN_MAX_HOP], w[QMANN_MAX_HOP], att[QMANN_MAX_HOP], bin;
    uint32_t softmax_shift, en_att_scale;     // in-hop softmax variants (qmann_net)
    uint32_t en_non_lin;                      // EN_NON_LINEARITY: see hop_relu() below
    float att_scale[QMANN_MAX_HOP];
    uint32_t ham_kinds;      // mode 3 under mixed quantisation: 2 bits per hop, kHamSame / kHamCoarse / kHamFine (ham_common.h)
};
}  // namespace qmann
using qmann::HopArgs;

namespace {


constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kUnrollDefault = 4;  // 16-byte loads per lane and tile in the key scan (two tiles in flight)
// dynamic LDS a kernel may use without hipFuncSetAttribute (static arrays come on top, hence the margin)
constexpr size_t kLdsDefaultLimit = 48 * 1024;
constexpr int kLiveCap = 256;     // surviving rows kept in LDS (at most 2^frac <= 128 can exist)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS carve-up (bytes); one dynamic allocation, 16-byte aligned pieces
constexpr uint32_t kOffHist = 0;                                   // u32 [4][256]
constexpr uint32_t kOffPtab = kOffHist + kWaves * 256 * 4;         // float [256]  p per score code
constexpr uint32_t kOffU = kOffPtab + 256 * 4;                     // float [256]  current u
constexpr uint32_t kOffO = kOffU + 256 * 4;                        // float [256]  read-out o
constexpr uint32_t kOffKu = kOffO + 256 * 4;                       // s16   [256]  Q_bin(u) codes
constexpr uint32_t kOffKp = kOffKu + 256 * 2;                      // u8    [256]  Q(p) per score code
constexpr uint32_t kOffLiveRow = kOffKp + 256;                     // u32   [kLiveCap]
constexpr uint32_t kOffLiveKp = kOffLiveRow + kLiveCap * 4;        // u8    [kLiveCap]
constexpr uint32_t kOffMisc = kOffLiveKp + kLiveCap;               // u32   [16]
constexpr uint32_t kOffRed = kOffMisc + 64;                        // double[8]
constexpr uint32_t kOffScores = kOffRed + 64;                      // i8    [slots]
static_assert(kOffScores % 16 == 0, "score bytes must start 16-byte aligned");


// raw buffer resources (bounds-checked loads: hops_mid.h)
constexpr int kRawBufferFlags = 0x00020000;         // dword 3 of a raw buffer resource on gfx9 / CDNA: 32-bit data format, no swizzle
constexpr int kBufferNt = 2;                        // cache policy of a buffer load: nt (streamed once)

// Stage marks for tools/stage_budget.py (a build with -DQM_STAGE_MARKS, never the shipped one): a comment in the assembly plus a
// scheduling barrier, so that the instructions between two marks are those of the stage the first one names.
#ifdef QM_STAGE_MARKS
#define QM_MARK(NAME) do { asm volatile("; QM_MARK " NAME ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define QM_MARK(NAME) do { } while (0)
#endif

// Stage CLOCKS for tools/stage_clocks.py (a build with -DQM_STAGE_CLOCKS, never the shipped one): where does a wavefront's TIME go
// in a latency-bound kernel?  QM_CLK(i) charges the shader cycles (s_memtime) since the wavefront's previous QM_CLK to stage i --
// wall time of that wavefront, waits and the other wavefronts' turns included --; QM_CLK_FLUSH() adds the wavefront's sums to
// qm_stage_clk[] (one atomic per stage and wavefront at kernel end), which qmann_debug_stage_clocks() reads and clears.
#ifdef QM_STAGE_CLOCKS
constexpr int kQmClkStages = 16;
__device__ unsigned long long qm_stage_clk[kQmClkStages];
#define QM_CLK_DECL() unsigned long long qm_clk_acc_[kQmClkStages] = {}; unsigned long long qm_clk_last_ = __builtin_amdgcn_s_memtime()
#define QM_CLK(I) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); qm_clk_acc_[I] += t_ - qm_clk_last_; qm_clk_last_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define QM_CLK_FLUSH() do { if ((threadIdx.x & 63u) == 0u) for (int i_ = 0; i_ < kQmClkStages; i_++) if (qm_clk_acc_[i_]) atomicAdd(&qm_stage_clk[i_], qm_clk_acc_[i_]); } while (0)
#else
#define QM_CLK_DECL() do { } while (0)
#define QM_CLK(I) do { } while (0)
#define QM_CLK_FLUSH() do { } while (0)
#endif

template <int CTRL>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// sum over the LPR adjacent lanes that share a row (every lane ends with the total)
template <int LPR>
__device__ __forceinline__ int row_lanes_sum(int v)
{
    if (LPR >= 2) v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    if (LPR >= 4) v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    if (LPR >= 8) v = dpp_add<0x141>(v);         // row_half_mirror
    if (LPR >= 16) v = dpp_add<0x140>(v);        // row_mirror
    return v;
}

// packed u16 x u16 -> u16 multiply that saturates at 0xFFFF
__device__ __forceinline__ uint32_t pk_mul_sat_u16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Per-lane constants of the key scan for the 16 columns a lane owns (4 dwords of 4 columns).
// Keys are sign-magnitude bytes.  With M = 2^wl - 1 the largest code of the attention format and
// f = frac of the query format, the reference's per-product term is
//     sgn(k) sgn(u) . min( floor(|k| |u| / 2^f), M )          (lib/layer_cuda.cu:120)
// Pre-shifting |u| left by s = 16 - wl - f turns the floor into a plain logical shift and makes
// the clamp coincide with 16-bit unsigned saturation:
//     min(floor(|k||u| / 2^f), M) == sat_u16(|k| . (|u| << s)) >> (16 - wl)
// (the product reaches 2^16 exactly when floor(.) reaches M + 1).
struct ScanConst {
    uint32_t ue[4];   // (|u| << s) for columns 4d+0 (low half) and 4d+2 (high half), saturated to 0xFFFF
    uint32_t uo[4];   // columns 4d+1 and 4d+3
    uint32_t s7[4];   // 0x80 in byte i where u[4d+i] < 0
};

// sum over this lane's 16 columns of Qm(Qm(k) . Qv(u)), in units of 2^-frac_m
__device__ __forceinline__ int lane_row_sum(const i32x4 x, const ScanConst &c, uint32_t sh)
{
    int acc = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t w = (uint32_t)x[d];
        const uint32_t ev = w & 0x007F007Fu;                    // |k| of bytes 0 and 2
        const uint32_t od = (w >> 8) & 0x007F007Fu;             // |k| of bytes 1 and 3
        const u16x2 te = __builtin_bit_cast(u16x2, pk_mul_sat_u16(ev, c.ue[d])) >> (unsigned short)sh;
        const u16x2 to = __builtin_bit_cast(u16x2, pk_mul_sat_u16(od, c.uo[d])) >> (unsigned short)sh;
        const uint32_t tb = __builtin_bit_cast(uint32_t, te) | (__builtin_bit_cast(uint32_t, to) << 8);
        const uint32_t sb = (w ^ c.s7[d]) & 0x80808080u;        // sign of each product
        // bytes +1 / -1: v_perm_b32 yields 0xFF for a selector byte >= 13 (0x80 here) and source
        // byte 0 (= 0x01) for selector 0
        const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, sb);
        acc = __builtin_amdgcn_sdot4((int)tb, (int)sg, acc, false);
    }
    return acc;
}

// ScanConst of the 16 columns starting at c0 from the Q(v) codes `ku` of the vector; wl = word
// length (iwl + frac) of the product format, fv = frac of the vector format.  Returns the result shift.
// `relu`: the codes are those of RELU(u) instead of u (negative codes become 0; a binarised operand,
// format (0,0), becomes +1 everywhere because RELU(u) >= 0) -- see hop_relu().
__device__ __forceinline__ uint32_t make_scan_const(ScanConst &c, const short *ku, uint32_t c0, uint32_t wl, int fv,
                                                    bool relu = false, bool sign_fmt = false)
{
    const uint32_t sh = 16u - wl;                       // result shift
    const int pre = (int)sh - fv;                       // |u| << pre (>= 0 since wl + fv <= 14)
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t m[4], sg = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int k = ku[c0 + 4 * d + i];
            if (relu && k < 0) k = sign_fmt ? 1 : 0;
            const uint32_t a = (uint32_t)(k < 0 ? -k : k) << pre;
            m[i] = a > 0xFFFFu ? 0xFFFFu : a;
            sg |= (k < 0 ? 0x80u : 0u) << (8 * i);
        }
        c.ue[d] = m[0] | (m[2] << 16);
        c.uo[d] = m[1] | (m[3] << 16);
        c.s7[d] = sg;
    }
    return sh;
}

// Word length 7 (every 8-bit configuration): the per-product clamp at 127 is the SIGNED 16-bit saturation of
// |k| . (|u| << (8 - frac_v)) (32767 >> 8 = 127), and the quotient is then the product's high byte: no shift, and
// one v_perm_b32 gathers the four high bytes.  9 VALU operations per 4 key bytes instead of 11.
__device__ __forceinline__ uint32_t pk_mul_sat_i16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// four columns: key dword w against the constants ue / uo / s7 of those columns
__device__ __forceinline__ int dword_sum7(uint32_t w, uint32_t ue, uint32_t uo, uint32_t s7, int acc)
{
    const uint32_t pe = pk_mul_sat_i16(w & 0x007F007Fu, ue);                    // bytes 0 and 2
    const uint32_t po = pk_mul_sat_i16((w >> 8) & 0x007F007Fu, uo);             // bytes 1 and 3
    const uint32_t tb = __builtin_amdgcn_perm(po, pe, 0x07030501u);             // the four high bytes, in column order
    const uint32_t sb = (w ^ s7) & 0x80808080u;
    const uint32_t sg = __builtin_amdgcn_perm(0x01010101u, 0x01010101u, sb);
    return __builtin_amdgcn_sdot4((int)tb, (int)sg, acc, false);
}
__device__ __forceinline__ int lane_row_sum7(const i32x4 x, const ScanConst &c)
{
    int acc = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) acc = dword_sum7((uint32_t)x[d], c.ue[d], c.uo[d], c.s7[d], acc);
    return acc;
}
// make_scan_const for that form: |u| << (8 - fv), saturated at 0x7FFF
__device__ __forceinline__ void make_scan_const7(ScanConst &c, const short *ku, uint32_t c0, int fv, bool relu = false, bool sign_fmt = false)
{
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t m[4], sg = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int k = ku[c0 + 4 * d + i];
            if (relu && k < 0) k = sign_fmt ? 1 : 0;
            const uint32_t a = (uint32_t)(k < 0 ? -k : k) << (8 - fv);
            m[i] = a > 0x7FFFu ? 0x7FFFu : a;
            sg |= (k < 0 ? 0x80u : 0u) << (8 * i);
        }
        c.ue[d] = m[0] | (m[2] << 16);
        c.uo[d] = m[1] | (m[3] << 16);
        c.s7[d] = sg;
    }
}

template <bool NT>
__device__ __forceinline__ i32x4 load16(const uint8_t *p)
{
    if (NT) return __builtin_nontemporal_load((const i32x4 *)p);
    return *(const i32x4 *)p;
}

// Memories shorter than one tile (bAbI-sized stories): one guarded pass, rows spread over the
// four wavefronts.  Not a bandwidth path.
template <int LPR, typename RowSum, typename Retire>
__device__ __forceinline__ void scan_rows_short(const uint8_t *__restrict__ kb, uint32_t S, RowSum row_sum,
                                                Retire retire, uint32_t lane, uint32_t wave, uint32_t nwaves)
{
    constexpr uint32_t RPW = kWave / LPR;
    constexpr uint32_t Dp = LPR * 16;
    const uint32_t sub = lane / LPR, chunk = lane % LPR;
    for (uint32_t base = wave * RPW; base < S; base += nwaves * RPW) {
        const uint32_t r = base + sub;
        i32x4 x = {0, 0, 0, 0};
        if (r < S) x = *(const i32x4 *)(kb + (size_t)r * Dp + chunk * 16);
        const int v = row_lanes_sum<LPR>(row_sum(x));
        if (chunk == 0 && r < S) retire(r, v);
    }
}

// One wavefront streams tiles of UN x (64 / LPR) rows; loads of the next tile are issued before
// the current one is reduced (addresses are clamped to the last row instead of being predicated,
// so the loop body has no divergent control flow and the compiler can count its vmcnt waits).
template <int LPR, int UN, bool NT, int NW, typename RowSum, typename Retire>
__device__ __forceinline__ void scan_rows(const uint8_t *__restrict__ kb, uint32_t S, RowSum row_sum, Retire retire,
                                          uint32_t lane, uint32_t wave)
{
    constexpr uint32_t nwaves = NW;                // wavefronts sharing the rows of this query
    constexpr uint32_t RPW = kWave / LPR;          // rows per wavefront instruction
    constexpr uint32_t TILE = RPW * UN;            // rows per wavefront iteration
    constexpr uint32_t Dp = LPR * 16;
    const uint32_t sub = lane / LPR, chunk = lane % LPR;
    const uint32_t n_tiles = (S + TILE - 1) / TILE;
    const uint8_t *lane_base = kb + chunk * 16;

    // A tile that would run past the last row is moved back to end exactly at row S (the rows it
    // then repeats are skipped when retiring), so every load is in range with no per-row clamping
    // and one tile needs a single address plus immediate offsets.  Needs S >= TILE.
    auto tile_start = [&](uint32_t base) { return base + TILE <= S ? base : S - TILE; };
    auto issue = [&](i32x4 (&x)[UN], uint32_t base) {
        const uint8_t *p = lane_base + (size_t)(tile_start(base) + sub) * Dp;
#pragma unroll
        for (int j = 0; j < UN; j++) x[j] = load16<NT>(p + (size_t)j * RPW * Dp);
    };
    auto reduce = [&](const i32x4 (&x)[UN], uint32_t base) {
        int s[UN];
#pragma unroll
        for (int j = 0; j < UN; j++) s[j] = row_lanes_sum<LPR>(row_sum(x[j]));
        // after the butterfly every lane of a row group holds that row's sum: lane (sub, chunk)
        // retires row j = chunk (+ LPR, ...) so that all 64 lanes store at once
        const uint32_t start = tile_start(base);
#pragma unroll
        for (int j0 = 0; j0 < UN; j0 += LPR) {
            int v = s[j0];
#pragma unroll
            for (int t = 1; t < LPR && j0 + t < UN; t++) v = (chunk == (uint32_t)t) ? s[j0 + t] : v;
            const uint32_t j = j0 + chunk;
            const uint32_t r = start + j * RPW + sub;
            if (j < (uint32_t)UN && r >= base) retire(r, v);
        }
    };

    if (wave >= n_tiles) return;
    i32x4 xa[UN], xb[UN];
    uint32_t base = wave * TILE;
    constexpr uint32_t STEP = nwaves * TILE;
    issue(xa, base);
    for (uint32_t t = wave; t < n_tiles; t += 2 * nwaves) {
        issue(xb, base + STEP);
        reduce(xa, base);
        if (t + nwaves >= n_tiles) break;
        issue(xa, base + 2 * STEP);
        reduce(xb, base + STEP);
        base += 2 * STEP;
    }
}

// sign-magnitude byte -> integer code
__device__ __forceinline__ int sm_decode(uint8_t b)
{
    const int m = b & 0x7F;
    return (b & 0x80) ? -m : m;
}

// block reductions; workgroups are 1 or 4 wavefronts (blockDim.x = 64 or 256)
template <typename T>
__device__ __forceinline__ T block_max(T v, T *scratch, uint32_t lane, uint32_t wave)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    const uint32_t nwaves = blockDim.x / kWave;
    if (nwaves == 1) return v;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T r = scratch[0];
    for (uint32_t i = 1; i < nwaves; i++) r = scratch[i] > r ? scratch[i] : r;
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_sum_double(double v, double *scratch, uint32_t lane, uint32_t wave)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const uint32_t nwaves = blockDim.x / kWave;
    if (nwaves == 1) return v;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
    for (uint32_t i = 1; i < nwaves; i++) r += scratch[i];
    __syncthreads();
    return r;
}

// EN_NON_LINEARITY (MemN2N/define.h, a RELU layer non_lin[h] after every sum_vec, MemN2N.c:894-896): from
// the second hop on the ATTENTION reads RELU(sv[h-1]) (:2435-2437) while lin_map[h] keeps reading sv[h-1]
// itself (:2471-2473), and the answer layer reads RELU(sv[H-1]) (:2535-2537).  The kernels therefore keep
// the sum_vec state and apply the RELU where it is read.
__device__ __forceinline__ bool hop_relu(const HopArgs &a, uint32_t h) { return a.en_non_lin && h > 0; }
__device__ __forceinline__ float relu_if(float x, bool on) { return (on && !(x > 0.0f)) ? 0.0f : x; }

// ---- softmax variants over slots ----------------------------------------------------------------
// base EXP : e^(x-max), double normaliser, float quotient of the double division (lib/layer_cuda.cu:2006-2042);
//            shift-based: divided by the integer llrint(log2(total)) instead (:2038)
// base POW2: 2^(x-max), float quotient (lib/layer.c:1225-1243); shift-based: 2^(x-max+1) (:1217)
// base PLAN: four-segment piece-wise linear exp (lib/common.c:51-73, table lib/common.h:270-286)
// optional scale layer in front: x = score . w, one float product (lib/layer_cuda.cu:1551-1558, 4822)
struct SmCfg {
    uint32_t base;
    bool shift, en_scale;
    float scale;
};
__device__ __forceinline__ SmCfg sm_cfg(const HopArgs &a, uint32_t h)
{
    return SmCfg{a.softmax_base, a.softmax_shift != 0, a.en_att_scale != 0, a.att_scale[h]};
}
__device__ __forceinline__ float sm_scaled(float score, const SmCfg &c) { return c.en_scale ? score * c.scale : score; }
__device__ __forceinline__ float sm_exp_plan(float x)
{
    float o = 0.597226f * x + 0.933989f;
    o = fmaxf(o, 0.141642f * x + 0.43981f);
    o = fmaxf(o, 0.070265f * x + 0.10888f);
    return fmaxf(o, 0.0f * x + 0.0f);
}
// x = (scaled) score - max
__device__ __forceinline__ float sm_exp(float x, const SmCfg &c)
{
    if (c.base == QMANN_SOFTMAX_EXP) return expf(x);
    if (c.base == QMANN_SOFTMAX_EXP_PLAN) return sm_exp_plan(x);
    return exp2f(c.shift ? x + 1.0f : x);
}
__device__ __forceinline__ float sm_quot(float e, double total, const SmCfg &c)
{
    if (c.base == QMANN_SOFTMAX_EXP)
        return c.shift ? e / (float)llrintf(log2f((float)total)) : (float)((double)e / total);
    return e / (float)total;
}

// The reference's CPU softmax -- the 2^x and exp_plan bases, lib/layer.c:1195-1243 -- adds its terms to a FLOAT total in
// slot order (`float tot`, :1161).  Where a memory fits a wavefront (slot r in lane r) that sum is reproduced as it is: S
// dependent float additions.  (It matters at the truncation steps of Q(p): with one dominant slot and a runner-up 2^-24
// below it the float total stays 1 and p = 1 exactly, a double total gives p = 1 - 2^-23 and Q(p) one code less.)  The
// streaming kernels take the same serial float total for those bases through wave_serial_total_f32 / block_serial_total_f32
// below (one wavefront walks the slots, one dependent addition per slot; its measured cost is in DESIGN.md section 5).
__device__ __forceinline__ float wave_serial_sum_f32(float e, uint32_t S)
{
    float tot = 0.0f;
    for (uint32_t r = 0; r < S; r++) tot += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), (int)r));
    return tot;
}

// The same total for memories of any length: ONE wavefront walks the slots 64 at a time.  Every lane fetches its slot's term
// (e_of(slot): a table look-up or an exp, 64 in flight), then the 64 dependent additions run in the last lane of each 16-lane
// row through DPP row shifts (lane 15 of a row reads lane 15 - k), the four rows one after the other, a row's sum handed to the
// next row by row_bcast:15 and a group's to the next group through lane 63: one vector instruction per slot, and exactly the
// float additions of lib/layer.c:1236 in their order.  (An order-free double sum -- what these kernels did through round 3 --
// agrees within 1e-7 relative, which is enough to move a weight code where p sits on a truncation step of Q(p).)
// Every lane of the calling wavefront gets the total.
template <typename EOf>
__device__ __forceinline__ float wave_serial_total_f32(uint32_t S, uint32_t lane, EOf e_of)
{
    float t = 0.0f;
    for (uint32_t g = 0; g < S; g += (uint32_t)kWave) {
        const uint32_t r = g + lane;
        const float e = r < S ? e_of(r) : 0.0f;                        // (+0 leaves the sum as it is)
        const int eb = __builtin_bit_cast(int, e);
#define QM_SER_STEP(K) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, eb, 0x110 + K, 0xF, 0xF, true))
#define QM_SER_ROW()                                                                                                   \
        QM_SER_STEP(15); QM_SER_STEP(14); QM_SER_STEP(13); QM_SER_STEP(12); QM_SER_STEP(11); QM_SER_STEP(10); QM_SER_STEP(9);    \
        QM_SER_STEP(8); QM_SER_STEP(7); QM_SER_STEP(6); QM_SER_STEP(5); QM_SER_STEP(4); QM_SER_STEP(3); QM_SER_STEP(2);        \
        QM_SER_STEP(1); t += e
#define QM_SER_NEXT(ROWMASK) t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, t), __builtin_bit_cast(int, t), 0x142, ROWMASK, 0xF, false))
        QM_SER_ROW(); QM_SER_NEXT(0x2);                                 // row 0, its sum to lane 31
        QM_SER_ROW(); QM_SER_NEXT(0x4);
        QM_SER_ROW(); QM_SER_NEXT(0x8);
        QM_SER_ROW();                                                   // lane 63: the total so far
#undef QM_SER_NEXT
#undef QM_SER_ROW
#undef QM_SER_STEP
        t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
    }
    return t;
}
// ... for a workgroup: wavefront 0 walks, the others wait; `slot` one LDS double all of them can see
template <typename EOf>
__device__ __forceinline__ double block_serial_total_f32(uint32_t S, uint32_t lane, uint32_t wave, double *slot, EOf e_of)
{
    __syncthreads();                                                    // the tables e_of reads are complete
    if (wave == 0) {
        const float t = wave_serial_total_f32(S, lane, e_of);
        if (lane == 0) slot[0] = (double)t;
    }
    __syncthreads();
    const double r = slot[0];
    __syncthreads();
    return r;
}

// Stages after the softmax, shared by every attention mode:
//   read-out  o[c] = Qa( sum_r Qa( Qa(p[r]) . Qa(C[r][c]) ) )   (lib/layer_cuda.cu:547-635 via :2430/:2512)
//             over the rows whose weight code Q(p) is non-zero (the others contribute exact zeros),
//   lin_map   lu = Qw( sum_i Qw( Qw(H[o][i]) . Qbin(u[i]) ) )     (lib/layer_cuda.cu:49-83 via :3184)
//   update    u' = Qa( Qa(lu) + Qa(o) )                          (lib/layer_cuda.cu:1535-1542)
// `ku` holds Q_bin(u) codes, `kp_of_row` gives the weight code of a row (used only if the
// survivor list overflowed).  Ends with a barrier; u_f holds u' afterwards.
template <uint32_t Dp, typename KpOfRow>
__device__ __forceinline__ void readout_sparse(const HopArgs &a, uint32_t h, uint32_t r0, uint32_t S, uint32_t n_live,
                                               const uint32_t *live_row, const uint8_t *live_kp, KpOfRow kp_of_row,
                                               float *o_f, uint32_t tid)
{
    const QFmt fa = a.act[h];
    const int maxa = (1 << (fa.iwl + fa.frac)) - 1;
    const uint32_t nthreads = blockDim.x;
    for (uint32_t c = tid; c < Dp; c += nthreads) {
        const uint8_t *vb = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + c;
        int acc = 0;
        if (n_live <= (uint32_t)kLiveCap) {
            // four surviving rows per step, their (cold, HBM) value bytes requested together
            for (uint32_t i0 = 0; i0 < n_live; i0 += 4) {
                uint8_t vbyte[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) vbyte[j] = (i0 + j < n_live) ? vb[(size_t)live_row[i0 + j] * Dp] : (uint8_t)0;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (i0 + j < n_live) acc += qm_mul_code((int)live_kp[i0 + j], sm_decode(vbyte[j]), fa.frac, maxa);
            }
        } else {                                   // cannot happen for p summing to 1; kept exact anyway
            for (uint32_t r = 0; r < S; r++) {
                const int kp = kp_of_row(r);
                if (kp) acc += qm_mul_code(kp, sm_decode(vb[(size_t)r * Dp]), fa.frac, maxa);
            }
        }
        acc = acc > maxa ? maxa : (acc < -maxa ? -maxa : acc);
        o_f[c] = qm_scale_down((float)acc, fa.frac);
    }
    __syncthreads();
}

// lin_map + hop update from o_f (any read-out) and the Q_bin(u) codes in `ku`; ends with a barrier.
// H is sign-magnitude int8 [D][Dp]: the matrix-vector product is the key scan's arithmetic (same
// per-product form Qw(Qw(H).Qbin(u)), lib/layer_cuda.cu:71), Dp/16 lanes per output row.  After the
// butterfly all Dp/16 lanes of a row group hold the row's sum, so lane `chunk` keeps the sum of the
// chunk-th row of a group of Dp/16 iterations and the hop update runs once per group on all lanes.
// The update u' = Qa(Qa(lu) + Qa(o)) is done on integer codes of the activation format (the float
// form adds two grid values exactly, so both give the same code).
// `pre` + `use_pre`: ScanConst already built for (word length of w[h], frac_bin) and this lane's chunk.
// `rows0`: this lane's 16-byte pieces of the first group's rows, already in registers (a one-wavefront
// caller that issued the loads early), or nullptr.
template <uint32_t Dp>
__device__ __forceinline__ void linmap_update(const HopArgs &a, uint32_t q, uint32_t h, const short *ku, float *u_f,
                                              const float *o_f, uint32_t tid, const ScanConst *pre = nullptr,
                                              uint32_t pre_sh = 0, bool use_pre = false, const i32x4 *rows0 = nullptr)
{
    constexpr uint32_t LPR = Dp / 16, RPW = kWave / LPR;
    const QFmt fa = a.act[h], fw = a.w[h], fb = a.bin;
    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
    const int maxa = (1 << (fa.iwl + fa.frac)) - 1;
    const uint32_t D = a.D;
    const uint32_t nthreads = blockDim.x, nwaves = nthreads / kWave;
    const uint32_t lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t sub = lane / LPR, chunk = lane % LPR;
    auto update = [&](uint32_t o_i, int lu_a) {          // lu_a: code of Qa(lu)
        const float o = o_f[o_i];
        int un = lu_a + qm_code(o, fa.iwl, fa.frac);
        un = un > maxa ? maxa : (un < -maxa ? -maxa : un);
        const float unf = qm_scale_down((float)un, fa.frac);
        if (a.tap_o) a.tap_o[((size_t)q * a.n_hop + h) * D + o_i] = o;
        if (a.tap_u) a.tap_u[((size_t)q * a.n_hop + h) * D + o_i] = unf;
        u_f[o_i] = unf;
    };
    if (a.en_lin_map) {
        ScanConst c;
        uint32_t sh;
        // (`pre` is never a conditional pointer: selecting between two addresses would force the caller's
        // constants out of registers)
        if (pre && use_pre) { c = *pre; sh = pre_sh; }
        else sh = make_scan_const(c, ku, chunk * 16, fw.iwl + fw.frac, (int)fb.frac);
        const uint8_t *hb = (const uint8_t *)a.lin_map[h] + chunk * 16;
        const uint32_t n_it = (D + nwaves * RPW - 1) / (nwaves * RPW);
        for (uint32_t g = 0; g < n_it; g += LPR) {
            int keep = 0;
            uint32_t keep_r = D;
            if (rows0) {                                 // (known at compile time per call site) rows already in registers
#pragma unroll
                for (uint32_t t = 0; t < LPR; t++) {
                    if (g + t < n_it) {                  // wavefront-uniform
                        const uint32_t r = ((g + t) * nwaves + wave) * RPW + sub;
                        i32x4 x = {0, 0, 0, 0};
                        if (g == 0) x = rows0[t < 4 ? t : 3];
                        else if (r < D) x = *(const i32x4 *)(hb + (size_t)r * Dp);
                        const int acc = row_lanes_sum<LPR>(lane_row_sum(x, c, sh));
                        if (chunk == t) { keep = acc; keep_r = r; }
                    }
                }
            } else {
                // rows are fetched four at a time before any of them is reduced: the loads are L2 hits
                // (~0.6 us each) and would otherwise be paid one after the other
                constexpr uint32_t BATCH = LPR < 4 ? LPR : 4;
#pragma unroll
                for (uint32_t t0 = 0; t0 < LPR; t0 += BATCH) {
                    i32x4 x[BATCH];
#pragma unroll
                    for (uint32_t j = 0; j < BATCH; j++) {
                        const uint32_t r = ((g + t0 + j) * nwaves + wave) * RPW + sub;
                        x[j] = i32x4{0, 0, 0, 0};
                        if (g + t0 + j < n_it && r < D) x[j] = *(const i32x4 *)(hb + (size_t)r * Dp);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < BATCH; j++) {
                        const uint32_t t = t0 + j;
                        if (g + t < n_it) {              // wavefront-uniform
                            const uint32_t r = ((g + t) * nwaves + wave) * RPW + sub;
                            const int acc = row_lanes_sum<LPR>(lane_row_sum(x[j], c, sh));
                            if (chunk == t) { keep = acc; keep_r = r; }
                        }
                    }
                }
            }
            if (keep_r < D) {
                // Qw of the row sum, then Qa of that value: shift between the two grids, toward zero
                const int kw = keep > maxw ? maxw : (keep < -maxw ? -maxw : keep);
                const uint32_t mag = (uint32_t)(kw < 0 ? -kw : kw);
                const uint32_t ma = fa.frac >= fw.frac ? mag << (fa.frac - fw.frac) : mag >> (fw.frac - fa.frac);
                const int la = ma > (uint32_t)maxa ? maxa : (int)ma;
                update(keep_r, kw < 0 ? -la : la);
            }
        }
    } else {
        for (uint32_t o_i = tid; o_i < D; o_i += nthreads) update(o_i, qm_code(u_f[o_i], fa.iwl, fa.frac));
    }
    __syncthreads();
}

template <uint32_t Dp, typename KpOfRow>
__device__ __forceinline__ void finish_hop(const HopArgs &a, uint32_t q, uint32_t h, uint32_t r0, uint32_t S,
                                           uint32_t n_live, const uint32_t *live_row, const uint8_t *live_kp,
                                           KpOfRow kp_of_row, const short *ku, float *u_f, float *o_f, uint32_t tid)
{
    readout_sparse<Dp>(a, h, r0, S, n_live, live_row, live_kp, kp_of_row, o_f, tid);
    linmap_update<Dp>(a, q, h, ku, u_f, o_f, tid);
}

}  // namespace
