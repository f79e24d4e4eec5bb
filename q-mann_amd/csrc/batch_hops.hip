// batch_hops.hip -- the hot path: every hop of every query in one launch.
//
// One 256-thread workgroup (4 wavefronts) owns one query and walks its hops in
// order (hops are sequentially dependent: u_{h+1} = H.u_h + o_h).  Per hop:
//
//   1. key scan (HBM-bound): the query's key plane, sign-magnitude int8 [S][Dp]
//      row-major, is streamed with 16-byte non-temporal loads -- a wavefront
//      instruction covers 1 KiB = 64/LPR whole rows (LPR = Dp/16 lanes per row);
//      each wavefront keeps two tiles of four such loads in flight (the next
//      tile is issued before the current one is reduced).  Each lane combines
//      its 16 key codes with the 16 query codes it keeps in registers, with the
//      reference's per-product quantisation Qm(Qm(k).Qv(u)) (lib/layer_cuda.cu:120):
//      magnitudes are multiplied in packed 16-bit lanes by the pre-shifted |u| so
//      that truncation is a logical shift and the clamp is the multiplier's own
//      unsigned saturation (see ScanConst), signs become +-1 bytes through one
//      v_perm_b32 and are applied by one v_dot4 per 4 bytes (12 VALU ops per 4 bytes).  Row sums meet across the LPR lanes
//      with DPP adds (no LDS), are clamped to the attention format (:135) and
//      land in LDS as one byte per slot, plus a per-wavefront histogram of the
//      (at most 255) score codes.
//   2. softmax over slots from the histogram: scores live on an 8-bit grid, so
//      exp(x - max) takes at most 255 distinct values; the normaliser is
//      sum_d count[d].e[d] in double (lib/layer_cuda.cu:2024-2042) and the
//      read-out weight code Q(p) is a 255-entry table.
//   3. weighted read-out: Q(p) is zero for every slot with p < 2^-frac, so only
//      the few surviving rows of the value plane are read (bit-identical to
//      summing all rows: the skipped terms are exact zeros, :562).
//   4. linear map H.u (int8 [D][Dp], L2-resident) and u' = Q(Q(Hu) + Q(o)).
//
// All integer work is exact; the only floating-point step is the softmax table.
#include "qfmt.h"
#include "rt.h"
#include "../../include/qmann_batch.h"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kUnrollDefault = 4;  // 16-byte loads per lane and tile in the key scan (two tiles in flight)
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

struct HopArgs {
    const int8_t *keys;
    const int8_t *vals;
    size_t hop_stride;
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
    uint32_t n_hop, D, Dp, softmax_base, en_lin_map;
    QFmt act[QMANN_MAX_HOP], w[QMANN_MAX_HOP], att[QMANN_MAX_HOP], bin;
};

template <int CTRL>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// sum over the LPR adjacent lanes that share a row (every lane ends with the total)
template <int LPR>
__device__ __forceinline__ int row_lanes_sum(int v)
{
    v = dpp_add<0xB1>(v);                        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                        // quad_perm [2,3,0,1]
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

template <bool NT>
__device__ __forceinline__ i32x4 load16(const uint8_t *p)
{
    if (NT) return __builtin_nontemporal_load((const i32x4 *)p);
    return *(const i32x4 *)p;
}

// Memories shorter than one tile (bAbI-sized stories): one guarded pass, rows spread over the
// four wavefronts.  Not a bandwidth path.
template <int LPR>
__device__ __forceinline__ void scan_keys_short(const uint8_t *__restrict__ kb, uint32_t S, const ScanConst &c,
                                                uint32_t sh, int maxm, int8_t *sc, uint32_t *hist, uint32_t lane,
                                                uint32_t wave)
{
    constexpr uint32_t RPW = kWave / LPR;
    constexpr uint32_t Dp = LPR * 16;
    const uint32_t sub = lane / LPR, chunk = lane % LPR;
    for (uint32_t base = wave * RPW; base < S; base += kWaves * RPW) {
        const uint32_t r = base + sub;
        i32x4 x = {0, 0, 0, 0};
        if (r < S) x = *(const i32x4 *)(kb + (size_t)r * Dp + chunk * 16);
        const int v = row_lanes_sum<LPR>(lane_row_sum(x, c, sh));
        if (chunk == 0 && r < S) {
            const int code = v > maxm ? maxm : (v < -maxm ? -maxm : v);
            sc[r] = (int8_t)code;
            atomicAdd(&hist[code + 127], 1u);
        }
    }
}

// One wavefront streams tiles of UN x (64 / LPR) rows; loads of the next tile are issued before
// the current one is reduced (addresses are clamped to the last row instead of being predicated,
// so the loop body has no divergent control flow and the compiler can count its vmcnt waits).
template <int LPR, int UN, bool NT>
__device__ __forceinline__ void scan_keys(const uint8_t *__restrict__ kb, uint32_t S, const ScanConst &c,
                                          uint32_t sh, int maxm, int8_t *sc, uint32_t *hist, uint32_t lane,
                                          uint32_t wave)
{
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
        for (int j = 0; j < UN; j++) s[j] = row_lanes_sum<LPR>(lane_row_sum(x[j], c, sh));
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
            if (j < (uint32_t)UN && r >= base) {
                const int code = v > maxm ? maxm : (v < -maxm ? -maxm : v);
                sc[r] = (int8_t)code;
                atomicAdd(&hist[code + 127], 1u);
            }
        }
    };

    if (wave >= n_tiles) return;
    i32x4 xa[UN], xb[UN];
    uint32_t base = wave * TILE;
    constexpr uint32_t STEP = kWaves * TILE;
    issue(xa, base);
    for (uint32_t t = wave; t < n_tiles; t += 2 * kWaves) {
        issue(xb, base + STEP);
        reduce(xa, base);
        if (t + kWaves >= n_tiles) break;
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

__device__ __forceinline__ int block_max_int(int v, int *scratch, uint32_t lane, uint32_t wave)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    int r = scratch[0];
    for (int i = 1; i < kWaves; i++) r = scratch[i] > r ? scratch[i] : r;
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_sum_double(double v, double *scratch, uint32_t lane, uint32_t wave)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    const double r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return r;
}

template <int LPR, int kUnroll, bool NT, int MINW>
__global__ void __launch_bounds__(kBlock, MINW)
k_hops_fixed(const HopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = (uint32_t *)(smem + kOffHist);
    float *ptab = (float *)(smem + kOffPtab);
    float *u_f = (float *)(smem + kOffU);
    float *o_f = (float *)(smem + kOffO);
    short *ku = (short *)(smem + kOffKu);
    uint8_t *kplut = (uint8_t *)(smem + kOffKp);
    uint32_t *live_row = (uint32_t *)(smem + kOffLiveRow);
    uint8_t *live_kp = (uint8_t *)(smem + kOffLiveKp);
    uint32_t *misc = (uint32_t *)(smem + kOffMisc);
    double *red = (double *)(smem + kOffRed);
    int8_t *sc = (int8_t *)(smem + kOffScores);

    constexpr uint32_t Dp = LPR * 16;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t q = blockIdx.x;
    const uint32_t r0 = a.row_off[q];
    const uint32_t S = a.row_off[q + 1] - r0;
    const uint32_t D = a.D;

    u_f[tid] = (tid < D) ? a.u0[(size_t)q * D + tid] : 0.0f;
    __syncthreads();

    for (uint32_t h = 0; h < a.n_hop; h++) {
        const QFmt fa = a.act[h], fw = a.w[h], fm = a.att[h], fb = a.bin;
        const int fv = (int)fb.frac;
        const int maxm = (1 << (fm.iwl + fm.frac)) - 1;
        const int maxa = (1 << (fa.iwl + fa.frac)) - 1;
        const int maxw = (1 << (fw.iwl + fw.frac)) - 1;

        // query codes Q_bin(u), histogram reset
        const int kuc = (tid < D) ? qm_code(u_f[tid], fb.iwl, fb.frac) : 0;
        ku[tid] = (short)kuc;
#pragma unroll
        for (int i = 0; i < kWaves; i++) hist[i * 256 + tid] = 0u;
        if (tid == 0) misc[0] = 0u;
        __syncthreads();

        if (S > 0) {
            // lane constants: |u| pre-shifted so that saturation == the per-product clamp
            const uint32_t wl = fm.iwl + fm.frac;
            const uint32_t sh = 16u - wl;                       // result shift
            const int pre = (int)sh - fv;                       // |u| << pre (>= 0 since wl + fv <= 14)
            ScanConst c;
            const uint32_t c0 = (lane % LPR) * 16;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t m[4], sg = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = ku[c0 + 4 * d + i];
                    const uint32_t a = (uint32_t)(k < 0 ? -k : k) << pre;
                    m[i] = a > 0xFFFFu ? 0xFFFFu : a;
                    sg |= (k < 0 ? 0x80u : 0u) << (8 * i);
                }
                c.ue[d] = m[0] | (m[2] << 16);
                c.uo[d] = m[1] | (m[3] << 16);
                c.s7[d] = sg;
            }
            const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.hop_stride + (size_t)r0 * Dp;
            if (S >= (kWave / LPR) * kUnroll)
                scan_keys<LPR, kUnroll, NT>(kb, S, c, sh, maxm, sc, hist + wave * 256, lane, wave);
            else
                scan_keys_short<LPR>(kb, S, c, sh, maxm, sc, hist + wave * 256, lane, wave);
        }
        __syncthreads();

        // softmax over slots from the histogram of score codes (bin d <-> code d - 127)
        uint32_t n_live = 0;
        if (S > 0) {
            const uint32_t cnt = hist[tid] + hist[256 + tid] + hist[512 + tid] + hist[768 + tid];
            const int dmax = block_max_int(cnt ? (int)tid : -1, (int *)red, lane, wave);
            const float x = (float)((int)tid - dmax) / (float)(1 << fm.frac);   // score - max, exact
            float e, p;
            if (a.softmax_base == QMANN_SOFTMAX_EXP) {
                e = expf(x);
                const double total = block_sum_double(cnt ? (double)cnt * (double)e : 0.0, red, lane, wave);
                p = (float)((double)e / total);
            } else {
                e = exp2f(x);
                const float total = (float)block_sum_double(cnt ? (double)cnt * (double)e : 0.0, red, lane, wave);
                p = e / total;
            }
            if (!cnt) p = 0.0f;
            ptab[tid] = p;
            kplut[tid] = (uint8_t)qm_code(p, fa.iwl, fa.frac);
            __syncthreads();

            if (a.tap_codes || a.tap_scores || a.tap_probs) {
                const size_t tb = (size_t)h * a.rows_total + r0;
                for (uint32_t r = tid; r < S; r += kBlock) {
                    const int code = sc[r];
                    if (a.tap_codes) a.tap_codes[tb + r] = code;
                    if (a.tap_scores) a.tap_scores[tb + r] = (float)code / (float)(1 << fm.frac);
                    if (a.tap_probs) a.tap_probs[tb + r] = ptab[code + 127];
                }
            }
            // rows whose quantised weight is non-zero
            for (uint32_t r = tid; r < S; r += kBlock) {
                const uint8_t kp = kplut[(int)sc[r] + 127];
                if (kp) {
                    const uint32_t i = atomicAdd(&misc[0], 1u);
                    if (i < (uint32_t)kLiveCap) { live_row[i] = r; live_kp[i] = kp; }
                }
            }
            __syncthreads();
            n_live = misc[0];
        }

        // weighted read-out o[c] = Qa( sum_r Qa( Qa(p[r]) . Qa(C[r][c]) ) ), column-parallel
        if (tid < Dp) {
            const uint8_t *vb = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + tid;
            int acc = 0;
            if (n_live <= (uint32_t)kLiveCap) {
                for (uint32_t i = 0; i < n_live; i++)
                    acc += qm_mul_code((int)live_kp[i], sm_decode(vb[(size_t)live_row[i] * Dp]), fa.frac, maxa);
            } else {                                   // cannot happen for p summing to 1; kept exact anyway
                for (uint32_t r = 0; r < S; r++) {
                    const int kp = kplut[(int)sc[r] + 127];
                    if (kp) acc += qm_mul_code(kp, sm_decode(vb[(size_t)r * Dp]), fa.frac, maxa);
                }
            }
            acc = acc > maxa ? maxa : (acc < -maxa ? -maxa : acc);
            o_f[tid] = (float)acc / (float)(1 << fa.frac);
        }
        __syncthreads();

        // lu = lin_map[h] . u with the dense layer's per-product quantisation, then u' = Q(Q(lu) + Q(o))
        if (tid < D) {
            float lu = u_f[tid];
            if (a.en_lin_map) {
                const int8_t *hr = a.lin_map[h] + (size_t)tid * Dp;
                int acc = 0;
                for (uint32_t i = 0; i < Dp; i += 16) {
                    const i32x4 wv = *(const i32x4 *)(hr + i);
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        const int kh = (int)(int8_t)((uint32_t)wv[b >> 2] >> (8 * (b & 3)));
                        acc += qm_mul_code(kh, (int)ku[i + b], fb.frac, maxw);
                    }
                }
                acc = acc > maxw ? maxw : (acc < -maxw ? -maxw : acc);
                lu = (float)acc / (float)(1 << fw.frac);
            }
            const float o = o_f[tid];
            const float un = qm_quant(qm_quant(lu, fa.iwl, fa.frac) + qm_quant(o, fa.iwl, fa.frac), fa.iwl, fa.frac);
            if (a.tap_o) a.tap_o[((size_t)q * a.n_hop + h) * D + tid] = o;
            if (a.tap_u) a.tap_u[((size_t)q * a.n_hop + h) * D + tid] = un;
            u_f[tid] = un;
        }
        __syncthreads();
    }
    if (tid < D) a.u_out[(size_t)q * D + tid] = u_f[tid];
}

__global__ void k_quantize_i8(const float *__restrict__ src, int8_t *__restrict__ dst, size_t rows, uint32_t cols,
                              uint32_t pitch, QFmt f, int signmag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * pitch) return;
    const size_t r = i / pitch;
    const uint32_t c = (uint32_t)(i % pitch);
    int k = 0;
    uint32_t neg = 0;
    if (c < cols) {
        const float x = src[r * cols + c];
        k = qm_code(x, f.iwl, f.frac);
        neg = !(x >= 0.0f);                      // the reference keys the sign on the float (lib/common.h:210)
    }
    if (signmag) dst[i] = (int8_t)((uint32_t)(k < 0 ? -k : k) | (neg ? 0x80u : 0u));
    else dst[i] = (int8_t)k;
}

inline bool fmt8(qmann_fmt f) { return f.iwl + f.frac >= 1 && f.iwl + f.frac <= 7; }

int g_tune = 0;

}  // namespace

extern "C" {

// development switch: selects a compiled tuning variant of the D=128 scan (0 = shipped)
void qmann_debug_set_tune(int v) { g_tune = v; }

size_t qmann_hops_lds_bytes(uint32_t max_slots)
{
    return (size_t)kOffScores + (((size_t)max_slots + 15) & ~(size_t)15);
}

int qmann_quantize_i8(const float *src, int8_t *dst, size_t rows, uint32_t cols, uint32_t pitch, qmann_fmt fmt,
                      int layout, void *stream)
{
    if (!src || !dst || pitch < cols) return QMANN_EINVAL;
    if (!fmt8(fmt)) return QMANN_ERANGE;
    if (layout != QMANN_CODE_TWOS && layout != QMANN_CODE_SIGNMAG) return QMANN_EINVAL;
    const size_t n = rows * pitch;
    if (n == 0) return QMANN_OK;
    k_quantize_i8<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, rows, cols, pitch,
                                                                              QFmt{fmt.iwl, fmt.frac}, layout);
    QM_LAUNCH_CHECK();
    return QMANN_OK;
}

int qmann_hops_i8(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                  const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                  const qmann_taps *taps, uint32_t n_query, void *stream)
{
    if (!net || !keys || !vals || !row_off || !u0 || !u_out) return QMANN_EINVAL;
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP) return QMANN_EINVAL;
    if (net->dim_emb == 0 || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (net->dim_emb_pad != 64 && net->dim_emb_pad != 128 && net->dim_emb_pad != 256) return QMANN_EUNSUPPORTED;
    if (net->attention_mode != QMANN_ATT_FIXED) return QMANN_EUNSUPPORTED;
    if (net->softmax_base > QMANN_SOFTMAX_POW2) return QMANN_EINVAL;
    if (!fmt8(net->bin)) return QMANN_ERANGE;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        if (net->en_lin_map && !net->lin_map[h]) return QMANN_EINVAL;
    }
    const size_t lds = qmann_hops_lds_bytes(max_slots);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;   // a little static LDS is used besides
    if (n_query == 0) return QMANN_OK;

    HopArgs a{};
    a.keys = keys; a.vals = vals; a.hop_stride = hop_stride; a.row_off = row_off; a.u0 = u0; a.u_out = u_out;
    if (taps) {
        a.tap_codes = taps->score_codes; a.tap_scores = taps->scores; a.tap_probs = taps->probs;
        a.tap_o = taps->o; a.tap_u = taps->u;
    }
    a.rows_total = (uint32_t)(hop_stride / net->dim_emb_pad);
    a.n_hop = net->n_hop; a.D = net->dim_emb; a.Dp = net->dim_emb_pad;
    a.softmax_base = net->softmax_base; a.en_lin_map = net->en_lin_map;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        a.lin_map[h] = net->lin_map[h];
        a.act[h] = QFmt{net->act[h].iwl, net->act[h].frac};
        a.w[h] = QFmt{net->w[h].iwl, net->w[h].frac};
        a.att[h] = QFmt{net->att[h].iwl, net->att[h].frac};
    }
    a.bin = QFmt{net->bin.iwl, net->bin.frac};

    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(n_query), block(kBlock);
#define QM_LAUNCH_HOPS(LPR, UN, NT, MINW)                                                               \
    do {                                                                                                \
        static size_t attr_bytes = 0;       /* raise the dynamic-LDS limit only when it grows */        \
        if (lds > attr_bytes) {                                                                         \
            QM_HIP(hipFuncSetAttribute((const void *)k_hops_fixed<LPR, UN, NT, MINW>,                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
            attr_bytes = lds;                                                                           \
        }                                                                                               \
        k_hops_fixed<LPR, UN, NT, MINW><<<grid, block, lds, st>>>(a);                                   \
    } while (0)
    if (net->dim_emb_pad == 64) QM_LAUNCH_HOPS(4, kUnrollDefault, true, 1);
    else if (net->dim_emb_pad == 256) QM_LAUNCH_HOPS(16, kUnrollDefault, true, 1);
    else {
        switch (g_tune) {                 // tuning variants (bench_variants.py); 0 is the shipped one
        case 1: QM_LAUNCH_HOPS(8, 4, false, 1); break;
        case 2: QM_LAUNCH_HOPS(8, 8, false, 1); break;
        case 3: QM_LAUNCH_HOPS(8, 8, true, 1); break;
        case 4: QM_LAUNCH_HOPS(8, 8, false, 1); break;
        case 5: QM_LAUNCH_HOPS(8, 4, true, 8); break;
        case 6: QM_LAUNCH_HOPS(8, 8, true, 4); break;
        case 7: QM_LAUNCH_HOPS(8, 6, true, 1); break;
        case 8: QM_LAUNCH_HOPS(8, 12, true, 1); break;
        case 9: QM_LAUNCH_HOPS(8, 2, true, 8); break;
        default: QM_LAUNCH_HOPS(8, kUnrollDefault, true, 1); break;
        }
    }
#undef QM_LAUNCH_HOPS
    QM_LAUNCH_CHECK();
    return QMANN_OK;
}

}  // extern "C"
