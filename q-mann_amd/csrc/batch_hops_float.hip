// batch_hops_float.hip -- fused hop kernel for ATTENTION_MODE 1 ("normal" attention over quantized
// embeddings): float dot-product scores, float softmax, float read-out (lib/layer.c:177-195 forces
// f_fixed = false for both dot_mat_vec layers), while the embeddings, the linear map and the hop
// update stay fixed point.
//
// Memories are the same sign-magnitude int8 planes; here they carry the embedding outputs on the
// WEIGHT grid Q(w[h]) (no attention re-quantisation happens in this mode).  Because both operands
// sit on grids, a score is an exact integer (|sum| < 2^24) times 2^-(frac_w + frac_u): it is
// computed as sum |k| . (+-u) with one v_dot4 per 4 bytes (the key's sign selects u or -u through
// v_perm + v_bfi) and is bit-identical to the reference's float sum in any order.  The read-out
// o[c] = sum_r p[r] . C[r][c] is a genuine float sum over ALL slots (p is not quantised), so this
// mode streams the value plane as well; its result carries the float-order tolerance (1e-5).
#include "hops_common.h"

namespace {

struct DotConst {
    uint32_t up[4];   // two's-complement bytes of  u for this lane's 16 columns
    uint32_t un[4];   // two's-complement bytes of -u
};

__device__ __forceinline__ int dot_lane_sum(const i32x4 x, const DotConst &c)
{
    int acc = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t w = (uint32_t)x[d];
        const uint32_t neg = __builtin_amdgcn_perm(0u, 0u, w & 0x80808080u);      // 0xFF where the key is negative
        const uint32_t uu = (c.un[d] & neg) | (c.up[d] & ~neg);                   // -u there, u elsewhere
        acc = __builtin_amdgcn_sdot4((int)(w & 0x7F7F7F7Fu), (int)uu, acc, false);
    }
    return acc;
}

template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_hops_float(const HopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *u_f = (float *)(smem + kOffU);
    float *o_f = (float *)(smem + kOffO);
    short *ku = (short *)(smem + kOffKu);
    int8_t *kuq = (int8_t *)(smem + kOffHist);            // integer codes of u on its own grid (reuses hist area)
    float *part = (float *)(smem + kOffHist + 1024);      // [kWaves * 8][Dp <= 256]... see below (reuses hist/ptab)
    double *red = (double *)(smem + kOffRed);
    int32_t *sc = (int32_t *)(smem + kOffScores);         // scores, then p (float, in place)

    constexpr uint32_t Dp = LPR * 16;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t q = blockIdx.x;
    const uint32_t r0 = a.row_off[q];
    const uint32_t S_in = a.row_off[q + 1] - r0;
    const uint32_t S = S_in < a.max_slots ? S_in : a.max_slots;   // never index LDS past what the launch reserved
    const uint32_t D = a.D;

    u_f[tid] = (tid < D) ? a.u0[(size_t)q * D + tid] : 0.0f;
    __syncthreads();

    for (uint32_t h = 0; h < a.n_hop; h++) {
        const QFmt fw = a.w[h], fb = a.bin;
        const QFmt fsrc = h == 0 ? a.w[0] : a.act[h - 1];              // grid u sits on (emb_q or sv[h-1])
        const float uv = u_f[tid];
        ku[tid] = (short)((tid < D) ? qm_code_or_sign(uv, fb.iwl, fb.frac) : 0);
        kuq[tid] = (int8_t)((tid < D) ? qm_code(relu_if(uv, hop_relu(a, h)), fsrc.iwl, fsrc.frac) : 0);   // exact: u is on that grid
        __syncthreads();

        const float scale = qm_scale_down(1.0f, fw.frac + fsrc.frac);
        if (S > 0) {
            DotConst c;
            const uint32_t c0 = (lane % LPR) * 16;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t up = 0, un = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = kuq[c0 + 4 * d + i];
                    up |= ((uint32_t)k & 0xFFu) << (8 * i);
                    un |= ((uint32_t)(-k) & 0xFFu) << (8 * i);
                }
                c.up[d] = up; c.un[d] = un;
            }
            const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.hop_stride + (size_t)r0 * Dp;
            auto row_sum = [&](const i32x4 x) { return dot_lane_sum(x, c); };
            auto retire = [&](uint32_t r, int v) { sc[r] = v; };
            if (S >= (kWave / LPR) * 4) scan_rows<LPR, 4, true, kWaves>(kb, S, row_sum, retire, lane, wave);
            else scan_rows_short<LPR>(kb, S, row_sum, retire, lane, wave, kWaves);
        }
        __syncthreads();

        // softmax per slot (lib/layer_cuda.cu:1895-1916, 1969-2060); p replaces the score in place
        float *pf = (float *)sc;
        if (S > 0) {
            const SmCfg smc = sm_cfg(a, h);
            auto slot_x = [&](int sv) { return sm_scaled((float)sv * scale, smc); };   // (float)sv . scale is exact
            float xmax = -INFINITY;
            for (uint32_t r = tid; r < S; r += kBlock) xmax = fmaxf(xmax, slot_x(sc[r]));
            xmax = block_max<float>(xmax, (float *)red, lane, wave);
            double psum = 0.0;
            for (uint32_t r = tid; r < S; r += kBlock) psum += (double)sm_exp(slot_x(sc[r]) - xmax, smc);
            const double total = smc.base == QMANN_SOFTMAX_EXP ? block_sum_double(psum, red, lane, wave)       // (CPU bases: the CPU softmax's serial float total)
                                                               : block_serial_total_f32(S, lane, wave, red, [&](uint32_t r) { return sm_exp(slot_x(sc[r]) - xmax, smc); });
            // stock e^x softmax: e . (1/total) is within an ulp of e / total, and p is not quantised in this
            // mode (tolerance 1e-5); the variants go through the general quotient
            const bool stock = smc.base == QMANN_SOFTMAX_EXP && !smc.shift;
            const double inv_total = 1.0 / total;
            const size_t tb = (size_t)h * a.rows_total + r0;
            for (uint32_t r = tid; r < S; r += kBlock) {
                const int sv = sc[r];
                const float e = sm_exp(slot_x(sv) - xmax, smc);
                const float p = stock ? (float)((double)e * inv_total) : sm_quot(e, total, smc);
                if (a.tap_codes) a.tap_codes[tb + r] = sv;
                if (a.tap_scores) a.tap_scores[tb + r] = (float)sv * scale;
                if (a.tap_probs) a.tap_probs[tb + r] = p;
                pf[r] = p;
            }
        }
        __syncthreads();

        // dense float read-out: lane (sub, chunk) owns 16 columns of every (64 / LPR)-th row; tiles of
        // UN loads per lane, the next tile in flight while the current one is accumulated (as the key scan)
        {
            constexpr uint32_t RPW = kWave / LPR;
            constexpr int UN = 4;
            constexpr uint32_t TILE = RPW * UN, STEP = kWaves * TILE;
            const uint32_t sub = lane / LPR, chunk = lane % LPR;
            const float vscale = qm_scale_down(1.0f, fw.frac);
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
            const uint8_t *vb = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + chunk * 16;
            // acc[i] += (p . 2^-frac) . code.  Four sign-magnitude bytes become two's complement in 6 word-wide steps
            // (offset binary first: 128 + m or 128 - m never carries into the neighbour byte, "minus zero" included; then the
            // top bits flip), each byte converts with one sign-extending v_cvt_f32_i32 (SDWA byte select), one FMA accumulates
            auto accumulate = [&](const i32x4 x, float ps) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t w = (uint32_t)x[d];
                    const uint32_t t = w & 0x80808080u, one = t >> 7;                    // 0x80 / 0x01 in the negative bytes
                    const uint32_t ob = (w ^ 0x80808080u ^ (t - one)) + one;             // 128 + m | (127 - m) + 1
                    const uint32_t tc = ob ^ 0x80808080u;
                    float v0, v1, v2, v3;
                    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(v0) : "v"(tc));
                    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(v1) : "v"(tc));
                    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(v2) : "v"(tc));
                    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(v3) : "v"(tc));
                    acc[4 * d + 0] = __builtin_fmaf(ps, v0, acc[4 * d + 0]);
                    acc[4 * d + 1] = __builtin_fmaf(ps, v1, acc[4 * d + 1]);
                    acc[4 * d + 2] = __builtin_fmaf(ps, v2, acc[4 * d + 2]);
                    acc[4 * d + 3] = __builtin_fmaf(ps, v3, acc[4 * d + 3]);
                }
            };
            if (S >= TILE) {
                const uint32_t n_tiles = (S + TILE - 1) / TILE;
                auto tile_start = [&](uint32_t base) { return base + TILE <= S ? base : S - TILE; };
                auto issue = [&](i32x4 (&x)[UN], uint32_t base) {
                    const uint8_t *pp = vb + (size_t)(tile_start(base) + sub) * Dp;
#pragma unroll
                    for (int j = 0; j < UN; j++) x[j] = __builtin_nontemporal_load((const i32x4 *)(pp + (size_t)j * RPW * Dp));
                };
                auto consume = [&](const i32x4 (&x)[UN], uint32_t base) {
                    const uint32_t start = tile_start(base);
#pragma unroll
                    for (int j = 0; j < UN; j++) {
                        const uint32_t r = start + j * RPW + sub;
                        accumulate(x[j], r >= base ? pf[r] * vscale : 0.0f);     // rows a moved-back tile repeats add zero
                    }
                };
                if (wave < n_tiles) {
                    i32x4 xa[UN], xb[UN];
                    uint32_t base = wave * TILE;
                    issue(xa, base);
                    for (uint32_t t = wave; t < n_tiles; t += 2 * kWaves) {
                        issue(xb, base + STEP);
                        consume(xa, base);
                        if (t + kWaves >= n_tiles) break;
                        issue(xa, base + 2 * STEP);
                        consume(xb, base + STEP);
                        base += 2 * STEP;
                    }
                }
            } else {
                for (uint32_t r = wave * RPW + sub; r < S; r += kWaves * RPW)
                    accumulate(*(const i32x4 *)(vb + (size_t)r * Dp), pf[r] * vscale);
            }
            // fold the RPW row groups of a wavefront (lanes with equal chunk), then the wavefronts
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float v = acc[i];
                for (uint32_t o = LPR; o < kWave; o <<= 1) v += __shfl_xor(v, (int)o);
                acc[i] = v;
            }
            __syncthreads();                                   // the histogram area is free now
            if (sub == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) part[wave * 256 + chunk * 16 + i] = acc[i];
            }
            __syncthreads();
            if (tid < Dp) {
                float s = 0.0f;
                for (int w = 0; w < kWaves; w++) s += part[w * 256 + tid];
                o_f[tid] = s;
            }
            __syncthreads();
        }
        linmap_update<Dp>(a, q, h, ku, u_f, o_f, tid);
    }
    if (tid < D) a.u_out[(size_t)q * D + tid] = relu_if(u_f[tid], a.en_non_lin != 0);
}

template <int LPR>
void launch_float(const HopArgs &a, size_t lds, uint32_t n_query, hipStream_t st)
{
    if (lds > kLdsDefaultLimit)
        QM_HIP(hipFuncSetAttribute((const void *)k_hops_float<LPR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    k_hops_float<LPR><<<n_query, kBlock, lds, st>>>(a);
}

}  // namespace

extern "C" int qmann_hops_float_impl(const HopArgs &a, uint32_t Dp, uint32_t max_slots, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    static_assert(kOffHist + 1024 + kWaves * 256 * 4 <= kOffU, "partial sums must fit the histogram + p-table area");
    const size_t lds = (size_t)kOffScores + (((size_t)max_slots * 4 + 15) & ~(size_t)15);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    if (Dp == 64) launch_float<4>(a, lds, n_query, st);
    else if (Dp == 128) launch_float<8>(a, lds, n_query, st);
    else launch_float<16>(a, lds, n_query, st);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}
