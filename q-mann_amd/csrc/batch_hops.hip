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
#include "hops_mid.h"

namespace {

// W7: every attention format of the launch has word length 7 -- the clamp comes from the signed saturation of the
// multiply and the quotient is a byte gather (hops_common.h::lane_row_sum7): 9 instead of 11 operations per 4 key bytes
template <int LPR, int kUnroll, bool NT, int MINW, bool W7>
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
    const uint32_t nthreads = blockDim.x, nwaves = nthreads / kWave;      // 256 / 4, or 64 / 1 for short memories
    const uint32_t q = blockIdx.x;
    const uint32_t r0 = a.row_off[q];
    const uint32_t S_in = a.row_off[q + 1] - r0;
    const uint32_t S = S_in < a.max_slots ? S_in : a.max_slots;   // never index LDS past what the launch reserved
    const uint32_t D = a.D;

    for (uint32_t c = tid; c < 256; c += nthreads) u_f[c] = (c < D) ? a.u0[(size_t)q * D + c] : 0.0f;
    __syncthreads();

    for (uint32_t h = 0; h < a.n_hop; h++) {
        const QFmt fa = a.act[h], fm = a.att[h], fb = a.bin;
        const int fv = (int)fb.frac;
        const int maxm = (1 << (fm.iwl + fm.frac)) - 1;

        // query codes Q_bin(u), histogram reset
        for (uint32_t c = tid; c < 256; c += nthreads) {
            ku[c] = (short)((c < D) ? qm_code_or_sign(u_f[c], fb.iwl, fb.frac) : 0);
            for (uint32_t i = 0; i < nwaves; i++) hist[i * 256 + c] = 0u;
        }
        if (tid == 0) misc[0] = 0u;
        __syncthreads();

        if (S > 0) {
            // lane constants: |u| pre-shifted so that saturation == the per-product clamp
            ScanConst c;
            uint32_t sh = 0;
            if (W7) make_scan_const7(c, ku, (lane % LPR) * 16, fv, hop_relu(a, h), fb.iwl + fb.frac == 0);
            else sh = make_scan_const(c, ku, (lane % LPR) * 16, fm.iwl + fm.frac, fv, hop_relu(a, h), fb.iwl + fb.frac == 0);
            const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.hop_stride + (size_t)r0 * Dp;
            uint32_t *hw = hist + wave * 256;
            auto row_sum = [&](const i32x4 x) { return W7 ? lane_row_sum7(x, c) : lane_row_sum(x, c, sh); };
            auto retire = [&](uint32_t r, int v) {          // Qm of the row sum (lib/layer_cuda.cu:135)
                const int code = v > maxm ? maxm : (v < -maxm ? -maxm : v);
                sc[r] = (int8_t)code;
                atomicAdd(&hw[code + 127], 1u);
            };
            if (S < (kWave / LPR) * kUnroll) scan_rows_short<LPR>(kb, S, row_sum, retire, lane, wave, nwaves);
            else if (nwaves == kWaves) scan_rows<LPR, kUnroll, NT, kWaves>(kb, S, row_sum, retire, lane, wave);
            else scan_rows<LPR, kUnroll, NT, 1>(kb, S, row_sum, retire, lane, wave);
        }
        __syncthreads();

        // softmax over slots from the histogram of score codes (bin d <-> code d - 127)
        uint32_t n_live = 0;
        if (S > 0) {
            // each thread owns bins tid, tid + nthreads, ... (4 bins per thread in a one-wavefront group)
            const SmCfg smc = sm_cfg(a, h);
            auto bin_x = [&](uint32_t d) { return sm_scaled(qm_scale_down((float)((int)d - 127), fm.frac), smc); };
            float xmax = -INFINITY;
            for (uint32_t d = tid; d < 256; d += nthreads) {
                uint32_t cnt = 0;
                for (uint32_t i = 0; i < nwaves; i++) cnt += hist[i * 256 + d];
                hist[d] = cnt;                                  // bins are thread-private from here on
                if (cnt) xmax = fmaxf(xmax, bin_x(d));
            }
            xmax = block_max<float>(xmax, (float *)red, lane, wave);
            double part = 0.0;
            for (uint32_t d = tid; d < 256; d += nthreads) {
                const float e = sm_exp(bin_x(d) - xmax, smc);  // score - max: exact on the score grid
                ptab[d] = e;
                if (hist[d]) part += (double)hist[d] * (double)e;
            }
            // e^x base: the CUDA kernel's double total; the CPU bases: the CPU softmax's float total, slot by slot (hops_common.h)
            const double total = smc.base == QMANN_SOFTMAX_EXP ? block_sum_double(part, red, lane, wave)
                                                               : block_serial_total_f32(S, lane, wave, red, [&](uint32_t r) { return ptab[(int)sc[r] + 127]; });
            for (uint32_t d = tid; d < 256; d += nthreads) {
                const float p = hist[d] ? sm_quot(ptab[d], total, smc) : 0.0f;
                ptab[d] = p;
                kplut[d] = (uint8_t)qm_code(p, fa.iwl, fa.frac);
            }
            __syncthreads();

            if (a.tap_codes || a.tap_scores || a.tap_probs) {
                const size_t tb = (size_t)h * a.rows_total + r0;
                for (uint32_t r = tid; r < S; r += nthreads) {
                    const int code = sc[r];
                    if (a.tap_codes) a.tap_codes[tb + r] = code;
                    if (a.tap_scores) a.tap_scores[tb + r] = qm_scale_down((float)code, fm.frac);
                    if (a.tap_probs) a.tap_probs[tb + r] = ptab[code + 127];
                }
            }
            // rows whose quantised weight is non-zero: 16 score bytes per LDS read and their 16 table look-ups in flight
            // together (one score -> look-up -> branch chain per row left every row two LDS latencies long)
            for (uint32_t rb = tid * 16; rb < S; rb += nthreads * 16) {
                const i32x4 v = *(const i32x4 *)(sc + rb);           // (the score array is padded to 16 rows)
                uint8_t kp[16];
#pragma unroll
                for (int i = 0; i < 16; i++) kp[i] = kplut[(int)(int8_t)((uint32_t)v[i / 4] >> (8 * (i % 4))) + 127];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (rb + i < S && kp[i]) {
                        const uint32_t n = atomicAdd(&misc[0], 1u);
                        if (n < (uint32_t)kLiveCap) { live_row[n] = rb + i; live_kp[n] = kp[i]; }
                    }
                }
            }
            __syncthreads();
            n_live = misc[0];
        }

        // weighted read-out over the surviving rows, linear map, hop update (hops_common.h)
        auto kp_of_row = [&](uint32_t r) { return (int)kplut[(int)sc[r] + 127]; };
        finish_hop<Dp>(a, q, h, r0, S, n_live, live_row, live_kp, kp_of_row, ku, u_f, o_f, tid);
    }
    for (uint32_t c = tid; c < D; c += nthreads) a.u_out[(size_t)q * D + c] = relu_if(u_f[c], a.en_non_lin != 0);
}

__global__ void k_quantize_i8(const float *__restrict__ src, int8_t *__restrict__ dst, size_t rows, uint32_t cols,
                              uint32_t pitch, QFmt f, int signmag)
{
    const size_t n = rows * pitch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;          // grid-stride: n may exceed 2^32
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t r = i / pitch;
        const uint32_t c = (uint32_t)(i % pitch);
        int k = 0;
        uint32_t neg = 0;
        if (c < cols) {
            const float x = src[r * cols + c];
            k = qm_code(x, f.iwl, f.frac);
            neg = !(x >= 0.0f);                  // the reference keys the sign on the float (lib/common.h:210)
        }
        if (signmag) dst[i] = (int8_t)((uint32_t)(k < 0 ? -k : k) | (neg ? 0x80u : 0u));
        else dst[i] = (int8_t)k;
    }
}

inline bool fmt8(qmann_fmt f) { return f.iwl + f.frac >= 1 && f.iwl + f.frac <= 7; }

// qmann_check_slots: how many stories exceed the caller's bound (and would be cut by the hop kernels)
__global__ void k_check_slots(const uint32_t *__restrict__ row_off, uint32_t n_query, uint32_t max_slots, uint32_t *n_over)
{
    uint32_t mine = 0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_query; q += (size_t)gridDim.x * blockDim.x)
        mine += (row_off[q + 1] - row_off[q] > max_slots) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(n_over, mine);
}

}  // namespace

// rt.h::QmSplitReady: the launch rule of hops_lean.h::launch_lean for "this batch will be split" (stories capped between 17 and 64
// rows whose mean length is within the short form's 16), and the two lists computed ahead of the launch
bool qm_split_applies(size_t rows_total, uint32_t n_query, uint32_t max_slots)
{
    return n_query && max_slots > (uint32_t)kQuadSlots && max_slots <= (uint32_t)kWave && rows_total / n_query <= (size_t)kQuadSlots &&
           !qm_tuning().no_quad && !qm_tuning().no_lean;
}

uint32_t *qm_split_early(const uint32_t *row_off, uint32_t n_query, uint32_t max_slots, hipStream_t owner, hipStream_t run_on)
{
    return split_lists(row_off, n_query, max_slots, owner, run_on);
}

extern "C" {

int qmann_hops_appx_impl(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                         const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                         const qmann_taps *taps, uint32_t n_query, void *stream);   // batch_hops_ham.hip

int qmann_hops_hambytes_impl(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                             const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                             const qmann_taps *taps, uint32_t n_query, void *stream);   // batch_hops_ham.hip

int qmann_hops_float_impl(const HopArgs &a, uint32_t Dp, uint32_t max_slots, uint32_t n_query, void *stream);  // batch_hops_float.hip

int qmann_check_slots(const uint32_t *row_off, uint32_t n_query, uint32_t max_slots, uint32_t *n_over, void *stream)
{
    QmBatched qm_scope;
    if (!row_off || !n_over) return QMANN_EINVAL;
    if (n_query == 0) return QMANN_OK;
    const uint32_t blocks = (n_query + 255u) / 256u;
    k_check_slots<<<blocks < 1024u ? blocks : 1024u, 256, 0, (hipStream_t)stream>>>(row_off, n_query, max_slots, n_over);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

size_t qmann_hops_lds_bytes(uint32_t max_slots)
{
    return (size_t)kOffScores + (((size_t)max_slots + 15) & ~(size_t)15);
}

int qmann_quantize_i8(const float *src, int8_t *dst, size_t rows, uint32_t cols, uint32_t pitch, qmann_fmt fmt,
                      int layout, void *stream)
{
    QmBatched qm_scope;
    if (!src || !dst || pitch < cols) return QMANN_EINVAL;
    if (!fmt8(fmt)) return QMANN_ERANGE;
    if (layout != QMANN_CODE_TWOS && layout != QMANN_CODE_SIGNMAG) return QMANN_EINVAL;
    const size_t n = rows * pitch;
    if (n == 0) return QMANN_OK;
    const size_t blocks = (n + 255) / 256;
    k_quantize_i8<<<(unsigned)(blocks < 262144 ? blocks : 262144), 256, 0, (hipStream_t)stream>>>(src, dst, rows, cols, pitch,
                                                                              QFmt{fmt.iwl, fmt.frac}, layout);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_hops_i8(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                  const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                  const qmann_taps *taps, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    if (!net || !keys || !vals || !row_off || !u0 || !u_out) return QMANN_EINVAL;
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP) return QMANN_EINVAL;
    if (net->dim_emb == 0 || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (net->dim_emb_pad != 64 && net->dim_emb_pad != 128 && net->dim_emb_pad != 256) return QMANN_EUNSUPPORTED;
    if (taps && hop_stride == 0) return QMANN_EINVAL;       // taps are indexed [hop][row]: they need distinct hop planes
    if (net->attention_mode == QMANN_ATT_APPX)
        return qmann_hops_appx_impl(net, keys, vals, hop_stride, row_off, max_slots, u0, u_out, taps, n_query, stream);
    if (net->attention_mode == QMANN_ATT_HAMMING_V0 || net->attention_mode == QMANN_ATT_HAMMING_V1)
        return qmann_hops_hambytes_impl(net, keys, vals, hop_stride, row_off, max_slots, u0, u_out, taps, n_query, stream);
    if (net->attention_mode != QMANN_ATT_FIXED && net->attention_mode != QMANN_ATT_FLOAT) return QMANN_EUNSUPPORTED;
    if (net->softmax_base > QMANN_SOFTMAX_EXP_PLAN) return QMANN_EINVAL;
    if (!fmt8(net->bin) && net->bin.iwl + net->bin.frac != 0) return QMANN_ERANGE;     // (0,0) = BINARY_MODE: u binarised
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        if (net->en_lin_map && !net->lin_map[h]) return QMANN_EINVAL;
    }
    const size_t lds = qmann_hops_lds_bytes(max_slots);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;   // a little static LDS is used besides
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads

    HopArgs a{};
    a.keys = keys; a.vals = vals; a.hop_stride = hop_stride; a.key_hop_stride = hop_stride;
    a.row_off = row_off; a.u0 = u0; a.u_out = u_out;
    if (taps) {
        a.tap_codes = taps->score_codes; a.tap_scores = taps->scores; a.tap_probs = taps->probs;
        a.tap_o = taps->o; a.tap_u = taps->u;
    }
    a.rows_total = hop_stride ? (uint32_t)(hop_stride / net->dim_emb_pad) : (uint32_t)qm_rows_hint;      // (tied hops: the caller's hint, or 0 = unknown)
    a.max_slots = max_slots;
    a.n_hop = net->n_hop; a.D = net->dim_emb; a.Dp = net->dim_emb_pad;
    a.softmax_base = net->softmax_base; a.en_lin_map = net->en_lin_map;
    a.softmax_shift = net->softmax_shift_based; a.en_att_scale = net->en_att_scale; a.en_non_lin = net->en_non_linearity;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        a.att_scale[h] = net->att_scale[h];
        a.lin_map[h] = net->lin_map[h];
        a.act[h] = QFmt{net->act[h].iwl, net->act[h].frac};
        a.w[h] = QFmt{net->w[h].iwl, net->w[h].frac};
        a.att[h] = QFmt{net->att[h].iwl, net->att[h].frac};
    }
    a.bin = QFmt{net->bin.iwl, net->bin.frac};

    if (net->attention_mode == QMANN_ATT_FLOAT)
        return qmann_hops_float_impl(a, net->dim_emb_pad, max_slots, n_query, stream);

    hipStream_t st = (hipStream_t)stream;
    if (lean_supported(a, max_slots, 64)) {                 // hops_lean.h
        launch_lean<kModeFixed, 8>(a, max_slots, n_query, st);
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    if (max_slots <= (uint32_t)kWave) {                     // hops_small.h
        if (net->dim_emb_pad == 64) k_hops_small<4, 4, kModeFixed, 8><<<n_query, kWave, 0, st>>>(a, 64);
        else if (net->dim_emb_pad == 128) k_hops_small<8, 8, kModeFixed, 8><<<n_query, kWave, 0, st>>>(a, 128);
        else k_hops_small<16, 16, kModeFixed, 8><<<n_query, kWave, 0, st>>>(a, 256);
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    if (mid_supported(a, max_slots)) {                      // hops_mid.h: 65 .. 1 024 slots at bAbI width
        launch_mid(a, max_slots, n_query, st);
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    // memories of 65..256 slots run one wavefront per query with the histogram softmax: four times
    // as many queries resident per CU and no cross-wavefront barriers
    const dim3 grid(n_query), block(max_slots <= 256 ? kWave : kBlock);
#define QM_LAUNCH_HOPS_W(LPR, UN, NT, MINW, W7)                                                         \
    do {                                                                                                \
        if (lds > kLdsDefaultLimit)         /* beyond the default 64 KB: raise the limit (no cached state:  */ \
            QM_HIP(hipFuncSetAttribute((const void *)k_hops_fixed<LPR, UN, NT, MINW, W7>, /* thread- and  */ \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); /* device-safe */ \
        k_hops_fixed<LPR, UN, NT, MINW, W7><<<grid, block, lds, st>>>(a);                               \
    } while (0)
#define QM_LAUNCH_HOPS(LPR, UN, NT, MINW)                                                               \
    do { if (w7) QM_LAUNCH_HOPS_W(LPR, UN, NT, MINW, true); else QM_LAUNCH_HOPS_W(LPR, UN, NT, MINW, false); } while (0)
    bool w7 = !qm_tuning().no_w7;
    for (uint32_t h = 0; h < net->n_hop; h++) w7 = w7 && net->att[h].iwl + net->att[h].frac == 7;
    // (bounding the kernel to 96 VGPRs for a fifth wavefront per SIMD measured 1-3 % slower: MINW stays 1)
    if (net->dim_emb_pad == 64) QM_LAUNCH_HOPS(4, kUnrollDefault, true, 1);
    else if (net->dim_emb_pad == 256) QM_LAUNCH_HOPS(16, kUnrollDefault, true, 1);
    else QM_LAUNCH_HOPS(8, kUnrollDefault, true, 1);
#undef QM_LAUNCH_HOPS
#undef QM_LAUNCH_HOPS_W
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

}  // extern "C"
