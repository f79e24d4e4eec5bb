// hops_mid.h -- memories of 65 .. 1 024 slots at bAbI width (64-byte rows), fixed-point attention: between the sizes the
// reference reaches (MAX_SEN_LEN <= 64: hops_lean.h) and the long memories the streaming kernel (batch_hops.hip) is built for.
//
// At these sizes the streaming kernel's per-hop fixed costs are as large as its scan: a 256-bin score histogram with its
// per-bin exponentials, divisions and atomics (sized for 10 000 slots), the lane constants rebuilt from 16 codes per lane, the
// linear-map rows fetched from L2 every hop.  This kernel is hops_lean.h stretched over more rows instead:
//   * persistent workgroups of 8 wavefronts, one wavefront per query at a time; exp table per hop and the pre-split linear maps
//     in LDS for the kernel's lifetime (lean_stage_tables), operand constants published once per column (publish_const);
//   * the scan streams the key plane in groups of 64 rows (4 x 16-byte loads per lane), two groups in flight, the next hop's
//     (or the next query's) first two groups requested before the softmax / read-out / linear map of the current hop;
//   * scores are 8-bit codes in a per-wavefront LDS array (1 KB); the softmax is one table look-up per slot -- exp(x - max)
//     takes at most 255 values on the score grid -- and a double total (lib/layer_cuda.cu:2024-2042); the CPU softmax's bases
//     (2^x, exp_plan: lib/layer.c:1196-1243) take their tables and the serial float total of hops_common.h;
//   * a slot can carry a non-zero read-out weight Q(p) only if e >= total . 2^-frac (up to float rounding): only those few
//     slots get the exact quotient (float)((double)e / total) of the reference, and only their value rows are read (the others
//     contribute exact zeros, :562);
//   * the hop's tail (linear map on the LDS images, u' = Qa(Qa(Hu) + Qa(o))) is hops_lean.h's.
// Same arithmetic, stage by stage, as k_hops_fixed; tests/test_gpu_mid.py holds the two against each other and this kernel
// against the oracle.  One stage is NOT order-identical between the two: the e^x base's double total is a per-slot tree here and
// a histogram sum (count . e per score code) there -- equal to the last bit or two of a double, which shows only where a weight
// sits exactly on a truncation step of Q(p) (two tied top scores at one or two fraction bits: 1 in ~240 random-format soak
// cases; tools/soak.py reports the count).  Neither order is the reference's serial one (lib/layer_cuda.cu:2024); the oracle
// sums serially and such cases are the ones the tests excuse with the oracle's evidence that a p lies on a step.  Taken by qmann_hops_i8 for 64 < max_slots <= 1 024 when mid_supported() holds.
#pragma once
#include "hops_lean.h"

namespace {

constexpr uint32_t kMidMaxSlots = 1024;
#ifndef QM_MID_WAVES
#define QM_MID_WAVES 8                                    // (10 -- five wavefronts per SIMD at 96 registers -- measured 35 % slower)
#endif
constexpr int kMidWaves = QM_MID_WAVES;                 // wavefronts per workgroup; two workgroups per CU
constexpr int kMidBlock = kMidWaves * kWave;

template <bool W7>
__global__ void __launch_bounds__(kMidBlock, (2 * kMidWaves + 3) / 4)      // two workgroups per CU
k_hops_mid(const HopArgs a, const LeanArgs la)
{
    constexpr uint32_t Dp = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));   // (uniform: query bookkeeping stays in SGPRs)
    const uint32_t sub = lane >> 2, chunk = lane & 3u;
    const uint32_t D = a.D, H = a.n_hop;
    float *etab = (float *)smem;                                        // [H][256]
    uint8_t *lmap = smem + H * 1024u;                                   // [H][64][3][64] (lean_stage_tables)
    uint8_t *wbase = lmap + (la.lm_in_lds ? H * kLmHopBytes : 0u);
    const uint32_t wslice = kLwBytes + la.rows_pad;                     // rows_pad: score bytes per wavefront (multiple of 64)
    uint8_t *lw = wbase + wave * wslice;                                // small arrays
    int8_t *sc = (int8_t *)(lw + kLwBytes);                             // score codes, slot r
    lean_stage_tables(a, la, etab, lmap, tid, kMidBlock);
    __syncthreads();

    const uint32_t q_stride = gridDim.x * kMidWaves, n_query = a.rows_total;   // (rows_total carries the query count, see launcher)
    uint32_t q = blockIdx.x * kMidWaves + wave;
    if (q >= n_query) return;
    auto slots_of = [&](uint32_t qq, uint32_t &r0_, uint32_t &S_) {
        r0_ = a.row_off[qq];
        const uint32_t S_in = a.row_off[qq + 1] - r0_;
        S_ = S_in < a.max_slots ? S_in : a.max_slots;                   // a story longer than the caller's bound is cut to it
    };
    uint32_t r0, S;
    slots_of(q, r0, S);

    // Group g of a query = its rows 64 g .. 64 g + 63: four 16-byte BUFFER loads per lane (rows 64 g + 16 j + sub, piece
    // `chunk`) through a raw buffer resource that spans exactly the story's key bytes of the hop: a row past the story's end
    // reads as zeros and costs no memory traffic, whatever the group.  Round 3 used plain global loads and had to keep them
    // inside the plane by hand -- a short last group was moved back to end at the story's end (its repeated rows re-read: 1.10 x
    // the algorithmic bytes at 200 slots), a story shorter than a group read into its neighbour, the batch's last stories took
    // a separate clamped loop.  All of that is gone: one loop, algorithmic traffic, and the compiler still counts the loads
    // exactly (no branch around a load: s_waitcnt vmcnt(4) before a group's sums, the other group still in flight).
    const uint32_t lane_off = sub * Dp + chunk * 16u;
    i32x4 xa[4], xb[4];
    auto key_rsrc = [&](uint32_t h, uint32_t r0_, uint32_t S_) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)a.keys + (size_t)h * a.key_hop_stride + (size_t)r0_ * Dp), 0,
                                                 (int)(S_ * Dp), kRawBufferFlags);
    };
    auto issue = [&](i32x4 (&x)[4], __amdgpu_buffer_rsrc_t rs, uint32_t g) {
        const uint32_t off = lane_off + g * (64u * Dp);
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + (uint32_t)j * 16u * Dp), 0, kBufferNt);
    };
    // the passes of 16 rows group g really has (4 for a full group, fewer for a story's last one, none past its end): the
    // arithmetic of the passes it lacks is skipped (their loads ran, branch-free, and brought zeros)
    auto passes_of = [&](uint32_t g, uint32_t S_) { const uint32_t left = S_ > g * 64u ? S_ - g * 64u : 0u; return left >= 64u ? 4u : (left + 15u) / 16u; };
    { const __amdgpu_buffer_rsrc_t rs = key_rsrc(0, r0, S); issue(xa, rs, 0); issue(xb, rs, 1); }
    float u_next = (lane < D) ? a.u0[(size_t)q * D + lane] : 0.0f;

    for (; q < n_query; q += q_stride) {
        const uint32_t qn = q + q_stride;
        uint32_t r0n = 0, Sn = 0;
        if (qn < n_query) slots_of(qn, r0n, Sn);
        float u = u_next;
        const uint32_t n_g = (S + 63u) / 64u;
        for (uint32_t h = 0; h < H; h++) {
            const QFmt fa = a.act[h], fm = a.att[h], fb = a.bin, fw = a.w[h];
            const int maxa = (1 << (fa.iwl + fa.frac)) - 1;
            const uint32_t wl_m = fm.iwl + fm.frac, wl_w = fw.iwl + fw.frac;
            const int maxm = (1 << wl_m) - 1;
            const bool relu = hop_relu(a, h);
            // ---- column c: operand codes, published for the scan --------------------------------------------------
            const int kb_code = (lane < D) ? qm_code_or_sign(u, fb.iwl, fb.frac) : 0;
            int ka = kb_code;
            if (relu && ka < 0) ka = (fb.iwl + fb.frac == 0) ? 1 : 0;               // see make_scan_const
            publish_const<W7>(lw, lane, ka, wl_m, (int)fb.frac);
            wave_sync();
            ScanConst csc;
            uint32_t csh = fetch_scan_const(csc, lw, chunk, wl_m);

            // ---- scan: scores of all rows, 64 rows per step ---------------------------------------------------------
            const __amdgpu_buffer_rsrc_t rs = key_rsrc(h, r0, S);
            int mx = -128;
            auto consume = [&](const i32x4 (&x)[4], uint32_t g, uint32_t p_hi) {
                int s[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (j < p_hi) s[j] = row_lanes_sum<4>(lane_sum_w<W7>(x[j], csc, csh));      // wavefront-uniform
                // every lane of a row group holds its row's sum: lane (sub, chunk) keeps row 64 g + 16 chunk + sub
                int v = s[0];
                v = chunk == 1u ? s[1] : v; v = chunk == 2u ? s[2] : v; v = chunk == 3u ? s[3] : v;
                const int code = v > maxm ? maxm : (v < -maxm ? -maxm : v);     // Qm of the row sum (lib/layer_cuda.cu:135)
                const uint32_t r = g * 64u + chunk * 16u + sub;
                if (r < S) { sc[r] = (int8_t)code; mx = code > mx ? code : mx; }
            };
            {
                __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): groups 0 and 1, requested a hop ago, have landed
                uint32_t g = 0;
                for (; g + 2 < n_g; g += 2) {                         // steady state: two groups in flight, no branch around a load
                    consume(xa, g, 4);
                    issue(xa, rs, g + 2);
                    consume(xb, g + 1, 4);
                    issue(xb, rs, g + 3);
                }
                consume(xa, g, passes_of(g, S));
                consume(xb, g + 1, passes_of(g + 1, S));              // (a group past the story's end: no pass, nothing stored)
                // in flight during the rest of the hop: the next hop's first two groups, or the next query's
                // (unconditionally: with nothing to come the resource is empty -- r0n = Sn = 0 -- and the loads bring zeros for free)
                const bool more = h + 1 < H;
                const __amdgpu_buffer_rsrc_t rn = key_rsrc(more ? h + 1 : 0u, more ? r0 : r0n, more ? S : Sn);
                issue(xa, rn, 0); issue(xb, rn, 1);
            }
            if (h + 1 == H && qn < n_query) u_next = (lane < D) ? a.u0[(size_t)qn * D + lane] : 0.0f;
            wave_sync();                                              // the score bytes are visible to every lane

            // ---- softmax over slots: table look-ups and a double total ------------------------------------------------
            int acc = 0;
            if (S > 0) {
                const int mxc = wave_max_i32(mx);
                const float *et = etab + h * 256u;
                const SmCfg smc = sm_cfg(a, h);
                double total;
                if (smc.base == QMANN_SOFTMAX_EXP) {                  // the CUDA kernel's double total (lib/layer_cuda.cu:2024-2042)
                    double part = 0.0;
                    for (uint32_t g = 0; g < n_g; g++) {
                        const uint32_t s = g * 64u + lane;
                        if (s < S) part += (double)et[(uint32_t)(mxc - (int)sc[s])];
                    }
                    total = wave_sum_f64(part);
                } else {                                              // 2^x, exp_plan: the CPU softmax's float total, slot by slot (lib/layer.c:1236)
                    total = (double)wave_serial_total_f32(S, lane, [&](uint32_t s) { return et[(uint32_t)(mxc - (int)sc[s])]; });
                }
                // Q(p) != 0 needs p >= 2^-frac; p = (float)(e / total) rounds by at most 2^-24 relative, so a slot with
                // e < total . 2^-frac . (1 - 2^-20) cannot reach it.  Only the others get the exact quotient.
                const float thr = (float)(total * (double)qm_scale_down(1.0f, fa.frac) * (1.0 - 9.5367431640625e-07));
                const uint8_t *vb = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + lane;
                for (uint32_t g = 0; g < n_g; g++) {
                    const uint32_t s = g * 64u + lane;
                    const float e = (s < S) ? et[(uint32_t)(mxc - (int)sc[s])] : 0.0f;
                    const bool cand = (s < S) && e >= thr;
                    if (__ballot(cand) == 0) continue;                // wavefront-uniform
                    int kq = 0;
                    if (cand) {
                        const float p = sm_quot(e, total, smc);
                        kq = (int)__builtin_ldexpf(p, (int)fa.frac);   // Q(p) for 0 <= p <= 1: trunc(p . 2^frac), saturated
                        kq = kq > maxa ? maxa : kq;
                    }
                    // read-out over the rows of this group whose weight code is non-zero (lane c owns column c), four rows per
                    // round with their value bytes requested together
                    for (uint64_t m = __ballot(kq != 0); m;) {
                        int rr[4], kk[4];
                        uint32_t bb[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            rr[i] = m ? __builtin_ctzll(m) : -1;      // (wavefront-uniform)
                            m &= m - 1;                               // (0 stays 0)
                            kk[i] = rr[i] >= 0 ? __builtin_amdgcn_readlane(kq, rr[i] >= 0 ? rr[i] : 0) : 0;
                            bb[i] = rr[i] >= 0 ? vb[(size_t)(g * 64u + (uint32_t)rr[i]) * Dp] : 0u;
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            uint32_t t = ((uint32_t)kk[i] * (bb[i] & 0x7Fu)) >> fa.frac;   // |Q(p) . v| / 2^frac toward zero
                            t = t > (uint32_t)maxa ? (uint32_t)maxa : t;
                            acc += (bb[i] & 0x80u) ? -(int)t : (int)t;
                        }
                    }
                }
            }
            lean_finish_hop<W7>(a, h, lane, lw, lmap, u, [&]() { return acc; }, kb_code, csc, csh, wl_w == wl_m && !relu);
        }
        if (lane < D) a.u_out[(size_t)q * D + lane] = relu_if(u, a.en_non_lin != 0);
        r0 = r0n; S = Sn;
    }
}

// what this kernel covers: fixed-point attention with the softmax's exponential from a table -- e^x, or the CPU softmax's 2^x /
// exp_plan with its serial float total (no shift-based form, no scale layer),
// 64-byte rows, 65 .. 1 024 slots, no taps; everything else keeps the streaming kernel
inline bool mid_supported(const HopArgs &a, uint32_t max_slots)
{
    return a.Dp == 64 && max_slots > (uint32_t)kWave && max_slots <= kMidMaxSlots && a.softmax_base <= QMANN_SOFTMAX_EXP_PLAN &&
           !a.softmax_shift && !a.en_att_scale && !a.tap_codes && !a.tap_scores && !a.tap_probs && !a.tap_o && !a.tap_u &&
           !qm_tuning().no_mid;
}

template <bool W7>
inline void launch_mid_w(HopArgs a, uint32_t max_slots, uint32_t n_query, hipStream_t st)
{
    LeanArgs la{};
    la.rows_pad = (max_slots + 63u) & ~63u;
    la.exp_table = 1u;
    la.lm_in_lds = a.en_lin_map ? 1u : 0u;
    a.rows_total = n_query;                                               // (no taps: the field carries the query count)
    const size_t lds = a.n_hop * 1024u + (la.lm_in_lds ? a.n_hop * kLmHopBytes : 0u) + (size_t)kMidWaves * (kLwBytes + la.rows_pad);
    if (lds > kLdsDefaultLimit)
        QM_HIP(hipFuncSetAttribute((const void *)k_hops_mid<W7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t need = (n_query + kMidWaves - 1) / kMidWaves;
    const uint32_t resident = qm_resident_groups(kMidWaves, (2 * kMidWaves + 3) / 4, lds);   // (the kernel's __launch_bounds__)
    k_hops_mid<W7><<<need < resident ? need : resident, kMidBlock, lds, st>>>(a, la);
}

inline void launch_mid(const HopArgs &a, uint32_t max_slots, uint32_t n_query, hipStream_t st)
{
    bool w7 = true;
    for (uint32_t h = 0; h < a.n_hop; h++)
        w7 = w7 && a.att[h].iwl + a.att[h].frac == 7 && (!a.en_lin_map || a.w[h].iwl + a.w[h].frac == 7) &&
             a.act[h].iwl + a.act[h].frac == 7;                       // (lean_finish_hop<true> folds every word length to 7)
    if (w7) launch_mid_w<true>(a, max_slots, n_query, st);
    else launch_mid_w<false>(a, max_slots, n_query, st);
}

}  // namespace
