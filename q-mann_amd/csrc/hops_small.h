// hops_small.h -- every hop of a query whose memory has at most 64 slots (bAbI stories: 2..50
// sentences), one wavefront per query.  Slot r lives in lane r for the softmax, so there is no
// histogram, no survivor list and no cross-wavefront barrier; 3 KB of LDS.  Stage by stage the same
// arithmetic as the streaming kernels (batch_hops.hip, batch_hops_ham.hip); MODE selects the score.
#pragma once
#include "ham_common.h"

namespace {

enum { kModeFixed = 3 };      // beside kModeAppx / kModeV0 / kModeV1

// LPR: lanes per value / lin_map row (Dp / 16); LPRK: lanes per key row (key_row_bytes / 16)
template <int LPR, int LPRK, int MODE, int NB>
__global__ void __launch_bounds__(kWave)
k_hops_small(const HopArgs a, const uint32_t key_row_bytes)
{
    constexpr uint32_t Dp = LPR * 16, RPWK = kWave / LPRK;
    // bAbI width (Dp = 64): everything a hop reads from global memory -- its value rows, its linear-map
    // rows (at its start) and the NEXT hop's key rows (as soon as the scan is done; addresses do not depend on
    // the data) -- is requested early, so no global-memory latency sits between the stages of a hop.  Value rows pass through LDS
    // (row-major tile) because the read-out owns columns, not row pieces.
    constexpr bool PF = (LPR == 4 && LPRK <= 4);
    constexpr int NK = PF ? LPRK : 1;                   // key loads covering 64 rows
    __shared__ __attribute__((aligned(16))) uint8_t vt[PF ? kWave * 64 : 16];
    __shared__ float u_f[256];
    __shared__ float o_f[256];
    __shared__ short ku[256];
    __shared__ __attribute__((aligned(8))) uint8_t ub[256];
    __shared__ uint64_t upl[4 * 8];
    __shared__ int16_t sc[kWave];
    const uint32_t lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t r0 = a.row_off[q];
    const uint32_t S_in = a.row_off[q + 1] - r0;
    const uint32_t S = S_in < a.max_slots ? S_in : a.max_slots;           // cut to the caller's bound (<= 64 for this kernel)
    const uint32_t D = a.D;
    const uint32_t subk = lane / LPRK, chunkk = lane % LPRK;

    i32x4 kq[NK], vq[4], hq[4];
    auto load_keys = [&](i32x4 (&k)[NK], uint32_t h) {
        const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.key_hop_stride + (size_t)r0 * key_row_bytes + chunkk * 16;
#pragma unroll
        for (int j = 0; j < NK; j++) {
            const uint32_t r = j * RPWK + subk;
            k[j] = i32x4{0, 0, 0, 0};
            if (r < S) k[j] = *(const i32x4 *)(kb + (size_t)r * key_row_bytes);
        }
    };
    if (PF) load_keys(kq, 0);

    for (uint32_t c = lane; c < 256; c += kWave) u_f[c] = (c < D) ? a.u0[(size_t)q * D + c] : 0.0f;
    __syncthreads();

    for (uint32_t h = 0; h < a.n_hop; h++) {
        const QFmt fa = a.act[h], fm = a.att[h], fb = a.bin, fw = a.w[h];
        const int maxa = (1 << (fa.iwl + fa.frac)) - 1;
        if (PF) {
            const uint32_t sub = lane / LPR, chunk = lane % LPR;
            const uint8_t *vb = (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + chunk * 16;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t r = j * 16 + sub;
                vq[j] = i32x4{0, 0, 0, 0};
                if (r < S) vq[j] = *(const i32x4 *)(vb + (size_t)r * Dp);
            }
            if (a.en_lin_map) {
                const uint8_t *hb = (const uint8_t *)a.lin_map[h] + chunk * 16;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t r = j * 16 + sub;
                    hq[j] = i32x4{0, 0, 0, 0};
                    if (r < D) hq[j] = *(const i32x4 *)(hb + (size_t)r * Dp);
                }
            }
        }
        // codes of u: Q_bin for the scores (fixed) and the linear map; sign-magnitude Q_att bytes and
        // their bit planes for the Hamming forms
        for (uint32_t c = lane; c < Dp; c += kWave) {
            const float uv = u_f[c];
            ku[c] = (short)((c < D) ? qm_code_or_sign(uv, fb.iwl, fb.frac) : 0);
            if (MODE != kModeFixed) {
                const float ua = relu_if(uv, hop_relu(a, h));            // what the attention reads
                const uint32_t ubyte = ham_ubyte(ua, fm, c < D);
                ub[c] = (uint8_t)ubyte;
                if (mode_is_planes(MODE)) {
#pragma unroll
                    for (int i = 0; i < NB; i++) {
                        const uint64_t word = __ballot((ubyte >> (7 - i)) & 1u);
                        if (lane == 0) upl[(c / kWave) * 8 + i] = word;
                    }
                }
            }
        }
        __syncthreads();

        // scores: LPRK lanes per key row, 64 / LPRK rows per load
        ScanConst csc;
        uint32_t csh = 0;
        float unit = 1.0f;
        if (S > 0) {
            const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.key_hop_stride + (size_t)r0 * key_row_bytes + chunkk * 16;
            auto scan = [&](auto lane_sum, int lim, bool wrap = false) {     // wrap: mode 3's final quantiser (appx_clamp, ham_common.h)
                if (PF) {
#pragma unroll
                    for (int j = 0; j < NK; j++) {
                        const uint32_t r = j * RPWK + subk;
                        if (j * RPWK < S) {                                  // wavefront-uniform
                            const int v = row_lanes_sum<LPRK>(lane_sum(kq[j]));
                            if (chunkk == 0 && r < S) sc[r] = (int16_t)(v > lim ? lim : (v < -lim ? -lim : ((wrap && v == -lim) ? 0 : v)));
                        }
                    }
                    return;
                }
                for (uint32_t base = 0; base < S; base += RPWK) {
                    const uint32_t r = base + subk;
                    i32x4 x = {0, 0, 0, 0};
                    if (r < S) x = *(const i32x4 *)(kb + (size_t)r * key_row_bytes);
                    const int v = row_lanes_sum<LPRK>(lane_sum(x));
                    if (chunkk == 0 && r < S) sc[r] = (int16_t)(v > lim ? lim : (v < -lim ? -lim : ((wrap && v == -lim) ? 0 : v)));
                }
            };
            if (MODE == kModeFixed) {
                unit = qm_scale_down(1.0f, fm.frac);
                csh = make_scan_const(csc, ku, chunkk * 16, fm.iwl + fm.frac, (int)fb.frac, hop_relu(a, h), fb.iwl + fb.frac == 0);
                scan([&](const i32x4 x) { return lane_row_sum(x, csc, csh); }, (1 << (fm.iwl + fm.frac)) - 1);   // Qm, lib/layer_cuda.cu:135
            } else if (mode_is_appx(MODE)) {
                unit = 1.0f / 1024.0f;                          // 2^-(n-1) . 2^ATTENTION_CONST_SCALE, n = 8
                AppxConst c;
                make_appx_const(c, ub, chunkk * 16, D);
                const int lim = 1 << (fm.iwl + 10);             // Q(iwl, 31-iwl) clamps at +-2^iwl
                const uint32_t kind = MODE == kModeAppxMq ? ham_kind_of(a, h) : (uint32_t)kHamSame;     // (wavefront-uniform)
                if (kind == kHamFine) scan([&](const i32x4 x) { return appx_lane_sum_k<kHamFine>(x, c); }, lim, true);
                else if (kind == kHamCoarse) scan([&](const i32x4 x) { return appx_lane_sum_k<kHamCoarse>(x, c); }, lim, true);
                else scan([&](const i32x4 x) { return appx_lane_sum(x, c); }, lim, true);
            } else if (mode_is_planes(MODE)) {
                if (MODE == kModeV1) unit = qm_scale_down(1.0f, NB);
                PlaneConst c;
                make_plane_const<NB>(c, upl, chunkk, D);
                scan([&](const i32x4 x) { return plane_lane_sum<MODE, NB>(x, c); }, 32767);
            } else {
                if (MODE == kModeV1Bytes) unit = qm_scale_down(1.0f, NB);
                HamByteConst c;
                make_hambyte_const<MODE, NB>(c, ub, chunkk * 16, D);
                scan([&](const i32x4 x) { return hambyte_lane_sum<MODE, NB>(x, c); }, 32767);
            }
        }
        if (PF) {
            const uint32_t sub = lane / LPR, chunk = lane % LPR;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j * 16 < (int)S) *(i32x4 *)(vt + (size_t)(j * 16 + sub) * 64 + chunk * 16) = vq[j];
            if (h + 1 < a.n_hop) load_keys(kq, h + 1);      // the scan is done with this hop's keys
        }
        __syncthreads();

        // softmax over slots, slot r in lane r
        const bool live = lane < S;
        const int code = live ? (int)sc[lane] : 0;
        const SmCfg smc = sm_cfg(a, h);
        const float xs = live ? sm_scaled((float)code * unit, smc) : -INFINITY;      // (float)code . unit is exact
        float mx = xs;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = live ? sm_exp(xs - mx, smc) : 0.0f;              // score - max, exact on the score grid
        double total = (double)e;
        if (smc.base == QMANN_SOFTMAX_EXP) {             // the CUDA kernel's double total (lib/layer_cuda.cu:2024-2042)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        } else {
            total = (double)wave_serial_sum_f32(e, S);     // the CPU softmax's float total, in slot order (hops_common.h)
        }
        const float p = live ? sm_quot(e, total, smc) : 0.0f;
        const int kp = live ? qm_code(p, fa.iwl, fa.frac) : 0;
        if (live) {
            const size_t tb = (size_t)h * a.rows_total + r0 + lane;
            if (a.tap_codes) a.tap_codes[tb] = code;
            if (a.tap_scores) a.tap_scores[tb] = (float)code * unit;
            if (a.tap_probs) a.tap_probs[tb] = p;
        }

        // read-out over the rows whose weight code is non-zero (the others add exact zeros)
        const uint64_t survivors = __ballot(kp != 0);
        for (uint32_t c = lane; c < Dp; c += kWave) {
            const uint8_t *vb = PF ? (const uint8_t *)vt + c
                                   : (const uint8_t *)a.vals + (size_t)h * a.hop_stride + (size_t)r0 * Dp + c;
            int acc = 0;
            for (uint64_t m = survivors; m; m &= m - 1) {
                const int r = __builtin_ctzll(m);
                acc += qm_mul_code(__shfl(kp, r), sm_decode(vb[(size_t)r * Dp]), fa.frac, maxa);
            }
            acc = acc > maxa ? maxa : (acc < -maxa ? -maxa : acc);
            o_f[c] = qm_scale_down((float)acc, fa.frac);
        }
        __syncthreads();
        // the key scan's lane constants serve the linear map as they are when both formats have the
        // same word length (always so at BW_WL 8): same u codes, same shift
        const bool reuse = MODE == kModeFixed && S > 0 && (fw.iwl + fw.frac == fm.iwl + fm.frac) && !hop_relu(a, h);
        linmap_update<Dp>(a, q, h, ku, u_f, o_f, lane, &csc, csh, reuse, PF ? hq : nullptr);
    }
    for (uint32_t c = lane; c < D; c += kWave) a.u_out[(size_t)q * D + c] = relu_if(u_f[c], a.en_non_lin != 0);
}

}  // namespace
