// batch_hops_ham.hip -- fused hop kernel for the Hamming family of attention scores.
//
// Same skeleton as batch_hops.hip (one workgroup per query, all hops, streaming row scan,
// sparse exact read-out, in-kernel linear map); what changes is the score of a slot:
//
//   QMANN_ATT_APPX        the reference's live CUDA "approximate attention"
//                         (lib/layer_cuda.cu:355-541 with :218-326), on sign-magnitude int8 keys.
//                         For operands on the Q(iwl_att.7-iwl_att) grid the 32-bit word procedure
//                         reduces to byte arithmetic: same sign   -> +(127 - |ka - kb|)
//                                                     opposite    -> +-(127 - ((ka + kb) & 127)),
//                         positive only when ka + kb carries out of 7 bits and the larger
//                         operand (the key on ties) is the positive one; the row score is the sum
//                         in units of 2^-10, clamped at +-2^iwl.  Sums of |ka - kb| over the
//                         same-sign bytes of a dword are one v_sad_u8.
//   QMANN_ATT_HAMMING_V0  bit-agreement count over the top num_bit bits (lib/common.c:223-246)
//   QMANN_ATT_HAMMING_V1  signed, weighted bit agreement (lib/common.c:249-312)
//                         both on PACKED BINARY CODES: bit-plane i of a 64-column group is one
//                         uint64 (plane 0 = sign, plane i = magnitude bit 7-i), a row is
//                         [Dp/64 groups][num_bit planes]; agreement is ~(K ^ U), counted with
//                         popcount (v_bcnt), split by sign agreement for V1.
//
// Scores are not on an 8-bit grid here; every mode's row score is bounded by 127.D <= 32512 units,
// so they are kept as int16 in LDS.  V0 evaluates the softmax once per distinct count (histogram),
// the other modes per slot (max, sum of exp in double, quotient), as lib/layer_cuda.cu:1969-2060 does.
#include "ham_common.h"
#include "hops_lean.h"

namespace {

constexpr uint32_t kOffUb = kOffHist;                 // u8  [256]  sign-magnitude bytes of Q_att(u)   (reuses the
constexpr uint32_t kOffUpl = kOffHist + 256;          // u64 [4][8] bit-planes of u                    histogram area)
static_assert(kOffUpl % 8 == 0 && kOffUpl + 4 * 8 * 8 <= kOffPtab, "u planes must fit the histogram area");

// LPRK: lanes per key row; DP: padded embedding width; MODE; NB: planes (packed modes)
// (bounding the kernel to 96 VGPRs for a fifth workgroup per CU spills: 0.82 -> 0.65-0.70 of peak for V0, 0.75 -> 0.73-0.75 for APPX;
// removing the histogram atomics changes nothing: the kernel is not bound by them)
template <int LPRK, int DP, int MODE, int NB>
__global__ void __launch_bounds__(kBlock)
k_hops_ham(const HopArgs a, const uint32_t key_row_bytes, const uint32_t lds_slots)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t *ub = (uint8_t *)(smem + kOffUb);
    uint64_t *upl = (uint64_t *)(smem + kOffUpl);
    float *u_f = (float *)(smem + kOffU);
    float *o_f = (float *)(smem + kOffO);
    short *ku = (short *)(smem + kOffKu);
    uint32_t *live_row = (uint32_t *)(smem + kOffLiveRow);
    uint8_t *live_kp = (uint8_t *)(smem + kOffLiveKp);
    uint32_t *misc = (uint32_t *)(smem + kOffMisc);
    double *red = (double *)(smem + kOffRed);
    using score_t = int16_t;
    score_t *sc = (score_t *)(smem + kOffScores);

    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t q = blockIdx.x;
    const uint32_t r0 = a.row_off[q];
    const uint32_t S_in = a.row_off[q + 1] - r0;
    const uint32_t S = S_in < a.max_slots ? S_in : a.max_slots;   // never index LDS past what the launch reserved
    const uint32_t D = a.D;
    // V0 tables sit behind the score array (lds_slots is the launch's slot capacity)
    unsigned char *tab = smem + kOffScores + (((size_t)lds_slots * sizeof(score_t) + 15) & ~(size_t)15);
    uint32_t *v0_hist = (uint32_t *)tab;
    const uint32_t nbins = NB * D + 1;
    float *v0_p = (float *)(tab + v0_hist_bytes(nbins));
    uint8_t *v0_kp = (uint8_t *)(tab + 2 * v0_hist_bytes(nbins));

    u_f[tid] = (tid < D) ? a.u0[(size_t)q * D + tid] : 0.0f;
    __syncthreads();

    for (uint32_t h = 0; h < a.n_hop; h++) {
        const QFmt fa = a.act[h], fm = a.att[h], fb = a.bin;
        // codes of u: Q_bin for the linear map, sign-magnitude Q_att bytes for the attention
        const float uv = u_f[tid];
        ku[tid] = (short)((tid < D) ? qm_code_or_sign(uv, fb.iwl, fb.frac) : 0);
        const float ua = relu_if(uv, hop_relu(a, h));                    // what the attention reads
        const uint32_t ubyte = ham_ubyte(ua, fm, tid < D);
        ub[tid] = (uint8_t)ubyte;
        if (mode_is_planes(MODE)) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const uint64_t word = __ballot((ubyte >> (7 - i)) & 1u);
                if (lane == 0) upl[wave * 8 + i] = word;       // wavefront w covers columns 64w .. 64w+63
            }
        }
        if (tid == 0) misc[0] = 0u;
        if (mode_is_v0(MODE))
            for (uint32_t d = tid; d < nbins; d += kBlock) v0_hist[d] = 0u;
        __syncthreads();

        float scale = 1.0f;
        if (S > 0) {
            const uint8_t *kb = (const uint8_t *)a.keys + (size_t)h * a.key_hop_stride + (size_t)r0 * key_row_bytes;
            auto retire = [&](uint32_t r, int v) {
                sc[r] = (score_t)v;
                if (mode_is_v0(MODE)) atomicAdd(&v0_hist[v], 1u);
            };
            if (mode_is_appx(MODE)) {
                scale = 1.0f / 1024.0f;                         // 2^-(n-1) . 2^ATTENTION_CONST_SCALE, n = 8
                const int lim = 1 << (fm.iwl + 10);             // final Q(iwl, 31-iwl) clamps at +-2^iwl
                AppxConst c;
                make_appx_const(c, ub, (lane % LPRK) * 16, D);
                auto retire_c = [&](uint32_t r, int v) { sc[r] = (score_t)appx_clamp(v, lim); };
                auto run = [&](auto row_sum) {
                    if (S >= (kWave / LPRK) * 4) scan_rows<LPRK, 4, true, kWaves>(kb, S, row_sum, retire_c, lane, wave);
                    else scan_rows_short<LPRK>(kb, S, row_sum, retire_c, lane, wave, kWaves);
                };
                const uint32_t kind = MODE == kModeAppxMq ? ham_kind_of(a, h) : (uint32_t)kHamSame;     // (workgroup-uniform)
                if (kind == kHamFine) run([&](const i32x4 x) { return appx_lane_sum_k<kHamFine>(x, c); });
                else if (kind == kHamCoarse) run([&](const i32x4 x) { return appx_lane_sum_k<kHamCoarse>(x, c); });
                else run([&](const i32x4 x) { return appx_lane_sum(x, c); });
            } else if (mode_is_planes(MODE)) {
                if (MODE == kModeV1) scale = qm_scale_down(1.0f, NB);
                PlaneConst c;
                make_plane_const<NB>(c, upl, lane % LPRK, D);
                auto row_sum = [&](const i32x4 x) { return plane_lane_sum<MODE, NB>(x, c); };
                if (S >= (kWave / LPRK) * 4) scan_rows<LPRK, 4, true, kWaves>(kb, S, row_sum, retire, lane, wave);
                else scan_rows_short<LPRK>(kb, S, row_sum, retire, lane, wave, kWaves);
            } else {
                if (MODE == kModeV1Bytes) scale = qm_scale_down(1.0f, NB);
                HamByteConst c;
                make_hambyte_const<MODE, NB>(c, ub, (lane % LPRK) * 16, D);
                auto row_sum = [&](const i32x4 x) { return hambyte_lane_sum<MODE, NB>(x, c); };
                if (S >= (kWave / LPRK) * 4) scan_rows<LPRK, 4, true, kWaves>(kb, S, row_sum, retire, lane, wave);
                else scan_rows_short<LPRK>(kb, S, row_sum, retire, lane, wave, kWaves);
            }
        }
        __syncthreads();

        // softmax over slots, evaluated per slot (lib/layer_cuda.cu:1895-1916, 1969-2060)
        uint32_t n_live = 0;
        if (mode_is_v0(MODE) && S > 0) {
            // one exp per distinct count; normaliser sum_d count[d] . e[d] in double (lib/layer_cuda.cu:2024-2042)
            const SmCfg smc = sm_cfg(a, h);
            float xmax = -INFINITY;
            for (uint32_t d = tid; d < nbins; d += kBlock)
                if (v0_hist[d]) xmax = fmaxf(xmax, sm_scaled((float)d, smc));
            xmax = block_max<float>(xmax, (float *)red, lane, wave);
            double part = 0.0;
            for (uint32_t d = tid; d < nbins; d += kBlock) {
                const float e = sm_exp(sm_scaled((float)d, smc) - xmax, smc);
                v0_p[d] = e;
                if (v0_hist[d]) part += (double)v0_hist[d] * (double)e;
            }
            const double total = smc.base == QMANN_SOFTMAX_EXP ? block_sum_double(part, red, lane, wave)
                                                               : block_serial_total_f32(S, lane, wave, red, [&](uint32_t r) { return v0_p[sc[r]]; });
            for (uint32_t d = tid; d < nbins; d += kBlock) {
                const float p = v0_hist[d] ? sm_quot(v0_p[d], total, smc) : 0.0f;
                v0_p[d] = p;
                v0_kp[d] = (uint8_t)qm_code(p, fa.iwl, fa.frac);
            }
            __syncthreads();
            if (a.tap_codes || a.tap_scores || a.tap_probs) {
                const size_t tb = (size_t)h * a.rows_total + r0;
                for (uint32_t r = tid; r < S; r += kBlock) {
                    const int v = sc[r];
                    if (a.tap_codes) a.tap_codes[tb + r] = v;
                    if (a.tap_scores) a.tap_scores[tb + r] = (float)v;
                    if (a.tap_probs) a.tap_probs[tb + r] = v0_p[v];
                }
            }
            // rows whose quantised weight is non-zero: 8 scores per LDS read, their 8 table look-ups in flight together
            for (uint32_t rb = tid * 8; rb < S; rb += kBlock * 8) {
                const i32x4 v = *(const i32x4 *)(sc + rb);               // (the score array is padded to 8 rows)
                uint8_t kp[8];
#pragma unroll
                for (int i = 0; i < 8; i++) kp[i] = v0_kp[rb + i < S ? ((uint32_t)v[i / 2] >> (16 * (i % 2))) & 0xFFFFu : 0u];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (rb + i < S && kp[i]) {
                        const uint32_t n = atomicAdd(&misc[0], 1u);
                        if (n < (uint32_t)kLiveCap) { live_row[n] = rb + i; live_kp[n] = kp[i]; }
                    }
                }
            }
            __syncthreads();
            n_live = misc[0];
        } else if (S > 0) {
            const SmCfg smc = sm_cfg(a, h);
            // (float)score . scale is exact (integers below 2^24 times a power of two)
            auto slot_x = [&](uint32_t r) { return sm_scaled((float)sc[r] * scale, smc); };
            float xmax = -INFINITY;
            for (uint32_t r = tid; r < S; r += kBlock) xmax = fmaxf(xmax, slot_x(r));
            xmax = block_max<float>(xmax, (float *)red, lane, wave);
            double part = 0.0;
            for (uint32_t r = tid; r < S; r += kBlock) part += (double)sm_exp(slot_x(r) - xmax, smc);
            const double total = smc.base == QMANN_SOFTMAX_EXP ? block_sum_double(part, red, lane, wave)
                                                               : block_serial_total_f32(S, lane, wave, red, [&](uint32_t r) { return sm_exp(slot_x(r) - xmax, smc); });
            const size_t tb = (size_t)h * a.rows_total + r0;
            // Q(p) != 0 needs p >= 2^-frac, and p -- e / total in double then float, or e / (float)total -- is within 2^-23
            // relative of the exact quotient: a slot with e < total . 2^-frac . (1 - 2^-20) cannot reach it, and only the others
            // (a handful of 10 000) take the double-precision division and the quantiser.  That division per SLOT was a fifth of
            // this kernel's vector instructions in mode 3 (4.78 lane-operations per key byte against 3.9 in the scan loop).
            // With taps every slot's p is wanted, and the shift-based forms divide by a power of two up to sqrt(2) away from
            // the total: threshold 0, every slot takes the exact path.
            const bool every = a.tap_codes || a.tap_scores || a.tap_probs || smc.shift;
            const float thr = every ? 0.0f : (float)(total * (double)qm_scale_down(1.0f, fa.frac) * (1.0 - 9.5367431640625e-07));
            // ... and for the e^x base without a scale layer that test does not need the exponential either: e^(x - max) >= thr
            // needs x - max >= log(thr) (up to the rounding of expf and logf, far inside the 1e-3 taken off here), x = score .
            // scale exactly, so a slot below the integer `code_lo` is out without an exp -- 2 instead of ~20 vector operations
            // for all but a handful of the 10 000 slots of a hop.  The slots that pass still take the test on e itself.
            int code_lo = INT_MIN;
            if (!every && smc.base == QMANN_SOFTMAX_EXP && !smc.en_scale)
                code_lo = (int)floorf((logf(thr) - 1e-3f + xmax) / scale) - 1;
            for (uint32_t r = tid; r < S; r += kBlock) {
                if ((int)sc[r] < code_lo) continue;
                const float e = sm_exp(slot_x(r) - xmax, smc);
                if (!(e >= thr)) continue;
                const float p = sm_quot(e, total, smc);
                if (a.tap_codes) a.tap_codes[tb + r] = sc[r];
                if (a.tap_scores) a.tap_scores[tb + r] = (float)sc[r] * scale;
                if (a.tap_probs) a.tap_probs[tb + r] = p;
                const int kp = qm_code(p, fa.iwl, fa.frac);
                if (kp) {
                    const uint32_t i = atomicAdd(&misc[0], 1u);
                    if (i < (uint32_t)kLiveCap) { live_row[i] = r; live_kp[i] = (uint8_t)kp; }
                }
            }
            __syncthreads();
            n_live = misc[0];
            if (n_live > (uint32_t)kLiveCap) {                  // keep the overflow path exact: park Q(p) in sc
                for (uint32_t r = tid; r < S; r += kBlock) {
                    const float p = sm_quot(sm_exp(slot_x(r) - xmax, smc), total, smc);
                    sc[r] = (score_t)qm_code(p, fa.iwl, fa.frac);
                }
                __syncthreads();
            }
        }
        auto kp_of_row = [&](uint32_t r) { return mode_is_v0(MODE) ? (int)v0_kp[sc[r]] : (int)sc[r]; };
        finish_hop<DP>(a, q, h, r0, S, n_live, live_row, live_kp, kp_of_row, ku, u_f, o_f, tid);
    }
    if (tid < D) a.u_out[(size_t)q * D + tid] = relu_if(u_f[tid], a.en_non_lin != 0);
}

// sign-magnitude bytes [rows][Dp] -> bit-planes [rows][Dp/64][nb].  A lane loads 16 columns (one 16-byte load; round 2 loaded
// one byte per lane and ran at 0.7 TB/s), a quad of lanes is one 64-column group.  Bit 7-i of the four bytes of a dword is
// gathered into a nibble by one multiply (m = (x >> (7-i)) & 0x01010101; m . 0x01020408 puts byte j's bit at 24 + j, no two
// partial products share a bit); four nibbles are a lane's 16 plane bits, two quad-permute moves make every lane of the quad
// hold the group's 64-bit word, and lane t keeps the words of planes t and t + 4: an item's nb words leave as one contiguous
// piece per quad.  Grid-stride (a launch may not exceed 2^32 threads, memories run to 10^8 rows).
__global__ void __launch_bounds__(kBlock)
k_pack_planes(const uint8_t *__restrict__ sm, uint64_t *__restrict__ planes, size_t rows, uint32_t Dp, uint32_t nb)
{
    const uint32_t t = threadIdx.x & 3u;
    const size_t n = rows * (Dp / 16);                   // 16-byte pieces, in memory order; piece p belongs to item p / 4
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t p0 = (size_t)blockIdx.x * kBlock + threadIdx.x; p0 - t < n; p0 += 2 * stride) {     // (a quad enters together: n is a multiple of 4)
        i32x4 x[2];
        bool ok[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const size_t p = p0 + r * stride;
            ok[r] = p < n;
            x[r] = ok[r] ? __builtin_nontemporal_load((const i32x4 *)(sm + p * 16)) : i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint64_t mine[2] = {0, 0};
            for (uint32_t i = 0; i < nb; i++) {
                uint32_t h = 0;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t m = ((uint32_t)x[r][d] >> (7u - i)) & 0x01010101u;
                    h |= ((m * 0x01020408u) >> 24) << (4 * d);
                }
                uint32_t v = h << (16u * (t & 1u));
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]: lanes 0,1 low half, lanes 2,3 high half
                const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]: the other half
                const uint64_t word = (t & 2u) ? ((uint64_t)v << 32) | o : ((uint64_t)o << 32) | v;
                if ((i & 3u) == t) mine[i >> 2] = word;
            }
            const size_t item = (p0 + r * stride) / 4;
            if (ok[r] && t < nb) planes[item * nb + t] = mine[0];
            if (ok[r] && t + 4 < nb) planes[item * nb + t + 4] = mine[1];
        }
    }
}

inline bool fmt8(qmann_fmt f) { return f.iwl + f.frac >= 1 && f.iwl + f.frac <= 7; }

int fill_args(HopArgs &a, const qmann_net *net, const void *keys, const int8_t *vals, size_t key_hop_stride,
              size_t val_hop_stride, const uint32_t *row_off, const float *u0, float *u_out, const qmann_taps *taps)
{
    if (!net || !keys || !vals || !row_off || !u0 || !u_out) return QMANN_EINVAL;
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP) return QMANN_EINVAL;
    if (net->dim_emb == 0 || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (net->dim_emb_pad != 64 && net->dim_emb_pad != 128 && net->dim_emb_pad != 256) return QMANN_EUNSUPPORTED;
    if (net->softmax_base > QMANN_SOFTMAX_EXP_PLAN) return QMANN_EINVAL;
    if (!fmt8(net->bin) && net->bin.iwl + net->bin.frac != 0) return QMANN_ERANGE;     // (0,0) = BINARY_MODE: u binarised
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        if (net->att[h].iwl + net->att[h].frac != 7 || net->att[h].iwl < 1) return QMANN_EUNSUPPORTED;
        if (net->en_lin_map && !net->lin_map[h]) return QMANN_EINVAL;
    }
    a = HopArgs{};
    // Mode 3 does word arithmetic on its operands before it compares bits: a byte per operand carries that only for the grid
    // combinations of qfmt.h::ham_hop_kind.  u entering hop h comes from emb_q (format w[0]) or from sv[h-1] (format act[h-1]);
    // the keys of hop h lie on w[h].  (Modes 10 / 11 compare the words' top bits as they are: any grid.)
    if (net->attention_mode == QMANN_ATT_APPX)
        for (uint32_t h = 0; h < net->n_hop; h++) {
            const qmann_fmt src = h == 0 ? net->w[0] : net->act[h - 1];
            const int kind = ham_hop_kind(QFmt{src.iwl, src.frac}, QFmt{net->w[h].iwl, net->w[h].frac}, QFmt{net->att[h].iwl, net->att[h].frac});
            if (kind == kHamNone) return QMANN_EUNSUPPORTED;
            a.ham_kinds |= (uint32_t)kind << (2u * h);
        }
    a.keys = (const int8_t *)keys; a.vals = vals; a.row_off = row_off;
    a.hop_stride = val_hop_stride; a.key_hop_stride = key_hop_stride;
    a.u0 = u0; a.u_out = u_out;
    if (taps) {
        a.tap_codes = taps->score_codes; a.tap_scores = taps->scores; a.tap_probs = taps->probs;
        a.tap_o = taps->o; a.tap_u = taps->u;
    }
    a.rows_total = val_hop_stride ? (uint32_t)(val_hop_stride / net->dim_emb_pad) : (uint32_t)qm_rows_hint;   // (tied hops: the caller's hint, or 0 = unknown)
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
    return QMANN_OK;
}

template <int LPRK, int DP, int MODE, int NB>
void launch(const HopArgs &a, uint32_t key_row_bytes, size_t lds, uint32_t lds_slots, uint32_t n_query, hipStream_t st)
{
    if (lds > kLdsDefaultLimit)
        QM_HIP(hipFuncSetAttribute((const void *)k_hops_ham<LPRK, DP, MODE, NB>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_hops_ham<LPRK, DP, MODE, NB><<<n_query, kBlock, lds, st>>>(a, key_row_bytes, lds_slots);
}

size_t ham_lds_bytes(uint32_t max_slots, uint32_t v0_bins)
{
    return (size_t)kOffScores + (((size_t)max_slots * 2 + 15) & ~(size_t)15) + (v0_bins ? v0_table_bytes(v0_bins) : 0);
}

}  // namespace

extern "C" {

int qmann_pack_bitplanes(const int8_t *sm_codes, uint64_t *planes, size_t rows, uint32_t dim_emb_pad,
                         uint32_t num_bit, void *stream)
{
    QmBatched qm_scope;
    if (!sm_codes || !planes) return QMANN_EINVAL;
    if (dim_emb_pad % 64 != 0 || num_bit < 1 || num_bit > 8) return QMANN_EINVAL;
    const size_t pieces = rows * (dim_emb_pad / 16);
    if (pieces == 0) return QMANN_OK;
    if (((uintptr_t)sm_codes & 15u) || ((uintptr_t)planes & 7u)) return QMANN_EINVAL;       // 16-byte loads, 8-byte stores
    const size_t blocks = (pieces + 2 * kBlock - 1) / (2 * kBlock);
    k_pack_planes<<<(unsigned)(blocks < 65536 ? blocks : 65536), kBlock, 0, (hipStream_t)stream>>>(
        (const uint8_t *)sm_codes, planes, rows, dim_emb_pad, num_bit);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

// APPX on sign-magnitude int8 keys (called from qmann_hops_i8)
int qmann_hops_appx_impl(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                         const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                         const qmann_taps *taps, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    HopArgs a;
    const int rc = fill_args(a, net, keys, vals, hop_stride, hop_stride, row_off, u0, u_out, taps);
    if (rc) return rc;
    a.max_slots = max_slots;
    const size_t lds = ham_lds_bytes(max_slots, 0);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    hipStream_t st = (hipStream_t)stream;
    const bool mq = a.ham_kinds != 0;                       // some hop's operands leave the attention grid (EN_MQ): the kernels
                                                            // that carry all three lane sums (ham_common.h)
#define QM_APPX(M)                                                                                                    \
    do {                                                                                                              \
        if (lean_supported(a, max_slots, 64)) launch_lean<M, 8>(a, max_slots, n_query, st);       /* hops_lean.h */   \
        else if (max_slots <= (uint32_t)kWave) {                                                  /* hops_small.h */  \
            if (net->dim_emb_pad == 64) k_hops_small<4, 4, M, 8><<<n_query, kWave, 0, st>>>(a, 64);                   \
            else if (net->dim_emb_pad == 128) k_hops_small<8, 8, M, 8><<<n_query, kWave, 0, st>>>(a, 128);            \
            else k_hops_small<16, 16, M, 8><<<n_query, kWave, 0, st>>>(a, 256);                                       \
        } else if (net->dim_emb_pad == 64) launch<4, 64, M, 8>(a, 64, lds, max_slots, n_query, st);                   \
        else if (net->dim_emb_pad == 128) launch<8, 128, M, 8>(a, 128, lds, max_slots, n_query, st);                  \
        else launch<16, 256, M, 8>(a, 256, lds, max_slots, n_query, st);                                              \
    } while (0)
    if (mq) QM_APPX(kModeAppxMq); else QM_APPX(kModeAppx);
#undef QM_APPX
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

// V0 / V1 straight from sign-magnitude int8 keys (called from qmann_hops_i8)
int qmann_hops_hambytes_impl(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                             const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                             const qmann_taps *taps, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    const uint32_t nb = net->num_bit, Dp = net->dim_emb_pad;
    if (nb != 1 && nb != 2 && nb != 4 && nb != 8) return QMANN_EUNSUPPORTED;
    HopArgs a;
    const int rc = fill_args(a, net, keys, vals, hop_stride, hop_stride, row_off, u0, u_out, taps);
    if (rc) return rc;
    a.max_slots = max_slots;
    const bool v1 = net->attention_mode == QMANN_ATT_HAMMING_V1;
    const size_t lds = ham_lds_bytes(max_slots, v1 ? 0u : nb * net->dim_emb + 1u);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    hipStream_t st = (hipStream_t)stream;
    if (lean_supported(a, max_slots, Dp)) {                 // hops_lean.h
        if (v1) { if (nb == 1) launch_lean<kModeV1Bytes, 1>(a, max_slots, n_query, st); else if (nb == 2) launch_lean<kModeV1Bytes, 2>(a, max_slots, n_query, st);
                  else if (nb == 4) launch_lean<kModeV1Bytes, 4>(a, max_slots, n_query, st); else launch_lean<kModeV1Bytes, 8>(a, max_slots, n_query, st); }
        else { if (nb == 1) launch_lean<kModeV0Bytes, 1>(a, max_slots, n_query, st); else if (nb == 2) launch_lean<kModeV0Bytes, 2>(a, max_slots, n_query, st);
               else if (nb == 4) launch_lean<kModeV0Bytes, 4>(a, max_slots, n_query, st); else launch_lean<kModeV0Bytes, 8>(a, max_slots, n_query, st); }
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
#define QM_HAMB(DP, NB)                                                                                   \
    do {                                                                                                  \
        constexpr int L = DP / 16;                                                                        \
        if (max_slots <= (uint32_t)kWave) {                 /* hops_small.h */                            \
            if (v1) k_hops_small<L, L, kModeV1Bytes, NB><<<n_query, kWave, 0, st>>>(a, DP);               \
            else k_hops_small<L, L, kModeV0Bytes, NB><<<n_query, kWave, 0, st>>>(a, DP);                  \
        } else if (v1) launch<L, DP, kModeV1Bytes, NB>(a, DP, lds, max_slots, n_query, st);               \
        else launch<L, DP, kModeV0Bytes, NB>(a, DP, lds, max_slots, n_query, st);                         \
    } while (0)
#define QM_HAMB_NB(DP)                                                                                    \
    do { if (nb == 1) QM_HAMB(DP, 1); else if (nb == 2) QM_HAMB(DP, 2); else if (nb == 4) QM_HAMB(DP, 4); else QM_HAMB(DP, 8); } while (0)
    if (Dp == 64) QM_HAMB_NB(64); else if (Dp == 128) QM_HAMB_NB(128); else QM_HAMB_NB(256);
#undef QM_HAMB_NB
#undef QM_HAMB
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_hops_packed(const qmann_net *net, const uint64_t *key_planes, size_t key_hop_stride, const int8_t *vals,
                      size_t val_hop_stride, const uint32_t *row_off, uint32_t max_slots, const float *u0,
                      float *u_out, const qmann_taps *taps, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    if (!net) return QMANN_EINVAL;
    if (net->attention_mode != QMANN_ATT_HAMMING_V0 && net->attention_mode != QMANN_ATT_HAMMING_V1)
        return QMANN_EUNSUPPORTED;
    const uint32_t nb = net->num_bit, Dp = net->dim_emb_pad;
    if (nb != 1 && nb != 2 && nb != 4 && nb != 8) return QMANN_EUNSUPPORTED;
    if (key_hop_stride != val_hop_stride / Dp * (Dp / 64) * nb * 8) return QMANN_EINVAL;
    if (taps && val_hop_stride == 0) return QMANN_EINVAL;   // taps are indexed [hop][row]: they need distinct hop planes
    HopArgs a;
    const int rc = fill_args(a, net, key_planes, vals, key_hop_stride, val_hop_stride, row_off, u0, u_out, taps);
    if (rc) return rc;
    a.max_slots = max_slots;
    const bool v1 = net->attention_mode == QMANN_ATT_HAMMING_V1;
    const size_t lds = ham_lds_bytes(max_slots, v1 ? 0u : nb * net->dim_emb + 1u);
    if (lds > 160 * 1024 - 1024) return QMANN_ERANGE;
    const uint32_t row_bytes = Dp / 64 * nb * 8;
    if (row_bytes < 16) return QMANN_EUNSUPPORTED;      // Dp = 64 with a single plane
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    hipStream_t st = (hipStream_t)stream;
#define QM_HAM(DP, NB)                                                                                   \
    do {                                                                                                 \
        constexpr int LPRK = (DP / 64) * NB * 8 / 16;                                                    \
        if (max_slots <= (uint32_t)kWave) {                 /* hops_small.h */                           \
            if (v1) k_hops_small<DP / 16, LPRK, kModeV1, NB><<<n_query, kWave, 0, st>>>(a, row_bytes);   \
            else k_hops_small<DP / 16, LPRK, kModeV0, NB><<<n_query, kWave, 0, st>>>(a, row_bytes);      \
        } else if (v1) launch<LPRK, DP, kModeV1, NB>(a, row_bytes, lds, max_slots, n_query, st);         \
        else launch<LPRK, DP, kModeV0, NB>(a, row_bytes, lds, max_slots, n_query, st);                   \
    } while (0)
    if (Dp == 64) { if (nb == 2) QM_HAM(64, 2); else if (nb == 4) QM_HAM(64, 4); else QM_HAM(64, 8); }
    else if (Dp == 128) { if (nb == 1) QM_HAM(128, 1); else if (nb == 2) QM_HAM(128, 2); else if (nb == 4) QM_HAM(128, 4); else QM_HAM(128, 8); }
    else { if (nb == 1) QM_HAM(256, 1); else if (nb == 2) QM_HAM(256, 2); else if (nb == 4) QM_HAM(256, 4); else QM_HAM(256, 8); }
#undef QM_HAM
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

}  // extern "C"
