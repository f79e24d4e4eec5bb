// hops_quad.h -- stories of at most 16 rows at the bAbI width (64-byte rows): FOUR queries per wavefront, a 16-lane DPP row each.
//
// The real bAbI stories are short (task 1: 2 .. 10 sentences, mean 6; the 20-task set: 91 % of the stories have <= 16) and the
// one-wavefront-per-query kernel (hops_lean.h) is bound by vector-instruction issue, not by memory: for a 6-row story it still
// pays a 16-row scan pass, 64-lane reductions, a softmax over 64 lanes and the fixed cost of every stage -- 0.07 of the HBM
// roofline on task 1 (round 4).  Here a wavefront carries four queries through the hops together:
//   * lane (g, l), g = lane >> 4 the query, owns COLUMNS 4l .. 4l+3 of that query's hop state, read-out and linear-map output;
//   * the key scan covers 4 rows x 4 lanes x 16 bytes per query and pass, so a pass serves four queries and a quad of stories
//     needs ceil(longest / 4) passes instead of four 16-row passes;
//   * after pass j the lanes with (lane & 3) == j keep their row's score: slot 4j + s sits in lane 4s + j of the row (a
//     transposed order -- the softmax, the weight codes and the survivor search do not care about the order of the slots);
//   * every reduction is a 4-step butterfly inside a DPP row and serves four queries (maximum, the double total);
//   * the survivors of Q(p) (at most 2^frac rows per query) are found per row group, their weight codes fetched with
//     ds_bpermute, their value rows -- one dword of four columns per lane -- with bounds-checked buffer loads while the linear
//     map runs; the products go through the packed 16-bit saturating multiply the scan uses (|v| . Q(p) >> frac, clamp at 127);
//   * the linear map keeps the scan's arithmetic (4 rows x 4 lanes per query and pass, 16 passes on the pre-split rows in LDS);
//     row 16s + 4(t & 3) + (t >> 2) in pass t lands the sums of rows 4l .. 4l+3 in lane l, which owns those columns.
// Same arithmetic as hops_lean.h / hops_small.h stage by stage (which stay the reference implementation inside the library and
// the path for taps); bit-identical to them (tests/test_gpu_quad.py).  Word length 7 in every format of the launch and the e^x
// softmax base (the stock 8-bit configurations): anything else keeps the lean kernel (quad_supported()).
// Included by hops_lean.h in front of its launchers.
#pragma once

namespace {

constexpr int kQuadWaves = 8;                       // wavefronts per workgroup: 32 queries in flight
constexpr int kQuadBlock = kQuadWaves * kWave;
constexpr int kQuadWps = 6;                         // wavefronts per SIMD the kernel is compiled for (80 registers): three workgroups per CU
constexpr int kQuadWpsLong = 4;                     // ... the form for stories of up to 64 rows (two key register sets, four scores per lane): 128 registers
constexpr uint32_t kQuadSlots = 16;                 // rows per story
constexpr uint32_t kQwBytes = 256;                  // a query's constant images: kLwE, kLwO, kLwS, kLwUb of hops_lean.h
constexpr uint32_t kOobOffset = 0x80000000u;        // a buffer offset no plane reaches (planes are bounded to 2 GiB here): reads zeros

struct QuadArgs {
    const uint32_t *list;                           // query indices (nullptr: queries 0 .. n_items - 1 as they are)
    const uint32_t *n_list;                         // device word with the list's length (nullptr: n_items)
    uint32_t n_items;
    const uint32_t *n_other;                        // side-by-side launch (hops_lean.h::corun_groups): the other list's length; nullptr otherwise
    uint32_t *publish;                              // pinned host words that receive n_list[0 .. 1] (both list lengths: rt.h::QmSide::last_counts), or nullptr
};

typedef short s16x2 __attribute__((ext_vector_type(2)));

// a story's survivor mask: 16 bits per slot set
template <int NC> struct QuadMask { typedef uint64_t type; };
template <> struct QuadMask<1> { typedef uint32_t type; };
__device__ __forceinline__ uint32_t quad_ctz(uint32_t m) { return (uint32_t)__builtin_ctz(m); }
__device__ __forceinline__ uint32_t quad_ctz(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }

// butterflies over the 16 lanes of a DPP row (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror): every lane
// ends with the row's result
#define QM_QUAD_ROW_STEPS(X) X(0xB1) X(0x4E) X(0x141) X(0x140)
__device__ __forceinline__ int quad_row_max_i32(int v)
{
#define QM_STEP(C) { const int t = __builtin_amdgcn_update_dpp(0, v, C, 0xF, 0xF, true); v = t > v ? t : v; }
    QM_QUAD_ROW_STEPS(QM_STEP)
#undef QM_STEP
    return v;
}
__device__ __forceinline__ float quad_row_max_f32(float v)
{
#define QM_STEP(C) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), C, 0xF, 0xF, true)));
    QM_QUAD_ROW_STEPS(QM_STEP)
#undef QM_STEP
    return v;
}
__device__ __forceinline__ double quad_row_sum_f64(double v)
{
#define QM_STEP(C) {                                                                                                   \
        const uint64_t b_ = __builtin_bit_cast(uint64_t, v);                                                           \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b_, C, 0xF, 0xF, true);           \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b_ >> 32), C, 0xF, 0xF, true);   \
        v += __builtin_bit_cast(double, (uint64_t)lo_ | ((uint64_t)hi_ << 32)); }
    QM_QUAD_ROW_STEPS(QM_STEP)
#undef QM_STEP
    return v;
}
#undef QM_QUAD_ROW_STEPS

__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sign_mask_i16(uint32_t a)                 // 0xFFFF in a half whose bit 15 is set
{
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) >> (short)15));
}

// qm_code / qm_code_or_sign / ham_ubyte (qfmt.h, ham_common.h) for word lengths up to 7, without their early returns (four
// columns per lane and hop: the branches would be divergent ones): clamp on the float side, then the truncating conversion
__device__ __forceinline__ int quad_code(float x, QFmt f)
{
    const float m = (float)((1 << (f.iwl + f.frac)) - 1);               // (wavefront-uniform)
    const float t = __builtin_ldexpf(x, (int)f.frac);                   // exact
    const int k = (int)__builtin_fminf(__builtin_fmaxf(t, -m), m);
    return t != t ? 0 : k;                                              // (a NaN converts to 0 in the general form)
}
__device__ __forceinline__ int quad_code_or_sign(float x, QFmt f)
{
    if (f.iwl + f.frac == 0) return (x >= 0.0f) ? 1 : -1;               // (wavefront-uniform: the format is a launch constant)
    return quad_code(x, f);
}
__device__ __forceinline__ uint32_t quad_ubyte(float ua, QFmt fm, bool real)
{
    const int kc = quad_code(ua, fm);
    const uint32_t mag = ua == -(float)(1u << fm.iwl) ? 0u : (uint32_t)(kc < 0 ? -kc : kc);
    return real ? (mag | (!(ua >= 0.0f) ? 0x80u : 0u)) : 0u;
}

// NC: chunks of 16 rows a story may have -- 1: the short stories (every row of a hop in 16 key registers, requested a hop ahead);
// 4: up to 64 rows (the |mem| = 50 cap of BASELINE's metric): the key chunks alternate between two register sets, the next one
// requested before the current one is scored; a lane keeps NC scores (slot 16k + 4 (lane & 3) + ((lane >> 2) & 3) in set k)
template <int MODE, int NB, int WPS, int NC>
__global__ void __launch_bounds__(kQuadBlock, WPS)
k_hops_quad(const HopArgs a, const LeanArgs la, const QuadArgs qa)
{
    constexpr uint32_t Dp = 64;
    constexpr uint32_t kSlots = kQuadSlots * NC;
    constexpr bool W7 = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    const uint32_t g = lane >> 4, l = lane & 15u, sub4 = (lane >> 2) & 3u, chunk = lane & 3u;
    const uint32_t D = a.D, H = a.n_hop;
    float *etab = (float *)smem;                                        // [H][256]
    uint8_t *lmap = smem + (la.exp_table ? H * 1024u : 0u);             // [H][64][3][64] (lean_stage_tables)
    uint8_t *lw = lmap + (la.lm_in_lds ? H * kLmHopBytes : 0u) + (wave * 4u + g) * kQwBytes;      // this lane's query: its constant images
    // (a 25-us copy on the stream otherwise: the launcher's hint costs one lane two stores)
    if (qa.publish && blockIdx.x == 0 && tid == 0) { qa.publish[0] = qa.n_list[0]; qa.publish[1] = qa.n_list[1]; }
    uint32_t n_groups = gridDim.x;
    if (qa.n_other) {                                                   // (side by side with the long stories' kernel)
        n_groups = corun_groups(*qa.n_list, *qa.n_other, gridDim.x, false);
        if (blockIdx.x >= n_groups) return;
    }
    lean_stage_tables(a, la, etab, lmap, tid, kQuadBlock);
    __syncthreads();

    const uint32_t n_items = qa.n_list ? *qa.n_list : qa.n_items;
    const uint32_t stride = n_groups * kQuadWaves * 4u;
    const uint32_t cap = a.max_slots < kSlots ? a.max_slots : kSlots;
    uint32_t i0 = (blockIdx.x * kQuadWaves + wave) * 4u;
    if (i0 >= n_items) return;
    // this lane's query of the quad that starts at item i: its index, first row and length (0 rows for a missing query)
    auto item_of = [&](uint32_t i, uint32_t &q_, uint32_t &r0_, uint32_t &S_) {
        const uint32_t it = i + g;
        q_ = 0xFFFFFFFFu; r0_ = 0; S_ = 0;
        if (it < n_items) {
            q_ = qa.list ? qa.list[it] : it;
            r0_ = a.row_off[q_];
            const uint32_t s = a.row_off[q_ + 1] - r0_;
            S_ = s < cap ? s : cap;
        }
    };
    uint32_t q, r0, S;
    item_of(i0, q, r0, S);
    i32x4 kq[NC > 1 ? 2 : 1][4];
    // keys of hop h for the four stories (r0_, S_ per lane), chunk c_ of 16 rows: row 16 c_ + 4j + sub4 of the lane's story, 16
    // bytes per lane.  One raw buffer resource spans the hop's whole key plane (wavefront-uniform, as a buffer load needs it); a
    // row the story does not have is asked at an offset beyond every plane and reads as zeros without touching memory.
    auto load_chunk = [&](i32x4 (&k_)[4], uint32_t h, uint32_t r0_, uint32_t S_, uint32_t c_) {
        const uint8_t *k0 = (const uint8_t *)a.keys + (size_t)h * a.key_hop_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)k0, 0, (int)0x7FFFFFFF, kRawBufferFlags);
        const uint32_t row = c_ * 16u + sub4, base = (r0_ + row) * Dp + chunk * 16u;
#pragma unroll
        for (int j = 0; j < 4; j++)
            k_[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row + (uint32_t)j * 4u < S_ ? base + (uint32_t)j * 4u * Dp : kOobOffset), 0, kBufferNt);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_keys_of = [&](uint32_t h, uint32_t r0_, uint32_t S_) { load_chunk(kq[0], h, r0_, S_, 0u); };
    load_keys_of(0, r0, S);
    const QFmt fb = a.bin;
    for (; i0 < n_items; i0 += stride) {
        uint32_t qn = 0xFFFFFFFFu, r0n = 0, Sn = 0;
        float uf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) uf[i] = (q != 0xFFFFFFFFu && 4u * l + i < D) ? a.u0[(size_t)q * D + 4u * l + i] : 0.0f;
        // the longest of the four stories bounds the passes (wavefront-uniform)
        uint32_t maxS;
        {
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)S, 0), s1 = (uint32_t)__builtin_amdgcn_readlane((int)S, 16);
            const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)S, 32), s3 = (uint32_t)__builtin_amdgcn_readlane((int)S, 48);
            const uint32_t m01 = s0 > s1 ? s0 : s1, m23 = s2 > s3 ? s2 : s3;
            maxS = m01 > m23 ? m01 : m23;
        }
        const uint32_t slot0 = 4u * chunk + sub4;                       // the slots whose scores this lane keeps: 16 k + slot0
        bool live[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) live[k] = 16u * k + slot0 < S;
        for (uint32_t h = 0; h < H; h++) {
            const QFmt fa = a.act[h], fm = a.att[h], fw = a.w[h];
            const int maxa = 127;
            if (h + 1 == H) item_of(i0 + stride, qn, r0n, Sn);         // the next quad (past the end: no rows, its key prefetch brings zeros for free)
            const bool relu = hop_relu(a, h);
            QM_MARK("operand codes + publish");
            // ---- operand codes of this lane's four columns, published into the query's images -----------------------
            int kb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) kb[i] = (4u * l + i < D) ? quad_code_or_sign(uf[i], fb) : 0;
            auto publish = [&](const int (&k)[4]) {                     // ScanConst images of hops_lean.h, word length 7
                uint32_t m[4], sg = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    m[i] = (uint32_t)(k[i] < 0 ? -k[i] : k[i]) << (8 - (int)fb.frac);
                    sg |= (k[i] < 0 ? 0x80u : 0u) << (8 * i);
                }
                *(uint32_t *)(lw + kLwE + l * 4u) = m[0] | (m[2] << 16);
                *(uint32_t *)(lw + kLwO + l * 4u) = m[1] | (m[3] << 16);
                *(uint32_t *)(lw + kLwS + l * 4u) = sg;
            };
            if (MODE == kModeFixed) {
                int ka[4];
#pragma unroll
                for (int i = 0; i < 4; i++) ka[i] = (relu && kb[i] < 0) ? ((fb.iwl + fb.frac == 0) ? 1 : 0) : kb[i];
                publish(ka);
            } else {
                uint32_t ub4 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) ub4 |= quad_ubyte(relu_if(uf[i], relu), fm, 4u * l + i < D) << (8 * i);
                *(uint32_t *)(lw + kLwUb + l * 4u) = ub4;
            }
            wave_sync();

            QM_MARK("scan");
            // ---- scores: pass j covers rows 4j .. 4j+3 of every story; lanes with chunk == j keep the pass's score ----
            ScanConst csc;
            uint32_t csh = 0;
            float unit = 1.0f;
            int code[NC];
            auto scan = [&](auto lane_sum, int lim, bool wrap) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    code[k] = 0;
                    if (16u * k < maxS) {                               // wavefront-uniform
                        // (NC > 1) the next chunk's keys into the other register set before this chunk is scored
                        if (NC > 1 && k + 1 < NC && 16u * (k + 1) < maxS) load_chunk(kq[(k + 1) & 1], h, r0, S, (uint32_t)(k + 1));
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (16u * k + (uint32_t)j * 4u < maxS) {    // wavefront-uniform
                                const int v = row_lanes_sum<4>(lane_sum(kq[k & 1][j]));
                                const int c = v > lim ? lim : (v < -lim ? -lim : ((wrap && v == -lim) ? 0 : v));
                                code[k] = chunk == (uint32_t)j ? c : code[k];
                            }
                        }
                    }
                }
            };
            if (MODE == kModeFixed) {
                unit = qm_scale_down(1.0f, fm.frac);
                csh = fetch_scan_const(csc, lw, chunk, 7u);
                scan([&](const i32x4 x) { return lane_row_sum7(x, csc); }, 127, false);
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
                scan([&](const i32x4 x) { return hambyte_lane_sum<MODE, NB>(x, c); }, 32767, false);
            }

            QM_MARK("softmax + weight codes");
            // ---- softmax over the slots of each story: one DPP row per story -------------------------------------------
            const SmCfg smc = sm_cfg(a, h);
            float e[NC];
            if (MODE == kModeFixed && la.exp_table) {
                int mxc = -32768;
#pragma unroll
                for (int k = 0; k < NC; k++) mxc = (live[k] && code[k] > mxc) ? code[k] : mxc;
                mxc = quad_row_max_i32(mxc);
#pragma unroll
                for (int k = 0; k < NC; k++) e[k] = live[k] ? etab[h * 256u + (uint32_t)(mxc - code[k])] : 0.0f;
            } else {
                float xs[NC], mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < NC; k++) { xs[k] = live[k] ? sm_scaled((float)code[k] * unit, smc) : -INFINITY; mx = fmaxf(mx, xs[k]); }
                mx = quad_row_max_f32(mx);
#pragma unroll
                for (int k = 0; k < NC; k++) e[k] = live[k] ? sm_exp(xs[k] - mx, smc) : 0.0f;
            }
            double tsum = 0.0;
#pragma unroll
            for (int k = 0; k < NC; k++) tsum += (double)e[k];
            const double total = quad_row_sum_f64(tsum);                // the CUDA kernel's double total (lib/layer_cuda.cu:2024-2042)
            // weight codes Q(p) of the lane's NC slots, a byte each (lean_weight_code's arithmetic: the quotient's side of the
            // truncation steps from e . 2^frac . rcp(total), the exact double division only when some lane sits within 2^-11 of a step)
            uint32_t kp4 = 0;
            {
                const float r2 = __builtin_ldexpf(__builtin_amdgcn_rcpf((float)total), (int)fa.frac);
                float x[NC];
                bool near = false;
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    x[k] = live[k] ? e[k] * r2 : 0.0f;
                    const float xr = __builtin_rintf(x[k]);
                    near = near || (xr >= 1.0f && __builtin_fabsf(x[k] - xr) <= 4.8828125e-04f);
                }
                if (__builtin_expect(__ballot(near) != 0, 0)) {
#pragma unroll
                    for (int k = 0; k < NC; k++) x[k] = __builtin_ldexpf(live[k] ? (float)((double)e[k] / total) : 0.0f, (int)fa.frac);
                }
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const int kc = (int)x[k];
                    kp4 |= (uint32_t)(kc > maxa ? maxa : kc) << (8 * k);
                }
            }

            QM_MARK("survivors: pick, fetch");
            // ---- survivors of Q(p): up to four per round, their value dwords requested at once -------------------------
            // this story's survivors: bit 16 k + (lane of the row) for slot set k
            typedef typename QuadMask<NC>::type mask_t;
            mask_t m16 = 0;
#pragma unroll
            for (int k = 0; k < NC; k++) m16 |= (mask_t)((uint32_t)(__ballot(((kp4 >> (8 * k)) & 0xFFu) != 0u) >> (16u * g)) & 0xFFFFu) << (16 * k);
            const uint8_t *v0 = (const uint8_t *)a.vals + (size_t)h * a.hop_stride;
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)v0, 0, (int)0x7FFFFFFF, kRawBufferFlags);
            uint32_t kk4, bb[4];                                        // kk4: the four weight codes, a byte each
            uint32_t n_pick = 0;                                        // how many of the four are in use by some story (wavefront-uniform)
            auto pick_fetch = [&]() {
                kk4 = 0;
                n_pick = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // (a peaked softmax leaves one survivor: the later picks, their loads and their products are skipped together)
                    if (i > 0 && __ballot(m16 != 0u) == 0) { bb[i] = 0u; continue; }
                    n_pick = (uint32_t)i + 1u;
                    const bool has = m16 != 0u;
                    const uint32_t bit = has ? quad_ctz(m16) : 0u, pos = bit & 15u, set = bit >> 4;
                    m16 &= m16 - (mask_t)1;                             // (0 stays 0)
                    const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 48u) | pos) << 2), (int)kp4);
                    kk4 |= (has ? (w >> (8u * set)) & 0xFFu : 0u) << (8 * i);
                    const uint32_t r = (set << 4) | ((pos & 3u) << 2) | (pos >> 2);       // the slot that lane `pos` of the row keeps in that set
                    bb[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rv, (int)(has ? (r0 + r) * Dp + l * 4u : kOobOffset), 0, 0);
                }
            };
            uint32_t acc02 = 0, acc13 = 0;                              // columns 4l, 4l+2 | 4l+1, 4l+3: packed 16-bit sums
            auto add = [&]() {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if ((uint32_t)i >= n_pick) continue;                // (wavefront-uniform)
                    // sign(v) . min(|v| . Q(p) >> frac, 127): the scan's packed multiply (Q(p) pre-shifted, signed 16-bit saturation)
                    const uint32_t kc = ((kk4 >> (8 * i)) & 0xFFu) << (8 - (int)fa.frac), kc2 = kc | (kc << 16);
                    const uint32_t pe = pk_mul_sat_i16(bb[i] & 0x007F007Fu, kc2), po = pk_mul_sat_i16((bb[i] >> 8) & 0x007F007Fu, kc2);
                    const uint32_t t02 = __builtin_amdgcn_perm(0u, pe, 0x0C030C01u), t13 = __builtin_amdgcn_perm(0u, po, 0x0C030C01u);
                    const uint32_t n02 = pk_sign_mask_i16(bb[i] << 8), n13 = pk_sign_mask_i16(bb[i]);
                    acc02 = pk_sub_i16(pk_add_i16(acc02, t02 ^ n02), n02);
                    acc13 = pk_sub_i16(pk_add_i16(acc13, t13 ^ n13), n13);
                }
            };
            pick_fetch();                                               // requested first: older than the key prefetch below
            // the next hop's keys, or the next quad's first keys
            auto prefetch_keys = [&]() {
                const bool more = h + 1 < H;
                load_keys_of(more ? h + 1 : 0u, more ? r0 : r0n, more ? S : Sn);
            };

            QM_MARK("linear map");
            // ---- linear map: 16 passes of 4 rows x 4 lanes per query on the pre-split rows in LDS ------------------------
            const bool reuse = MODE == kModeFixed && !relu;             // (word length 7 everywhere: the scan's constants serve)
            int keep[4] = {0, 0, 0, 0};
            if (a.en_lin_map) {
                if (!reuse) {
                    wave_sync();                                        // every lane is done with the previous images
                    publish(kb);
                    wave_sync();
                    csh = fetch_scan_const(csc, lw, chunk, 7u);
                }
                const uint8_t *hb = lmap + h * kLmHopBytes + (16u * sub4) * 192u + chunk * 16u;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint8_t *hr = hb + (4 * (t & 3) + (t >> 2)) * 192;
                    const int s = row_lanes_sum<4>(lane_sum_split<W7>(*(const i32x4 *)hr, *(const i32x4 *)(hr + 64), *(const i32x4 *)(hr + 128), csc, csh));
                    keep[t >> 2] = chunk == (uint32_t)(t & 3) ? s : keep[t >> 2];
                }
            }
            QM_MARK("key prefetch");
            // behind the linear map: in front of it the 16 key registers are live across the 16 passes (8 spilled registers at the
            // 80-register budget) and the forwards measured 1.5-4.5 % SLOWER (task 1 419 against 427 M q/s, joint 222 against 232;
            // interleaved A/B, ROUND_NOTES.md)
            prefetch_keys();
            QM_MARK("read-out");
            // ---- read-out ---------------------------------------------------------------------------------------------------
            add();
            while (__ballot(m16 != 0u)) { pick_fetch(); add(); }        // more than four survivors: formats with frac > 2, rarely
            int oc[4];
            oc[0] = (int)(short)(acc02 & 0xFFFFu); oc[2] = (int)acc02 >> 16;
            oc[1] = (int)(short)(acc13 & 0xFFFFu); oc[3] = (int)acc13 >> 16;
            QM_MARK("hop update");
            // ---- hop update u' = Qa(Qa(lu) + Qa(o)), this lane's four columns ------------------------------------------------
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int o = oc[i] > maxa ? maxa : (oc[i] < -maxa ? -maxa : oc[i]);
                int la_;
                if (a.en_lin_map) {
                    const int kw = keep[i] > 127 ? 127 : (keep[i] < -127 ? -127 : keep[i]);      // Qw of the row sum
                    const uint32_t mag = (uint32_t)(kw < 0 ? -kw : kw);
                    const uint32_t ma = fa.frac >= fw.frac ? mag << (fa.frac - fw.frac) : mag >> (fw.frac - fa.frac);
                    const int lam = ma > (uint32_t)maxa ? maxa : (int)ma;                   // Qa of that value
                    la_ = kw < 0 ? -lam : lam;
                } else {
                    la_ = (4u * l + i < D) ? quad_code(uf[i], fa) : 0;
                }
                int un = la_ + o;
                un = un > maxa ? maxa : (un < -maxa ? -maxa : un);
                uf[i] = qm_scale_down((float)un, fa.frac);
            }
            wave_sync();                                                // the next hop rewrites the images
            QM_MARK("end of hop");
        }
        if (q != 0xFFFFFFFFu) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4u * l + i < D) a.u_out[(size_t)q * D + 4u * l + i] = relu_if(uf[i], a.en_non_lin != 0);
        }
        q = qn; r0 = r0n; S = Sn;
    }
}

// Stories of a batch by length: the indices of those with at most kQuadSlots rows (after the cut to the caller's bound) go to
// `short_list`, the others to `long_list`.  A workgroup's 1 024 consecutive queries are appended together -- ONE atomic per list
// and workgroup (an atomic per wavefront serialised 8 192 of them on two words: 62 us for 262 144 queries) --, in order, so the
// lists stay nearly sorted.  counts[0] / counts[1]: the list lengths, zeroed by the launcher.
constexpr int kSplitBlock = 1024;
__global__ void __launch_bounds__(kSplitBlock)
k_split_by_length(const uint32_t *__restrict__ row_off, uint32_t n_query, uint32_t max_slots, uint32_t *__restrict__ counts,
                  uint32_t *__restrict__ short_list, uint32_t *__restrict__ long_list)
{
    __shared__ uint32_t wsum[2][kSplitBlock / 64], wbase[2][kSplitBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (size_t q0 = (size_t)blockIdx.x * kSplitBlock; q0 < n_query; q0 += (size_t)gridDim.x * kSplitBlock) {
        const size_t q = q0 + threadIdx.x;
        uint32_t s = 0;
        const bool ok = q < n_query;
        if (ok) { s = row_off[q + 1] - row_off[q]; s = s < max_slots ? s : max_slots; }
        const bool is_short = ok && s <= kQuadSlots, is_long = ok && s > kQuadSlots;
        const uint64_t ms = __ballot(is_short), ml = __ballot(is_long);
        if (lane == 0) { wsum[0][wave] = (uint32_t)__popcll(ms); wsum[1][wave] = (uint32_t)__popcll(ml); }
        __syncthreads();
        if (threadIdx.x < 2) {                                          // thread k: list k's offsets of the 16 wavefronts, one atomic
            uint32_t tot = 0;
            for (int w = 0; w < kSplitBlock / 64; w++) { wbase[threadIdx.x][w] = tot; tot += wsum[threadIdx.x][w]; }
            const uint32_t b = tot ? atomicAdd(&counts[threadIdx.x], tot) : 0u;
            for (int w = 0; w < kSplitBlock / 64; w++) wbase[threadIdx.x][w] += b;
        }
        __syncthreads();
        const uint64_t below = (1ull << lane) - 1ull;
        if (is_short) short_list[wbase[0][wave] + (uint32_t)__popcll(ms & below)] = (uint32_t)q;
        if (is_long) long_list[wbase[1][wave] + (uint32_t)__popcll(ml & below)] = (uint32_t)q;
        __syncthreads();                                                // (the next round rewrites wsum / wbase)
    }
}

// [0] short count, [1] long count, then the two lists (n_query words each), in the scratch words of stream `owner`; the work
// itself goes to `run_on` (the same stream, or the second one beside it -- rt.h::QmSplitReady)
inline uint32_t *split_lists(const uint32_t *row_off, uint32_t n_query, uint32_t max_slots, hipStream_t owner, hipStream_t run_on)
{
    uint32_t *ws = qm_scratch_u32(2u + 2u * (size_t)n_query, owner);
    if (!ws) return nullptr;
    QM_HIP(hipMemsetAsync(ws, 0, 2 * sizeof(uint32_t), run_on));
    const uint32_t blocks = (n_query + kSplitBlock - 1u) / kSplitBlock;
    k_split_by_length<<<blocks < 1024u ? blocks : 1024u, kSplitBlock, 0, run_on>>>(row_off, n_query, max_slots, ws, ws + 2, ws + 2 + n_query);
    return ws;
}

// what the quad kernel covers: the lean kernel's shapes (lean_supported) with word length 7 in every format of the launch, the
// e^x softmax base without the shift-based normaliser, and planes of known size below 2 GiB (its buffer offsets are 32-bit)
inline bool quad_supported(const HopArgs &a, int mode, uint32_t max_slots, uint32_t n_query)
{
    if (qm_tuning().no_quad || a.softmax_base != QMANN_SOFTMAX_EXP || a.softmax_shift) return false;
    for (uint32_t h = 0; h < a.n_hop; h++) {
        if (a.act[h].iwl + a.act[h].frac != 7) return false;
        if (mode == kModeFixed && a.att[h].iwl + a.att[h].frac != 7) return false;
        if (a.en_lin_map && a.w[h].iwl + a.w[h].frac != 7) return false;
    }
    if (a.bin.frac > 7u) return false;
    // The kernel addresses rows by 32-bit byte offsets from the start of a hop's plane (one buffer resource per hop): the plane --
    // every row the batch's offsets can name, cut stories' unused rows included -- must stay below 2 GiB, and its size must be
    // KNOWN: from hop_stride, or for tied hops (hop_stride = 0) from the caller's hint (rt.h::QmRowsHint, qmann_model gives it)
    return a.rows_total != 0u && (uint64_t)a.rows_total * 64u < 0x7FFF0000ull;
}

template <int MODE, int NB, int NC>
inline void launch_quad(HopArgs a, const QuadArgs &qa, uint32_t n_max, hipStream_t st)
{
    LeanArgs la{};
    la.rows_pad = 0;
    la.exp_table = (MODE == kModeFixed && !a.en_att_scale) ? 1u : 0u;
    la.lm_in_lds = a.en_lin_map ? 1u : 0u;
    const size_t lds = (la.exp_table ? a.n_hop * 1024u : 0u) + (la.lm_in_lds ? a.n_hop * kLmHopBytes : 0u) + (size_t)kQuadWaves * 4u * kQwBytes;
    const uint32_t need = (n_max + kQuadWaves * 4u - 1u) / (kQuadWaves * 4u);
    constexpr int wps = NC == 1 ? kQuadWps : kQuadWpsLong;
    auto kernel = k_hops_quad<MODE, NB, wps, NC>;
    if (lds > kLdsDefaultLimit) QM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t resident = qm_resident_groups(kQuadWaves, (unsigned)wps, lds);
    kernel<<<need < resident ? need : resident, kQuadBlock, lds, st>>>(a, la, qa);
}

}  // namespace
