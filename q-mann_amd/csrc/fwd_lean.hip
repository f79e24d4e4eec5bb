// fwd_lean.hip -- the whole test-phase forward of a bAbI-sized query in ONE kernel: word indices -> story and question
// embedding (gather-sum into LDS) -> every hop (hops_lean.h) -> answer layer -> prediction.  One wavefront per query,
// persistent workgroups.  Keys and values never touch HBM: a hop's two memories are built in the wavefront's LDS tiles
// right before they are used (MemN2N/MemN2N.c:2626-2697 does dense_mat_fwd x 2 per hop as well).
//
// Bit-identical to the staged pipeline (qmann_embed_story_idx -> qmann_embed_query_idx -> qmann_hops_i8 ->
// qmann_answer_f32), which remains the path for long stories, wide embeddings and taps; tests/test_gpu_fused.py
// compares the two.  Stage by stage the arithmetic is that of batch_io.hip (embedding, answer) and hops_lean.h (hops).
#include "fwd_lean.h"

#include "hops_lean.h"

namespace {

constexpr uint32_t kFwWords = 16;                       // word slots per row (qmann_embed_story_idx: max_words <= 16)

typedef short s16x2 __attribute__((ext_vector_type(2)));

// ---- per-wavefront LDS slice -------------------------------------------------------------------------------------
//   kt [rows_pad][64]  keys of the current hop      vt [rows_pad][64]  values of the current hop
//   wd [rows_pad][16]  u16 per story word slot: the word whose table row is to be added, 0xFFFF = nothing (unused slot,
//                      out-of-range word, or a repeat of a word that an earlier slot of the row carries with its count)
//   ct [rows_pad][16]  u8 count of that word in the row (1 unless a word repeats)
//   lw                 hops_lean.h's small arrays, then the duplicate-detection bitmaps [4 rows][8] u32
constexpr uint32_t kFwBm = kLwBytes;                    // offset inside lw
constexpr uint32_t kFwLwBytes = kLwBytes + 128;
constexpr uint32_t kFwRowBytes = 2u * 64u + kFwWords * 2u + kFwWords;      // tiles + wd + ct, per story row

// integer count c >= 0 as a code of the weight format (saturating): Qw(c)
__device__ __forceinline__ int fw_count_code(uint32_t c, uint32_t frac, int maxw)
{
    const uint64_t k = (uint64_t)c << frac;
    return k > (uint64_t)maxw ? maxw : (int)k;
}

// One row group (4 rows, 16 lanes each): which word slots add what.  Same rules as k_embed_story_idx
// (MemN2N/sample.c:466-475, 544-548): word entries COUNT occurrences, the time entry (the row's last valid slot)
// SETS its bag-of-words entry to 1, out-of-range words are ignored.  The usual case -- no word twice in a row -- is
// found with one LDS atomic per lane on a 256-bit hash bitmap per row; otherwise the slots are compared pairwise.
// Returns word | count << 16, or 0xFFFF (count 0) for a slot that adds nothing.
__device__ __forceinline__ uint32_t fw_pack_row(uint32_t w, uint32_t V, bool time_last, uint32_t nw, uint32_t lane, uint8_t *lw)
{
    const uint32_t sub = lane & 15u, grp = lane >> 4;
    const uint32_t m16 = (uint32_t)(__ballot(w != 0xFFFFu) >> (16 * grp)) & 0xFFFFu;
    const uint32_t n_valid = 32u - (uint32_t)__clz(m16);
    const bool valid = w != 0xFFFFu && w < V;
    const bool is_time = time_last && valid && (sub + 1 == n_valid);
    uint32_t *bm = (uint32_t *)(lw + kFwBm);
    if (lane < 32u) bm[lane] = 0u;
    wave_sync();
    bool clash = false;
    if (valid) {
        const uint32_t bit = 1u << (w & 31u);
        clash = (atomicOr(&bm[grp * 8u + ((w >> 5) & 7u)], bit) & bit) != 0u;
    }
    uint32_t cnt = 1;
    bool dup = false;
    if (__any(clash)) {                                  // some row of this group may hold a word twice: the exact, slow way
        const uint32_t me = w | (valid ? 1u << 16 : 0u) | (is_time ? 1u << 17 : 0u);
        bool timed = false;
        cnt = 0;
        for (uint32_t j = 0; j < nw; j++) {
            const uint32_t o = (uint32_t)__shfl((int)me, (int)j, 16);
            const bool same = (((o ^ me) & 0xFFFFu) == 0u) && ((o >> 16) & 1u);
            const bool o_time = (o >> 17) & 1u;
            cnt += (same && !o_time) ? 1u : 0u;
            timed |= same && o_time;
            dup |= same && j < sub;
        }
        if (timed) cnt = 1;
    }
    return (valid && !dup) ? (w | (cnt << 16)) : 0xFFFFu;
}

// sign-magnitude memory bytes from packed int16 sums in the weight format (batch_io.hip::to_bytes): Qw of the sum, the
// magnitude moved to the target grid toward zero and clamped, the sign bit from the VALUE (minus zero stays minus zero)
__device__ __forceinline__ uint32_t fw_to_bytes(s16x2 x, int maxw, QFmt fw, QFmt dst)
{
    const short mw = (short)maxw;
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, s16x2{(short)-mw, (short)-mw}), s16x2{mw, mw});
    u16x2 mag = __builtin_bit_cast(u16x2, __builtin_elementwise_max(x, (s16x2)(-x)));
    mag = dst.frac >= fw.frac ? (u16x2)(mag << (unsigned short)(dst.frac - fw.frac)) : (u16x2)(mag >> (unsigned short)(fw.frac - dst.frac));
    const unsigned short md = (unsigned short)((1u << (dst.iwl + dst.frac)) - 1u);
    mag = __builtin_elementwise_min(mag, u16x2{md, md});
    const u16x2 sgn = __builtin_bit_cast(u16x2, (s16x2)(x >> 8)) & (unsigned short)0x0080;
    return __builtin_bit_cast(uint32_t, (u16x2)(mag | sgn));
}

// Gather-sum of hop h's two tables (keys: ta -> kt on the attention grid, values: tc -> vt on the activation grid) over
// the rows of a story.  16 lanes per row (a lane owns the 4 columns of dword `sub`), 4 rows per step.  The 8 (16) word
// slots of a row are read with one (two) 16-byte loads and all their table reads are in flight together.
// TAB16: the tables sit in LDS expanded to int16 -- per word 16 x {even-columns pair, odd-columns pair}, one ds_read_b64
// per slot, and row V is all zero so that an empty slot needs no predicate; else int8 [V][64] two's complement in global
// memory (L2).  `multi`: bit s is set when a row of step s repeats a word (then, and for purely fractional weight
// formats, every term is Qw(Qw(count) . kw) instead of a plain add).
template <bool TAB16>
__device__ __forceinline__ void fw_embed_hop(uint8_t *kt, uint8_t *vt, const uint16_t *wd, const uint8_t *ct, uint32_t S, uint32_t nw,
                                             uint32_t multi, const void *ta, const void *tc, uint32_t V, QFmt fw, QFmt f_att, QFmt f_act,
                                             uint32_t D, uint32_t lane)
{
    const uint32_t sub = lane & 15u, grp = lane >> 4;
    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
    const bool one_ok = (1 << fw.frac) <= maxw;          // 1.0 is a value of the format: Qw(1 . kw) = kw
    uint32_t colmask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) colmask |= (4 * sub + (uint32_t)k < D ? 0xFFu : 0u) << (8 * k);
    for (uint32_t s0 = 0, step = 0; s0 < S; s0 += 4, step++) {
        const uint32_t row = s0 + grp;
        const bool row_ok = row < S;
        const bool slow = !one_ok || ((multi >> step) & 1u);              // wavefront-uniform
        s16x2 ae = {0, 0}, ao = {0, 0}, ce = {0, 0}, co = {0, 0};
        for (uint32_t e0 = 0; e0 < nw; e0 += 8) {                         // (one pass for sentences of up to 8 slots)
            i32x4 wv = {-1, -1, -1, -1};
            if (row_ok) wv = *(const i32x4 *)(wd + row * kFwWords + e0);
            s16x2 xa_e[8], xa_o[8], xc_e[8], xc_o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t we = ((uint32_t)wv[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
                if (TAB16) {
                    const uint32_t off = (we < V ? we : V) * 128u + sub * 8u;
                    const uint2 t1 = *(const uint2 *)((const uint8_t *)ta + off), t2 = *(const uint2 *)((const uint8_t *)tc + off);
                    xa_e[e] = __builtin_bit_cast(s16x2, t1.x); xa_o[e] = __builtin_bit_cast(s16x2, t1.y);
                    xc_e[e] = __builtin_bit_cast(s16x2, t2.x); xc_o[e] = __builtin_bit_cast(s16x2, t2.y);
                } else {
                    const bool ok = we < V;
                    const size_t off = (size_t)(ok ? we : 0u) * 16u + sub;
                    const uint32_t t1 = ok ? ((const uint32_t *)ta)[off] : 0u, t2 = ok ? ((const uint32_t *)tc)[off] : 0u;
                    xa_e[e] = (__builtin_bit_cast(s16x2, t1) << 8) >> 8; xa_o[e] = __builtin_bit_cast(s16x2, t1) >> 8;
                    xc_e[e] = (__builtin_bit_cast(s16x2, t2) << 8) >> 8; xc_o[e] = __builtin_bit_cast(s16x2, t2) >> 8;
                }
            }
            if (slow) {                                                   // rare: per-slot counts
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t c1 = row_ok ? ct[row * kFwWords + e0 + e] : 1u;
                    const int cc = fw_count_code(c1 ? c1 : 1u, fw.frac, maxw);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        xa_e[e][k] = (short)qm_mul_code(cc, xa_e[e][k], fw.frac, maxw); xa_o[e][k] = (short)qm_mul_code(cc, xa_o[e][k], fw.frac, maxw);
                        xc_e[e][k] = (short)qm_mul_code(cc, xc_e[e][k], fw.frac, maxw); xc_o[e][k] = (short)qm_mul_code(cc, xc_o[e][k], fw.frac, maxw);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e++) { ae += xa_e[e]; ao += xa_o[e]; ce += xc_e[e]; co += xc_o[e]; }
        }
        if (row_ok) {
            *(uint32_t *)(kt + row * 64u + sub * 4u) = (fw_to_bytes(ae, maxw, fw, f_att) | (fw_to_bytes(ao, maxw, fw, f_att) << 8)) & colmask;
            *(uint32_t *)(vt + row * 64u + sub * 4u) = (fw_to_bytes(ce, maxw, fw, f_act) | (fw_to_bytes(co, maxw, fw, f_act) << 8)) & colmask;
        }
    }
}

// question embedding u0[c] = Qw0(sum_k Qw0(Qw0(W[c][k]) . Qw0(count_k))) -- k_embed_query_idx; lane c owns column c
template <bool TAB16>
__device__ __forceinline__ float fw_embed_query(const uint16_t *qwords, uint32_t nqw, const void *tab, uint32_t V, QFmt fw, uint32_t D, uint32_t lane)
{
    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
    const bool one_ok = (1 << fw.frac) <= maxw;
    uint32_t w = 0xFFFFu;
    if (lane < nqw) w = qwords[lane];
    const bool valid = w != 0xFFFFu && w < V;
    int acc = 0;
    for (uint32_t e = 0; e < nqw; e++) {
        const uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)e);      // (e is wavefront-uniform)
        if (we == 0xFFFFu || we >= V) continue;
        // count of this word over the slots, first occurrence only
        const uint64_t same = __ballot(valid && w == we);
        if ((uint32_t)__builtin_ctzll(same) != e) continue;
        const uint32_t ce = (uint32_t)__builtin_popcountll(same);
        int kw;
        if (TAB16) {
            // column c = 4 q + i: even columns pair (i = 0, 2) then odd columns pair (i = 1, 3)
            const uint32_t q4 = lane >> 2, i = lane & 3u;
            kw = ((const int16_t *)tab)[(size_t)we * 64u + q4 * 4u + (i & 1u) * 2u + (i >> 1)];
        } else {
            kw = ((const int8_t *)tab)[(size_t)we * 64u + lane];
        }
        acc += (ce == 1u && one_ok) ? kw : qm_mul_code(fw_count_code(ce, fw.frac, maxw), kw, fw.frac, maxw);
    }
    const int v = acc > maxw ? maxw : (acc < -maxw ? -maxw : acc);
    return lane < D ? qm_scale_down((float)v, fw.frac) : 0.0f;
}

// ---- the kernel --------------------------------------------------------------------------------------------------
// TAB16: embedding tables in LDS (int16); VPT > 0: answer layer in the kernel (W^T in LDS, VPT logits per lane).
// Every matrix format has word length 7 (W7 of hops_lean.h); other word lengths take the staged pipeline.
constexpr int kFwMaxWaves = 16;
template <int MODE, int NB, bool TAB16, int VPT>
__global__ void __launch_bounds__(kFwMaxWaves * kWave)
k_fwd_lean(const HopArgs a, const LeanArgs la, const qmann::FwdArgs f)
{
    constexpr bool ANS = VPT > 0;
    constexpr uint32_t VP = 64u * (VPT > 0 ? VPT : 1);
    typedef float fvec __attribute__((ext_vector_type(VPT > 0 ? VPT : 1)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, nthreads = blockDim.x, nwaves = nthreads / kWave;
    const uint32_t D = a.D, H = a.n_hop, V = f.V;
    // workgroup tables
    float *etab = (float *)smem;
    uint8_t *lmap = smem + (la.exp_table ? H * 1024u : 0u);
    uint8_t *tabs = lmap + (la.lm_in_lds ? H * 4096u : 0u);               // [1 + 2H][V + 1][64] int16 (TAB16; row V is zero)
    const uint32_t tab_bytes = TAB16 ? (V + 1u) * 128u : 0u;
    float *wt = (float *)(tabs + (1u + 2u * H) * tab_bytes);              // [D][VP] W^T (ANS)
    uint8_t *wbase = (uint8_t *)wt + (ANS ? D * VP * 4u : 0u);
    const uint32_t tile = la.rows_pad * 64u;
    const uint32_t wslice = la.rows_pad * kFwRowBytes + kFwLwBytes;
    uint8_t *kt = wbase + wave * wslice, *vt = kt + tile;
    uint16_t *wd = (uint16_t *)(vt + tile);
    uint8_t *ct = (uint8_t *)(wd + la.rows_pad * kFwWords);
    uint8_t *lw = ct + la.rows_pad * kFwWords;

    lean_stage_tables(a, la, etab, lmap, tid, nthreads);
    if (TAB16) {
        // int8 [V][64] two's complement -> int16 [V + 1][16]{even pair, odd pair}
        for (uint32_t t = 0; t < 1u + 2u * H; t++) {
            const uint32_t *src = (const uint32_t *)(t == 0 ? f.t_q : (t <= H ? f.t_a[t - 1] : f.t_c[t - 1 - H]));
            uint2 *dst = (uint2 *)(tabs + t * tab_bytes);
            for (uint32_t i = tid; i < (V + 1u) * 16u; i += nthreads) {
                const uint32_t x = i < V * 16u ? src[i] : 0u;
                const s16x2 ev = (__builtin_bit_cast(s16x2, x) << 8) >> 8, od = __builtin_bit_cast(s16x2, x) >> 8;
                dst[i] = uint2{__builtin_bit_cast(uint32_t, ev), __builtin_bit_cast(uint32_t, od)};
            }
        }
    }
    if (ANS)
        for (uint32_t i = tid; i < D * VP; i += nthreads) {
            const uint32_t c = i / VP, v = i % VP;
            wt[i] = v < V ? f.w_ans[(size_t)v * D + c] : 0.0f;
        }
    __syncthreads();

    auto table = [&](uint32_t t) -> const void * {                       // 0: question, 1 + h: keys of hop h, 1 + H + h: values
        if (TAB16) return tabs + t * tab_bytes;
        return t == 0 ? (const void *)f.t_q : (t <= H ? (const void *)f.t_a[t - 1] : (const void *)f.t_c[t - 1 - H]);
    };
    const SmCfg smo{a.softmax_base, false, false, 1.0f};                  // sf_out is never shift-based (MemN2N.c:910)
    const uint32_t v0 = lane * (VPT > 0 ? VPT : 1);
    float cost_acc = 0.0f;
    uint32_t match_acc = 0;
    const uint32_t nw = f.max_words, sub = lane & 15u, grp = lane >> 4;

    for (size_t q = (size_t)blockIdx.x * nwaves + wave; q < f.n_query; q += (size_t)gridDim.x * nwaves) {
        const uint32_t r0 = a.row_off[q];
        const uint32_t S_in = a.row_off[q + 1] - r0;
        const uint32_t S = S_in < la.rows_pad ? S_in : la.rows_pad;          // (rows_pad <= 64; a longer story is cut like everywhere else)
        // ---- the story's word slots -> what each adds (once per query) -------------------------------------------
        uint32_t multi = 0;
        for (uint32_t s0 = 0, step = 0; s0 < S; s0 += 4, step++) {
            const uint32_t row = s0 + grp;
            uint32_t w = 0xFFFFu;
            if (row < S && sub < nw) w = f.story_words[(size_t)(r0 + row) * nw + sub];
            const uint32_t p = fw_pack_row(w, V, f.time_last != 0, nw, lane, lw);
            if (row < S) { wd[row * kFwWords + sub] = (uint16_t)p; ct[row * kFwWords + sub] = (uint8_t)(p >> 16); }
            if (__any((p >> 16) > 1u)) multi |= 1u << step;
        }
        float u = fw_embed_query<TAB16>(f.question_words + q * f.max_q_words, f.max_q_words, table(0), V, a.w[0], D, lane);
        wave_sync();

        for (uint32_t h = 0; h < H; h++) {
            // hop h's memories: E = Qw(sum of table rows), re-read on the attention / activation grid (batch_io.hip)
            fw_embed_hop<TAB16>(kt, vt, wd, ct, S, nw, multi, table(1 + h), table(1 + H + h), V, f.emb_w[h], f.emb_att[h], f.emb_act[h], D, lane);
            wave_sync();
            lean_hop<MODE, NB, true>(a, la, h, S, lane, vt, lw, lmap, etab, u,
                                     [&](int j) { return *(const i32x4 *)(kt + j * 1024 + lane * 16); }, [&]() {});
        }
        const float uo = relu_if(u, a.en_non_lin != 0);
        if (lane < D) a.u_out[q * D + lane] = uo;
        if (!ANS) continue;

        // ---- answer layer: k_answer_small's arithmetic (serial float sum over the embedding axis per logit) ---------
        fvec acc = 0.0f;
        for (uint32_t c = 0; c < D; c++) {
            const float uc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uo), (int)c));
            const fvec w = *(const fvec *)(wt + c * VP + v0);
            const fvec t = w * uc;
            acc += t;
        }
        float sum[VPT > 0 ? VPT : 1], e[VPT > 0 ? VPT : 1], p[VPT > 0 ? VPT : 1];
        bool live[VPT > 0 ? VPT : 1];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < VPT; k++) { sum[k] = acc[k]; live[k] = v0 + k < V; mx = (live[k] && sum[k] > mx) ? sum[k] : mx; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
        double total = 0.0;
#pragma unroll
        for (int k = 0; k < VPT; k++) { e[k] = live[k] ? sm_exp(sum[k] - mx, smo) : 0.0f; total += (double)e[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        float bv = -INFINITY;
        uint32_t bi = 0;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            p[k] = (a.softmax_base == QMANN_SOFTMAX_EXP) ? (float)((double)e[k] / total) : e[k] / (float)total;
            if (live[k] && !(bv > p[k])) { bv = p[k]; bi = v0 + k; }           // later index wins a tie
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float tv = __shfl_xor(bv, o);
            const uint32_t ti = __shfl_xor(bi, o);
            if (tv > bv || (tv == bv && ti > bi)) { bv = tv; bi = ti; }       // ties go to the highest index
        }
        if (lane == 0) f.pred[q] = bi;
        if (f.answer) {
            const uint32_t y = f.answer[q];
            if (y < V) {
                float py = 0.0f;
#pragma unroll
                for (int k = 0; k < VPT; k++) py = (y % (VPT > 0 ? VPT : 1) == (uint32_t)k) ? __shfl(p[k], (int)(y / (VPT > 0 ? VPT : 1))) : py;
                cost_acc += -py;
                match_acc += (y == bi) ? 1u : 0u;
            }
        }
    }
    if (ANS && f.answer && lane == 0) {
        if (f.cost) atomicAdd(f.cost, cost_acc);
        if (f.match && match_acc) atomicAdd(f.match, match_acc);
    }
}

struct Plan {
    bool tab16;
    int vpt;        // 0: answer layer stays a separate launch
    int nw;
    size_t lds;
};

// LDS the kernel needs for a choice of (tables in LDS, answer in kernel, wavefronts per workgroup)
size_t plan_lds(const HopArgs &a, const LeanArgs &la, uint32_t V, bool tab16, int vpt, int nw)
{
    const size_t tables = (la.exp_table ? a.n_hop * 1024u : 0u) + (la.lm_in_lds ? a.n_hop * 4096u : 0u) +
                          (tab16 ? (size_t)(1u + 2u * a.n_hop) * (V + 1u) * 128u : 0u) + (vpt ? (size_t)a.D * 64u * vpt * 4u : 0u);
    return tables + (size_t)nw * (la.rows_pad * kFwRowBytes + kFwLwBytes);
}

template <int MODE, int NB, bool TAB16, int VPT>
void launch_k(const HopArgs &a, const LeanArgs &la, const qmann::FwdArgs &f, const Plan &p, hipStream_t st)
{
    const uint32_t need = (f.n_query + p.nw - 1) / p.nw;
    const uint32_t per_cu_lds = (uint32_t)(160u * 1024u / (p.lds + 256u)), per_cu_w = 32u / (uint32_t)p.nw;
    const uint32_t per_cu = per_cu_lds < per_cu_w ? per_cu_lds : per_cu_w;
    const uint32_t grid = need < 256u * (per_cu ? per_cu : 1u) ? need : 256u * (per_cu ? per_cu : 1u);
    const void *fn = (const void *)k_fwd_lean<MODE, NB, TAB16, VPT>;
    if (p.lds > kLdsDefaultLimit) QM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    k_fwd_lean<MODE, NB, TAB16, VPT><<<grid, p.nw * kWave, p.lds, st>>>(a, la, f);
}

template <int MODE, int NB>
void launch_mode(const HopArgs &a, const LeanArgs &la, const qmann::FwdArgs &f, const Plan &p, hipStream_t st)
{
#define QM_FWD_V(T16)                                                                                            \
    do {                                                                                                         \
        if (p.vpt == 0) launch_k<MODE, NB, T16, 0>(a, la, f, p, st);                                             \
        else if (p.vpt == 1) launch_k<MODE, NB, T16, 1>(a, la, f, p, st);                                        \
        else launch_k<MODE, NB, T16, 4>(a, la, f, p, st);                                                        \
    } while (0)
    if (p.tab16) QM_FWD_V(true); else QM_FWD_V(false);
#undef QM_FWD_V
}

inline bool fmt8(qmann_fmt f) { return f.iwl + f.frac >= 1 && f.iwl + f.frac <= 7; }

}  // namespace

namespace qmann {

int fwd_lean(const qmann_net *net, const qmann_net *emb_net, FwdArgs f, uint32_t max_slots, float *u_out, int *answer_done, void *stream)
{
    *answer_done = 0;
    if (getenv("QMANN_NO_FUSED")) return QMANN_EUNSUPPORTED;
    if (net->dim_emb_pad != 64 || max_slots > 64 || f.max_words == 0 || f.max_words > kFwWords || f.max_q_words == 0 || f.max_q_words > 64)
        return QMANN_EUNSUPPORTED;
    const uint32_t mode = net->attention_mode;
    if (mode != QMANN_ATT_FIXED && mode != QMANN_ATT_APPX && mode != QMANN_ATT_HAMMING_V0 && mode != QMANN_ATT_HAMMING_V1) return QMANN_EUNSUPPORTED;
    const uint32_t nb = net->num_bit;
    if ((mode == QMANN_ATT_HAMMING_V0 || mode == QMANN_ATT_HAMMING_V1) && nb != 1 && nb != 2 && nb != 4 && nb != 8) return QMANN_EUNSUPPORTED;
    if (net->softmax_base > QMANN_SOFTMAX_EXP_PLAN) return QMANN_EINVAL;
    if (!fmt8(net->bin) && net->bin.iwl + net->bin.frac != 0) return QMANN_ERANGE;
    HopArgs a{};
    a.row_off = f.row_off; a.u_out = u_out;
    a.n_hop = net->n_hop; a.D = net->dim_emb; a.Dp = 64;
    a.softmax_base = net->softmax_base; a.en_lin_map = net->en_lin_map;
    a.softmax_shift = net->softmax_shift_based; a.en_att_scale = net->en_att_scale; a.en_non_lin = net->en_non_linearity;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        if (!fmt8(emb_net->act[h]) || !fmt8(emb_net->w[h]) || !fmt8(emb_net->att[h])) return QMANN_ERANGE;
        if (net->en_lin_map && !net->lin_map[h]) return QMANN_EINVAL;
        // the kernel is built for word length 7 (hops_lean.h, W7); shorter words take the staged pipeline
        if ((mode == QMANN_ATT_FIXED && net->att[h].iwl + net->att[h].frac != 7) || (net->en_lin_map && net->w[h].iwl + net->w[h].frac != 7))
            return QMANN_EUNSUPPORTED;
        if (mode != QMANN_ATT_FIXED) {                      // the byte forms of the Hamming family (batch_hops_ham.hip::fill_args)
            if (net->att[h].iwl + net->att[h].frac != 7 || net->att[h].iwl < 1) return QMANN_EUNSUPPORTED;
            const qmann_fmt src = h == 0 ? net->w[0] : net->act[h - 1];
            if (!(src.iwl <= net->att[h].iwl && src.frac <= net->att[h].frac)) return QMANN_EUNSUPPORTED;
        }
        a.att_scale[h] = net->att_scale[h];
        a.lin_map[h] = net->lin_map[h];
        a.act[h] = QFmt{net->act[h].iwl, net->act[h].frac};
        a.w[h] = QFmt{net->w[h].iwl, net->w[h].frac};
        a.att[h] = QFmt{net->att[h].iwl, net->att[h].frac};
        f.emb_w[h] = QFmt{emb_net->w[h].iwl, emb_net->w[h].frac};
        f.emb_att[h] = QFmt{emb_net->att[h].iwl, emb_net->att[h].frac};
        f.emb_act[h] = QFmt{emb_net->act[h].iwl, emb_net->act[h].frac};
    }
    a.bin = QFmt{net->bin.iwl, net->bin.frac};
    if (f.n_query == 0) return QMANN_OK;
    LeanArgs la{};
    la.rows_pad = ((max_slots ? max_slots : 1u) + 15u) & ~15u;
    la.exp_table = (mode == QMANN_ATT_FIXED && a.softmax_base == QMANN_SOFTMAX_EXP && !a.softmax_shift && !a.en_att_scale) ? 1u : 0u;
    la.lm_in_lds = a.en_lin_map ? 1u : 0u;
    f.V = net->dim_input;

    // what goes into LDS: prefer many wavefronts per CU, then the tables, then the answer layer
    const int vpt_want = f.w_ans && f.pred ? (f.V <= 64 ? 1 : (f.V <= 256 ? 4 : 0)) : 0;
    Plan best{false, 0, 0, 0};
    int best_score = -1;
    for (int tab16 = 1; tab16 >= 0; tab16--)
        for (int ans = 1; ans >= 0; ans--)
            for (int nw : {16, 8, 4}) {
                const int vpt = ans ? vpt_want : 0;
                if (ans && !vpt) continue;
                const size_t lds = plan_lds(a, la, f.V, tab16 != 0, vpt, nw);
                if (lds > 158u * 1024u) continue;
                const int per_cu = (int)(160u * 1024u / (lds + 256u));
                const int waves = nw * (per_cu < 32 / nw ? per_cu : 32 / nw);
                const int score = (waves >= 16 ? 16 : waves) * 4 + tab16 * 2 + ans;     // 16 wavefronts per CU are as many as the registers allow
                if (score > best_score) { best_score = score; best = Plan{tab16 != 0, vpt, nw, lds}; }
            }
    if (best_score < 0) return QMANN_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define QM_FWD_NB(M)                                                                                             \
    do { if (nb == 1) launch_mode<M, 1>(a, la, f, best, st); else if (nb == 2) launch_mode<M, 2>(a, la, f, best, st); \
         else if (nb == 4) launch_mode<M, 4>(a, la, f, best, st); else launch_mode<M, 8>(a, la, f, best, st); } while (0)
    if (mode == QMANN_ATT_FIXED) launch_mode<kModeFixed, 8>(a, la, f, best, st);
    else if (mode == QMANN_ATT_APPX) launch_mode<kModeAppx, 8>(a, la, f, best, st);
    else if (mode == QMANN_ATT_HAMMING_V0) QM_FWD_NB(kModeV0Bytes);
    else QM_FWD_NB(kModeV1Bytes);
#undef QM_FWD_NB
    QM_LAUNCH_CHECK();
    *answer_done = best.vpt != 0;
    return QMANN_OK;
}

}  // namespace qmann
