// batch_io.hip -- the stages either side of the hop loop, batched over queries:
// story / question embedding into int8 memories, and the answer layer
// (projection, softmax, arg-max, test-phase bookkeeping).
#include "hops_common.h"

#include <limits.h>
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------
// Answer layer.  One workgroup per query.  Thread v owns logit v and sums over
// the embedding axis serially in float (separate multiply and add), which is the
// exact operation order of the reference's serial loop (lib/layer_cuda.cu:70-80),
// so logits are bit-identical to it.  Softmax: e^(x-max), double normaliser,
// float quotient (:2006-2042) or the CPU form 2^(x-max) with float arithmetic
// (lib/layer.c:1225-1243).  Arg-max ties go to the highest index (:1918-1939).
// ---------------------------------------------------------------------------
// PRECOMPUTED: logits come from a buffer (written by the MFMA projection below) instead of being
// computed here.
template <bool PRECOMPUTED>
__global__ void __launch_bounds__(kBlock)
k_answer(const float *__restrict__ w_ans, const float *__restrict__ u, const uint32_t *__restrict__ answer,
         uint32_t *__restrict__ pred, float *__restrict__ probs, float *cost, uint32_t *match, uint32_t D,
         uint32_t V, uint32_t softmax_base, uint32_t n_query)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *us = (float *)smem;            // [D]
    float *lg = us + ((D + 3) & ~3u);     // [V]
    __shared__ double red_d[kWaves];
    __shared__ float red_f[kWaves];
    __shared__ uint32_t red_i[kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    float cost_acc = 0.0f;              // thread 0 only; added once per workgroup (see k_answer_small)
    uint32_t match_acc = 0;
    for (uint32_t q = blockIdx.x; q < n_query; q += gridDim.x) {
    __syncthreads();                    // the previous query's LDS rows are free again
    if (!PRECOMPUTED) {
        for (uint32_t c = tid; c < D; c += kBlock) us[c] = u[(size_t)q * D + c];
        __syncthreads();
    }

    float mx = -INFINITY;
    for (uint32_t v = tid; v < V; v += kBlock) {
        float sum = 0.0f;
        if (PRECOMPUTED) {
            sum = w_ans[(size_t)q * V + v];          // w_ans aliases the logits buffer [n_query][V]
        } else {
            const float *wr = w_ans + (size_t)v * D;
#pragma unroll 8
            for (uint32_t c = 0; c < D; c++) {       // unrolled: loads of a row run ahead of the serial adds
                const float t = wr[c] * us[c];
                sum += t;
            }
        }
        lg[v] = sum;
        mx = sum > mx ? sum : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(mx, o);
        mx = t > mx ? t : mx;
    }
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    for (int i = 1; i < kWaves; i++) mx = red_f[i] > mx ? red_f[i] : mx;

    double part = 0.0;
    for (uint32_t v = tid; v < V; v += kBlock) {
        const float e = sm_exp(lg[v] - mx, SmCfg{softmax_base, false, false, 1.0f});    // sf_out is never shift-based
        lg[v] = e;
        part += (double)e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) red_d[wave] = part;
    __syncthreads();
    const double total = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);

    float bv = -INFINITY;
    uint32_t bi = 0;
    for (uint32_t v = tid; v < V; v += kBlock) {
        const float p = (softmax_base == QMANN_SOFTMAX_EXP) ? (float)((double)lg[v] / total)
                                                            : lg[v] / (float)total;
        lg[v] = p;
        if (probs) probs[(size_t)q * V + v] = p;
        if (!(bv > p)) { bv = p; bi = v; }           // later index wins a tie
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float tv = __shfl_xor(bv, o);
        const uint32_t ti = __shfl_xor(bi, o);
        if (tv > bv || (tv == bv && ti > bi)) { bv = tv; bi = ti; }
    }
    __syncthreads();
    if (lane == 0) { red_f[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < kWaves; i++)
            if (red_f[i] > bv || (red_f[i] == bv && red_i[i] > bi)) { bv = red_f[i]; bi = red_i[i]; }
        pred[q] = bi;
        if (answer) {
            const uint32_t y = answer[q];
            if (y < V) {
                cost_acc += -lg[y];
                match_acc += (y == bi) ? 1u : 0u;
            }
        }
    }
    }
    if (tid == 0 && answer) {
        if (cost) atomicAdd(cost, cost_acc);
        if (match && match_acc) atomicAdd(match, match_acc);
    }
}

// Answer layer for small dictionaries (V <= 256: bAbI single-task and joint sizes): SIXTEEN LANES per query, four
// queries per wavefront; lane s of a group owns the VPT adjacent logits VPT.s .. VPT.s + VPT - 1 (16 . VPT >= V).  The
// arithmetic is k_answer's (serial float sum over the embedding axis per logit, softmax with a double normaliser,
// arg-max with ties to the highest index) without block barriers.  W is staged transposed in LDS once per (persistent)
// workgroup; per embedding column a lane reads its VPT weights and its query's component from LDS and the products /
// sums go through the packed fp32 pipeline.  The kernel is bound by vector-instruction issue: with one query per
// wavefront (round 1) the reductions and the per-logit divisions cost ~450 instructions per query whatever V was; a
// 16-lane group is one DPP row, so every reduction is four DPP steps serving four queries.
// 16 wavefronts per workgroup share one copy of W^T (up to 61 KB): 2 workgroups fill a CU's 32 wavefront slots
constexpr int kAnsBlock = 1024, kAnsWaves = kAnsBlock / kWave;

// butterfly over the 16 lanes of a DPP row: after the four steps every lane holds the row's result
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror)
#define QM_ROW_STEPS(X) X(0xB1) X(0x4E) X(0x141) X(0x140)
template <int CTRL> __device__ __forceinline__ float row_peer_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ uint32_t row_peer_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ double row_peer_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = row_peer_u32<CTRL>((uint32_t)b), hi = row_peer_u32<CTRL>((uint32_t)(b >> 32));
    return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

template <int LPQ, int VPT, int QB>
__global__ void __launch_bounds__(kAnsBlock, 8)             // 64 registers: two workgroups per CU
k_answer_small(const float *__restrict__ w_ans, const float *__restrict__ u, const uint32_t *__restrict__ answer,
               uint32_t *__restrict__ pred, float *__restrict__ probs, float *cost, uint32_t *match, uint32_t D,
               uint32_t V, uint32_t softmax_base, uint32_t n_query)
{
    static_assert(VPT % 2 == 0, "logits are handled in pairs");
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(LPQ == 16 || LPQ == 32 || LPQ == 64, "a query's lanes are whole DPP rows");
    constexpr uint32_t VP = LPQ * VPT;                  // logits padded to a whole group (zero columns)
    constexpr uint32_t QPW = (kWave / LPQ) * QB;        // queries per wavefront: QB per lane group, sharing each weight read
    typedef float fq __attribute__((ext_vector_type(QB)));
    float *wt = (float *)smem;                          // [D][VP]: W transposed
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));      // uniform: query bookkeeping in SGPRs (no spill at 64 registers)
    float *us = wt + (size_t)D * VP + wave_u * QPW * D;   // [kAnsWaves][D][QPW]: this wavefront's queries, interleaved
    const uint32_t lane = threadIdx.x & (kWave - 1), grp = lane / LPQ, sub = lane % LPQ;
    const size_t stride = (size_t)gridDim.x * kAnsWaves * QPW;
    for (uint32_t i = threadIdx.x; i < D * VP; i += kAnsBlock) {
        const uint32_t c = i / VP, v = i % VP;
        wt[i] = v < V ? w_ans[(size_t)v * D + c] : 0.0f;
    }
    __syncthreads();
    const SmCfg smc{softmax_base, false, false, 1.0f};  // sf_out is never shift-based (MemN2N.c:910)
    const uint32_t v0 = sub * VPT;                      // first logit of this lane
    bool live[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) live[k] = v0 + k < V;
    // cost / match are summed per wavefront and added once: one device-scope atomic per query on a
    // single word would serialise the whole batch (~12 ns each)
    float cost_acc = 0.0f;
    uint32_t match_acc = 0;
    for (size_t qb = ((size_t)blockIdx.x * kAnsWaves + wave_u) * QPW; qb < n_query; qb += stride) {
        const uint32_t nq = n_query - qb < QPW ? (uint32_t)(n_query - qb) : QPW;
        for (uint32_t i = lane; i < nq * D; i += kWave) us[(i % D) * QPW + i / D] = u[qb * D + i];   // consecutive queries: one contiguous block
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f2 acc[QB][VPT / 2];
#pragma unroll
        for (int j = 0; j < QB; j++)
#pragma unroll
            for (int k = 0; k < VPT / 2; k++) acc[j][k] = f2{0.0f, 0.0f};
        const float *uq = us + grp * QB;
        const float *wl = wt + v0;
#pragma unroll 2
        for (uint32_t c = 0; c < D; c++) {               // serial over the embedding axis, as the reference sums
            const fq uc = *(const fq *)(uq + c * QPW);
#pragma unroll
            for (int k = 0; k < VPT / 2; k++) {
                const f2 w = *(const f2 *)(wl + c * VP + 2 * k);
#pragma unroll
                for (int j = 0; j < QB; j++) {
                    const f2 t = w * uc[j];
                    acc[j][k] += t;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // reads of us[] done before the next queries overwrite it
#pragma unroll
        for (int j = 0; j < QB; j++) {
            const bool q_ok = grp * QB + j < nq;         // (a missing query's lanes compute on stale LDS and store nothing)
            const size_t q = qb + grp * QB + j;
            float sum[VPT];
#pragma unroll
            for (int k = 0; k < VPT; k++) sum[k] = acc[j][k / 2][k % 2];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < VPT; k++) mx = (live[k] && sum[k] > mx) ? sum[k] : mx;
#define QM_STEP(C) { const float t = row_peer_f32<C>(mx); mx = t > mx ? t : mx; }
            QM_ROW_STEPS(QM_STEP)
#undef QM_STEP
#pragma unroll
            for (int o = 16; o < LPQ; o <<= 1) { const float t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
            float e[VPT];
            double total = 0.0;
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                e[k] = live[k] ? sm_exp(sum[k] - mx, smc) : 0.0f;
                total += (double)e[k];
            }
#define QM_STEP(C) total += row_peer_f64<C>(total);
            QM_ROW_STEPS(QM_STEP)
#undef QM_STEP
#pragma unroll
            for (int o = 16; o < LPQ; o <<= 1) total += __shfl_xor(total, o);
            float bv = -INFINITY;
            uint32_t bi = 0;
            float p[VPT];
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                p[k] = (softmax_base == QMANN_SOFTMAX_EXP) ? (float)((double)e[k] / total) : e[k] / (float)total;
                if (live[k]) {
                    if (probs && q_ok) probs[q * V + v0 + k] = p[k];
                    if (!(bv > p[k])) { bv = p[k]; bi = v0 + k; }               // later index wins a tie
                }
            }
#define QM_STEP(C) { const float tv = row_peer_f32<C>(bv); const uint32_t ti = row_peer_u32<C>(bi); \
                     if (tv > bv || (tv == bv && ti > bi)) { bv = tv; bi = ti; } }   /* ties go to the highest index */
            QM_ROW_STEPS(QM_STEP)
#undef QM_STEP
#pragma unroll
            for (int o = 16; o < LPQ; o <<= 1) {
                const float tv = __shfl_xor(bv, o);
                const uint32_t ti = __shfl_xor(bi, o);
                if (tv > bv || (tv == bv && ti > bi)) { bv = tv; bi = ti; }
            }
            if (sub == 0 && q_ok) pred[q] = bi;
            if (answer) {
                const uint32_t y = q_ok ? answer[q] : 0xFFFFFFFFu;
                const uint32_t ys = y < V ? y : 0u;
                float psel = p[0];
#pragma unroll
                for (int k = 1; k < VPT; k++) psel = (ys % VPT == (uint32_t)k) ? p[k] : psel;
                const float py = __shfl(psel, (int)(grp * LPQ + ys / VPT));
                if (sub == 0 && y < V) {
                    cost_acc += -py;
                    match_acc += (y == bi) ? 1u : 0u;
                }
            }
        }
    }
    if (answer) {
#pragma unroll
        for (int o = LPQ; o < kWave; o <<= 1) { cost_acc += __shfl_xor(cost_acc, o); match_acc += __shfl_xor(match_acc, o); }
        if (lane == 0) {
            if (cost) atomicAdd(cost, cost_acc);
            if (match && match_acc) atomicAdd(match, match_acc);
        }
    }
}
#undef QM_ROW_STEPS

// ---------------------------------------------------------------------------
// The FLOAT answer layer on the bf16 matrix cores (bAbI widths: 64-byte rows, dictionaries up to 256), within north_star's 1e-5
// on the float softmax instead of bit-equal to the reference's serial sum.  The kernels above reproduce
// lib/layer_cuda.cu:70-80's order of additions exactly and pay for it: 545 vector instructions per query at the joint
// dictionary (profiles/r05_units_j20_k_answer_small.txt), a fifth of that forward.  Here:
//   * u, the last hop's output, lies on an 8-bit grid (|code| <= 127): every component is exactly a bf16;
//   * W (arbitrary float32) is split once per workgroup into three bf16 images W1 + W2 + W3 = W exactly (each takes the top 8
//     significant bits of what the one before left), 144-byte rows in LDS (16 B of padding: fragment reads without bank conflicts);
//   * logits = u . W3 + u . W2 + u . W1 on v_mfma_f32_16x16x32_bf16, rows = 16 answers, columns = 16 queries: every product is
//     exact (15 significant bits), only the float accumulation rounds -- the result sits closer to the exact sum than the serial
//     float loop does; against that loop: a few units in the last place of the logit (measured: probabilities within 5e-6
//     relative at the tests' and the bench's magnitudes, tools/answer_fused_error.py);
//   * a lane ends with its query's logits of answers 16t + 4(lane >> 4) + r in registers: maximum, 2^((l - max) log2 e) through
//     v_exp_f32, float total, arg-max on the LOGITS (equal logits <=> equal probabilities; ties to the highest index as
//     lib/layer_cuda.cu:1918-1939) and the label's probability never leave them; two cross-lane steps join the four lane groups.
// 36 vector instructions per query instead of 545 at V = 238.  QMANN_ANSWER_EXACT (or the drop-in queue, which promises the
// serial loop's results) keeps the serial-order kernels.
// ---------------------------------------------------------------------------
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kAmWaves = 16, kAmBlock = kAmWaves * kWave;
constexpr uint32_t kAmPitch = 144;                  // bytes per row of a W image: 64 bf16 + 16 B (rows 16 B apart modulo 256)

template <int T>                                     // tiles of 16 answers: V <= 16 T
__global__ void __launch_bounds__(kAmBlock, 4)
k_answer_mfma(const float *__restrict__ w_ans, const float *__restrict__ u, const uint32_t *__restrict__ answer,
              uint32_t *__restrict__ pred, float *__restrict__ probs, float *cost, uint32_t *match, uint32_t D, uint32_t V,
              uint32_t softmax_base, uint32_t n_query)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t kImg = T * 16u * kAmPitch;        // one bf16 image of W
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    for (uint32_t i = tid; i < T * 16u * 64u; i += kAmBlock) {
        const uint32_t v = i >> 6, c = i & 63u;
        const float w = (v < V && c < D) ? w_ans[(size_t)v * D + c] : 0.0f;
        const uint32_t b1 = __builtin_bit_cast(uint32_t, w) & 0xFFFF0000u;
        const float r1 = w - __builtin_bit_cast(float, b1);                 // exact
        const uint32_t b2 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, b2);                // exact; at most 8 significant bits are left
        const uint32_t b3 = __builtin_bit_cast(uint32_t, r2) & 0xFFFF0000u;
        uint8_t *dst = smem + v * kAmPitch + c * 2u;
        *(uint16_t *)dst = (uint16_t)(b1 >> 16);
        *(uint16_t *)(dst + kImg) = (uint16_t)(b2 >> 16);
        *(uint16_t *)(dst + 2u * kImg) = (uint16_t)(b3 >> 16);
    }
    __syncthreads();
    const uint32_t qi = lane & 15u, grp = lane >> 4;
    const bool vec4 = (D & 3u) == 0u;
    const uint8_t *arow = smem + qi * kAmPitch + grp * 16u;          // this lane's piece of an answer row: k = 8 grp .. + 7 of a K step
    float cost_acc = 0.0f;
    uint32_t match_acc = 0;
    const uint32_t n_task = (n_query + 15u) / 16u;
    for (uint32_t task = blockIdx.x * kAmWaves + wave; task < n_task; task += gridDim.x * kAmWaves) {
        asm volatile("" ::: "memory");                                   // (the W fragments are read per task: hoisted out of this loop they would take 24 T registers)
        const uint32_t q = task * 16u + qi;
        const bool q_ok = q < n_query;
        // B fragments: u[q][32 ks + 8 grp + j], j = 0 .. 7, as bf16 (exact: u lies on an 8-bit grid)
        bf16x8_t bf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const uint32_t k0 = 32u * ks + 8u * grp;
            uint32_t w[8];
            if (vec4) {
                const float *src = u + (size_t)q * D + k0;
                // (plain ifs: `cond ? *(const f32x4_t *)p : zero` compiles to ONE dword load splatted over the vector with hipcc 7.2)
                float4 lo = {0.0f, 0.0f, 0.0f, 0.0f}, hi = {0.0f, 0.0f, 0.0f, 0.0f};
                if (q_ok && k0 + 3u < D) lo = *(const float4 *)src;
                if (q_ok && k0 + 7u < D) hi = *(const float4 *)(src + 4);
#pragma unroll
                for (int j = 0; j < 4; j++) { w[j] = __builtin_bit_cast(uint32_t, (&lo.x)[j]); w[4 + j] = __builtin_bit_cast(uint32_t, (&hi.x)[j]); }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = (q_ok && k0 + j < D) ? __builtin_bit_cast(uint32_t, u[(size_t)q * D + k0 + j]) : 0u;
            }
            i32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; j++) pk[j] = (int)__builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);    // the two high halves
            bf[ks] = __builtin_bit_cast(bf16x8_t, pk);
        }
        f32x4_t acc[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            acc[t] = f32x4_t{0, 0, 0, 0};
#pragma unroll
            for (int s_ = 2; s_ >= 0; s_--)                              // the smallest terms first
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const bf16x8_t af = *(const bf16x8_t *)(arow + s_ * kImg + t * 16u * kAmPitch + ks * 64u);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[ks], acc[t], 0, 0, 0);
                }
        }
        // ---- softmax statistics of this lane's T x 4 logits (answer 16 t + 4 grp + r), then across the four lane groups ----
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < T; t++) {
            if ((uint32_t)t * 16u + 16u > V) {                           // (wavefront-uniform) the dictionary ends inside this tile
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t][r] = ((uint32_t)t * 16u + grp * 4u + r < V) ? acc[t][r] : -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) mx = fmaxf(mx, acc[t][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // arg-max on the logits: the highest position of this lane that holds the maximum (positions ascend with the answer index)
        int bpos = -1;
#pragma unroll
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) bpos = acc[t][r] == mx ? 4 * t + r : bpos;
            __builtin_amdgcn_sched_barrier(0);                           // (a tile's compares next to its selects: all 4 T lane masks at once do not fit the scalar file)
        }
        uint32_t bi = bpos >= 0 ? (uint32_t)(bpos >> 2) * 16u + grp * 4u + (uint32_t)(bpos & 3) : 0u;
        { const uint32_t o = (uint32_t)__shfl_xor((int)bi, 16); const int ob = __shfl_xor(bpos, 16); if (ob >= 0 && (bpos < 0 || o > bi)) { bi = o; bpos = ob; } }
        { const uint32_t o = (uint32_t)__shfl_xor((int)bi, 32); const int ob = __shfl_xor(bpos, 32); if (ob >= 0 && (bpos < 0 || o > bi)) { bi = o; bpos = ob; } }
        // the label's logit, if this lane holds it (a binary tree of selects over the lane's positions)
        const uint32_t y = (answer && q_ok) ? answer[q] : 0xFFFFFFFFu;
        float ly = -INFINITY;
        if (answer) {                                                    // (wavefront-uniform)
            const bool mine = y < V && ((y >> 2) & 3u) == grp;
            const uint32_t ty = y >> 4, ry = y & 3u;
            f32x4_t sel = acc[0];
#pragma unroll
            for (int t = 1; t < T; t++) {
                const bool pick = ty == (uint32_t)t;
#pragma unroll
                for (int r = 0; r < 4; r++) sel[r] = pick ? acc[t][r] : sel[r];
                __builtin_amdgcn_sched_barrier(0);
            }
            float l1 = (ry & 1u) ? sel[1] : sel[0], l2 = (ry & 1u) ? sel[3] : sel[2];
            ly = mine ? ((ry & 2u) ? l2 : l1) : -INFINITY;
            ly = fmaxf(ly, __shfl_xor(ly, 16));
            ly = fmaxf(ly, __shfl_xor(ly, 32));
        }
        // e = base^(l - max); the 2^x base needs no scaling (lib/layer.c:1225), e^x goes through 2^(x log2 e)
        const float sc = softmax_base == QMANN_SOFTMAX_EXP ? 1.44269504088896341f : 1.0f;
        float tot = 0.0f;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[t][r] = __builtin_amdgcn_exp2f((acc[t][r] - mx) * sc);
                tot += acc[t][r];
            }
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        const float inv = __builtin_amdgcn_rcpf(tot);
        if (probs && q_ok) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t v = (uint32_t)t * 16u + grp * 4u + r;
                    if (v < V) probs[(size_t)q * V + v] = acc[t][r] * inv;
                }
        }
        if (grp == 0 && q_ok) {
            pred[q] = bi;
            if (y < V) {
                cost_acc += -(__builtin_amdgcn_exp2f((ly - mx) * sc) * inv);
                match_acc += (y == bi) ? 1u : 0u;
            }
        }
    }
    if (answer) {
        // cost / match: summed per workgroup, one atomic each
        __shared__ float red_c[kAmWaves];
        __shared__ uint32_t red_m[kAmWaves];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { cost_acc += __shfl_xor(cost_acc, o); match_acc += __shfl_xor(match_acc, o); }   // lanes 0 .. 15 hold the figures
        if (lane == 0) { red_c[wave] = cost_acc; red_m[wave] = match_acc; }
        __syncthreads();
        if (tid == 0) {
            float c = 0.0f; uint32_t m = 0;
            for (int i = 0; i < kAmWaves; i++) { c += red_c[i]; m += red_m[i]; }
            if (cost) atomicAdd(cost, c);
            if (match && m) atomicAdd(match, m);
        }
    }
}

// ---------------------------------------------------------------------------
// Answer projection on the matrix cores, for an answer matrix that lives on an int8 grid:
// logits[q][v] = sum_c U[q][c] . W[v][c] with both operands small integers (codes), so the
// reference's float serial sum (lib/layer_cuda.cu:70-80) is an exact integer times
// 2^-(frac_u + frac_w) (|sum| <= 256.127.127 < 2^24) and an int32 MFMA accumulation reproduces it
// bit for bit.  One wavefront owns a 16 x 16 tile (16 queries x 16 answers) and issues
// v_mfma_i32_16x16x64_i8 over the embedding axis; both fragments are plain 16-byte row segments
// (lane l: row l & 15, bytes 16.(l >> 4) .. +15 of the 64-deep K step), so they are loaded straight
// from global memory -- the query side is converted from the hop kernel's float-on-grid vector in
// registers once and stays resident while the wavefront walks the answer tiles.
// C/D: lane l holds column l & 15, rows 4.(l >> 4) + r.
// ---------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kTilesPerBlockY = 16;      // answer tiles (of 16) one wavefront walks with its query fragment resident

template <int KSTEPS>
__global__ void __launch_bounds__(kBlock)
k_logits_mfma_i8(const float *__restrict__ u, const int8_t *__restrict__ w, float *__restrict__ logits,
                 uint32_t n_query, uint32_t D, uint32_t V, QFmt fu, float scale)
{
    constexpr uint32_t Dp = KSTEPS * 64;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t m0 = (blockIdx.x * kWaves + wave) * 16;       // this wavefront's 16 queries
    if (m0 >= n_query) return;                                   // whole wavefront
    const uint32_t row = lane & 15, kq = lane >> 4;
    const uint32_t qrow = m0 + row;
    // A fragments (query codes) for every K step, converted once
    i32x4 a[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
        const uint32_t kb = ks * 64 + 16 * kq;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t c = kb + 4 * d + i;
                const int code = (qrow < n_query && c < D) ? qm_code(u[(size_t)qrow * D + c], fu.iwl, fu.frac) : 0;
                pk |= ((uint32_t)code & 0xFFu) << (8 * i);
            }
            a[ks][d] = (int)pk;
        }
    }
    const uint32_t t0 = blockIdx.y * kTilesPerBlockY;
    for (uint32_t t = t0; t < t0 + kTilesPerBlockY && t * 16 < V; t++) {
        const uint32_t n0 = t * 16;
        const uint32_t vrow = n0 + row;
        i32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            i32x4 b = {0, 0, 0, 0};
            if (vrow < V) b = *(const i32x4 *)(w + (size_t)vrow * Dp + ks * 64 + 16 * kq);
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks], b, acc, 0, 0, 0);
        }
        const uint32_t col = n0 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t qr = m0 + 4 * (lane >> 4) + r;
            if (qr < n_query && col < V) logits[(size_t)qr * V + col] = (float)acc[r] * scale;
        }
    }
}

// ---------------------------------------------------------------------------
// The int8 answer layer in one pass, without the [n_query][V] logits round trip: projection on the matrix cores, softmax
// statistics and arg-max in the accumulator registers.  What makes a one-pass form exact enough:
//   * logits are integers k times 2^-(frac_u + frac_w), so the arg-max of the rounded probabilities, ties to the highest
//     index (lib/layer_cuda.cu:1918-1939), is the highest index whose k equals the maximum: two different k cannot round
//     to the same probability (their ratio is at least e^(2^-14));
//   * the normaliser sum_v e^(l_v - max) is accumulated against the RUNNING maximum and rescaled when the maximum moves
//     (relative error ~1e-7, inside the 1e-5 softmax tolerance); it only enters the cost -p[answer], an atomically
//     accumulated float anyway.  A caller that wants the probabilities themselves takes the two-kernel path.
// A workgroup of 4 wavefronts owns 64 queries (their code fragments stay in registers) and a slice of the dictionary; the
// slice's W rows pass through a double-buffered LDS tile shared by the 4 wavefronts (64 rows, pitch Dp + 16).  Slices keep
// every CU busy at 8 192 queries; a second, tiny kernel merges the per-slice records.
// ---------------------------------------------------------------------------
struct AnsPart {            // per (slice, query)
    int m;                  // maximum logit code of the slice
    uint32_t idx;           // its highest index
    double sum;             // sum over the slice of e^((k - m) . scale)
};
constexpr uint32_t kAnsTile = 64;                       // dictionary rows per LDS tile = 4 accumulator blocks of 16
constexpr int kAnsFloor = -(1 << 28);                   // below every logit code (|sum of 256 products of 7-bit codes| < 2^23); e^(floor . scale) = 0

// e^((k - m) . scale) for a code difference dk <= 0.  Base e^x: the hardware exponential with scale . log2 e folded into one
// factor (v_cvt, v_mul, v_exp; ~2 ulp -- the reference's own kernel uses the fast __expf there, lib/layer_cuda.cu:2006);
// the other bases through the shared definitions.
template <bool EXPB>
__device__ __forceinline__ float ans_term(int dk, float scale, float scale_log2e, const SmCfg &c)
{
    if (EXPB) return __builtin_amdgcn_exp2f((float)dk * scale_log2e);
    return sm_exp((float)dk * scale, c);
}

template <int KSTEPS, bool EXPB>
__global__ void __launch_bounds__(kBlock)
k_answer_i8_part(const float *__restrict__ u, const int8_t *__restrict__ w, AnsPart *__restrict__ part,
                 uint32_t n_query, uint32_t D, uint32_t V, QFmt fu, float scale, uint32_t softmax_base, uint32_t tiles_per_slice)
{
    constexpr uint32_t Dp = KSTEPS * 64, PITCH = Dp + 16, NB = kAnsTile / 16;
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][kAnsTile * PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t m0 = (blockIdx.x * kWaves + wave) * 16;
    const uint32_t row = lane & 15, kq = lane >> 4;
    const SmCfg smc{softmax_base, false, false, 1.0f};
    const float scale_log2e = scale * 1.44269504088896341f;
    // A fragments: the wavefront's 16 query rows are read coalesced, turned into codes and passed through a private
    // corner of the second tile buffer (nobody writes it before the barrier below), so that each lane can pick up its
    // row's 16-byte pieces; they stay in registers for the whole slice
    i32x4 a[KSTEPS];
    {
        uint8_t *stage = tile[1] + wave * 16 * PITCH;
        const bool vec = (D & 3u) == 0;
#pragma unroll 4
        for (uint32_t i = lane; i < 16 * (Dp / 4); i += kWave) {
            const uint32_t r = i / (Dp / 4), c = (i % (Dp / 4)) * 4;
            const bool ok = m0 + r < n_query;
            const float *src = u + (size_t)(m0 + r) * D + c;
            float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (vec) {
                if (ok && c < D) { const float4 v = *(const float4 *)src; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (ok && c + j < D) x[j] = src[j];
            }
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) pk |= ((uint32_t)qm_code(x[j], fu.iwl, fu.frac) & 0xFFu) << (8 * j);
            *(uint32_t *)(stage + r * PITCH + c) = pk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) a[ks] = *(const i32x4 *)(stage + row * PITCH + ks * 64 + 16 * kq);
    }
    // accumulator layout: lane l holds answer column l & 15 of a 16-answer block and query rows 4 (l >> 4) + r
    int mx[4] = {kAnsFloor, kAnsFloor, kAnsFloor, kAnsFloor};       // (no logit is that low: the first one becomes the maximum)
    uint32_t bi[4] = {0, 0, 0, 0};
    double sm[4] = {0.0, 0.0, 0.0, 0.0};

    const uint32_t n_tiles = (V + kAnsTile - 1) / kAnsTile;
    const uint32_t t_lo = blockIdx.y * tiles_per_slice, t_hi = (t_lo + tiles_per_slice < n_tiles) ? t_lo + tiles_per_slice : n_tiles;
    constexpr uint32_t PIECES = kAnsTile * Dp / 16 / kBlock;             // 16-byte pieces of a tile per thread: 1, 2 or 4
    i32x4 stage_r[PIECES];
    auto fetch = [&](uint32_t t) {
#pragma unroll
        for (uint32_t p = 0; p < PIECES; p++) {
            const uint32_t i = p * kBlock + tid, r = i / (Dp / 16), c = i % (Dp / 16);
            const uint32_t v = t * kAnsTile + r;
            stage_r[p] = i32x4{0, 0, 0, 0};
            if (v < V) stage_r[p] = *(const i32x4 *)(w + (size_t)v * Dp + c * 16);
        }
    };
    auto put = [&](uint32_t b) {
#pragma unroll
        for (uint32_t p = 0; p < PIECES; p++) {
            const uint32_t i = p * kBlock + tid, r = i / (Dp / 16), c = i % (Dp / 16);
            *(i32x4 *)(tile[b] + r * PITCH + c * 16) = stage_r[p];
        }
    };
    if (t_lo < t_hi) { fetch(t_lo); put(0); }
    __syncthreads();
    for (uint32_t t = t_lo; t < t_hi; t++) {
        const uint32_t b = (t - t_lo) & 1u;
        if (t + 1 < t_hi) fetch(t + 1);
        // four independent accumulator chains: the matrix pipe runs them back to back
        i32x4 acc[NB];
#pragma unroll
        for (uint32_t nb = 0; nb < NB; nb++) acc[nb] = i32x4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
            for (uint32_t nb = 0; nb < NB; nb++) {
                const i32x4 bm = *(const i32x4 *)(tile[b] + (nb * 16 + row) * PITCH + ks * 64 + kq * 16);
                acc[nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks], bm, acc[nb], 0, 0, 0);
            }
        const uint32_t v0 = t * kAnsTile + row;                          // this lane's column in block nb: v0 + 16 nb
        if (t * kAnsTile + kAnsTile > V) {                               // the ragged last tile: columns past V drop out
#pragma unroll
            for (uint32_t nb = 0; nb < NB; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[nb][r] = (v0 + 16 * nb < V) ? acc[nb][r] : kAnsFloor;
        }
        // Per query row: the new maximum over the four columns first, ONE rescale of the running sum, then the four terms
        // (summed in float: four values <= 1) -- 2 conversions and 1 multiply-add in double per row and tile.
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int mn = mx[r];
#pragma unroll
            for (uint32_t nb = 0; nb < NB; nb++) mn = acc[nb][r] > mn ? acc[nb][r] : mn;
            const float fs = ans_term<EXPB>(mx[r] - mn, scale, scale_log2e, smc);      // 1 when the maximum stays (no overflow: |codes| < 2^23)
            float e = 0.0f;
#pragma unroll
            for (uint32_t nb = 0; nb < NB; nb++) {
                e += ans_term<EXPB>(acc[nb][r] - mn, scale, scale_log2e, smc);
                bi[r] = (acc[nb][r] == mn) ? v0 + 16 * nb : bi[r];        // columns arrive in ascending order: the later index wins a tie
            }
            sm[r] = sm[r] * (double)fs + (double)e;
            mx[r] = mn;
        }
        if (t + 1 < t_hi) put(b ^ 1u);
        __syncthreads();
    }
    // the 16 lanes that hold the same query rows (equal l >> 4) merge their columns
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m2 = __shfl_xor(mx[r], o);
            const uint32_t i2 = __shfl_xor(bi[r], o);
            const double s2 = __shfl_xor(sm[r], o);
            const int M = m2 > mx[r] ? m2 : mx[r];
            const double f1 = (double)ans_term<EXPB>(mx[r] - M, scale, scale_log2e, smc);
            const double f2 = (double)ans_term<EXPB>(m2 - M, scale, scale_log2e, smc);
            sm[r] = sm[r] * f1 + s2 * f2;
            bi[r] = (m2 == mx[r]) ? (i2 > bi[r] ? i2 : bi[r]) : (m2 > mx[r] ? i2 : bi[r]);
            mx[r] = M;
        }
    }
    if (row == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t q = m0 + 4 * kq + r;
            if (q < n_query) part[(size_t)blockIdx.y * n_query + q] = AnsPart{mx[r], bi[r], sm[r]};
        }
    }
}

// merges the slices of a query (ascending dictionary ranges: the later slice wins a tie), writes the prediction and
// accumulates cost / match (cross_entropy_run mode 3).  16 lanes per query: they also form the labelled answer's logit
// code (one row of W against the query's codes -- the same integer the matrix cores produced for that column), so the
// projection kernel carries no label bookkeeping.
__global__ void __launch_bounds__(kBlock)
k_answer_i8_combine(const AnsPart *__restrict__ part, const float *__restrict__ u, const int8_t *__restrict__ w,
                    const uint32_t *__restrict__ answer, uint32_t *__restrict__ pred, float *cost, uint32_t *match,
                    uint32_t n_query, uint32_t n_slice, uint32_t D, uint32_t Dp, uint32_t V, QFmt fu, float scale, uint32_t softmax_base)
{
    const SmCfg smc{softmax_base, false, false, 1.0f};
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & 15;
    const uint32_t wave_g = (blockIdx.x * kBlock + threadIdx.x) / kWave, n_wave = gridDim.x * (kBlock / kWave);
    float cost_acc = 0.0f;
    uint32_t match_acc = 0;
    for (uint32_t qb = wave_g * 4; qb < n_query; qb += n_wave * 4) {              // wavefront-uniform trip count
        const uint32_t q = qb + (lane >> 4);
        const bool q_ok = q < n_query;
        const uint32_t y = (answer && q_ok) ? answer[q] : 0xFFFFFFFFu;
        int ky = 0;
        if (y < V) {
            const uint32_t per = Dp / 16;
            for (uint32_t i = 0; i < per; i++) {
                const uint32_t c = sub * per + i;
                const int code = c < D ? qm_code(u[(size_t)q * D + c], fu.iwl, fu.frac) : 0;
                ky += code * (int)w[(size_t)y * Dp + c];
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) ky += __shfl_xor(ky, o);
        if (sub != 0 || !q_ok) continue;
        int M = kAnsFloor;
        uint32_t bi = 0;
        for (uint32_t s = 0; s < n_slice; s++) {
            const AnsPart p = part[(size_t)s * n_query + q];
            if (p.m >= M && p.m != kAnsFloor) { M = p.m; bi = p.idx; }
        }
        double total = 0.0;
        for (uint32_t s = 0; s < n_slice; s++) {
            const AnsPart p = part[(size_t)s * n_query + q];
            total += p.sum * (p.m == M ? 1.0 : (double)sm_exp((float)(p.m - M) * scale, smc));
        }
        pred[q] = bi;
        if (y < V) {
            const float e = sm_exp((float)(ky - M) * scale, smc);
            const float py = (softmax_base == QMANN_SOFTMAX_EXP) ? (float)((double)e / total) : e / (float)total;
            cost_acc += -py;
            match_acc += (y == bi) ? 1u : 0u;
        }
    }
    if (answer) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cost_acc += __shfl_xor(cost_acc, o); match_acc += __shfl_xor(match_acc, o); }
        if (lane == 0) {
            if (cost) atomicAdd(cost, cost_acc);
            if (match && match_acc) atomicAdd(match, match_acc);
        }
    }
}

// ---------------------------------------------------------------------------
// Story embedding: E[s][j] = Qw( sum_k Qw( Qw(X[s][k]) . Qw(W[j][k]) ) )
// (lib/layer_cuda.cu:105-172 via :3531) for the A and C tables of every hop, then
// re-quantised to the format its consumer applies (keys: att[h], lib/layer_cuda.cu:120;
// values: act[h], :562) and stored as sign-magnitude int8.  X is a bag of words: a handful of
// non-zeros per row, so rows are first compacted to (index, value) pairs in LDS
// and the sum becomes a short gather over table columns.  One workgroup per
// story row; thread j < D owns output column j for all 2.n_hop tables.
// ---------------------------------------------------------------------------
constexpr int kMaxNnz = 64;

// sign-magnitude byte of Q(f)(x): the top byte of the reference's FLOAT2FIXED word
// (`mz`: see ew_to_bytes -- exactly -2^iwl is "minus zero" for the keys of a Hamming-family attention on a wider weight grid)
__device__ __forceinline__ int8_t sm_byte(float x, QFmt f, bool mz = false)
{
    const int k = qm_code(x, f.iwl, f.frac);
    const uint32_t mag = (mz && x == -(float)(1u << f.iwl)) ? 0u : (uint32_t)(k < 0 ? -k : k);
    return (int8_t)(mag | ((x >= 0.0f) ? 0u : 0x80u));
}

struct EmbedArgs {
    const float *story;
    const float *w_a[QMANN_MAX_HOP];
    const float *w_c[QMANN_MAX_HOP];
    int8_t *keys;
    int8_t *vals;
    size_t hop_stride;
    uint32_t n_hop, D, Dp, V, rows;
    QFmt act[QMANN_MAX_HOP], w[QMANN_MAX_HOP], att[QMANN_MAX_HOP];   // att: the format of the KEY BYTES (fill_key_formats)
    uint32_t key_mz;              // bit h: hop h's keys follow the minus-zero rule
    const uint32_t *row_list;     // optional: only these rows (device array; its length is read from *n_list)
    const uint32_t *n_list;
};

__global__ void __launch_bounds__(kBlock)
k_embed_story(const EmbedArgs a)
{
    __shared__ uint32_t nz_idx[kMaxNnz];
    __shared__ float nz_val[kMaxNnz];
    __shared__ uint32_t nnz;
    const uint32_t tid = threadIdx.x;
    const size_t n_it = a.row_list ? (size_t)*a.n_list : (size_t)a.rows;
    for (size_t it = blockIdx.x; it < n_it; it += gridDim.x) {     // grid-stride over story rows
    const size_t s = a.row_list ? (size_t)a.row_list[it] : it;
    const float *x = a.story + s * a.V;
    __syncthreads();
    if (tid == 0) nnz = 0;
    __syncthreads();
    for (uint32_t k = tid; k < a.V; k += kBlock) {
        const float v = x[k];
        if (v != 0.0f) {
            const uint32_t i = atomicAdd(&nnz, 1u);
            if (i < (uint32_t)kMaxNnz) { nz_idx[i] = k; nz_val[i] = v; }
        }
    }
    __syncthreads();
    const uint32_t n = nnz;
    const bool dense = n > (uint32_t)kMaxNnz;       // not a bag of words: fall back to the full row
    for (uint32_t j = tid; j < a.Dp; j += kBlock) {
        for (uint32_t h = 0; h < a.n_hop; h++) {
            const QFmt fw = a.w[h];
            int8_t kcode = 0, vcode = 0;
            if (j < a.D) {
                const float *wa = a.w_a[h] + (size_t)j * a.V;
                const float *wc = a.w_c[h] + (size_t)j * a.V;
                float sa = 0.0f, sc = 0.0f;           // exact: multiples of 2^-frac below 2^24 units
                if (!dense) {
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t k = nz_idx[i];
                        sa += qm_fixed_mul(nz_val[i], wa[k], fw, fw);
                        sc += qm_fixed_mul(nz_val[i], wc[k], fw, fw);
                    }
                } else {
                    for (uint32_t k = 0; k < a.V; k++) {
                        sa += qm_fixed_mul(x[k], wa[k], fw, fw);
                        sc += qm_fixed_mul(x[k], wc[k], fw, fw);
                    }
                }
                const float ea = qm_quant(sa, fw.iwl, fw.frac);
                const float ec = qm_quant(sc, fw.iwl, fw.frac);
                kcode = sm_byte(ea, a.att[h], (a.key_mz >> h) & 1u);
                vcode = sm_byte(ec, a.act[h]);
            }
            a.keys[(size_t)h * a.hop_stride + s * a.Dp + j] = kcode;
            a.vals[(size_t)h * a.hop_stride + s * a.Dp + j] = vcode;
        }
    }
    }
}

// Question embedding u0[j] = Qw0( sum_k Qw0( Qw0(W[j][k]) . Qw0(q[k]) ) ) (lib/layer_cuda.cu:49-83 via :3184)
__global__ void __launch_bounds__(kBlock)
k_embed_query(const float *__restrict__ question, const float *__restrict__ w_q, float *__restrict__ u0,
              uint32_t D, uint32_t V, QFmt fw, uint32_t n_query, const uint32_t *__restrict__ row_list,
              const uint32_t *__restrict__ n_list)
{
    __shared__ uint32_t nz_idx[kMaxNnz];
    __shared__ float nz_val[kMaxNnz];
    __shared__ uint32_t nnz;
    const uint32_t tid = threadIdx.x;
    const size_t n_it = row_list ? (size_t)*n_list : (size_t)n_query;
    for (size_t it = blockIdx.x; it < n_it; it += gridDim.x) {     // grid-stride over questions
    const size_t q = row_list ? (size_t)row_list[it] : it;
    const float *x = question + q * V;
    __syncthreads();
    if (tid == 0) nnz = 0;
    __syncthreads();
    for (uint32_t k = tid; k < V; k += kBlock) {
        const float v = x[k];
        if (v != 0.0f) {
            const uint32_t i = atomicAdd(&nnz, 1u);
            if (i < (uint32_t)kMaxNnz) { nz_idx[i] = k; nz_val[i] = v; }
        }
    }
    __syncthreads();
    const uint32_t n = nnz;
    for (uint32_t j = tid; j < D; j += kBlock) {
        const float *wr = w_q + (size_t)j * V;
        float s = 0.0f;
        if (n <= (uint32_t)kMaxNnz) {
            for (uint32_t i = 0; i < n; i++) s += qm_fixed_mul(wr[nz_idx[i]], nz_val[i], fw, fw);
        } else {
            for (uint32_t k = 0; k < V; k++) s += qm_fixed_mul(wr[k], x[k], fw, fw);
        }
        u0[q * D + j] = qm_quant(s, fw.iwl, fw.frac);
    }
    }
}

// ---------------------------------------------------------------------------
// Bag-of-words rows -> word lists.  The reference hands its stories over as float rows of dim_input entries
// (cuda_data_in pools, MemN2N.c:2337-2349) although sample.c knows every sentence as a handful of word indices; a row
// whose non-zero entries are small positive integers (counts; the time entry is a 1) IS such a list, and the word-index
// kernels embed it on the integer / matrix-core path, bit-identical to the float path.  One wavefront per row: the
// non-zeros are compacted in ascending index order, an index repeated by its count; a row that is not a plain bag of
// words (fractional or negative entries -- position encoding --, more than 16 words) gets an empty list and is
// recorded in `irr_rows` for the float kernel to redo.
// ---------------------------------------------------------------------------
// L lanes per row (16 / 32 / 64: the smallest that covers a dictionary of up to 64 words; longer rows take passes of 64),
// 64 / L rows per wavefront.  A pass whose entries are all 0.0 or 1.0 -- every pass of a real bAbI row but the few with a word
// said twice -- takes its positions from one ballot instead of a six-step scan.
template <int L>
__global__ void __launch_bounds__(kBlock)
k_bow_to_words(const float *__restrict__ bow, uint32_t rows, uint32_t V, uint16_t *__restrict__ words,
               uint32_t *__restrict__ irr_rows, uint32_t *__restrict__ n_irr)
{
    constexpr int RPW = kWave / L;
    constexpr uint64_t kGroupMask = L == 64 ? ~0ull : ((1ull << (L & 63)) - 1ull);
    __shared__ __attribute__((aligned(16))) uint16_t buf[kWaves][RPW][24];       // 16 list slots + a dump slot (index 16) for the lanes with nothing to write
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, sub = lane & (L - 1), grp = lane / L;
    const uint64_t below = (1ull << sub) - 1ull;
    for (size_t r0 = ((size_t)blockIdx.x * kWaves + wave) * RPW; r0 < rows; r0 += (size_t)gridDim.x * kWaves * RPW) {
        const size_t r = r0 + grp;
        const bool r_ok = r < rows;
        if (sub < 16) buf[wave][grp][sub] = 0xFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t base = 0;
        bool bad = false;
        for (uint32_t c0 = 0; c0 < V; c0 += L) {                         // wavefront-uniform
            const uint32_t k = c0 + sub;
            float x = 0.0f;
            if (r_ok && k < V) x = bow[r * V + k];
            // nothing but 0.0 and 1.0 in the pass (told by the bit patterns; the compare's lane mask IS the ballot): positions from
            // the ballot, a branch-free write; anything else -- counts, fractions, -0.0, NaN -- takes the general tests and the scan
            const uint32_t bits = __builtin_bit_cast(uint32_t, x);
            const bool one = bits == 0x3F800000u;
            if (!__any(bits != 0u && !one)) {
                const uint64_t gm = (__ballot(one) >> (grp * L)) & kGroupMask;
                const uint32_t pos = base + (uint32_t)__popcll(gm & below);
                buf[wave][grp][(one && pos < 16u) ? pos : 16u] = (uint16_t)k;
                base += (uint32_t)__popcll(gm);
                continue;
            }
            const bool nz = x != 0.0f;                                   // (a NaN is "non-zero" and fails the next test)
            const int c = (x >= 1.0f && x <= 16.0f) ? (int)x : 0;
            const bool ok = nz && c > 0 && (float)c == x;
            bad |= nz && !ok;
            const uint32_t cnt = ok ? (uint32_t)c : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int o = 1; o < L; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, L);
                if (sub >= (uint32_t)o) incl += t;
            }
            const uint32_t pos = base + incl - cnt;
            for (uint32_t t = 0; t < cnt; t++)
                if (pos + t < 16u) buf[wave][grp][pos + t] = (uint16_t)k;
            base += __shfl(incl, L - 1, L);
        }
        const bool irregular = ((__ballot(bad) >> (grp * L)) & kGroupMask) != 0ull || base > 16u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (r_ok && sub < 8) ((uint32_t *)(words + r * 16))[sub] = irregular ? 0xFFFFFFFFu : ((const uint32_t *)buf[wave][grp])[sub];
        if (r_ok && irregular && sub == 0) irr_rows[atomicAdd(n_irr, 1u)] = (uint32_t)r;
        __builtin_amdgcn_wave_barrier();                                 // the next rows rewrite the buffer
    }
}

// Dictionaries of more than 64 words (the 20-task set: 238): a lane takes FOUR consecutive entries of the row in one 16-byte
// buffer load (rows are only 4-byte aligned: 952 bytes at V = 238; the resource ends with the array, so the last rows' overhang
// reads as zeros), a pass covers 256 entries, and the next row's load is issued before this row is worked on -- the kernel is
// bound by the float rows it reads (0.96 ms -> see DESIGN.md for 2.44 M rows of 238).  Ascending order = lane-major, then the
// lane's four entries: positions from four ballots (all counts 0 / 1) or from a scan of the lanes' totals.
__global__ void __launch_bounds__(kBlock)
k_bow_to_words_wide(const float *__restrict__ bow, uint32_t rows, uint32_t V, uint16_t *__restrict__ words,
                    uint32_t *__restrict__ irr_rows, uint32_t *__restrict__ n_irr)
{
    __shared__ __attribute__((aligned(16))) uint16_t buf[kWaves][24];    // 16 list slots + a dump slot (index 16) for the lanes with nothing to write
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const size_t stride = (size_t)gridDim.x * kWaves;
    const size_t row_bytes = (size_t)V * 4u, total = (size_t)rows * row_bytes;
    const uint32_t n_pass = (V + 255u) / 256u;
    // one resource over the whole array while 32-bit offsets reach every row (< 4 GiB: 4.5 M rows of 238); per row beyond
    const bool one_rsrc = total <= 0xFFFFFFFFull;
    const __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc((void *)bow, 0, (int)(uint32_t)(one_rsrc ? total : 0u), kRawBufferFlags);
    auto request = [&](size_t r, uint32_t pass) -> i32x4 {
        const uint32_t in_row = (pass * 256u + lane * 4u) * 4u;
        // (the row's offset in the VECTOR offset: that is the one the bounds check covers, and the last rows' overhang must read as zeros)
        if (one_rsrc) return __builtin_amdgcn_raw_buffer_load_b128(rs_all, (int)((uint32_t)(r * row_bytes) + in_row), 0, kBufferNt);
        const size_t off = r * row_bytes, left = total - off;           // (wavefront-uniform)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)bow + off), 0,
                                                                             (int)(left < 0x7FFFFFFFu ? left : 0x7FFFFFFFu), kRawBufferFlags);
        return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)in_row, 0, kBufferNt);
    };
    size_t r = (size_t)blockIdx.x * kWaves + wave;
    if (r >= rows) return;
    const uint64_t below = (1ull << lane) - 1ull;
    i32x4 x = request(r, 0);
    for (; r < rows; r += stride) {
        if (lane < 16) buf[wave][lane] = 0xFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t base = 0;
        bool bad = false;
        for (uint32_t pass = 0; pass < n_pass; pass++) {                 // wavefront-uniform
            const i32x4 cur = x;
            // the next load: this row's next pass, or the next row's first
            if (pass + 1u < n_pass) x = request(r, pass + 1u);
            else if (r + stride < rows) x = request(r + stride, 0);
            const uint32_t k0 = pass * 256u + lane * 4u;
            // almost every pass of a real row holds nothing but 0.0 and 1.0: told by the bit patterns, one compare per entry each
            // (its lane mask IS the ballot); anything else -- counts, fractions, -0.0, NaN -- takes the general tests
            uint32_t bits[4];
            bool one[4], other = false;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // (element by element, a plain `if`: `cond ? bit_cast(cur[e]) : 0` compiled to cur[0] for every e -- hipcc 7.2)
                bits[e] = (uint32_t)cur[e];
                if (k0 + (uint32_t)e >= V) bits[e] = 0u;                 // (past the row's end the load holds the next row's entries)
                one[e] = bits[e] == 0x3F800000u;
                other |= bits[e] != 0u && !one[e];
            }
            if (!__any(other)) {
                const uint64_t m0 = __ballot(one[0]), m1 = __ballot(one[1]), m2 = __ballot(one[2]), m3 = __ballot(one[3]);
                uint32_t pos = base + (uint32_t)(__popcll(m0 & below) + __popcll(m1 & below) + __popcll(m2 & below) + __popcll(m3 & below));
#pragma unroll
                for (int e = 0; e < 4; e++) {                            // no branches: a lane without an entry writes the dump slot
                    buf[wave][(one[e] && pos < 16u) ? pos : 16u] = (uint16_t)(k0 + (uint32_t)e);
                    pos += one[e] ? 1u : 0u;
                }
                base += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                continue;
            }
            uint32_t cnt[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v = __builtin_bit_cast(float, bits[e]);
                const bool nz = v != 0.0f;                               // (a NaN is "non-zero" and fails the next test)
                const int c = (v >= 1.0f && v <= 16.0f) ? (int)v : 0;
                const bool ok = nz && c > 0 && (float)c == v;
                bad |= nz && !ok;
                cnt[e] = ok ? (uint32_t)c : 0u;
            }
            const uint32_t mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += t;
            }
            uint32_t pos = base + incl - mine;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                for (uint32_t t = 0; t < cnt[e]; t++)
                    if (pos + t < 16u) buf[wave][pos + t] = (uint16_t)(k0 + (uint32_t)e);
                pos += cnt[e];
            }
            base += __shfl(incl, kWave - 1);
        }
        const bool irregular = __any(bad) || base > 16u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 8) ((uint32_t *)(words + r * 16))[lane] = irregular ? 0xFFFFFFFFu : ((const uint32_t *)buf[wave])[lane];
        if (irregular && lane == 0) irr_rows[atomicAdd(n_irr, 1u)] = (uint32_t)r;
        __builtin_amdgcn_wave_barrier();                                 // the next row rewrites the buffer
    }
}

// ---------------------------------------------------------------------------
// Story embedding from WORD INDICES.  A bag-of-words row has at most a
// handful of non-zeros, so the dense X . W^T of dense_mat_fwd is a gather-sum over table rows.
// Wire format: uint16 [rows][max_words], unused entries 0xFFFF; with `time_last` the last valid
// entry of a row is its time-encoding index, whose bag-of-words entry is SET to 1 while word
// entries COUNT occurrences (MemN2N/sample.c:466-475, 544-548).  Tables are int8 [V][Dp]
// two's-complement codes of Q(w[h]) (transposed, so a word's row is contiguous).  For an integer
// count c the reference term Qw(Qw(c) . Qw(W)) is clamp(c' . kw, +-max) with c' = the count
// saturated to the format (a code needs no truncation when one factor is an integer), so the
// result is bit-identical to the float path.  One wavefront per story row, lane = column.
// ---------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
constexpr int kMaxWords = 16;

struct EmbedIdxArgs {
    const uint16_t *words;
    const int8_t *t_a[QMANN_MAX_HOP];
    const int8_t *t_c[QMANN_MAX_HOP];
    int8_t *keys;
    int8_t *vals;
    size_t hop_stride;
    uint32_t n_hop, D, Dp, V, rows, max_words, time_last;
    QFmt act[QMANN_MAX_HOP], w[QMANN_MAX_HOP], att[QMANN_MAX_HOP];   // att: the format of the KEY BYTES (fill_key_formats)
    uint32_t key_mz;              // bit h: hop h's keys follow the minus-zero rule
    uint32_t qkinds;              // 2 bits per (hop, table): how the matrix-core kernel's epilogue may quantise (kQk*, below)
};

// integer count c >= 0 as a code of the format in units of 2^-frac: Qw(c), saturating at the format maximum
__device__ __forceinline__ int count_code(uint32_t c, uint32_t frac, int maxw)
{
    const uint64_t k = (uint64_t)c << frac;
    return k > (uint64_t)maxw ? maxw : (int)k;
}
// 16 lanes per story row (a lane owns 4 adjacent columns = one dword of a table row), 4 rows per wavefront step,
// persistent workgroups.  The kernel is bound by VALU issue, so the work per word slot is kept small:
//   * which slot adds what (word, count, first occurrence) is settled once per row group: the usual case -- no word
//     twice in a row -- is found with one LDS atomic per lane on a 256-bit hash bitmap per row; only a group with a
//     possible repeat compares its slots pairwise;
//   * the 8 (16) slots of a row are read back as one (two) 16-byte LDS loads and the table reads of all slots are in
//     flight together, for the A and the C table of a hop at once;
//   * small dictionaries: all 2.n_hop tables sit in LDS expanded to int16 (one ds_read_b64 per slot and table, no
//     unpacking; row V is zero so that an empty slot needs no predicate).  Large ones are gathered from L2 as int8.
constexpr uint32_t kEwWords = 16;                       // word slots per row
// per-wavefront LDS: wd u16 [4][16], ct u8 [4][16], duplicate-detection bitmaps u32 [4][8]
constexpr uint32_t kEwWd = 0, kEwCt = 128, kEwBm = 192, kEwWaveBytes = 320;

// One row group (4 rows, 16 lanes each): returns word | count << 16 for a slot that adds, 0xFFFF for one that does not
// (unused, out of range, or a repeat of a word an earlier slot of the row carries with its count).
__device__ __forceinline__ uint32_t ew_pack_row(uint32_t w, uint32_t V, bool time_last, uint32_t nw, uint32_t lane, uint32_t *bm)
{
    const uint32_t sub = lane & 15u, grp = lane >> 4;
    const uint32_t m16 = (uint32_t)(__ballot(w != 0xFFFFu) >> (16 * grp)) & 0xFFFFu;
    const uint32_t n_valid = 32u - (uint32_t)__clz(m16);
    const bool valid = w != 0xFFFFu && w < V;
    const bool is_time = time_last && valid && (sub + 1 == n_valid);
    if (lane < 32u) bm[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool clash = false;
    if (valid) {
        const uint32_t bit = 1u << (w & 31u);
        clash = (atomicOr(&bm[grp * 8u + ((w >> 5) & 7u)], bit) & bit) != 0u;
    }
    uint32_t cnt = 1;
    bool dup = false;
    if (__any(clash)) {                                  // a row of this group may hold a word twice: the exact, slow way
        const uint32_t me = w | (valid ? 1u << 16 : 0u) | (is_time ? 1u << 17 : 0u);
        bool timed = false;
        cnt = 0;
        for (uint32_t j = 0; j < nw; j++) {
            const uint32_t o = (uint32_t)__shfl((int)me, (int)j, 16);
            const bool same = (((o ^ me) & 0xFFFFu) == 0u) && ((o >> 16) & 1u);
            const bool o_time = (o >> 17) & 1u;
            cnt += (same && !o_time) ? 1u : 0u;          // word entries COUNT occurrences ...
            timed |= same && o_time;                     // ... the time entry SETS its slot to 1
            dup |= same && j < sub;
        }
        if (timed) cnt = 1;
    }
    return (valid && !dup) ? (w | (cnt << 16)) : 0xFFFFu;
}

// Qw of the sum, then the memory byte: magnitude moved to the target grid (toward zero), clamped, sign bit from the
// VALUE (a negative sum that truncates to zero is "minus zero")
// `mz`: the key bytes of a Hamming-family attention whose weight grid is wider than the attention grid -- a value of exactly
// -2^iwl_att is "minus zero" in the reference's operand word (ham_common.h::ham_ubyte): magnitude 0, sign kept
__device__ __forceinline__ uint32_t ew_to_bytes(s16x2 x, int maxw, QFmt fw, QFmt dst, bool mz = false)
{
    const short mw = (short)maxw;
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, s16x2{(short)-mw, (short)-mw}), s16x2{mw, mw});
    u16x2 mag = __builtin_bit_cast(u16x2, __builtin_elementwise_max(x, (s16x2)(-x)));
    mag = dst.frac >= fw.frac ? (u16x2)(mag << (unsigned short)(dst.frac - fw.frac)) : (u16x2)(mag >> (unsigned short)(fw.frac - dst.frac));
    const unsigned short md = (unsigned short)((1u << (dst.iwl + dst.frac)) - 1u);
    if (mz) {
        const unsigned short edge = (unsigned short)(md + 1u);
        const u16x2 hit = __builtin_bit_cast(u16x2, (s16x2)(mag == u16x2{edge, edge})) & __builtin_bit_cast(u16x2, (s16x2)(x < s16x2{0, 0}));
        mag = mag & ~hit;
    }
    mag = __builtin_elementwise_min(mag, u16x2{md, md});
    const u16x2 sgn = __builtin_bit_cast(u16x2, (s16x2)(x >> 8)) & (unsigned short)0x0080;
    return __builtin_bit_cast(uint32_t, (u16x2)(mag | sgn));
}

// ew_to_bytes for the usual format pairs, on FOUR values at once in the byte domain (the matrix-core kernel's epilogue is a
// third of its instructions).  With word lengths of at most 8 the clamped sum fits a byte; its sign-magnitude form is
// (x ^ s) + (s & 1) per byte (s = 0xFF where x < 0: a negative byte becomes |x| - 1, then + 1, never a carry), and the target
// grid is at most one bit away in EN_MQ (MemN2N.c:748-754) and equal without it:
//   kQkSame   dst.frac == fw.frac, dst clamp not below the source's:   nothing to do
//   kQkLeft   dst.frac == fw.frac + 1, dst word length 8:              m + m per byte (<= 254), bytes >= 128 saturate to 127
//   kQkRight  dst.frac == fw.frac - 1, dst clamp not below maxw >> 1:  (m >> 1) per byte; the sign stays the VALUE's
//   kQkGeneral anything else (and keys under the minus-zero rule):     ew_to_bytes
enum { kQkSame = 0, kQkLeft = 1, kQkRight = 2, kQkGeneral = 3 };
__host__ __device__ inline uint32_t qkind_of(QFmt fw, QFmt dst, bool mz)
{
    const uint32_t maxw = (1u << (fw.iwl + fw.frac)) - 1u, md = (1u << (dst.iwl + dst.frac)) - 1u;
    if (mz || maxw > 127u) return kQkGeneral;
    if (dst.frac == fw.frac && md >= maxw) return kQkSame;
    if (dst.frac == fw.frac + 1u && md == 127u) return kQkLeft;
    if (dst.frac + 1u == fw.frac && md >= (maxw >> 1)) return kQkRight;
    return kQkGeneral;
}
// x01 / x23: the four sums as packed int16 pairs (columns 0, 1 and 2, 3 of a dword); returns the dword of memory bytes
__device__ __forceinline__ uint32_t ew_to_bytes4(s16x2 x01, s16x2 x23, int maxw, uint32_t kind)
{
    const short mw = (short)maxw;
    x01 = __builtin_elementwise_min(__builtin_elementwise_max(x01, s16x2{(short)-mw, (short)-mw}), s16x2{mw, mw});
    x23 = __builtin_elementwise_min(__builtin_elementwise_max(x23, s16x2{(short)-mw, (short)-mw}), s16x2{mw, mw});
    const uint32_t d = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, x23), __builtin_bit_cast(uint32_t, x01), 0x06040200u);   // two's complement bytes
    const uint32_t sg = d & 0x80808080u;
    const uint32_t sm = __builtin_amdgcn_perm(0u, 0u, sg);               // 0xFF where negative (selector bytes >= 0x80 give 0xFF, 0 gives byte 0 = 0)
    uint32_t m = (d ^ sm) + (sm & 0x01010101u);                           // |x| per byte
    if (kind == kQkLeft) {
        m += m;
        m = (m | __builtin_amdgcn_perm(0u, 0u, m & 0x80808080u)) & 0x7F7F7F7Fu;
    } else if (kind == kQkRight) {
        m = (m >> 1) & 0x3F3F3F3Fu;
    }
    return m | sg;
}

template <bool TAB16>
__global__ void __launch_bounds__(kBlock)
k_embed_story_idx(const EmbedIdxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, sub = lane & 15u, grp = lane >> 4;
    const uint32_t V = a.V, H = a.n_hop, nw = a.max_words;
    const uint32_t dw = a.Dp / 4;                                        // dwords (4 columns) per row: 16, 32 or 64
    const uint32_t dw_sh = 31u - (uint32_t)__clz(dw);                    // (a full 32-bit multiply costs four plain operations)
    const uint32_t tab_bytes = TAB16 ? (V + 1u) * dw * 8u : 0u;          // int16 [V + 1][dw]{even-columns pair, odd-columns pair}
    uint8_t *tabs = smem;                                                // [H][2] tables (A, C)
    uint8_t *ws = smem + 2u * H * tab_bytes + wave * kEwWaveBytes;
    uint16_t *wd = (uint16_t *)(ws + kEwWd);
    uint8_t *ct = ws + kEwCt;
    uint32_t *bm = (uint32_t *)(ws + kEwBm);
    if (TAB16) {
        for (uint32_t t = 0; t < 2u * H; t++) {
            const uint32_t *src = (const uint32_t *)((t & 1u) ? a.t_c[t >> 1] : a.t_a[t >> 1]);
            uint2 *dst = (uint2 *)(tabs + t * tab_bytes);
            for (uint32_t i = tid; i < (V + 1u) * dw; i += kBlock) {
                const uint32_t x = i < V * dw ? src[i] : 0u;
                const s16x2 ev = (__builtin_bit_cast(s16x2, x) << 8) >> 8, od = __builtin_bit_cast(s16x2, x) >> 8;
                dst[i] = uint2{__builtin_bit_cast(uint32_t, ev), __builtin_bit_cast(uint32_t, od)};
            }
        }
        __syncthreads();
    }

    const size_t rows_per_pass = (size_t)gridDim.x * kWaves * 4;
    for (size_t s0 = ((size_t)blockIdx.x * kWaves + wave) * 4; s0 < a.rows; s0 += rows_per_pass) {
        const size_t s = s0 + grp;
        const bool row_ok = s < a.rows;
        uint32_t w = 0xFFFFu;
        if (row_ok && sub < nw) w = a.words[s * nw + sub];
        const uint32_t pk = ew_pack_row(w, V, a.time_last != 0, nw, lane, bm);
        wd[grp * kEwWords + sub] = (uint16_t)pk;
        ct[grp * kEwWords + sub] = (uint8_t)(pk >> 16);
        const bool multi = __any((pk >> 16) > 1u);
        // slots 8..15 matter only when some row of the group uses them
        const uint32_t n_pass = __any(sub >= 8u && (pk & 0xFFFFu) != 0xFFFFu) ? 2u : 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (uint32_t c0 = 0; c0 < dw; c0 += 16)                          // 64 columns per round (one round at bAbI width)
        for (uint32_t h = 0; h < H; h++) {
            const uint32_t c4 = c0 + sub;                                 // (dw is a multiple of 16: Dp is 64, 128 or 256)
            uint32_t colmask = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) colmask |= (4 * c4 + (uint32_t)k < a.D ? 0xFFu : 0u) << (8 * k);
            const QFmt fw = a.w[h];
            const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
            // Qw(1 . kw) = kw needs 1.0 to be a value of the format (iwl >= 1); a purely fractional format saturates the
            // count itself, like a repeated word: then every term is Qw(Qw(count) . kw) per column (rare)
            const bool slow = multi || (1 << fw.frac) > maxw;
            const void *ta = TAB16 ? (const void *)(tabs + (2u * h) * tab_bytes) : (const void *)a.t_a[h];
            const void *tc = TAB16 ? (const void *)(tabs + (2u * h + 1u) * tab_bytes) : (const void *)a.t_c[h];
            s16x2 ae = {0, 0}, ao = {0, 0}, ce = {0, 0}, co = {0, 0};
            for (uint32_t pass = 0; pass < n_pass; pass++) {
                const i32x4 wv = *(const i32x4 *)(wd + grp * kEwWords + pass * 8u);
                s16x2 xa_e[8], xa_o[8], xc_e[8], xc_o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t we = ((uint32_t)wv[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
                    if (TAB16) {
                        const uint32_t off = (((we < V ? we : V) << dw_sh) + c4) * 8u;
                        const uint2 t1 = *(const uint2 *)((const uint8_t *)ta + off), t2 = *(const uint2 *)((const uint8_t *)tc + off);
                        xa_e[e] = __builtin_bit_cast(s16x2, t1.x); xa_o[e] = __builtin_bit_cast(s16x2, t1.y);
                        xc_e[e] = __builtin_bit_cast(s16x2, t2.x); xc_o[e] = __builtin_bit_cast(s16x2, t2.y);
                    } else {
                        const bool ok = we < V;
                        const uint32_t off = ((ok ? we : 0u) << dw_sh) + c4;
                        const uint32_t t1 = ok ? ((const uint32_t *)ta)[off] : 0u, t2 = ok ? ((const uint32_t *)tc)[off] : 0u;
                        // sign-extended bytes: even columns (0, 2) and odd columns (1, 3)
                        xa_e[e] = (__builtin_bit_cast(s16x2, t1) << 8) >> 8; xa_o[e] = __builtin_bit_cast(s16x2, t1) >> 8;
                        xc_e[e] = (__builtin_bit_cast(s16x2, t2) << 8) >> 8; xc_o[e] = __builtin_bit_cast(s16x2, t2) >> 8;
                    }
                }
                if (slow) {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const uint32_t c1 = ct[grp * kEwWords + pass * 8u + e];
                        const int cc = count_code(c1 ? c1 : 1u, fw.frac, maxw);
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            xa_e[e][k] = (short)qm_mul_code(cc, xa_e[e][k], fw.frac, maxw); xa_o[e][k] = (short)qm_mul_code(cc, xa_o[e][k], fw.frac, maxw);
                            xc_e[e][k] = (short)qm_mul_code(cc, xc_e[e][k], fw.frac, maxw); xc_o[e][k] = (short)qm_mul_code(cc, xc_o[e][k], fw.frac, maxw);
                        }
                    }
                }
                // Sums of <= 16 codes fit 16 bits: columns 0/2 and 1/3 of the dword are kept as packed int16 pairs
#pragma unroll
                for (int e = 0; e < 8; e++) { ae += xa_e[e]; ao += xa_o[e]; ce += xc_e[e]; co += xc_o[e]; }
            }
            if (row_ok) {
                const bool kmz = (a.key_mz >> h) & 1u;
                *(uint32_t *)(a.keys + (size_t)h * a.hop_stride + s * a.Dp + 4 * c4) = (ew_to_bytes(ae, maxw, fw, a.att[h], kmz) | (ew_to_bytes(ao, maxw, fw, a.att[h], kmz) << 8)) & colmask;
                *(uint32_t *)(a.vals + (size_t)h * a.hop_stride + s * a.Dp + 4 * c4) = (ew_to_bytes(ce, maxw, fw, a.act[h]) | (ew_to_bytes(co, maxw, fw, a.act[h]) << 8)) & colmask;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // the next group rewrites wd / ct
    }
}

// ---------------------------------------------------------------------------
// The same embedding on the int8 matrix cores, for dictionaries of up to 256 entries (every bAbI configuration).
// A tile of 16 story rows is a bag-of-words matrix X [16][K] of small counts (K = dictionary size padded to 64) and
//     E^T [columns][rows] = T^T [columns][K] . X^T
// is one v_mfma_i32_16x16x64_i8 per 16 columns and 64 dictionary entries, exact in int32.  The operands fall out of the
// wire format: X is built in LDS with one byte-add per word slot (the time entry is stored as 1 afterwards: it SETS its
// slot), T^T is staged once per workgroup.  The accumulator layout hands every lane 4 adjacent columns of one row -- one
// output dword -- so the quantisation epilogue (ew_to_bytes) runs on 2 packed registers per 16 x 16 tile and table.
// Per 16 rows and hop: ~200 vector instructions instead of ~2 000 for the gather-sum above (that kernel is bound by VALU
// issue: SQ_INSTS_VALU x 4.2 cycles is 70 % of its run time).
// What the product cannot express is the per-product clamp of a REPEATED word, Qw(Qw(count) . kw) != count . kw when
// |count . kw| exceeds the format: the second occurrence of a word in a row is noticed by the byte-add itself (it returns
// the old count) and those few (row, word) pairs get a per-column correction.  Purely fractional weight formats
// (1.0 not representable, so even a single word is Qw(Qw(1) . kw)) keep the gather-sum kernel.
// Grid: x = workgroups over tiles (persistent), y = hop.
// ---------------------------------------------------------------------------
constexpr uint32_t kEmRows = 16;                        // story rows per tile
// LDS images of the two MFMA operands, laid out for ds_read_b128's lane groups.  The hardware serves a wavefront's b128 read in
// four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and a
// fragment's lane l reads row l & 15, K bytes 16 (l >> 4) .. + 15: every group holds each row 0 .. 15 exactly once, with a K
// piece that depends on the row.  So a row's pieces must not share banks across K pieces: each (16 rows x 16 bytes) block of
// one K piece is 256 contiguous bytes, row r at 16 r -- a group's 16 lanes then cover the 64 banks once whatever pieces they
// read.  (Through round 4 these images were row-major with a 16-byte pad: rows 11 and 12 of adjacent pieces met in every
// group, SQ_LDS_BANK_CONFLICT was half of the LDS-active cycles and the LDS array was busy 64 % of the kernel.)
//   T^T: [table][16-column block][K / 16 pieces][16 columns][16 bytes]       em_tt_off(K, table, column, k)
//   X  : [K / 16 pieces][16 rows][16 bytes]                                   em_x_off(row, k)
__device__ __forceinline__ uint32_t em_tt_off(uint32_t K, uint32_t t, uint32_t col, uint32_t k)
{
    return ((((t * 4u + (col >> 4)) * (K >> 4) + (k >> 4)) * 16u + (col & 15u)) << 4) + (k & 15u);
}
__device__ __forceinline__ uint32_t em_x_off(uint32_t row, uint32_t k) { return ((((k >> 4) << 4) + row) << 4) + (k & 15u); }
constexpr uint32_t kEmDupCap = 128;                     // repeated (row, word) pairs a tile can hold: 16 rows x 16 slots / 2

// wavefronts per SIMD each instantiation is compiled for (its register budget; the launcher sizes the persistent grid by it)
// (KS = 1, task-1 dictionaries: 80 registers buy a third 8-wavefront workgroup per CU -- the kernel is latency-bound: +4.5 % on the task-1 forward
// in an interleaved A/B against five per SIMD; the 16-wavefront forms are held to one workgroup per CU by their LDS tiles)
constexpr int em_waves_per_simd(int KS) { return KS == 1 ? 6 : 4; }

template <int KS, int NW>                               // K / 64: 1, 2 or 4; wavefronts per workgroup (they share T^T)
__global__ void __launch_bounds__(NW * kWave, em_waves_per_simd(KS))
k_embed_story_mfma(const EmbedIdxArgs a)
{
    constexpr uint32_t kBlockEm = NW * kWave;
    constexpr uint32_t K = 64u * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (readfirstlane: the compiler treats threadIdx.x / 64 as divergent and would keep every per-wavefront address in vector registers)
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    constexpr uint32_t Dp = 64;                         // bAbI width (wider embeddings take the gather-sum kernel): the product loops
                                                        // unroll, so a tile's fragment loads and MFMAs are in flight together
    const uint32_t h = blockIdx.y, V = a.V, nw = a.max_words;
    int8_t *tt = (int8_t *)smem;                                         // [2][Dp][K]: A then C of this hop, transposed (em_tt_off)
    constexpr uint32_t SP = Dp + 16u;                                    // row pitch of the output staging tile (conflict-free dword writes)
    uint8_t *ws = smem + 2u * Dp * K + wave * (kEmRows * K + kEmRows * SP + kEmDupCap * 4u + 16u);
    uint32_t *X = (uint32_t *)ws;                                        // [16][K] bytes (em_x_off)
    uint8_t *stage = ws + kEmRows * K;                                   // [16][SP] one table's output rows
    uint32_t *dup = (uint32_t *)(stage + kEmRows * SP);                  // [kEmDupCap] row << 16 | word
    uint32_t *n_dup = dup + kEmDupCap;

    // ---- T^T of the hop's two tables ------------------------------------------------------------------------------
    for (uint32_t t = 0; t < 2; t++) {
        const uint32_t *src = (const uint32_t *)(t ? a.t_c[h] : a.t_a[h]);       // [V][Dp] two's complement
        for (uint32_t i = tid; i < K * (Dp / 4); i += kBlockEm) {
            const uint32_t k = i / (Dp / 4), c4 = i % (Dp / 4);
            const uint32_t x = k < V ? src[(size_t)k * (Dp / 4) + c4] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) tt[em_tt_off(K, t, 4 * c4 + j, k)] = (int8_t)(x >> (8 * j));
        }
    }
    __syncthreads();

    const QFmt fw = a.w[h], f_att = a.att[h], f_act = a.act[h];
    const bool kmz = (a.key_mz >> h) & 1u;
    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
    const uint32_t r4 = lane >> 2, qd = lane & 3u;                       // word phase: row r4 of the tile, slots 4 qd .. 4 qd + 3
    const uint32_t nrow = lane & 15u, kq = lane >> 4;                    // matrix phase: story row nrow, K bytes 16 kq .. / columns 4 kq ..
    const size_t n_tiles = ((size_t)a.rows + kEmRows - 1) / kEmRows;
    // This lane's 4 word slots of a tile (row r4, slots 4 qd ..): one 8-byte load (the launcher sends word lists whose
    // pitch is not a multiple of 4 slots to the gather-sum kernel).  Requested one tile ahead so that a wavefront does
    // not start every tile with a round trip to HBM: the load is unconditional (address clamped into the array) and its
    // result is not touched before the next iteration -- a predicated load, or any use of the value, makes the compiler
    // wait for it on the spot.
    const uint32_t slot0 = 4 * qd < nw ? 4 * qd : 0u;
    uint2 raw_next = {0u, 0u};
    auto request_words = [&](size_t tile_) {
        size_t row = tile_ * kEmRows + r4;
        row = row < a.rows ? row : a.rows - 1;
        raw_next = *(const uint2 *)(a.words + row * nw + slot0);
    };
    auto take_words = [&](size_t tile_, uint32_t (&w_)[4]) {
        const bool in = tile_ * kEmRows + r4 < a.rows;
        w_[0] = raw_next.x & 0xFFFFu; w_[1] = raw_next.x >> 16; w_[2] = raw_next.y & 0xFFFFu; w_[3] = raw_next.y >> 16;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) w_[i] = (in && 4 * qd + i < nw) ? w_[i] : 0xFFFFu;
    };
    const size_t tile_step = (size_t)gridDim.x * NW;
    request_words((size_t)blockIdx.x * NW + wave);
    // A tile's finished rows (16 bytes per lane and table) are stored at the START of the next iteration, behind that
    // iteration's word request: vector loads and stores retire in order through one counter, so waiting for the words
    // of a tile also waits for every older store -- stores issued a whole iteration earlier have long completed, stores
    // issued just before the wait would put their full latency in front of every tile.
    i32x4 pend[2];
    size_t pend_row0 = 0;
    bool pending = false;
    // (raw BUFFER stores on a resource that spans the tile's rows: a row past the plane's last one is dropped by the bounds check and
    // the stores stand under no lane predicate -- with one the compiler cannot count the stores in flight and waits for ALL of them,
    // `vmcnt(0)`, wherever it waits for the next tile's words: k_embed_story_mfma_hops below has the measurement)
    auto flush_pending = [&]() {
        if (!pending) return;
        const size_t left = (size_t)a.rows - pend_row0;
        const int bytes = (int)((left < kEmRows ? left : (size_t)kEmRows) * Dp);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(a.keys + (size_t)h * a.hop_stride + pend_row0 * Dp), 0, bytes, kRawBufferFlags);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(a.vals + (size_t)h * a.hop_stride + pend_row0 * Dp), 0, bytes, kRawBufferFlags);
        // (nt: 0.6-0.9 GB of rows per batch, read back once by the hop kernel after the whole plane is written -- streamed past
        // the caches: task-1 forward +2.2 %, 20-task +0.5 % in an interleaved A/B against the default policy)
        __builtin_amdgcn_raw_buffer_store_b128(pend[0], rk, (int)(lane * 16u), 0, kBufferNt);
        __builtin_amdgcn_raw_buffer_store_b128(pend[1], rv, (int)(lane * 16u), 0, kBufferNt);
        pending = false;
    };
    for (size_t tile = (size_t)blockIdx.x * NW + wave; tile < n_tiles; tile += tile_step) {
        const size_t row0 = tile * kEmRows;
        QM_MARK("X: zero, words of the tile");
        // ---- X: counts per (row, word) -----------------------------------------------------------------------------
        for (uint32_t i = lane; i < kEmRows * K / 16u; i += kWave) *(i32x4 *)((uint8_t *)X + i * 16u) = i32x4{0, 0, 0, 0};
        if (lane == 0) *n_dup = 0u;
        uint32_t w[4];
        take_words(tile, w);
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));    // (the words are taken HERE, before the next request is issued: see k_embed_story_mfma_hops)
        __builtin_amdgcn_sched_barrier(0);
        request_words(tile + tile_step);
        flush_pending();                                                 // the previous tile's rows (see below)
        QM_MARK("X: last slot, byte adds, time entry");
        uint32_t last = 0;                                               // 1 + this row's last non-empty slot
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) last = w[i] != 0xFFFFu ? 4 * qd + i + 1 : last;
        {   // maximum over the 4 lanes of the row
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
            last = o > last ? o : last;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last, 0x4E, 0xF, 0xF, true);                // quad_perm [2,3,0,1]
            last = o > last ? o : last;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t time_w = 0xFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (w[i] >= V) continue;                                     // empty (0xFFFF) or out of range: ignored
            if (a.time_last && 4 * qd + i + 1 == last) { time_w = w[i]; continue; }
            const uint32_t sh = 8u * (w[i] & 3u);
            const uint32_t old = atomicAdd(&X[em_x_off(r4, w[i]) >> 2], 1u << sh);
            if (((old >> sh) & 0xFFu) == 1u) {                           // the second occurrence announces the repeat, once
                const uint32_t n = atomicAdd(n_dup, 1u);
                if (n < kEmDupCap) dup[n] = (r4 << 16) | w[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (time_w != 0xFFFFu) ((uint8_t *)X)[em_x_off(r4, time_w)] = 1;  // the time entry SETS its slot (sample.c:474)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        QM_MARK("B fragments, repeated words");
        i32x4 bx[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bx[ks] = *(const i32x4 *)((const uint8_t *)X + em_x_off(nrow, ks * 64u + kq * 16u));

        // ---- repeated words: does any product Qw(Qw(count) . kw) differ from count . kw?  (it does only when the product
        // leaves the format: |count . kw| > max_w -- rare; lane = column) ------------------------------------------------
        const uint32_t nd = *n_dup < kEmDupCap ? *n_dup : kEmDupCap;
        uint32_t fix_rows = 0;                                           // rows to be summed term by term (wavefront-uniform)
        for (uint32_t d = 0; d < nd; d++) {
            const uint32_t e = dup[d], r = e >> 16, wd = e & 0xFFFFu;
            const int c = (int)((const uint8_t *)X)[em_x_off(r, wd)];
            const int cc = count_code((uint32_t)c, fw.frac, maxw);
            const int ka = (int)tt[em_tt_off(K, 0, lane, wd)], kc = (int)tt[em_tt_off(K, 1, lane, wd)];
            if (__any(qm_mul_code(cc, ka, fw.frac, maxw) != c * ka || qm_mul_code(cc, kc, fw.frac, maxw) != c * kc)) fix_rows |= 1u << r;
        }

        QM_MARK("products (MFMA)");
        // ---- the products, the epilogue, the stores ---------------------------------------------------------------
        // A lane's 4 columns are one dword; the 16 rows of a tile are contiguous in memory, so the tile leaves through an
        // LDS staging tile as whole rows, 16 bytes per lane (a store of 64 scattered dwords per 16 columns ran at the
        // rate of one cache line per lane group: the kernel was bound by it).
        i32x4 acc[2][4];
#pragma unroll
        for (uint32_t t = 0; t < 2; t++)
#pragma unroll
            for (uint32_t cb = 0; cb < 4; cb++) {
                acc[t][cb] = i32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const i32x4 am = *(const i32x4 *)(tt + em_tt_off(K, t, cb * 16u + nrow, ks * 64u + kq * 16u));
                    acc[t][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(am, bx[ks], acc[t][cb], 0, 0, 0);
                }
            }
        QM_MARK("epilogue: quantise, stage, rows back");
#pragma unroll
        for (uint32_t t = 0; t < 2; t++) {
            const QFmt dstf = t ? f_act : f_att;
            const uint32_t qk = (a.qkinds >> (4u * h + 2u * t)) & 3u;    // (workgroup-uniform)
#pragma unroll
            for (uint32_t cb = 0; cb < 4; cb++) {
                const i32x4 v = acc[t][cb];                              // v[r]: story row nrow, column 16 cb + 4 kq + r
                const s16x2 x01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u));
                const s16x2 x23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u));
                uint32_t out;
                if (qk != kQkGeneral) {
                    out = ew_to_bytes4(x01, x23, maxw, qk);
                } else {
                    const uint32_t b01 = ew_to_bytes(x01, maxw, fw, dstf, t == 0 && kmz), b23 = ew_to_bytes(x23, maxw, fw, dstf, t == 0 && kmz);
                    out = __builtin_amdgcn_perm(b23, b01, 0x06040200u);
                }
                *(uint32_t *)(stage + nrow * SP + cb * 16u + kq * 4u) = out;
            }
            // a row whose repeated word leaves the format: its sums term by term, Qw(Qw(count) . kw) over the row's distinct
            // words (the non-zero bytes of its X row), lane = column; replaces the row in the staging tile
            for (uint32_t m = fix_rows; m; m &= m - 1) {
                const uint32_t r = (uint32_t)__builtin_ctz(m);
                const uint32_t xr = lane < K / 4 ? X[em_x_off(r, 4u * lane) >> 2] : 0u;       // counts of words 4 lane .. 4 lane + 3
                int sum = 0;
                for (uint64_t nz = __ballot(xr != 0u); nz; nz &= nz - 1) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(nz);
                    const uint32_t xv = (uint32_t)__builtin_amdgcn_readlane((int)xr, (int)j);
#pragma unroll
                    for (uint32_t b = 0; b < 4; b++) {
                        const uint32_t c = (xv >> (8 * b)) & 0xFFu;
                        if (c) sum += qm_mul_code(count_code(c, fw.frac, maxw), (int)tt[em_tt_off(K, t, lane, 4 * j + b)], fw.frac, maxw);
                    }
                }
                __builtin_amdgcn_wave_barrier();                         // (every lane's dword of this row is written)
                stage[r * SP + lane] = (uint8_t)ew_to_bytes(s16x2{(short)sum, (short)0}, maxw, fw, dstf, t == 0 && kmz);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // whole rows, 16 bytes per lane: lane i holds piece i % 4 of row i / 4 (16 rows x 64 bytes = 64 lanes x 16 bytes)
            pend[t] = *(const i32x4 *)(stage + (lane / (Dp / 16u)) * SP + (lane % (Dp / 16u)) * 16u);
            __builtin_amdgcn_wave_barrier();                             // the second table reuses the staging tile
        }
        pend_row0 = row0;
        pending = true;
        __builtin_amdgcn_wave_barrier();                                 // the next tile rewrites X
        QM_MARK("end of tile");
    }
    flush_pending();
}

// ---------------------------------------------------------------------------
// The same for the joint-task dictionaries (129 .. 256 entries), EVERY HOP of a tile in one workgroup.  With one workgroup per
// hop (the kernel above) a tile's count matrix X is rebuilt for every hop -- 135 of its 349 vector instructions per tile and hop
// (tools/stage_budget.py; the kernel's vector and matrix issue together keep a SIMD 75 % busy, profiles/r05_units_j20_*) -- because
// T^T of three hops (104 KB) and sixteen wavefronts' whole X tiles (70 KB) do not fit one CU's LDS together.  They do once X is
// built in four CHUNKS of 64 dictionary entries: a chunk is 16 rows x 80 bytes, becomes the wavefront's B fragment of that K
// step at once (4 registers), and its LDS bytes are reused by the next chunk and, after the products, by the staging tile.
// Per tile: 4 short build rounds (zero 1.3 KB, byte-add the words of the chunk, set the time entry, read the fragment) for all
// hops together instead of a 4.3 KB build per hop; then per hop 32 MFMAs, the epilogue and the stores as above.  Hops before
// the last store their rows at once (younger than the next tile's word request, so the wait for those words does not wait for
// them); the last hop's rows wait in registers for the start of the next tile, as above.
// Repeated words: noticed by the byte-add as above, checked per chunk against every hop's tables; a row whose repeated word
// leaves a hop's weight format is summed term by term from the tile's word list (kept in LDS for that, 512 bytes).
// ---------------------------------------------------------------------------
template <int NW, int KS_, int WPS>                     // wavefronts per workgroup; K / 64 (chunks); wavefronts per SIMD compiled for
__global__ void __launch_bounds__(NW * kWave, WPS)
k_embed_story_mfma_hops(const EmbedIdxArgs a)
{
    constexpr uint32_t kBlockEm = NW * kWave;
    constexpr uint32_t KS = (uint32_t)KS_, K = 64u * KS, Dp = 64u;
    constexpr uint32_t XC = kEmRows * 64u;              // an X chunk: [4 pieces][16 rows][16 bytes] (em_x_off on k & 63)
    constexpr uint32_t SP = Dp + 16u;                   // row pitch of the staging tile, which follows the two chunks in their bytes
    constexpr uint32_t kWaveLds = 2u * XC + kEmDupCap * 4u + 16u + kEmRows * 16u * 2u;
    static_assert(2u * XC >= kEmRows * SP, "the staging tile lies over the two X chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    const uint32_t H = a.n_hop, V = a.V, nw = a.max_words;
    int8_t *tt = (int8_t *)smem;                                         // [H][2][Dp][K]: A then C of every hop, transposed (em_tt_off)
    uint8_t *ws = smem + (size_t)H * 2u * Dp * K + wave * kWaveLds;
    uint8_t *xs = ws;                                                    // two X chunks in turn (one is zeroed while the other is filled); later one table's output rows [16][SP]
    uint32_t *dup = (uint32_t *)(ws + 2u * XC);                          // [kEmDupCap] row << 16 | word
    uint32_t *n_dup = dup + kEmDupCap;
    uint16_t *wl = (uint16_t *)(n_dup + 4);                              // [16][16] the tile's words (the term-by-term path reads them)

    for (uint32_t t = 0; t < 2u * H; t++) {
        const uint32_t *src = (const uint32_t *)((t & 1u) ? a.t_c[t >> 1] : a.t_a[t >> 1]);      // [V][Dp] two's complement
        int8_t *dst = tt + (size_t)(t >> 1) * 2u * Dp * K;
        for (uint32_t i = tid; i < K * (Dp / 4); i += kBlockEm) {
            const uint32_t k = i / (Dp / 4), c4 = i % (Dp / 4);
            const uint32_t x = k < V ? src[(size_t)k * (Dp / 4) + c4] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) dst[em_tt_off(K, t & 1u, 4 * c4 + j, k)] = (int8_t)(x >> (8 * j));
        }
    }
    __syncthreads();
    QM_CLK_DECL();
    QM_CLK(0);                                                           // 0: T^T staging

    const uint32_t r4 = lane >> 2, qd = lane & 3u;                       // word phase: row r4 of the tile, slots 4 qd .. 4 qd + 3
    const uint32_t nrow = lane & 15u, kq = lane >> 4;                    // matrix phase: story row nrow, K bytes 16 kq .. / columns 4 kq ..
    const size_t n_tiles = ((size_t)a.rows + kEmRows - 1) / kEmRows;
    const uint32_t slot0 = 4 * qd < nw ? 4 * qd : 0u;
    uint2 raw_next = {0u, 0u};
    auto request_words = [&](size_t tile_) {
        size_t row = tile_ * kEmRows + r4;
        row = row < a.rows ? row : a.rows - 1;
        raw_next = *(const uint2 *)(a.words + row * nw + slot0);
    };
    auto take_words = [&](size_t tile_, uint32_t (&w_)[4]) {
        const bool in = tile_ * kEmRows + r4 < a.rows;
        w_[0] = raw_next.x & 0xFFFFu; w_[1] = raw_next.x >> 16; w_[2] = raw_next.y & 0xFFFFu; w_[3] = raw_next.y >> 16;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) w_[i] = (in && 4 * qd + i < nw) ? w_[i] : 0xFFFFu;
    };
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const size_t tile_step = (size_t)gridDim.x * NW;
    request_words((size_t)blockIdx.x * NW + wave);
    // the last hop's rows of the previous tile (keeping the last TWO hops' rows back, so that the youngest stores in flight at the
    // wait for a tile's words are two thirds of a tile old, measured 3 % SLOWER on the joint forwards)
    i32x4 pend[2];
    size_t pend_row0 = 0;
    bool pending = false;
    const uint32_t srow = lane / (Dp / 16u), spiece = lane % (Dp / 16u);  // whole rows out: lane i holds piece i % 4 of row i / 4
    // Rows leave through raw BUFFER stores on a resource that spans exactly the tile's rows inside the plane: a row past the
    // last one (the last tile only) is dropped by the bounds check, so the stores stand under no branch.  With a lane predicate
    // around them the compiler cannot count the stores in flight, and the wait for the NEXT tile's words -- a load older than
    // these stores -- became `s_waitcnt vmcnt(0)`: every tile waited for its predecessor's stores to reach memory (15 % of a
    // wavefront's time, tools/stage_clocks.py).
    auto store_rows = [&](uint32_t h, size_t row0_, const i32x4 (&rows_)[2]) {
        const size_t left = (size_t)a.rows - row0_;                      // (row0_ < rows for every tile that runs)
        const int bytes = (int)((left < kEmRows ? left : (size_t)kEmRows) * Dp);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(a.keys + (size_t)h * a.hop_stride + row0_ * Dp), 0, bytes, kRawBufferFlags);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(a.vals + (size_t)h * a.hop_stride + row0_ * Dp), 0, bytes, kRawBufferFlags);
        __builtin_amdgcn_raw_buffer_store_b128(rows_[0], rk, (int)(lane * 16u), 0, kBufferNt);
        __builtin_amdgcn_raw_buffer_store_b128(rows_[1], rv, (int)(lane * 16u), 0, kBufferNt);
    };
    for (size_t tile = (size_t)blockIdx.x * NW + wave; tile < n_tiles; tile += tile_step) {
        const size_t row0 = tile * kEmRows;
        QM_MARK("words of the tile");
        QM_CLK(1);                                                       // 1: loop overhead / end of the previous tile
        uint32_t w[4];
        take_words(tile, w);
        // (the words are TAKEN before anything else is issued: hoisted above them, the next tile's request and the pending stores
        // stood between the old load and its wait, and the compiler -- which cannot count the stores of the hop loop behind it --
        // waited `vmcnt(1)`: for the request it had just issued.  15 % of a wavefront's time, tools/stage_clocks.py)
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));    // (materialised HERE: the wait for the old load sits in front of the new request)
        __builtin_amdgcn_sched_barrier(0);
        request_words(tile + tile_step);
        if (pending) { store_rows(H - 1u, pend_row0, pend); pending = false; }
        *(uint2 *)(wl + r4 * 16u + 4u * qd) = uint2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
        if (lane == 0) *n_dup = 0u;
        uint32_t last = 0;                                               // 1 + this row's last non-empty slot
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) last = w[i] != 0xFFFFu ? 4 * qd + i + 1 : last;
        {   // maximum over the 4 lanes of the row
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
            last = o > last ? o : last;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last, 0x4E, 0xF, 0xF, true);                // quad_perm [2,3,0,1]
            last = o > last ? o : last;
        }
        // The time entry (the row's last non-empty slot, if its word is in range) SETS its byte to 1 (sample.c:474), the other
        // slots count.  Here the time slot simply adds its 1 and every other slot of the row that holds the same word is dropped:
        // the same byte, without a separate store (and its wait) behind the adds.
        uint32_t time_w = 0xFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (w[i] >= V) { w[i] = 0xFFFFu; continue; }                 // empty or out of range: ignored
            if (a.time_last && 4 * qd + i + 1 == last) time_w = w[i];
        }
        {   // the row's time word to its four lanes (one of them holds it, the others 0xFFFF: a minimum)
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)time_w, 0xB1, 0xF, 0xF, true);
            uint32_t tw = o < time_w ? o : time_w;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tw, 0x4E, 0xF, 0xF, true);
            tw = o < tw ? o : tw;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (w[i] == tw && !(a.time_last && 4 * qd + i + 1 == last)) w[i] = 0xFFFFu;
        }
        QM_CLK(2);                                                       // 2: words taken (waits for the prefetched load), pending stores, time word
        // ---- X in four chunks of 64 dictionary entries: each becomes the B fragment of its K step ------------------------
        i32x4 bx[KS];
        uint32_t fix_rows[QMANN_MAX_HOP] = {};                           // per hop: rows to be summed term by term (wavefront-uniform)
        uint32_t nd_seen = 0;
        bool repeats = false;                                            // some lane's byte add met a count of 1 (wavefront-uniform once balloted)
#pragma unroll
        for (uint32_t ks = 0; ks < KS; ks++) {
            QM_MARK("X chunk: zero the next, byte adds, fragment");
            uint8_t *xc = xs + (ks & 1u) * XC, *xn = xs + ((ks + 1u) & 1u) * XC;
            if (ks == 0) {                                               // (the later chunks were zeroed a round ahead)
                *(i32x4 *)(xc + lane * 16u) = i32x4{0, 0, 0, 0};
                wsync();
            }
            if (ks + 1u < KS) *(i32x4 *)(xn + lane * 16u) = i32x4{0, 0, 0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                if (w[i] == 0xFFFFu || (w[i] >> 6) != ks) continue;
                const uint32_t sh = 8u * (w[i] & 3u);
                const uint32_t old = atomicAdd((uint32_t *)(xc + (em_x_off(r4, w[i] & 63u) & ~3u)), 1u << sh);
                if (((old >> sh) & 0xFFu) == 1u) {                       // the second occurrence announces the repeat, once
                    const uint32_t n = atomicAdd(n_dup, 1u);
                    if (n < kEmDupCap) dup[n] = (r4 << 16) | w[i];
                    repeats = true;
                }
            }
            QM_CLK(3);                                                   // 3: chunk zero + byte adds issued
            wsync();
            QM_CLK(4);                                                   // 4: wait for the adds
            bx[ks] = *(const i32x4 *)(xc + em_x_off(nrow, kq * 16u));
            // repeated words of this chunk: does any product Qw(Qw(count) . kw) differ from count . kw in some hop?  (only when the
            // product leaves the format: rare; lane = column)
            // (the list's length is read only when some lane announced a repeat: otherwise an LDS round trip per chunk for nothing)
            uint32_t nd = nd_seen;
            if (__any(repeats)) { nd = *n_dup < kEmDupCap ? *n_dup : kEmDupCap; }
            for (uint32_t d = nd_seen; d < nd; d++) {
                const uint32_t e = dup[d], r = e >> 16, wd = e & 0xFFFFu;
                const int c = (int)xc[em_x_off(r, wd & 63u)];
                for (uint32_t h = 0; h < H; h++) {
                    const QFmt fw = a.w[h];
                    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
                    const int cc = count_code((uint32_t)c, fw.frac, maxw);
                    const int8_t *th = tt + (size_t)h * 2u * Dp * K;
                    const int ka = (int)th[em_tt_off(K, 0, lane, wd)], kc = (int)th[em_tt_off(K, 1, lane, wd)];
                    if (__any(qm_mul_code(cc, ka, fw.frac, maxw) != c * ka || qm_mul_code(cc, kc, fw.frac, maxw) != c * kc)) fix_rows[h] |= 1u << r;
                }
            }
            nd_seen = nd;
            QM_CLK(5);                                                   // 5: fragment read + repeated-word check
        }
        wsync();                                                         // the staging tile rewrites chunk bytes
        QM_CLK(6);
        // ---- per hop: the products, the epilogue, the stores ---------------------------------------------------------------
        for (uint32_t h = 0; h < H; h++) {
            QM_MARK("products (MFMA)");
            const int8_t *th = tt + (size_t)h * 2u * Dp * K;
            const QFmt fw = a.w[h], f_att = a.att[h], f_act = a.act[h];
            const bool kmz = (a.key_mz >> h) & 1u;
            const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
            i32x4 acc[2][4];
#pragma unroll
            for (uint32_t t = 0; t < 2; t++)
#pragma unroll
                for (uint32_t cb = 0; cb < 4; cb++) {
                    acc[t][cb] = i32x4{0, 0, 0, 0};
#pragma unroll
                    for (uint32_t ks = 0; ks < KS; ks++) {
                        const i32x4 am = *(const i32x4 *)(th + em_tt_off(K, t, cb * 16u + nrow, ks * 64u + kq * 16u));
                        acc[t][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(am, bx[ks], acc[t][cb], 0, 0, 0);
                    }
                }
            QM_CLK(7);                                                   // 7: 32 fragment reads + MFMAs issued
            QM_MARK("epilogue: quantise, stage, rows back");
            i32x4 rows[2];
#pragma unroll
            for (uint32_t t = 0; t < 2; t++) {
                const QFmt dstf = t ? f_act : f_att;
                const uint32_t qk = (a.qkinds >> (4u * h + 2u * t)) & 3u;    // (workgroup-uniform)
#pragma unroll
                for (uint32_t cb = 0; cb < 4; cb++) {
                    const i32x4 v = acc[t][cb];                              // v[r]: story row nrow, column 16 cb + 4 kq + r
                    const s16x2 x01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u));
                    const s16x2 x23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u));
                    uint32_t out;
                    if (qk != kQkGeneral) {
                        out = ew_to_bytes4(x01, x23, maxw, qk);
                    } else {
                        const uint32_t b01 = ew_to_bytes(x01, maxw, fw, dstf, t == 0 && kmz), b23 = ew_to_bytes(x23, maxw, fw, dstf, t == 0 && kmz);
                        out = __builtin_amdgcn_perm(b23, b01, 0x06040200u);
                    }
                    *(uint32_t *)(xs + nrow * SP + cb * 16u + kq * 4u) = out;
                }
                // a row whose repeated word leaves the format: its sums term by term, Qw(Qw(count) . kw) over the row's distinct
                // words (from the tile's word list; the time slot sets its word's count to 1), lane = column
                for (uint32_t m = fix_rows[h]; m; m &= m - 1) {
                    const uint32_t r = (uint32_t)__builtin_ctz(m);
                    uint32_t last_r = 0;
                    for (uint32_t j = 0; j < nw; j++) last_r = wl[r * 16u + j] != 0xFFFFu ? j + 1u : last_r;
                    int sum = 0;
                    for (uint32_t j = 0; j < nw; j++) {
                        const uint32_t wj = wl[r * 16u + j];
                        if (wj >= V) continue;
                        bool first = true, timed = false;
                        uint32_t cnt = 0;
                        for (uint32_t k = 0; k < nw; k++) {
                            if (wl[r * 16u + k] != wj) continue;
                            first = first && k >= j;
                            if (a.time_last && k + 1u == last_r) timed = true; else cnt++;
                        }
                        if (!first) continue;
                        if (timed) cnt = 1;
                        sum += qm_mul_code(count_code(cnt, fw.frac, maxw), (int)th[em_tt_off(K, t, lane, wj)], fw.frac, maxw);
                    }
                    __builtin_amdgcn_wave_barrier();                     // (every lane's dword of this row is written)
                    xs[r * SP + lane] = (uint8_t)ew_to_bytes(s16x2{(short)sum, (short)0}, maxw, fw, dstf, t == 0 && kmz);
                }
                QM_CLK(8);                                               // 8: quantise + staging writes (waits for the MFMA results)
                wsync();
                rows[t] = *(const i32x4 *)(xs + srow * SP + spiece * 16u);
                __builtin_amdgcn_wave_barrier();                         // the second table reuses the staging tile
                QM_CLK(9);                                               // 9: rows read back
            }
            if (h + 1u < H) store_rows(h, row0, rows);
            else { pend[0] = rows[0]; pend[1] = rows[1]; pend_row0 = row0; pending = true; }
            QM_CLK(10);                                                  // 10: stores issued
        }
        __builtin_amdgcn_wave_barrier();                                 // the next tile rewrites the chunk bytes and the word list
        QM_MARK("end of tile");
    }
    if (pending) store_rows(H - 1u, pend_row0, pend);
    QM_CLK(11);
    QM_CLK_FLUSH();
}

// question: word entries only (no time entry, sample.c:557-565); u0[j] = Qw0(sum_k Qw0(Qw0(W[j][k]) . Qw0(c_k))),
// written as floats on the Q(w[0]) grid.  Same lane layout as the story kernel: 16 lanes per question.
template <bool TAB_LDS>
__global__ void __launch_bounds__(kBlock)
k_embed_query_idx(const uint16_t *__restrict__ words, const int8_t *__restrict__ t_q, float *__restrict__ u0,
                  uint32_t n_query, uint32_t max_words, uint32_t D, uint32_t Dp, uint32_t V, QFmt fw, uint32_t pe_dim_word)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), sub = lane & 15u, grp = lane >> 4;
    const uint32_t dw = Dp / 4;
    uint32_t *tab = (uint32_t *)smem;
    if (TAB_LDS) {
#pragma unroll 4                                                 // (eight loads in flight were the kernel's 33rd register)
        for (uint32_t i = tid; i < V * dw; i += kBlock) tab[i] = ((const uint32_t *)t_q)[i];
        __syncthreads();
    }
    const int maxw = (1 << (fw.iwl + fw.frac)) - 1;
    const bool vec_rows = (D & 3u) == 0u && ((uintptr_t)u0 & 15u) == 0u;
    // (a 32-bit query counter, 64-bit only where an address is formed: the kernel fits 32 registers, and its workgroups find room
    // on a CU beside the story embedding's -- model_host.hip runs the two side by side)
    const uint32_t per_pass = gridDim.x * kWaves * 4;
    for (uint32_t q0 = (blockIdx.x * kWaves + tid / kWave) * 4; q0 < n_query; q0 += per_pass) {
        const uint32_t q = q0 + grp;
        const bool q_ok = q < n_query;
        uint32_t w = 0xFFFFu;
        if (q_ok && sub < max_words) w = words[(size_t)q * max_words + sub];
        const bool valid = w != 0xFFFFu && w < V;
        const uint32_t me = w | (valid ? 1u << 16 : 0u);
        // slots past the last valid word of the wavefront's four questions add nothing: the loops below stop there (the word
        // arrays are 8 or 16 slots wide, a bAbI question has 3 .. 5 words)
        uint32_t n_used = 0;
        {
            const uint64_t vm = __ballot(valid);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const uint32_t m = (uint32_t)(vm >> (16 * g4)) & 0xFFFFu;
                const uint32_t n = m ? 32u - (uint32_t)__builtin_clz(m) : 0u;
                n_used = n > n_used ? n : n_used;
            }
        }
        uint32_t cnt = 0;
        bool dup = false, later = false;
        for (uint32_t j = 0; j < n_used; j++) {
            const uint32_t o = (uint32_t)__shfl((int)me, (int)j, 16);
            const bool same = (((o ^ me) & 0xFFFFu) == 0u) && ((o >> 16) & 1u);
            cnt += same ? 1u : 0u;
            dup |= same && j < sub;
            later |= same && j > sub;
        }
        uint32_t pack = (w & 0xFFFFu) | (cnt << 16) | ((valid && !dup) ? 1u << 24 : 0u);
        if (pe_dim_word) {
            // EN_PE (MemN2N/define.h:298): the bag-of-words entry of a question word is SET to the position weight
            // pe_w[word][slot] = 1 + 4 (word / dim_input - 0.5)(slot / dim_word - 0.5) (MemN2N.c:615, float quotients, the
            // rest in double, stored as float; sample.c:559-560), so the last occurrence of a word decides; the term of
            // the word is then Qw(Qw(W) . Qw(weight)).  The slot field carries the weight's code instead of a count.
            const float pw = (float)(1.0 + (4.0 * ((double)((float)w / (float)V) - 0.5)) * ((double)((float)sub / (float)pe_dim_word) - 0.5));
            const uint32_t kx = valid ? (uint32_t)qm_code(pw, fw.iwl, fw.frac) : 0u;
            pack = (w & 0xFFFFu) | (kx << 16) | ((valid && !later) ? 1u << 24 : 0u);
        }
        for (uint32_t c0 = 0; c0 < dw; c0 += 16) {
            const uint32_t c4 = c0 + sub;
            const bool col_ok = c4 < dw;
            int acc[4] = {0, 0, 0, 0};
            for (uint32_t e = 0; e < n_used; e++) {
                const uint32_t pe = (uint32_t)__shfl((int)pack, (int)e, 16);
                if (!((pe >> 24) & 1u) || !col_ok) continue;
                const uint32_t we = pe & 0xFFFFu, ce = (pe >> 16) & 0xFFu;
                const uint32_t t = TAB_LDS ? tab[we * dw + c4] : ((const uint32_t *)t_q)[(size_t)we * dw + c4];
                const int cc = pe_dim_word ? (int)ce : count_code(ce, fw.frac, maxw);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int kw = (int)(int8_t)(t >> (8 * k));
                    acc[k] += (!pe_dim_word && ce == 1u && (1 << fw.frac) <= maxw) ? kw : qm_mul_code(cc, kw, fw.frac, maxw);   // see k_embed_story_idx
                }
            }
            if (q_ok && col_ok) {
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int v = acc[k] > maxw ? maxw : (acc[k] < -maxw ? -maxw : acc[k]);
                    o[k] = qm_scale_down((float)v, fw.frac);
                }
                if (vec_rows) {                                          // rows of whole 16-byte groups (bAbI: D = 60): one store per lane
                    if (4u * c4 < D) *(float4 *)(u0 + (size_t)q * D + 4u * c4) = float4{o[0], o[1], o[2], o[3]};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (4u * c4 + (uint32_t)k < D) u0[(size_t)q * D + 4u * c4 + (uint32_t)k] = o[k];
                }
            }
        }
    }
}

// float [D][V] -> int8 two's-complement codes, transposed to [V][Dp] (padding columns zero)
__global__ void k_quantize_transpose(const float *__restrict__ src, int8_t *__restrict__ dst, uint32_t D, uint32_t V,
                                     uint32_t Dp, QFmt f)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V * Dp) return;
    const uint32_t v = i / Dp, c = i % Dp;
    dst[i] = (c < D) ? (int8_t)qm_code(src[(size_t)c * V + v], f.iwl, f.frac) : (int8_t)0;
}

// the way back: int8 codes [V][Dp] -> the grid values as a float matrix [D][V] (exact: code . 2^-frac)
__global__ void k_dequantize_transpose(const int8_t *__restrict__ src, float *__restrict__ dst, uint32_t D, uint32_t V,
                                       uint32_t Dp, uint32_t frac)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * V) return;
    const uint32_t c = i / V, v = i % V;
    dst[i] = qm_decode((int32_t)src[(size_t)v * Dp + c], frac);
}

inline bool fmt8(qmann_fmt f) { return f.iwl + f.frac >= 1 && f.iwl + f.frac <= 7; }

// formats of a hop's memory bytes from the net: values on act[h]; keys on att[h] -- except mode 3 with a finer weight grid,
// whose keys keep the weight grid (qfmt.h::ham_key_format) -- and the hops whose keys follow the minus-zero rule
template <typename Args>
static void fill_key_formats(Args &a, const qmann_net *net)
{
    a.key_mz = 0;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        const qmann_fmt src = h == 0 ? net->w[0] : net->act[h - 1];
        a.act[h] = QFmt{net->act[h].iwl, net->act[h].frac};
        a.w[h] = QFmt{net->w[h].iwl, net->w[h].frac};
        bool mz = false;
        a.att[h] = ham_key_format(net->attention_mode, QFmt{src.iwl, src.frac}, a.w[h], QFmt{net->att[h].iwl, net->att[h].frac}, &mz);
        if (mz) a.key_mz |= 1u << h;
    }
}
static void fill_qkinds(EmbedIdxArgs &a)
{
    a.qkinds = 0;
    for (uint32_t h = 0; h < a.n_hop; h++) {
        a.qkinds |= qkind_of(a.w[h], a.att[h], (a.key_mz >> h) & 1u) << (4u * h);
        a.qkinds |= qkind_of(a.w[h], a.act[h], false) << (4u * h + 2u);
    }
    if (qm_tuning().embed_general_epilogue) a.qkinds = 0xFFFFFFFFu;     // A/B: ew_to_bytes everywhere
}

}  // namespace

extern "C" {

static int answer_f32_impl(const qmann_net *net, const float *w_ans, const float *u, const uint32_t *answer, uint32_t *pred,
                           float *probs, float *cost, uint32_t *match, uint32_t n_query, void *stream, bool allow_fused)
{
    QmBatched qm_scope;
    if (!net || !w_ans || !u || !pred) return QMANN_EINVAL;
    const uint32_t D = net->dim_emb, V = net->dim_input;
    if (D == 0 || V == 0) return QMANN_EINVAL;
    const size_t lds = ((size_t)((D + 3) & ~3u) + V) * sizeof(float);
    if (lds > 128 * 1024) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    // bAbI shapes (D <= 64, V <= 256), the e^x and 2^x bases: the fused form on the bf16 matrix cores (k_answer_mfma), within
    // 1e-5 on the softmax; qmann_answer_exact_scope / QMANN_ANSWER_EXACT keep the serial-order kernels below
    if (allow_fused && D <= 64u && V <= 256u && net->softmax_base != QMANN_SOFTMAX_EXP_PLAN && !qm_tuning().answer_exact && qm_answer_exact_depth == 0) {
        const uint32_t tiles = (V + 15u) / 16u;
        hipStream_t st = (hipStream_t)stream;
        const uint32_t n_task = (n_query + 15u) / 16u, need = (n_task + kAmWaves - 1) / kAmWaves;
#define QM_ANS_MFMA(TT)                                                                                                          \
    do {                                                                                                                         \
        const size_t l_ = 3u * (size_t)(TT) * 16u * kAmPitch;                                                                    \
        if (l_ > 48 * 1024)                                                                                                      \
            QM_HIP(hipFuncSetAttribute((const void *)k_answer_mfma<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_));   \
        const uint32_t cap = qm_resident_groups(kAmWaves, 4, l_ + 256);                                                          \
        k_answer_mfma<TT><<<need < cap ? need : cap, kAmBlock, l_, st>>>(w_ans, u, answer, pred, probs, cost, match, D, V, net->softmax_base, n_query); \
    } while (0)
        if (tiles <= 2) QM_ANS_MFMA(2); else if (tiles <= 3) QM_ANS_MFMA(3); else if (tiles <= 4) QM_ANS_MFMA(4);
        else if (tiles <= 5) QM_ANS_MFMA(5); else if (tiles <= 8) QM_ANS_MFMA(8); else if (tiles <= 12) QM_ANS_MFMA(12);
        else if (tiles <= 15) QM_ANS_MFMA(15); else QM_ANS_MFMA(16);
#undef QM_ANS_MFMA
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    if (lds > 48 * 1024)
        QM_HIP(hipFuncSetAttribute((const void *)k_answer<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // lanes per query / logits per lane / queries per lane group.  Short dictionaries: 16 lanes (one DPP row), four
    // lane groups side by side in a wavefront (the reductions are most of the work there); the joint-task sizes: the whole
    // wavefront on four queries at once, every weight read from LDS used four times (that kernel is bound by LDS reads)
    // (two queries per lane group up to 96 logits: -6 % on this kernel at V = 80; with 8 logits per lane the second query's state spills)
    const uint32_t lpq = V <= 128 ? 16u : 64u, qb = V <= 96 ? 2u : (V <= 128 ? 1u : 4u);
    uint32_t vpt = (V + lpq - 1) / lpq;
    vpt = vpt <= 2 ? 2u : vpt <= 4 ? 4u : vpt <= 6 ? 6u : 8u;
    const uint32_t qpw = kWave / lpq * qb;
    const size_t lds_small = ((size_t)D * lpq * vpt + (size_t)kAnsWaves * qpw * D) * sizeof(float);
    if (V <= 256u && lds_small <= 78 * 1024) {              // W^T fits LDS twice per CU
        const uint32_t need = (n_query + kAnsWaves * qpw - 1) / (kAnsWaves * qpw);
        const uint32_t cap = qm_resident_groups(kAnsWaves, 8, lds_small);      // persistent: two 16-wavefront workgroups per CU
        const uint32_t blocks = need < cap ? need : cap;
        hipStream_t st = (hipStream_t)stream;
#define QM_ANS_SMALL(L, N, Q)                                                                                                    \
    do {                                                                                                                         \
        if (lds_small > 48 * 1024)                                                                                               \
            QM_HIP(hipFuncSetAttribute((const void *)k_answer_small<L, N, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small)); \
        k_answer_small<L, N, Q><<<blocks, kAnsBlock, lds_small, st>>>(w_ans, u, answer, pred, probs, cost, match, D, V, net->softmax_base, n_query); \
    } while (0)
        if (lpq == 64) {
            QM_ANS_SMALL(64, 4, 4);                         // V in 129 .. 256
        } else {
            switch (vpt) {
            case 2: QM_ANS_SMALL(16, 2, 2); break;
            case 4: QM_ANS_SMALL(16, 4, 2); break;
            case 6: QM_ANS_SMALL(16, 6, 2); break;
            default: QM_ANS_SMALL(16, 8, 1); break;
            }
        }
#undef QM_ANS_SMALL
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    k_answer<false><<<n_query < 8192u ? n_query : 8192u, kBlock, lds, (hipStream_t)stream>>>(
        w_ans, u, answer, pred, probs, cost, match, D, V, net->softmax_base, n_query);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

#ifdef QM_STAGE_CLOCKS
// debug builds only (tools/stage_clocks.py): the per-stage shader-cycle sums of the instrumented kernels of this file, then cleared
int qmann_debug_stage_clocks(unsigned long long *out, int n)
{
    unsigned long long h[kQmClkStages] = {};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(qm_stage_clk), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < n && i < kQmClkStages; i++) out[i] = h[i];
    unsigned long long z[kQmClkStages] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(qm_stage_clk), z, sizeof z) != hipSuccess) return -1;
    return kQmClkStages;
}
#endif

int qmann_answer_f32(const qmann_net *net, const float *w_ans, const float *u, const uint32_t *answer,
                     uint32_t *pred, float *probs, float *cost, uint32_t *match, uint32_t n_query, void *stream)
{
    return answer_f32_impl(net, w_ans, u, answer, pred, probs, cost, match, n_query, stream, true);
}

int qmann_answer_f32_serial(const qmann_net *net, const float *w_ans, const float *u, const uint32_t *answer,
                            uint32_t *pred, float *probs, float *cost, uint32_t *match, uint32_t n_query, void *stream)
{
    return answer_f32_impl(net, w_ans, u, answer, pred, probs, cost, match, n_query, stream, false);
}

int qmann_answer_i8(const qmann_net *net, const int8_t *w_ans_i8, qmann_fmt w_fmt, const float *u, float *logits_ws,
                    const uint32_t *answer, uint32_t *pred, float *probs, float *cost, uint32_t *match,
                    uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    if (!net || !w_ans_i8 || !u || !logits_ws || !pred) return QMANN_EINVAL;
    const uint32_t D = net->dim_emb, Dp = net->dim_emb_pad, V = net->dim_input;
    if (D == 0 || V == 0 || Dp % 64 != 0 || D > Dp) return QMANN_EINVAL;
    const qmann_fmt fu = net->act[net->n_hop - 1];           // u is the last sum_vec output
    if (!fmt8(fu) || !fmt8(w_fmt)) return QMANN_ERANGE;
    const size_t lds = ((size_t)((D + 3) & ~3u) + V) * sizeof(float);
    if (lds > 128 * 1024) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    const float scale = 1.0f / (float)(1u << (fu.frac + w_fmt.frac));
    hipStream_t st = (hipStream_t)stream;
    const QFmt fuq{fu.iwl, fu.frac};
    // (the running-maximum normaliser needs a true exponential: the piece-wise linear exp_plan takes the two-pass form)
    if (!probs && net->softmax_base != QMANN_SOFTMAX_EXP_PLAN && !qm_tuning().answer_two_pass) {
        // one pass: no logits round trip; the workspace holds the per-slice records
        const uint32_t qblocks = (n_query + 16 * kWaves - 1) / (16 * kWaves), n_tiles = (V + kAnsTile - 1) / kAnsTile;
        // ~2 workgroups per CU (measured at 8 192 x 4 096 x 256: 128 workgroups 86 us, 256: 55, 512: 45, 1 024: 46)
        uint32_t n_slice = qblocks >= 512u ? 1u : (512u + qblocks - 1) / qblocks;
        if (n_slice > n_tiles) n_slice = n_tiles;
        while (n_slice > 1 && (size_t)n_slice * sizeof(AnsPart) > (size_t)V * sizeof(float)) n_slice--;   // records must fit logits_ws
        const uint32_t tps = (n_tiles + n_slice - 1) / n_slice;
        n_slice = (n_tiles + tps - 1) / tps;
        AnsPart *part = (AnsPart *)logits_ws;
        const dim3 grid1(qblocks, n_slice);
        const bool eb = net->softmax_base == QMANN_SOFTMAX_EXP;
#define QM_ANS_PART(KS)                                                                                                              \
    do {                                                                                                                             \
        if (eb) k_answer_i8_part<KS, true><<<grid1, kBlock, 0, st>>>(u, w_ans_i8, part, n_query, D, V, fuq, scale, net->softmax_base, tps);  \
        else k_answer_i8_part<KS, false><<<grid1, kBlock, 0, st>>>(u, w_ans_i8, part, n_query, D, V, fuq, scale, net->softmax_base, tps);   \
    } while (0)
        if (Dp == 64) QM_ANS_PART(1);
        else if (Dp == 128) QM_ANS_PART(2);
        else if (Dp == 256) QM_ANS_PART(4);
        else return QMANN_EUNSUPPORTED;
#undef QM_ANS_PART
        const uint32_t cb = (n_query * 16 + kBlock - 1) / kBlock;
        k_answer_i8_combine<<<cb < 2048u ? cb : 2048u, kBlock, 0, st>>>(part, u, w_ans_i8, answer, pred, cost, match, n_query, n_slice, D, Dp, V,
                                                                        fuq, scale, net->softmax_base);
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    const dim3 grid((n_query + 16 * kWaves - 1) / (16 * kWaves), (V + 16 * kTilesPerBlockY - 1) / (16 * kTilesPerBlockY));
    if (Dp == 64) k_logits_mfma_i8<1><<<grid, kBlock, 0, st>>>(u, w_ans_i8, logits_ws, n_query, D, V, fuq, scale);
    else if (Dp == 128) k_logits_mfma_i8<2><<<grid, kBlock, 0, st>>>(u, w_ans_i8, logits_ws, n_query, D, V, fuq, scale);
    else if (Dp == 256) k_logits_mfma_i8<4><<<grid, kBlock, 0, st>>>(u, w_ans_i8, logits_ws, n_query, D, V, fuq, scale);
    else return QMANN_EUNSUPPORTED;
    if (lds > 48 * 1024)
        QM_HIP(hipFuncSetAttribute((const void *)k_answer<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_answer<true><<<n_query < 8192u ? n_query : 8192u, kBlock, lds, (hipStream_t)stream>>>(
        logits_ws, nullptr, answer, pred, probs, cost, match, D, V, net->softmax_base, n_query);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

static int embed_story_impl(const qmann_net *net, const float *story, uint32_t rows_total, const uint32_t *row_list,
                            const uint32_t *n_list, const float *const *w_a, const float *const *w_c, int8_t *keys, int8_t *vals,
                            size_t hop_stride, void *stream)
{
    QmBatched qm_scope;
    if (!net || (!story && rows_total) || !w_a || !w_c || !keys || !vals) return QMANN_EINVAL;   // (no rows: no story array needed)
    if ((row_list == nullptr) != (n_list == nullptr)) return QMANN_EINVAL;
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (hop_stride < (size_t)rows_total * net->dim_emb_pad) return QMANN_EINVAL;
    EmbedArgs a{};
    a.story = story; a.keys = keys; a.vals = vals; a.hop_stride = hop_stride;
    a.n_hop = net->n_hop; a.D = net->dim_emb; a.Dp = net->dim_emb_pad; a.V = net->dim_input; a.rows = rows_total;
    a.row_list = row_list; a.n_list = n_list;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!w_a[h] || !w_c[h]) return QMANN_EINVAL;
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        a.w_a[h] = w_a[h]; a.w_c[h] = w_c[h];
    }
    fill_key_formats(a, net);
    if (rows_total == 0) return QMANN_OK;
    // a listed launch does not know the list's length on the host: a fixed grid walks it (and leaves at once if it is empty)
    const uint32_t grid = row_list ? (rows_total < 4096u ? rows_total : 4096u) : (rows_total < (1u << 22) ? rows_total : (1u << 22));
    k_embed_story<<<grid, kBlock, 0, (hipStream_t)stream>>>(a);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_embed_story(const qmann_net *net, const float *story, uint32_t rows_total, const float *const *w_a,
                      const float *const *w_c, int8_t *keys, int8_t *vals, size_t hop_stride, void *stream)
{
    return embed_story_impl(net, story, rows_total, nullptr, nullptr, w_a, w_c, keys, vals, hop_stride, stream);
}

int qmann_embed_story_rows(const qmann_net *net, const float *story, uint32_t rows_total, const uint32_t *row_list,
                           const uint32_t *n_list, const float *const *w_a, const float *const *w_c, int8_t *keys,
                           int8_t *vals, size_t hop_stride, void *stream)
{
    if (!row_list || !n_list) return QMANN_EINVAL;
    return embed_story_impl(net, story, rows_total, row_list, n_list, w_a, w_c, keys, vals, hop_stride, stream);
}

static int embed_query_impl(const qmann_net *net, const float *question, const uint32_t *row_list, const uint32_t *n_list,
                            const float *w_q, float *u0, uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    if (!net || !question || !w_q || !u0) return QMANN_EINVAL;
    if ((row_list == nullptr) != (n_list == nullptr)) return QMANN_EINVAL;
    if (!fmt8(net->w[0])) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (n_query >= (1u << 24)) return QMANN_ERANGE;      // one workgroup per query: a launch holds < 2^32 threads
    const uint32_t grid = row_list ? (n_query < 4096u ? n_query : 4096u) : n_query;
    k_embed_query<<<grid, kBlock, 0, (hipStream_t)stream>>>(question, w_q, u0, net->dim_emb, net->dim_input,
                                                           QFmt{net->w[0].iwl, net->w[0].frac}, n_query, row_list, n_list);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_embed_query(const qmann_net *net, const float *question, const float *w_q, float *u0, uint32_t n_query,
                      void *stream)
{
    return embed_query_impl(net, question, nullptr, nullptr, w_q, u0, n_query, stream);
}

int qmann_embed_query_rows(const qmann_net *net, const float *question, const uint32_t *row_list, const uint32_t *n_list,
                           const float *w_q, float *u0, uint32_t n_query, void *stream)
{
    if (!row_list || !n_list) return QMANN_EINVAL;
    return embed_query_impl(net, question, row_list, n_list, w_q, u0, n_query, stream);
}

int qmann_bow_to_words(const float *bow, uint32_t rows, uint32_t dim_input, uint16_t *words, uint32_t *irregular_rows,
                       uint32_t *n_irregular, void *stream)
{
    QmBatched qm_scope;
    if ((!bow && rows) || !words || !irregular_rows || !n_irregular) return QMANN_EINVAL;
    if (dim_input == 0 || dim_input >= 0xFFFFu) return QMANN_ERANGE;          // 0xFFFF marks an unused slot
    if (rows == 0) return QMANN_OK;
    auto go = [&](auto kernel, uint32_t rows_per_wave) {
        const uint32_t per_block = kWaves * rows_per_wave, need = (rows + per_block - 1) / per_block;
        kernel<<<need < 16384u ? need : 16384u, kBlock, 0, (hipStream_t)stream>>>(bow, rows, dim_input, words, irregular_rows, n_irregular);
    };
    if (dim_input <= 16u) go(k_bow_to_words<16>, 4u);
    else if (dim_input <= 32u) go(k_bow_to_words<32>, 2u);
    else if (dim_input <= 64u) go(k_bow_to_words<64>, 1u);
    else go(k_bow_to_words_wide, 1u);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_quantize_table_i8(const float *w, int8_t *table, uint32_t dim_emb, uint32_t dim_emb_pad, uint32_t dim_input,
                            qmann_fmt fmt, void *stream)
{
    QmBatched qm_scope;
    if (!w || !table || dim_emb > dim_emb_pad) return QMANN_EINVAL;
    if (!fmt8(fmt)) return QMANN_ERANGE;
    const uint32_t n = dim_input * dim_emb_pad;
    if (n == 0) return QMANN_OK;
    k_quantize_transpose<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(w, table, dim_emb, dim_input, dim_emb_pad,
                                                                         QFmt{fmt.iwl, fmt.frac});
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_dequantize_table_f32(const int8_t *table, float *w, uint32_t dim_emb, uint32_t dim_emb_pad, uint32_t dim_input,
                               qmann_fmt fmt, void *stream)
{
    QmBatched qm_scope;
    if (!w || !table || dim_emb > dim_emb_pad) return QMANN_EINVAL;
    if (!fmt8(fmt)) return QMANN_ERANGE;
    const uint32_t n = dim_input * dim_emb;
    if (n == 0) return QMANN_OK;
    k_dequantize_transpose<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(table, w, dim_emb, dim_input, dim_emb_pad, fmt.frac);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_embed_story_idx(const qmann_net *net, const uint16_t *words, uint32_t rows_total, uint32_t max_words,
                          int time_last, const int8_t *const *t_a, const int8_t *const *t_c, int8_t *keys, int8_t *vals,
                          size_t hop_stride, void *stream)
{
    QmBatched qm_scope;
    if (!net || (!words && rows_total) || !t_a || !t_c || !keys || !vals) return QMANN_EINVAL;   // (no rows: no word array needed)
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (max_words == 0 || max_words > (uint32_t)kMaxWords) return QMANN_ERANGE;
    if (hop_stride < (size_t)rows_total * net->dim_emb_pad) return QMANN_EINVAL;
    EmbedIdxArgs a{};
    a.words = words; a.keys = keys; a.vals = vals; a.hop_stride = hop_stride;
    a.n_hop = net->n_hop; a.D = net->dim_emb; a.Dp = net->dim_emb_pad; a.V = net->dim_input; a.rows = rows_total;
    a.max_words = max_words; a.time_last = time_last ? 1u : 0u;
    for (uint32_t h = 0; h < net->n_hop; h++) {
        if (!t_a[h] || !t_c[h]) return QMANN_EINVAL;
        if (!fmt8(net->act[h]) || !fmt8(net->w[h]) || !fmt8(net->att[h])) return QMANN_ERANGE;
        a.t_a[h] = t_a[h]; a.t_c[h] = t_c[h];
    }
    fill_key_formats(a, net);
    fill_qkinds(a);
    if (rows_total == 0) return QMANN_OK;
    if ((net->dim_emb_pad & 3u) || (hop_stride & 3u) || ((uintptr_t)keys & 3u) || ((uintptr_t)vals & 3u)) return QMANN_EINVAL;
    if (net->dim_emb_pad % 64 != 0) return QMANN_EUNSUPPORTED;          // 16 lanes x 4 columns per round
    // dictionaries of up to 256 entries with 1.0 representable in every weight format: the matrix-core kernel
    bool mfma_ok = net->dim_input <= 256 && net->dim_emb_pad == 64 && (max_words & 3u) == 0u && ((uintptr_t)words & 7u) == 0u &&
                   !qm_tuning().embed_valu;
    for (uint32_t h = 0; h < net->n_hop; h++) mfma_ok = mfma_ok && net->w[h].iwl >= 1;
    if (mfma_ok) {
        const uint32_t K = net->dim_input <= 64 ? 64u : (net->dim_input <= 128 ? 128u : 256u), Dp = net->dim_emb_pad;
        hipStream_t st = (hipStream_t)stream;
        const size_t tiles = ((size_t)rows_total + kEmRows - 1) / kEmRows;
        // joint-task dictionaries, several hops: every hop of a tile in one workgroup, X built once per tile in chunks
        // (k_embed_story_mfma_hops); QMANN_EMBED_PER_HOP keeps a workgroup per hop (A/B)
        // (64-entry dictionaries -- task 1 -- gain nothing from it: 444 against 443 M q/s in an interleaved A/B with the K = 64
        // instantiation at six wavefronts per SIMD; they keep a workgroup per hop)
        if (K == 256u && net->n_hop >= 2u && net->n_hop <= 3u && !qm_tuning().embed_per_hop) {
            constexpr uint32_t nwh = 16u;
            const size_t lds = (size_t)net->n_hop * 2u * Dp * K + (size_t)nwh * (2u * kEmRows * 64u + kEmDupCap * 4u + 16u + kEmRows * 32u);
            const uint32_t resident = qm_resident_groups(nwh, 4u, lds);
            const uint32_t nx = (uint32_t)((tiles + nwh - 1) / nwh < resident ? (tiles + nwh - 1) / nwh : resident);
            QM_HIP(hipFuncSetAttribute((const void *)k_embed_story_mfma_hops<16, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_embed_story_mfma_hops<16, 4, 4><<<nx, nwh * kWave, lds, st>>>(a);
            QM_LAUNCH_CHECK();
            return qm_scope.rc();
        }
        // T^T (2 x 64 x K bytes) is per workgroup: large dictionaries share it among 16 wavefronts
        const uint32_t nwv = K == 64 ? 8u : 16u;
        const size_t lds = 2u * (size_t)Dp * K + (size_t)nwv * (kEmRows * K + kEmRows * (Dp + 16u) + kEmDupCap * 4u + 16u);
        // persistent in x: never more workgroups than are resident at once (rt.h: through round 3 the cap was LDS-only and
        // rounded UP -- 86 x 3 = 258 workgroups for 256 one-per-CU slots on the joint dictionaries, 1 026 for 512 on task 1)
        const uint32_t resident = qm_resident_groups(nwv, (uint32_t)em_waves_per_simd(K == 64 ? 1 : (K == 128 ? 2 : 4)), lds);
        const uint32_t cap = resident / net->n_hop ? resident / net->n_hop : 1u;
        const uint32_t nx = (uint32_t)((tiles + nwv - 1) / nwv < cap ? (tiles + nwv - 1) / nwv : cap);
        const dim3 grid(nx, net->n_hop);
#define QM_EM_GO(KS_, NW_)                                                                                                       \
        do {                                                                                                                     \
            if (lds > 48 * 1024) QM_HIP(hipFuncSetAttribute((const void *)k_embed_story_mfma<KS_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            k_embed_story_mfma<KS_, NW_><<<grid, NW_ * kWave, lds, st>>>(a);                                                      \
        } while (0)
        if (K == 64) QM_EM_GO(1, 8); else if (K == 128) QM_EM_GO(2, 16); else QM_EM_GO(4, 16);
#undef QM_EM_GO
        QM_LAUNCH_CHECK();
        return qm_scope.rc();
    }
    const uint32_t need = (rows_total + kWaves * 4 - 1) / (kWaves * 4);
    // small dictionaries: all tables in LDS as int16; larger ones are gathered from L2 (for dictionaries whose tables
    // exceed LDS both alternatives to L2 gathers were measured on the 20-task joint set in round 1 and were slower:
    // staging one hop's tables at a time, and one copy per CU shared by a 1024-thread workgroup)
    const size_t tab_lds = (size_t)net->n_hop * 2 * (net->dim_input + 1u) * net->dim_emb_pad * 2u;
    const size_t wave_lds = (size_t)kWaves * kEwWaveBytes;
    if (tab_lds + wave_lds <= 64 * 1024) {
        const size_t lds = tab_lds + wave_lds;
        const uint32_t cap = qm_resident_groups(kWaves, 4, lds);          // (118 registers: four wavefronts per SIMD)
        if (lds > 48 * 1024) QM_HIP(hipFuncSetAttribute((const void *)k_embed_story_idx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_embed_story_idx<true><<<need < cap ? need : cap, kBlock, lds, (hipStream_t)stream>>>(a);
    } else {
        const uint32_t cap = qm_resident_groups(kWaves, 4, wave_lds);
        k_embed_story_idx<false><<<need < cap ? need : cap, kBlock, wave_lds, (hipStream_t)stream>>>(a);
    }
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

int qmann_embed_query_idx(const qmann_net *net, const uint16_t *words, uint32_t max_words, const int8_t *t_q, float *u0,
                          uint32_t n_query, void *stream)
{
    QmBatched qm_scope;
    if (!net || !words || !t_q || !u0) return QMANN_EINVAL;
    if (max_words == 0 || max_words > (uint32_t)kMaxWords) return QMANN_ERANGE;
    if (!fmt8(net->w[0])) return QMANN_ERANGE;
    if (n_query == 0) return QMANN_OK;
    if (net->dim_emb_pad & 3u) return QMANN_EINVAL;
    if (n_query > 0xFFF00000u) return QMANN_ERANGE;       // (the kernel's 32-bit query counter steps past n_query by up to a grid's worth)
    const uint32_t need = (n_query + kWaves * 4 - 1) / (kWaves * 4);
    const uint32_t blocks = need < 2048u ? need : 2048u;
    const size_t tab_lds = (size_t)net->dim_input * net->dim_emb_pad;
    const QFmt fw{net->w[0].iwl, net->w[0].frac};
    if (net->en_pe && net->pe_dim_word == 0) return QMANN_EINVAL;
    const uint32_t pe_dw = net->en_pe ? net->pe_dim_word : 0u;
    if (tab_lds <= 48 * 1024)
        k_embed_query_idx<true><<<blocks, kBlock, tab_lds, (hipStream_t)stream>>>(
            words, t_q, u0, n_query, max_words, net->dim_emb, net->dim_emb_pad, net->dim_input, fw, pe_dw);
    else
        k_embed_query_idx<false><<<blocks, kBlock, 0, (hipStream_t)stream>>>(
            words, t_q, u0, n_query, max_words, net->dim_emb, net->dim_emb_pad, net->dim_input, fw, pe_dw);
    QM_LAUNCH_CHECK();
    return qm_scope.rc();
}

}  // extern "C"
