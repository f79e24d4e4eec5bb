// abi_ops.hip -- boundary B, forward and utility verbs (include/qmann_abi.h).
//
// One-op-at-a-time entry points with the reference's float-on-a-grid calling
// convention: every tensor is a plain float* in HBM, Q(iwl.frac) formats come
// as scalars.  These are the compatibility face of the library -- the
// unmodified C host (MemN2N.c + layer.c) drives them one query at a time, so
// they are launch-latency bound by construction; the throughput path is the
// batched int8 API in batch_*.hip.  The kernels are written for wave64: one
// wavefront owns one output element (row kernels) or one 64-column stripe
// (column kernels) and reduces with cross-lane shuffles.  In fixed-point mode
// every partial product is an exact small multiple of 2^-frac, so the float
// accumulation is exact and independent of the reduction order -- results are
// bit-identical to the reference's serial sums.  In float mode the sum order
// differs from the reference's serial loop (documented tolerance 1e-5 rel.).
#include "qfmt.h"
#include "rt.h"
#include <vector>
#include "defer.h"
#include "../../include/qmann_abi.h"

#include <string.h>

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;

int g_softmax_base = 0;  // 0: e^x (CUDA path), 1: 2^x (CPU path)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// out[i] = Qo( sum_k Qa( Qa(A[(i / ncol) * K + k]) * Qb(B[(i % ncol) * K + k]) ) )
// covers the reference's _cuda_mat_vec_product (ncol = 1, B = the vector) and
// _cuda_mat_mat_trans_product (A.B^T); one wavefront per output element.
template <bool FIXED>
__global__ void __launch_bounds__(kBlock)
k_rows_dot(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ out,
           unsigned n_out, unsigned K, unsigned ncol, QFmt fa, QFmt fb, QFmt fo)
{
    const unsigned w = (blockIdx.x * kBlock + threadIdx.x) / kWave;
    const unsigned lane = threadIdx.x & (kWave - 1);
    if (w >= n_out) return;
    const float *a = A + (size_t)(w / ncol) * K;
    const float *b = B + (size_t)(w % ncol) * K;
    float acc = 0.0f;
    for (unsigned k = lane; k < K; k += kWave) {
        if (FIXED) acc += qm_fixed_mul(a[k], b[k], fa, fb);
        else acc += a[k] * b[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[w] = FIXED ? qm_quant(acc, fo.iwl, fo.frac) : acc;
}

// out[c] = Qo( sum_r Qm( Qm(p[r]) * Qm(M[r * C + c]) ) ): the weighted read-out
// (reference _cuda_mat_trans_mat_product).  A block owns 64 columns; its four
// wavefronts take every fourth row and meet in LDS.  Row-major M means the 64
// lanes of a wavefront read 256 contiguous bytes per row.
template <bool FIXED>
__global__ void __launch_bounds__(kBlock)
k_cols_dot(const float *__restrict__ p, const float *__restrict__ M, float *__restrict__ out,
           unsigned R, unsigned C, QFmt fm, QFmt fo)
{
    __shared__ float part[kBlock / kWave][kWave];
    const unsigned lane = threadIdx.x & (kWave - 1);
    const unsigned wv = threadIdx.x / kWave;
    const unsigned c = blockIdx.x * kWave + lane;
    float acc = 0.0f;
    if (c < C) {
        for (unsigned r = wv; r < R; r += kBlock / kWave) {
            const float pr = p[r];
            const float m = M[(size_t)r * C + c];
            if (FIXED) acc += qm_fixed_mul(pr, m, fm, fm);
            else acc += pr * m;
        }
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && c < C) {
        float s = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        out[c] = FIXED ? qm_quant(s, fo.iwl, fo.frac) : s;
    }
}

// Hamming-style "approximate" attention, reference _cuda_approximate_attention
// (lib/layer_cuda.cu:355-541): operands re-encoded left-aligned (frac = 31-iwl);
// the shared magnitude is removed (same sign) or moved onto the larger operand
// (opposite sign -- the 32-bit add may carry into the sign position); bits
// 1..num_bit-1 agree with weight 2^-i; the sign of the term is the agreement of
// the (possibly carried-into) sign bits; terms are scaled by 2^-3, quantised,
// summed over the row and quantised again.  One wavefront per memory row.
__device__ __forceinline__ float appx_pair(float a, float b, unsigned iwl, unsigned num_bit)
{
    const unsigned frac = 31u - iwl;
    uint32_t fa = qm_signmag(a, iwl, frac);
    uint32_t fb = qm_signmag(b, iwl, frac);
    const uint32_t sa = fa & 0x80000000u, sb = fb & 0x80000000u;
    const uint32_t ma = fa & 0x7FFFFFFFu, mb = fb & 0x7FFFFFFFu;
    const uint32_t mn = ma >= mb ? mb : ma;
    if (sa == sb) {
        fa = sa | (ma - mn);
        fb = sb | (mb - mn);
    } else if (ma >= mb) {
        fa = sa | (ma + mn);
        fb = sb;
    } else {
        fa = sa;
        fb = sb | (mb + mn);
    }
    const uint32_t diff = fa ^ fb;
    float acc = 0.0f;
    for (unsigned i = 1; i < num_bit; i++)
        if (((diff >> (31u - i)) & 1u) == 0u) acc += exp2f(-(float)i);
    const float sim = ((diff & 0x80000000u) == 0u) ? acc : -acc;
    return qm_quant(sim * 0.125f, iwl, frac);
}

__global__ void __launch_bounds__(kBlock)
k_appx_scores(const float *__restrict__ M, const float *__restrict__ v, float *__restrict__ out,
              unsigned R, unsigned C, unsigned iwl, unsigned num_bit)
{
    const unsigned w = (blockIdx.x * kBlock + threadIdx.x) / kWave;
    const unsigned lane = threadIdx.x & (kWave - 1);
    if (w >= R) return;
    const float *row = M + (size_t)w * C;
    float acc = 0.0f;
    for (unsigned c = lane; c < C; c += kWave) acc += appx_pair(row[c], v[c], iwl, num_bit);
    acc = wave_sum(acc);
    if (lane == 0) out[w] = qm_quant(acc, iwl, 31u - iwl);
}

// Softmax over `dim` elements by one workgroup, any dim (the reference stops at
// 1024 because the slot count is its blockDim).  base 0: e^(x-max), total kept
// in double, quotient rounded to float once (lib/layer_cuda.cu:2006-2042);
// base 1: 2^(x-max), float total, float division (lib/layer.c:1225-1243).
// f_shift_based divides by the integer llrint(log2(total)) (:2038).
template <typename T>
__device__ __forceinline__ T block_reduce(T v, T *scratch, bool is_max)
{
    for (int o = 32; o > 0; o >>= 1) {
        T other = __shfl_xor(v, o);
        v = is_max ? (other > v ? other : v) : v + other;
    }
    const unsigned lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    T r = scratch[0];
    for (unsigned i = 1; i < blockDim.x / kWave; i++)
        r = is_max ? (scratch[i] > r ? scratch[i] : r) : r + scratch[i];
    return r;
}

__global__ void __launch_bounds__(kBlock)
k_softmax(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ dev_max,
          unsigned dim, int base, bool shift_based)
{
    __shared__ double sd[kBlock / kWave];
    __shared__ float sf[kBlock / kWave];
    float mx = -INFINITY;
    for (unsigned i = threadIdx.x; i < dim; i += kBlock) mx = in[i] > mx ? in[i] : mx;
    mx = block_reduce<float>(mx, sf, true);
    if (threadIdx.x == 0 && dev_max) *dev_max = mx;
    if (base == 0) {
        double part = 0.0;
        for (unsigned i = threadIdx.x; i < dim; i += kBlock) {
            const float e = expf(in[i] - mx);
            out[i] = e;
            part += (double)e;
        }
        const double total = block_reduce<double>(part, sd, false);
        if (shift_based) {
            const float d = (float)llrintf(log2f((float)total));
            for (unsigned i = threadIdx.x; i < dim; i += kBlock) out[i] = out[i] / d;
        } else {
            for (unsigned i = threadIdx.x; i < dim; i += kBlock) out[i] = (float)((double)out[i] / total);
        }
    } else {
        for (unsigned i = threadIdx.x; i < dim; i += kBlock) {
            float e;
            if (base == 2) {                          // piece-wise linear exp (lib/common.c:51-73)
                const float x = in[i] - mx;
                e = fmaxf(fmaxf(0.597226f * x + 0.933989f, 0.141642f * x + 0.43981f),
                          fmaxf(0.070265f * x + 0.10888f, 0.0f * x + 0.0f));
            } else {
                e = shift_based ? exp2f(in[i] - mx + 1.0f) : exp2f(in[i] - mx);
            }
            out[i] = e;
        }
        // The CPU softmax these bases come from adds its total in a FLOAT, slot by slot (`float tot`, lib/layer.c:1161, :1236):
        // one thread walks the terms in that order.  A tree sum differs from it in the last bit, which decides Q(p) where p
        // sits on a truncation step -- e.g. a dominant slot with a runner-up 2^-24 below it: the serial float total is 1,
        // p = 1 exactly (tools/soak.py case 12750221; the short-memory kernels of the batched path do the same).
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
            for (unsigned i = 0; i < dim; i++) tot += out[i];
            sf[0] = tot;
        }
        __syncthreads();
        const float total = sf[0];
        for (unsigned i = threadIdx.x; i < dim; i += kBlock) out[i] = out[i] / total;
    }
}

// Prediction and test-phase bookkeeping, reference _cuda_max_i +
// _cuda_cross_entropy_cost + _cuda_cross_entropy_grad: arg-max with ties going
// to the highest index; where y == 1: cost += -h, match += (index == pred).
__global__ void __launch_bounds__(kBlock)
k_cross_entropy(const float *__restrict__ h, const float *__restrict__ y, float *cost, unsigned *m_cnt,
                unsigned *pred_i, float *__restrict__ grad, unsigned dim)
{
    __shared__ float sv[kBlock];
    __shared__ unsigned si[kBlock];
    float bv = -INFINITY;
    unsigned bi = 0;
    bool any = false;
    for (unsigned i = threadIdx.x; i < dim; i += kBlock) {
        // later indices win ties; NaN never wins (the reference keeps the right operand on a false '>')
        if (!any || !(bv > h[i])) { bv = h[i]; bi = i; any = true; }
    }
    sv[threadIdx.x] = any ? bv : -INFINITY;
    si[threadIdx.x] = any ? bi : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = sv[0];
        unsigned ix = si[0];
        const unsigned n = dim < kBlock ? dim : kBlock;
        for (unsigned t = 1; t < n; t++)
            if (sv[t] > v || (sv[t] == v && si[t] > ix)) { v = sv[t]; ix = si[t]; }
        si[0] = ix;
        *pred_i = ix;
    }
    __syncthreads();
    const unsigned pred = si[0];
    for (unsigned i = threadIdx.x; i < dim; i += kBlock) {
        const bool hit = (y[i] == 1.0f);
        if (hit) {
            atomicAdd(cost, -h[i]);
            if (i == pred) atomicAdd(m_cnt, 1u);
        }
        if (grad) grad[i] = hit ? 1.0f - h[i] : -h[i];
    }
}

enum { kActNull = 0, kActSigmoid = 1, kActRelu = 2 };

__global__ void k_activation(const float *in, float *out, unsigned n, int act, bool fixed, QFmt f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    if (act == kActSigmoid) v = (float)(1.0 / (1.0 + (double)expf(-v)));
    else if (act == kActRelu) v = v > 0.0f ? v : 0.0f;
    out[i] = fixed ? qm_quant(v, f.iwl, f.frac) : v;
}

__global__ void k_vec_sum(const float *a, const float *b, float *out, unsigned n, bool fixed, QFmt f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fixed) out[i] = qm_quant(qm_quant(a[i], f.iwl, f.frac) + qm_quant(b[i], f.iwl, f.frac), f.iwl, f.frac);
    else out[i] = a[i] + b[i];
}

__global__ void k_vec_scale(const float *in, const float *w, float *out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * (*w);
}

__global__ void k_vec_mul(const float *a, const float *b, float *out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// src [rows][cols] -> dest [rows][cols] or, transposed, [cols][rows]
template <bool ACCUM>
__global__ void k_copy_mat(const float *src, float *dest, unsigned cols, unsigned rows, bool trans)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const unsigned x = i % cols, y = i / cols;
    const unsigned d = trans ? x * rows + y : i;
    if (ACCUM) dest[d] += src[i];
    else dest[d] = src[i];
}

__global__ void k_set_value(float *dest, float value, unsigned n, unsigned start, unsigned stride)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i % stride == start) dest[i] = value;
}

inline void zero_f(float *p, size_t n)
{
    if (p && n) QM_HIP(hipMemsetAsync(p, 0, n * sizeof(float), 0));
}

inline int act_id(const char *s)
{
    if (s && !strcmp(s, "SIGMOID")) return kActSigmoid;
    if (s && !strcmp(s, "RELU")) return kActRelu;
    return kActNull;
}

inline void free_dev(void *p)
{
    if (p) QM_HIP(hipFree(p));
}

}  // namespace

// the launches behind the nine forward verbs (called at once, or when the deferred queue is drained op by op)
void qmdefer::run_now(const Op &op)
{
    switch (op.kind) {
    case kDot: {
        if (op.r == 0 || op.c == 0) return;
        float *in_mat = const_cast<float *>(op.in), *in_vec = const_cast<float *>(op.in2);
        if (op.trans) {
            const unsigned grid = qm_cdiv(op.c, kWave);
            if (op.fixed) k_cols_dot<true><<<grid, kBlock, 0, 0>>>(in_vec, in_mat, op.out, op.r, op.c, op.fa, op.fa);
            else k_cols_dot<false><<<grid, kBlock, 0, 0>>>(in_vec, in_mat, op.out, op.r, op.c, op.fa, op.fa);
        } else {
            const unsigned grid = qm_cdiv(op.r, kBlock / kWave);
            if (op.fixed) k_rows_dot<true><<<grid, kBlock, 0, 0>>>(in_mat, in_vec, op.out, op.r, op.c, 1, op.fa, op.fb, op.fa);
            else k_rows_dot<false><<<grid, kBlock, 0, 0>>>(in_mat, in_vec, op.out, op.r, op.c, 1, op.fa, op.fb, op.fa);
        }
        break;
    }
    case kDotAppx:
        if (op.r == 0 || op.c == 0) return;
        if (op.fa.iwl > 30) qm_fail("cuda_dot_mat_vec_fwd_appx", "iwl > 30 leaves no fraction bits");
        k_appx_scores<<<qm_cdiv(op.r, kBlock / kWave), kBlock, 0, 0>>>(op.in, op.in2, op.out, op.r, op.c, op.fa.iwl, op.k);
        break;
    case kSoftmax:
        if (op.r == 0) return;
        k_softmax<<<1, kBlock, 0, 0>>>(op.in, op.out, op.aux, op.r, g_softmax_base, op.shift);
        break;
    case kSumVec:
        if (op.r == 0) return;
        k_vec_sum<<<qm_cdiv(op.r, kBlock), kBlock, 0, 0>>>(op.in, op.in2, op.out, op.r, op.fixed, op.fa);
        break;
    case kDense: {
        if (op.c == 0 || op.r == 0) return;
        const unsigned grid = qm_cdiv(op.r, kBlock / kWave);
        if (op.fixed) k_rows_dot<true><<<grid, kBlock, 0, 0>>>(op.w, op.in, op.out, op.r, op.c, 1, op.fb, op.fa, op.fb);
        else k_rows_dot<false><<<grid, kBlock, 0, 0>>>(op.w, op.in, op.out, op.r, op.c, 1, op.fb, op.fa, op.fb);
        if (op.act != kActNull)
            k_activation<<<qm_cdiv(op.r, kBlock), kBlock, 0, 0>>>(op.out, op.out, op.r, op.act, op.fixed, op.fb);
        break;
    }
    case kDenseMat: {
        if (op.c == 0 || op.k == 0 || op.r == 0) return;
        const unsigned n_out = op.r * op.k;
        const unsigned grid = qm_cdiv(n_out, kBlock / kWave);
        if (op.fixed) k_rows_dot<true><<<grid, kBlock, 0, 0>>>(op.in, op.w, op.out, n_out, op.c, op.k, op.fa, op.fa, op.fa);
        else k_rows_dot<false><<<grid, kBlock, 0, 0>>>(op.in, op.w, op.out, n_out, op.c, op.k, op.fa, op.fa, op.fa);
        break;
    }
    case kCrossEntropy: {
        if (op.r == 0) return;
        const unsigned i = op.mode - 1u;
        k_cross_entropy<<<1, kBlock, 0, 0>>>(op.in, op.in2, op.cost[i], op.cnt[i], op.pred, op.out, op.r);
        break;
    }
    case kAct:
        if (op.r == 0) return;
        k_activation<<<qm_cdiv(op.r, kBlock), kBlock, 0, 0>>>(op.in, op.out, op.r, op.act, op.fixed, op.fa);
        break;
    case kScale:
        if (op.r == 0) return;
        k_vec_scale<<<qm_cdiv(op.r, kBlock), kBlock, 0, 0>>>(op.in, op.w, op.out, op.r);
        break;
    default:
        qm_fail("qmdefer::run_now", "unknown op");
    }
    QM_LAUNCH_CHECK();
}

int qmdefer::softmax_base() { return g_softmax_base; }

using qmdefer::Op;

namespace {

// verbose = true on a forward verb: what the reference prints there (lib/layer_cuda.cu:13-47 -- one value per "%f, ", the
// last of a row with a newline -- and the per-verb blocks :2450-2484, 2519-2553, 2868-2882, 3009-3027, 3210-3228, 3551-3567),
// with the same labels, on stdout.  The reference prints from the device after cudaDeviceSynchronize(); here the verb is a
// synchronisation point of the deferred queue (everything queued before it runs first), runs at once, and its operands are
// copied back and printed by the host.
void dump_head(const char *f_name)
{
    printf("\n< %s >\n", f_name);
}
void dump_vec(const char *label, const float *dev, unsigned n)
{
    printf("%s> dim: %u\n", label, n);
    std::vector<float> h(n);
    QM_HIP(hipDeviceSynchronize());
    if (n) QM_HIP(hipMemcpy(h.data(), dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    for (unsigned i = 0; i < n; i++) printf(i + 1 == n ? "%f\n" : "%f, ", h[i]);
}
void dump_mat(const char *head_fmt, const float *dev, unsigned r, unsigned c)
{
    printf(head_fmt, r, c);
    std::vector<float> h((size_t)r * c);
    QM_HIP(hipDeviceSynchronize());
    if (!h.empty()) QM_HIP(hipMemcpy(h.data(), dev, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (unsigned i = 0; i < r; i++)
        for (unsigned j = 0; j < c; j++) printf(j + 1 == c ? "%f\n" : "%f, ", h[(size_t)i * c + j]);
}
// a verbose verb: drain what was queued before it, run it now
void run_verbose(const Op &op)
{
    QM_SYNC_READS();
    qmdefer::run_now(op);
}
void dump_dot(const char *f_name, const Op &op)
{
    dump_head(f_name);
    printf(op.trans ? "f_trans: true\n" : "f_trans: false\n");
    dump_mat("dev_in_mat> dim_mat_r: %d, dim_mat_c: %d\n", op.in, op.r, op.c);
    dump_vec("dev_in_vec", op.in2, op.trans ? op.r : op.c);
    dump_vec("dev_out_vec", op.out, op.trans ? op.c : op.r);
    fflush(stdout);
}

}  // namespace

extern "C" {

void qmann_abi_set_softmax_base(int base)
{
    QM_SYNC_WRITES();                       // queued softmax ops run with the base they were issued under
    g_softmax_base = (base == 1 || base == 2) ? base : 0;
}
unsigned int qmann_abi_symbol_count(void) { return 66u; }

// ---------------------------------------------------------------- dot_mat_vec
void cuda_dot_mat_vec_constructor(float **dev_out_vec, float **dev_grad_out_vec, float **dev_grad_out_mat,
                                  float **dev_f_overflow, float **dev_cliff_marker, unsigned int r,
                                  unsigned int c, bool f_trans)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_out_vec, f_trans ? c : r);
    qm_alloc(dev_grad_out_vec, f_trans ? r : c);
    qm_alloc(dev_f_overflow, f_trans ? c : r);
    qm_alloc(dev_grad_out_mat, (size_t)r * c);
    qm_alloc(dev_cliff_marker, (size_t)r * c);
}

void cuda_dot_mat_vec_init(float *dev_out_vec, float *dev_grad_out_vec, float *dev_grad_out_mat,
                           float *dev_f_overflow, float *dev_cliff_marker, unsigned int r, unsigned int c,
                           bool f_trans)
{
    QM_SYNC_WRITES();
    zero_f(dev_out_vec, f_trans ? c : r);
    zero_f(dev_grad_out_vec, f_trans ? r : c);
    zero_f(dev_f_overflow, f_trans ? c : r);
    zero_f(dev_grad_out_mat, (size_t)r * c);
    zero_f(dev_cliff_marker, (size_t)r * c);
}

void cuda_dot_mat_vec_fwd(float *dev_in_mat, float *dev_in_vec, float *dev_out_vec, float *dev_f_overflow,
                          unsigned int r, unsigned int c, bool f_trans, bool f_fixed, unsigned int iwl_m,
                          unsigned int frac_m, unsigned int iwl_v, unsigned int frac_v, unsigned int f_mode,
                          bool verbose)
{
    (void)dev_f_overflow; (void)f_mode;
    Op op{};
    op.kind = qmdefer::kDot; op.in = dev_in_mat; op.in2 = dev_in_vec; op.out = dev_out_vec; op.r = r; op.c = c;
    op.trans = f_trans; op.fixed = f_fixed; op.fa = QFmt{iwl_m, frac_m}; op.fb = QFmt{iwl_v, frac_v};
    if (verbose) { run_verbose(op); dump_dot(__func__, op); return; }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_dot_mat_vec_fwd_appx(float *dev_in_mat, float *dev_in_vec, float *dev_out_vec, float *dev_f_overflow,
                               float *dev_cliff_marker, unsigned int r, unsigned int c, bool f_fixed,
                               unsigned int iwl, unsigned int frac, unsigned int f_mode,
                               unsigned int num_bit_attention, bool f_trans, bool verbose)
{
    (void)dev_cliff_marker;
    if (f_trans) {                               // the read-out of an ATTENTION_MODE 3 build is the plain weighted sum
        cuda_dot_mat_vec_fwd(dev_in_mat, dev_in_vec, dev_out_vec, dev_f_overflow, r, c, true, f_fixed, iwl, frac,
                             iwl, frac, f_mode, verbose);
        return;
    }
    Op op{};
    op.kind = qmdefer::kDotAppx; op.in = dev_in_mat; op.in2 = dev_in_vec; op.out = dev_out_vec; op.r = r; op.c = c;
    op.k = num_bit_attention; op.fixed = f_fixed; op.fa = QFmt{iwl, frac};
    if (verbose) { run_verbose(op); dump_dot(__func__, op); return; }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_dot_mat_vec_destructor(float *dev_out_vec, float *dev_grad_out_vec, float *dev_grad_out_mat,
                                 float *dev_f_overflow, float *dev_cliff_marker)
{
    QM_SYNC_WRITES();
    free_dev(dev_out_vec); free_dev(dev_grad_out_vec); free_dev(dev_grad_out_mat);
    free_dev(dev_f_overflow); free_dev(dev_cliff_marker);
}

// -------------------------------------------------------------------- softmax
void cuda_softmax_constructor(float **dev_out_vec, float **dev_grad_out, float **dev_max, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_out_vec, dim);
    qm_alloc(dev_grad_out, dim);
    qm_alloc(dev_max, 1);
}

void cuda_softmax_init(float *dev_out_vec, float *dev_grad_out, float *dev_max, unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_out_vec, dim);
    zero_f(dev_grad_out, dim);
    zero_f(dev_max, 1);
}

void cuda_softmax_fwd(float *dev_out_vec, float *dev_in_vec, float *out_vec, float *in_vec, float *dev_max,
                      unsigned int dim, bool f_shift_based, bool verbose)
{
    (void)out_vec; (void)in_vec;
    Op op{};
    op.kind = qmdefer::kSoftmax; op.in = dev_in_vec; op.out = dev_out_vec; op.aux = dev_max; op.r = dim; op.shift = f_shift_based;
    if (verbose) {
        run_verbose(op);
        dump_head(__func__); dump_vec("dev_in_vec", dev_in_vec, dim); dump_vec("dev_out_vec", dev_out_vec, dim); fflush(stdout);
        return;
    }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_softmax_destructor(float *dev_out_vec, float *dev_grad_out, float *dev_max)
{
    QM_SYNC_WRITES();
    free_dev(dev_out_vec); free_dev(dev_grad_out); free_dev(dev_max);
}

// -------------------------------------------------------------------- sum_vec
void cuda_sum_vec_constructor(float **dev_out_vec, float **dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_out_vec, dim);
    qm_alloc(dev_grad_out, dim);
}

void cuda_sum_vec_init(float *dev_out_vec, float *dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_out_vec, dim);
    zero_f(dev_grad_out, dim);
}

void cuda_sum_vec_fwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_out_vec, unsigned int dim,
                      bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    (void)f_mode;
    Op op{};
    op.kind = qmdefer::kSumVec; op.in = dev_in_vec_a; op.in2 = dev_in_vec_b; op.out = dev_out_vec; op.r = dim;
    op.fixed = f_fixed; op.fa = QFmt{iwl, frac};
    if (verbose) {
        run_verbose(op);
        dump_head(__func__); dump_vec("dev_in_vec_a", dev_in_vec_a, dim); dump_vec("dev_in_vec_b", dev_in_vec_b, dim);
        dump_vec("dev_out_vec", dev_out_vec, dim); fflush(stdout);
        return;
    }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_sum_vec_destructor(float *dev_out_vec, float *dev_grad_out)
{
    QM_SYNC_WRITES();
    free_dev(dev_out_vec); free_dev(dev_grad_out);
}

// ---------------------------------------------------------------------- dense
void cuda_dense_constructor(float **dev_w_mat, float **dev_w_mat_del, float **dev_w_mat_best, float **dev_bias,
                            float **dev_bias_del, float **dev_out_vec, float **dev_grad_out,
                            float **dev_grad_l2_norm, float **dev_grad_bias_l2_norm, float **dev_f_overflow,
                            unsigned int dim_in, unsigned int dim_out)
{
    QM_SYNC_WRITES();
    const size_t nw = (size_t)dim_in * dim_out;
    qm_alloc(dev_w_mat, nw);
    qm_alloc(dev_w_mat_del, nw);
    qm_alloc(dev_w_mat_best, nw);
    qm_alloc(dev_bias, dim_out);
    qm_alloc(dev_bias_del, dim_out);
    qm_alloc(dev_out_vec, dim_out);
    qm_alloc(dev_grad_out, dim_in);
    qm_alloc(dev_grad_l2_norm, 1);
    if (dev_grad_bias_l2_norm) qm_alloc(dev_grad_bias_l2_norm, 1);
    qm_alloc(dev_f_overflow, dim_out);
}

void cuda_dense_init(float *dev_out_vec, float *dev_grad_out, float *dev_w_mat_del, float *dev_w_mat,
                     float *dev_bias, float *dev_bias_del, float *w_mat, float *bias, float *dev_f_overflow,
                     unsigned int dim_in, unsigned int dim_out)
{
    QM_SYNC_WRITES();
    const size_t nw = (size_t)dim_in * dim_out;
    zero_f(dev_out_vec, dim_out);
    zero_f(dev_grad_out, dim_in);
    zero_f(dev_w_mat_del, nw);
    zero_f(dev_bias_del, dim_out);
    zero_f(dev_f_overflow, dim_out);
    QM_HIP(hipMemcpy(dev_w_mat, w_mat, nw * sizeof(float), hipMemcpyHostToDevice));
    QM_HIP(hipMemcpy(dev_bias, bias, dim_out * sizeof(float), hipMemcpyHostToDevice));
}

void cuda_dense_fwd(float *dev_w_mat, float *dev_bias, float *dev_in_vec, float *dev_out_vec,
                    float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out, char *activation,
                    bool f_fixed, unsigned int iwl_in, unsigned int frac_in, unsigned int iwl_w,
                    unsigned int frac_w, unsigned int f_mode, bool verbose)
{
    (void)dev_bias; (void)dev_f_overflow; (void)f_mode;
    if (f_fixed && iwl_w + frac_w == 0)
        qm_fail(__func__, "binary-weight (iwl_w+frac_w==0) rescale path is not part of this library");
    Op op{};
    op.kind = qmdefer::kDense; op.w = dev_w_mat; op.in = dev_in_vec; op.out = dev_out_vec; op.r = dim_out; op.c = dim_in;
    op.act = act_id(activation); op.fixed = f_fixed; op.fa = QFmt{iwl_in, frac_in}; op.fb = QFmt{iwl_w, frac_w};
    if (verbose) {
        run_verbose(op);
        dump_head(__func__); dump_vec("dev_in_vec", dev_in_vec, dim_in);
        dump_mat("dev_w_mat> dim_out: %d, dim_in: %d\n", dev_w_mat, dim_out, dim_in);
        dump_vec("dev_out_vec", dev_out_vec, dim_out); fflush(stdout);
        return;
    }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_dense_destructor(float *dev_w_mat, float *dev_w_mat_del, float *dev_w_mat_best, float *dev_out_vec,
                           float *dev_grad_out, float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm,
                           float *dev_f_overflow)
{
    QM_SYNC_WRITES();
    // the reference's constructor never allocates dev_grad_bias_l2_norm and its
    // destructor never frees dev_bias/dev_bias_del; free only what is certainly ours
    (void)dev_grad_bias_l2_norm;
    free_dev(dev_w_mat); free_dev(dev_w_mat_del); free_dev(dev_w_mat_best); free_dev(dev_out_vec);
    free_dev(dev_grad_out); free_dev(dev_grad_l2_norm); free_dev(dev_f_overflow);
}

// ------------------------------------------------------------------ dense_mat
void cuda_dense_mat_constructor(float **dev_w_mat, float **dev_w_mat_del, float **dev_w_mat_best,
                                float **dev_bias, float **dev_bias_del, float **dev_out_mat,
                                float **dev_grad_out, float **dev_grad_l2_norm, float **dev_grad_bias_l2_norm,
                                float **dev_f_overflow, unsigned int dim_in, unsigned int dim_out,
                                unsigned int dim_len)
{
    QM_SYNC_WRITES();
    const size_t nw = (size_t)dim_in * dim_out;
    qm_alloc(dev_w_mat, nw);
    qm_alloc(dev_w_mat_del, nw);
    qm_alloc(dev_w_mat_best, nw);
    qm_alloc(dev_bias, dim_out);
    qm_alloc(dev_bias_del, dim_out);
    qm_alloc(dev_out_mat, (size_t)dim_len * dim_out);
    qm_alloc(dev_grad_out, (size_t)dim_len * dim_in);
    qm_alloc(dev_grad_l2_norm, 1);
    if (dev_grad_bias_l2_norm) qm_alloc(dev_grad_bias_l2_norm, 1);
    qm_alloc(dev_f_overflow, (size_t)dim_len * dim_out);
}

void cuda_dense_mat_init(float *dev_out_mat, float *dev_grad_out, float *dev_w_mat, float *dev_w_mat_del,
                         float *dev_bias, float *dev_bias_del, float *w_mat, float *bias,
                         float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out,
                         unsigned int dim_len)
{
    QM_SYNC_WRITES();
    const size_t nw = (size_t)dim_in * dim_out;
    zero_f(dev_out_mat, (size_t)dim_len * dim_out);
    zero_f(dev_grad_out, (size_t)dim_len * dim_in);
    zero_f(dev_w_mat_del, nw);
    zero_f(dev_bias_del, dim_out);
    zero_f(dev_f_overflow, (size_t)dim_len * dim_out);
    QM_HIP(hipMemcpy(dev_w_mat, w_mat, nw * sizeof(float), hipMemcpyHostToDevice));
    QM_HIP(hipMemcpy(dev_bias, bias, dim_out * sizeof(float), hipMemcpyHostToDevice));
}

void cuda_dense_mat_fwd(float *dev_w_mat, float *dev_bias, float *dev_in_mat, float *dev_out_mat,
                        float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out, unsigned int dim_len,
                        bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    (void)dev_bias; (void)dev_f_overflow; (void)f_mode;
    if (f_fixed && iwl + frac == 0)
        qm_fail(__func__, "binary-weight (iwl+frac==0) rescale path is not part of this library");
    Op op{};
    op.kind = qmdefer::kDenseMat; op.w = dev_w_mat; op.in = dev_in_mat; op.out = dev_out_mat; op.r = dim_len; op.c = dim_in;
    op.k = dim_out; op.fixed = f_fixed; op.fa = QFmt{iwl, frac};
    if (verbose) {
        run_verbose(op);
        dump_head(__func__); dump_mat("dev_in_mat> dim_len: %d, dim_in: %d\n", dev_in_mat, dim_len, dim_in);
        dump_mat("dev_w_mat> dim_out: %d, dim_in: %d\n", dev_w_mat, dim_out, dim_in);
        dump_mat("dev_out_mat>dim_len: %d, dim_out: %d\n", dev_out_mat, dim_len, dim_out); fflush(stdout);
        return;
    }
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_dense_mat_destructor(float *dev_w_mat, float *dev_w_mat_del, float *dev_w_mat_best,
                               float *dev_out_mat, float *dev_grad_out, float *dev_grad_l2_norm,
                               float *dev_f_overflow)
{
    QM_SYNC_WRITES();
    free_dev(dev_w_mat); free_dev(dev_w_mat_del); free_dev(dev_w_mat_best); free_dev(dev_out_mat);
    free_dev(dev_grad_out); free_dev(dev_grad_l2_norm); free_dev(dev_f_overflow);
}

// -------------------------------------------------------------- cross entropy
void cuda_cross_entropy_constructor(float **dev_cost_train, float **dev_cost_valid, float **dev_cost_test,
                                    unsigned int **dev_m_cnt_train, unsigned int **dev_m_cnt_valid,
                                    unsigned int **dev_m_cnt_test, unsigned int **dev_pred_i,
                                    float **dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_cost_train, 1); qm_alloc(dev_cost_valid, 1); qm_alloc(dev_cost_test, 1);
    qm_alloc(dev_m_cnt_train, 1); qm_alloc(dev_m_cnt_valid, 1); qm_alloc(dev_m_cnt_test, 1);
    qm_alloc(dev_pred_i, 1);
    qm_alloc(dev_grad_out, dim);
}

void cuda_cross_entropy_init(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                             unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                             unsigned int *dev_m_cnt_test, float *dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_cost_train, 1); zero_f(dev_cost_valid, 1); zero_f(dev_cost_test, 1);
    zero_f((float *)dev_m_cnt_train, 1); zero_f((float *)dev_m_cnt_valid, 1); zero_f((float *)dev_m_cnt_test, 1);
    zero_f(dev_grad_out, dim);
}

void cuda_cross_entropy_run(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                            unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                            unsigned int *dev_m_cnt_test, unsigned int *dev_pred_i, float *cost, float *dev_h,
                            float *dev_y, float *h, float *y, float *dev_grad_out, float *grad_out,
                            unsigned int dim, unsigned int mode)
{
    (void)cost; (void)h; (void)y; (void)grad_out;
    if (mode < 1 || mode > 3) qm_fail(__func__, "mode must be 1 (train), 2 (valid) or 3 (test)");
    Op op{};
    op.kind = qmdefer::kCrossEntropy; op.in = dev_h; op.in2 = dev_y; op.out = dev_grad_out; op.r = dim; op.mode = mode;
    op.cost[0] = dev_cost_train; op.cost[1] = dev_cost_valid; op.cost[2] = dev_cost_test;
    op.cnt[0] = dev_m_cnt_train; op.cnt[1] = dev_m_cnt_valid; op.cnt[2] = dev_m_cnt_test; op.pred = dev_pred_i;
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_cross_entropy_cost_load(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                                  float *cost_train, float *cost_valid, float *cost_test)
{
    QM_SYNC_READS();
    QM_HIP(hipMemcpy(cost_train, dev_cost_train, sizeof(float), hipMemcpyDeviceToHost));
    QM_HIP(hipMemcpy(cost_valid, dev_cost_valid, sizeof(float), hipMemcpyDeviceToHost));
    QM_HIP(hipMemcpy(cost_test, dev_cost_test, sizeof(float), hipMemcpyDeviceToHost));
    zero_f(dev_cost_train, 1); zero_f(dev_cost_valid, 1); zero_f(dev_cost_test, 1);
}

void cuda_cross_entropy_m_cnt_load(unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                                   unsigned int *dev_m_cnt_test, unsigned int *m_cnt_train,
                                   unsigned int *m_cnt_valid, unsigned int *m_cnt_test)
{
    QM_SYNC_READS();
    QM_HIP(hipMemcpy(m_cnt_train, dev_m_cnt_train, sizeof(unsigned), hipMemcpyDeviceToHost));
    QM_HIP(hipMemcpy(m_cnt_valid, dev_m_cnt_valid, sizeof(unsigned), hipMemcpyDeviceToHost));
    QM_HIP(hipMemcpy(m_cnt_test, dev_m_cnt_test, sizeof(unsigned), hipMemcpyDeviceToHost));
    zero_f((float *)dev_m_cnt_train, 1); zero_f((float *)dev_m_cnt_valid, 1); zero_f((float *)dev_m_cnt_test, 1);
}

void cuda_cross_entropy_destructor(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                                   float *dev_m_cnt_train, float *dev_m_cnt_valid, float *dev_m_cnt_test,
                                   float *dev_pred_i, float *dev_grad_out)
{
    QM_SYNC_WRITES();
    free_dev(dev_cost_train); free_dev(dev_cost_valid); free_dev(dev_cost_test);
    free_dev(dev_m_cnt_train); free_dev(dev_m_cnt_valid); free_dev(dev_m_cnt_test);
    free_dev(dev_pred_i); free_dev(dev_grad_out);
}

// --------------------------------------------------------- activation / scale
void cuda_activation_constructor(float **dev_out, float **dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_out, dim);
    qm_alloc(dev_grad_out, dim);
}

void cuda_activation_init(float *dev_out, float *dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_out, dim);
    zero_f(dev_grad_out, dim);
}

void cuda_activation_fwd(float *dev_in, float *dev_out, char *type_act, unsigned int dim, bool f_fixed,
                         unsigned int iwl, unsigned int frac, unsigned int f_mode)
{
    (void)f_mode;
    Op op{};
    op.kind = qmdefer::kAct; op.in = dev_in; op.out = dev_out; op.act = act_id(type_act); op.r = dim; op.fixed = f_fixed;
    op.fa = QFmt{iwl, frac};
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_activation_destructor(float *dev_out, float *dev_grad_out)
{
    QM_SYNC_WRITES();
    free_dev(dev_out); free_dev(dev_grad_out);
}

void cuda_scale_constructor(float **dev_w, float **dev_w_del, float **dev_w_best, float **dev_out,
                            float **dev_grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_w, 1); qm_alloc(dev_w_del, 1); qm_alloc(dev_w_best, 1);
    qm_alloc(dev_out, dim);
    qm_alloc(dev_grad_out, dim);
}

void cuda_scale_init(float *dev_w, float *dev_w_del, float *dev_out, float *dev_grad_out, float *w,
                     unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_w_del, 1);
    zero_f(dev_out, dim);
    zero_f(dev_grad_out, dim);
    QM_HIP(hipMemcpy(dev_w, w, sizeof(float), hipMemcpyHostToDevice));
}

void cuda_scale_fwd(float *dev_in, float *dev_w, float *dev_out, unsigned int dim, bool f_fixed,
                    unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    (void)f_fixed; (void)iwl; (void)frac; (void)f_mode; (void)verbose;      // (the reference's scale verb prints nothing either: :4818-4826)
    Op op{};
    op.kind = qmdefer::kScale; op.in = dev_in; op.w = dev_w; op.out = dev_out; op.r = dim;
    if (!qmdefer::submit(op)) qmdefer::run_now(op);
}

void cuda_scale_destructor(float *dev_w, float *dev_w_del, float *dev_w_best, float *dev_out,
                           float *dev_grad_out)
{
    QM_SYNC_WRITES();
    free_dev(dev_w); free_dev(dev_w_del); free_dev(dev_w_best); free_dev(dev_out); free_dev(dev_grad_out);
}

// ------------------------------------------------------ mult_e_vec / mult_e_mat
void cuda_mult_e_vec_constructor(float **dev_out_vec, float **dev_grad_out_a, float **dev_grad_out_b,
                                 unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_out_vec, dim); qm_alloc(dev_grad_out_a, dim); qm_alloc(dev_grad_out_b, dim);
}

void cuda_mult_e_vec_init(float *dev_out_vec, float *dev_grad_out_a, float *dev_grad_out_b, unsigned int dim)
{
    QM_SYNC_WRITES();
    zero_f(dev_out_vec, dim); zero_f(dev_grad_out_a, dim); zero_f(dev_grad_out_b, dim);
}

void cuda_mult_e_vec_fwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_out_vec, float *in_vec_a,
                         float *in_vec_b, float *out_vec, unsigned int dim)
{
    QM_SYNC_WRITES();
    (void)in_vec_a; (void)in_vec_b; (void)out_vec;
    if (dim == 0) return;
    k_vec_mul<<<qm_cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_in_vec_a, dev_in_vec_b, dev_out_vec, dim);
    QM_LAUNCH_CHECK();
}

void cuda_mult_e_vec_destructor(void) {
    QM_SYNC_WRITES();}

void cuda_mult_e_mat_constructor(float **dev_out_mat, float **dev_grad_out_a, float **dev_grad_out_b,
                                 unsigned int dim_row, unsigned int dim_col)
{
    QM_SYNC_WRITES();
    const size_t n = (size_t)dim_row * dim_col;
    qm_alloc(dev_out_mat, n); qm_alloc(dev_grad_out_a, n); qm_alloc(dev_grad_out_b, n);
}

void cuda_mult_e_mat_init(float *dev_out_mat, float *dev_grad_out_a, float *dev_grad_out_b,
                          unsigned int dim_row, unsigned int dim_col)
{
    QM_SYNC_WRITES();
    const size_t n = (size_t)dim_row * dim_col;
    zero_f(dev_out_mat, n); zero_f(dev_grad_out_a, n); zero_f(dev_grad_out_b, n);
}

void cuda_mult_e_mat_fwd(float *dev_in_mat_a, float *dev_in_mat_b, float *dev_out_mat, float *in_mat_a,
                         float *in_mat_b, float *out_mat, unsigned int dim_row, unsigned int dim_col)
{
    QM_SYNC_WRITES();
    (void)in_mat_a; (void)in_mat_b; (void)out_mat;
    const unsigned n = dim_row * dim_col;
    if (n == 0) return;
    k_vec_mul<<<qm_cdiv(n, kBlock), kBlock, 0, 0>>>(dev_in_mat_a, dev_in_mat_b, dev_out_mat, n);
    QM_LAUNCH_CHECK();
}

void cuda_mult_e_mat_destructor(void) {
    QM_SYNC_WRITES();}

// ------------------------------------------------- helpers MemN2N.o calls itself
void cuda_data_constructor(float **dev_m, float **dev_q, float **dev_a, unsigned int dim_len,
                           unsigned int dim_in, unsigned int num_sample)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_m, (size_t)dim_len * dim_in);
    qm_alloc(dev_q, (size_t)num_sample * dim_in);
    qm_alloc(dev_a, (size_t)num_sample * dim_in);
}

void cuda_data_in(float *dev_m, float *dev_q, float *dev_a, float *m, float *q, float *a, unsigned int dim_len,
                  unsigned int dim_in, unsigned int num_sample)
{
    QM_SYNC_WRITES();
    QM_HIP(hipMemcpy(dev_m, m, (size_t)dim_len * dim_in * sizeof(float), hipMemcpyHostToDevice));
    QM_HIP(hipMemcpy(dev_q, q, (size_t)num_sample * dim_in * sizeof(float), hipMemcpyHostToDevice));
    QM_HIP(hipMemcpy(dev_a, a, (size_t)num_sample * dim_in * sizeof(float), hipMemcpyHostToDevice));
}

void cuda_data_destructor(float *dev_m, float *dev_q, float *dev_a)
{
    QM_SYNC_WRITES();
    free_dev(dev_m); free_dev(dev_q); free_dev(dev_a);
}

void cuda_dup_grad_constructor(float **dev_dup_grad, unsigned int num_hop, unsigned int dim)
{
    QM_SYNC_WRITES();
    qm_alloc(dev_dup_grad, (size_t)num_hop * dim);
}

void cuda_dup_grad_destructor(float *dev_dup_grad) {
    QM_SYNC_WRITES(); free_dev(dev_dup_grad); }

void cuda_copy_mat(float *dev_src, float *dev_dest, unsigned int dim_col, unsigned int dim_row, bool f_trans)
{
    QM_SYNC_WRITES();
    const unsigned n = dim_col * dim_row;
    if (n == 0) return;
    k_copy_mat<false><<<qm_cdiv(n, kBlock), kBlock, 0, 0>>>(dev_src, dev_dest, dim_col, dim_row, f_trans);
    QM_LAUNCH_CHECK();
}

void cuda_accum_mat(float *dev_src, float *dev_dest, unsigned int dim_col, unsigned int dim_row, bool f_trans)
{
    QM_SYNC_WRITES();
    const unsigned n = dim_col * dim_row;
    if (n == 0) return;
    k_copy_mat<true><<<qm_cdiv(n, kBlock), kBlock, 0, 0>>>(dev_src, dev_dest, dim_col, dim_row, f_trans);
    QM_LAUNCH_CHECK();
}

void cuda_set_value(float *dest, float value, unsigned int dim, unsigned int start_idx, unsigned int stride)
{
    QM_SYNC_WRITES();
    if (dim == 0 || stride == 0) return;
    k_set_value<<<qm_cdiv(dim, kBlock), kBlock, 0, 0>>>(dest, value, dim, start_idx, stride);
    QM_LAUNCH_CHECK();
}

void cuda_copy_dev2host(float *host, float *dev, unsigned int size)
{
    QM_SYNC_READS();
    QM_HIP(hipMemcpy(host, dev, (size_t)size * sizeof(float), hipMemcpyDeviceToHost));
}

}  // extern "C"
