// abi_train.hip -- backward / weight-update verbs of boundary B.
//
// These make the unmodified reference host (MemN2N.c: SGD with batch-32 gradient accumulation,
// weight tying, gradient-norm clipping) run end to end on the MI355X.  They are FUNCTIONAL, not
// tuned: training drives them one small op at a time (tens of launches per sample), so they are
// launch-latency bound whatever the kernel does.  Each output element is owned by one thread that
// walks its reduction serially in the reference's index order -- with separate multiply and add
// (fp-contract off) that reproduces the reference's thread-0 serial sums bit for bit, so the
// float-mode gradients match the oracle exactly, not just within a tolerance.
//
// Reference definitions: lib/layer_cuda.cu:2560 (dot_mat_vec_bwd), :2666 (_bwd_appx with the
// surrogate gradients :742-1463), :2885 (softmax_bwd), :3031 (sum_vec_bwd), :3232/:3317 (dense),
// :3570/:3612 (dense_mat), :3908 (dup_grad_bwd), :4586 (activation_bwd), :4829/:4860 (scale),
// :4265/:4396 (mult_e).
#include "qfmt.h"
#include "rt.h"
#include "defer.h"
#include "../../include/qmann_abi.h"

#include <string.h>

namespace {

constexpr int kBlock = 256;

// out[i] (+)= Qo( sum_{t<K} mul(a[ia], b[ib]) ),  i < n_out
//   PAT 0 (reference _cuda_mat_mat_product):       ia = (i / ncol) * K + t,     ib = (i % ncol) + ncol * t
//   PAT 1 (reference _cuda_mat_trans_mat_product): ia = (i / ncol) + nrow * t,  ib = (i % ncol) + ncol * t
// fixed: mul = Qa(Qa(a) . Qb(b)), result quantised to fo; float: plain product, no quantisation.
template <int PAT, bool ACCUM>
__global__ void __launch_bounds__(kBlock)
k_prod(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, unsigned n_out, unsigned K,
       unsigned nrow, unsigned ncol, bool fixed, QFmt fa, QFmt fb, QFmt fo)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_out) return;
    const unsigned r = i / ncol, c = i % ncol;
    float sum = 0.0f;
    for (unsigned t = 0; t < K; t++) {
        const float x = PAT == 0 ? a[(size_t)r * K + t] : a[r + (size_t)nrow * t];
        const float y = b[c + (size_t)ncol * t];
        const float p = fixed ? qm_fixed_mul(x, y, fa, fb) : x * y;
        sum += p;
    }
    const float v = fixed ? qm_quant(sum, fo.iwl, fo.frac) : sum;
    if (ACCUM) out[i] += v;
    else out[i] = v;
}

// out[i] = Qo( sum_k Qa(Qa(A[i][k]) . Qb(v[k])) ): rows of A against one vector (reference
// _cuda_mat_mat_trans_product with dim_out_col = 1), serial per output
__global__ void __launch_bounds__(kBlock)
k_rows_vec(const float *__restrict__ A, const float *__restrict__ v, float *__restrict__ out, unsigned n_out, unsigned K,
           bool fixed, QFmt fa, QFmt fb, QFmt fo)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_out) return;
    float sum = 0.0f;
    for (unsigned k = 0; k < K; k++) {
        const float p = fixed ? qm_fixed_mul(A[(size_t)i * K + k], v[k], fa, fb) : A[(size_t)i * K + k] * v[k];
        sum += p;
    }
    out[i] = fixed ? qm_quant(sum, fo.iwl, fo.frac) : sum;
}

// reference _cuda_softmax_bwd: g[i] = y[i] . (gin[i] - sum_j y[j] gin[j]); the shift-based form scales by 0.7
__global__ void __launch_bounds__(kBlock)
k_softmax_bwd(const float *__restrict__ y, const float *__restrict__ gin, float *__restrict__ gout, unsigned dim,
              bool shift_based)
{
    __shared__ float s_sum;
    if (threadIdx.x == 0) {
        float sum = 0.0f;
        for (unsigned j = 0; j < dim; j++) {
            const float t = y[j] * gin[j];
            sum += t;
        }
        s_sum = sum;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < dim; i += kBlock) {
        if (shift_based) gout[i] = (float)((0.7 * (double)y[i]) * (double)(gin[i] - s_sum));
        else gout[i] = y[i] * (gin[i] - s_sum);
    }
}

__global__ void k_copy(const float *src, float *dst, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void k_vec_sum_q(const float *a, const float *b, float *out, unsigned n, bool fixed, QFmt f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fixed) out[i] = qm_quant(qm_quant(a[i], f.iwl, f.frac) + qm_quant(b[i], f.iwl, f.frac), f.iwl, f.frac);
    else out[i] = a[i] + b[i];
}

// reference _cuda_l2_norm: *out += sqrt(sum of squares of one row), one row per block
__global__ void __launch_bounds__(64)
k_row_l2_accum(const float *__restrict__ in, float *out, unsigned cols)
{
    if (threadIdx.x != 0) return;
    const float *row = in + (size_t)blockIdx.x * cols;
    float sum = 0.0f;
    for (unsigned j = 0; j < cols; j++) {
        const float t = row[j] * row[j];
        sum += t;
    }
    atomicAdd(out, sqrtf(sum));
}

// reference _cuda_mat_w_up (:1783-1830), then the accumulated gradient is cleared
__global__ void k_mat_w_up(float *__restrict__ w_del, float *__restrict__ w, unsigned n, unsigned batch, float lr,
                           float lambda, float max_norm, const float *norm, bool fixed, QFmt f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float nv = *norm;
    float step;
    if (nv > max_norm) step = lr / batch * w_del[i] * max_norm / nv;
    else step = lr / batch * w_del[i];
    const float decay = lr * lambda * w[i];
    if (fixed) {
        float x = w[i] + (qm_quant(step, f.iwl, f.frac) + qm_quant(decay, f.iwl, f.frac));
        w[i] = qm_quant(x, f.iwl, f.frac);
    } else {
        w[i] += step + decay;
    }
    w_del[i] = 0.0f;
}

enum { kActNull = 0, kActSigmoid = 1, kActRelu = 2 };
inline int act_id(const char *s)
{
    if (s && !strcmp(s, "SIGMOID")) return kActSigmoid;
    if (s && !strcmp(s, "RELU")) return kActRelu;
    return kActNull;
}

// reference _cuda_sigmoid_bwd / _cuda_relu_bwd / _cuda_bypass (:1664-1727)
__global__ void k_act_bwd(const float *out, const float *gin, float *gout, unsigned n, int act, bool fixed, QFmt f)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g;
    if (act == kActSigmoid) g = (float)((double)(gin[i] * out[i]) * (1.0 - (double)out[i]));
    else if (act == kActRelu) g = out[i] > 0.0f ? gin[i] : 0.0f;
    else g = gin[i];
    gout[i] = fixed ? qm_quant(g, f.iwl, f.frac) : g;
}

// Surrogate gradients of the Hamming ("approximate") attention, reference
// _cuda_backprop_grad_out_mat (:742-...) and _cuda_backprop_grad_out_vec: the operands are
// re-encoded and transformed exactly as in the forward kernel, then every differing bit among the
// first num_bit contributes +-2^-3 with a sign rule that depends on the bit position.
struct AppxWords {
    uint32_t fm, fv;      // transformed words
    float sm, sv;         // signs of the ORIGINAL words
};

__device__ __forceinline__ AppxWords appx_words(float m, float v, unsigned iwl)
{
    const unsigned frac = 31u - iwl;
    uint32_t fm = qm_signmag(m, iwl, frac), fv = qm_signmag(v, iwl, frac);
    AppxWords w;
    w.sm = ((int32_t)fm >= 0) ? 1.0f : -1.0f;
    w.sv = ((int32_t)fv >= 0) ? 1.0f : -1.0f;
    const uint32_t sbm = fm & 0x80000000u, sbv = fv & 0x80000000u;
    const uint32_t am = fm & 0x7FFFFFFFu, av = fv & 0x7FFFFFFFu;
    const uint32_t mn = am >= av ? av : am;
    if (w.sm == w.sv) {
        fm = sbm | (am - mn);
        fv = sbv | (av - mn);
    } else if (am >= av) {
        fm = sbm | (am + mn);
        fv = sbv;
    } else {
        fm = sbm;
        fv = sbv | (av + mn);
    }
    w.fm = fm; w.fv = fv;
    return w;
}

// grad_out_mat[r][c] = g(r, c) . grad_in[r]
__global__ void __launch_bounds__(kBlock)
k_appx_grad_mat(const float *__restrict__ M, const float *__restrict__ v, const float *__restrict__ gin,
                float *__restrict__ gmat, unsigned R, unsigned C, unsigned iwl, unsigned num_bit)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= R * C) return;
    const unsigned r = i / C, c = i % C;
    const AppxWords w = appx_words(M[i], v[c], iwl);
    float ta = 0.0f;
    for (unsigned b = 0; b < num_bit && b < 32; b++) {
        const int bm = (int)((w.fm >> (31 - b)) & 1u), bv = (int)((w.fv >> (31 - b)) & 1u);
        if (bm != bv) {
            if (b == 0) ta += (float)(bm - bv) * w.sm * 0.125f;
            else ta += -1.0f * w.sv * 0.125f * (float)(bm - bv);
        }
    }
    gmat[i] = ta * gin[r];
}

// grad_out_vec[c] = sum_r g'(r, c) . grad_in[r]; g' re-adds the last bit term on every later bit
// (the reference does not reset its temporary inside the bit loop)
__global__ void __launch_bounds__(kBlock)
k_appx_grad_vec(const float *__restrict__ M, const float *__restrict__ v, const float *__restrict__ gin,
                float *__restrict__ gvec, unsigned R, unsigned C, unsigned iwl, unsigned num_bit)
{
    const unsigned c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    float sum = 0.0f;
    for (unsigned r = 0; r < R; r++) {
        const AppxWords w = appx_words(M[(size_t)r * C + c], v[c], iwl);
        float ta = 0.0f, ga = 0.0f;
        for (unsigned b = 0; b < num_bit && b < 32; b++) {
            const int bm = (int)((w.fm >> (31 - b)) & 1u), bv = (int)((w.fv >> (31 - b)) & 1u);
            if (bm != bv) {
                if (b == 0) ta = -1.0f * (float)(bm - bv) * w.sv * 0.125f;
                else ta = 1.0f * w.sm * 0.125f * (float)(bm - bv);
            }
            ga += ta;
        }
        const float t = ga * gin[r];
        sum += t;
    }
    gvec[c] = sum;
}

__global__ void k_vec_mul(const float *a, const float *b, float *out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

__global__ void k_scale_bwd(const float *gin, const float *in, const float *w, float *w_del, float *gout, unsigned n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // reference: assigns (not accumulates) the scalar
        float sum = 0.0f;
        for (unsigned i = 0; i < n; i++) {
            const float t = gin[i] * in[i];
            sum += t;
        }
        *w_del = sum;
    }
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gout[i] = gin[i] * (*w);
}

__global__ void k_scalar_w_up(float *w, float *w_del, unsigned batch, float lr, float lambda)
{
    *w += lr / batch * (*w_del) + lr * lambda * (*w);
    *w_del = 0.0f;
}

inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

}  // namespace

extern "C" {

void cuda_dot_mat_vec_bwd(float *dev_in_mat, float *dev_in_vec, float *dev_grad_in, float *dev_grad_out_mat,
                          float *dev_grad_out_vec, float *dev_f_overflow, unsigned int r, unsigned int c, bool f_trans,
                          bool f_fixed, unsigned int iwl_m, unsigned int frac_m, unsigned int iwl_v, unsigned int frac_v,
                          unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_f_overflow; (void)iwl_v; (void)frac_v; (void)f_mode; (void)verbose;
    if (r == 0 || c == 0) return;
    const QFmt fm{iwl_m, frac_m}, fg{1u, iwl_m + frac_m - 1u};
    if (f_trans) {
        // grad_out_mat[r][c] = in_vec[r] . grad_in[c] ; grad_out_vec[r] = sum_c in_mat[r][c] . grad_in[c]
        k_prod<0, false><<<cdiv(r * c, kBlock), kBlock, 0, 0>>>(dev_in_vec, dev_grad_in, dev_grad_out_mat, r * c, 1, 0, c,
                                                              f_fixed, fm, fm, fg);
        k_rows_vec<<<cdiv(r, kBlock), kBlock, 0, 0>>>(dev_in_mat, dev_grad_in, dev_grad_out_vec, r, c, f_fixed, fm, fm, fg);
    } else {
        // grad_out_mat[r][c] = grad_in[r] . in_vec[c] ; grad_out_vec[c] = sum_r grad_in[r] . in_mat[r][c]
        k_prod<0, false><<<cdiv(r * c, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_vec, dev_grad_out_mat, r * c, 1, 0, c,
                                                              f_fixed, fm, fm, fg);
        k_prod<1, false><<<cdiv(c, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_mat, dev_grad_out_vec, c, r, 1, c,
                                                          f_fixed, fm, fm, fg);
    }
    QM_LAUNCH_CHECK();
}

void cuda_dot_mat_vec_bwd_appx(float *dev_in_mat, float *dev_in_vec, float *dev_grad_in, float *dev_grad_out_mat,
                               float *dev_grad_out_vec, float *dev_f_overflow, float *dev_cliff_marker, unsigned int r,
                               unsigned int c, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode,
                               unsigned int num_bit_attention, bool f_trans, bool verbose, unsigned int hop)
{
    QM_SYNC_WRITES();
    (void)dev_cliff_marker; (void)hop;
    if (f_trans) {
        cuda_dot_mat_vec_bwd(dev_in_mat, dev_in_vec, dev_grad_in, dev_grad_out_mat, dev_grad_out_vec, dev_f_overflow, r,
                             c, true, f_fixed, iwl, frac, iwl, frac, f_mode, verbose);
        return;
    }
    if (r == 0 || c == 0) return;
    if (iwl > 30) qm_fail(__func__, "iwl > 30 leaves no fraction bits");
    k_appx_grad_mat<<<cdiv(r * c, kBlock), kBlock, 0, 0>>>(dev_in_mat, dev_in_vec, dev_grad_in, dev_grad_out_mat, r, c,
                                                         iwl, num_bit_attention);
    k_appx_grad_vec<<<cdiv(c, kBlock), kBlock, 0, 0>>>(dev_in_mat, dev_in_vec, dev_grad_in, dev_grad_out_vec, r, c, iwl,
                                                     num_bit_attention);
    QM_LAUNCH_CHECK();
}

void cuda_softmax_bwd(float *dev_grad_in, float *dev_out_vec, float *dev_grad_out, float *dev_in_vec, unsigned int dim,
                      bool f_shift_based, bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_in_vec; (void)verbose;
    if (dim == 0) return;
    k_softmax_bwd<<<1, kBlock, 0, 0>>>(dev_out_vec, dev_grad_in, dev_grad_out, dim, f_shift_based);
    QM_LAUNCH_CHECK();
}

void cuda_sum_vec_bwd(float *dev_grad_out, float *dev_grad_in, float *grad_in, float *grad_out, unsigned int dim)
{
    QM_SYNC_WRITES();
    (void)grad_in; (void)grad_out;
    if (dim == 0) return;
    k_copy<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_grad_out, dim);
    QM_LAUNCH_CHECK();
}

void cuda_dense_bwd(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del, float *dev_in_vec,
                    float *dev_out_vec, float *dev_grad_in, float *dev_grad_out, float *dev_f_overflow,
                    unsigned int dim_in, unsigned int dim_out, char *activation, bool f_fixed, unsigned int iwl_in,
                    unsigned int frac_in, unsigned int iwl_w, unsigned int frac_w, unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_bias; (void)dev_bias_del; (void)dev_f_overflow; (void)f_mode; (void)verbose;
    if (dim_in == 0 || dim_out == 0) return;
    const QFmt fw{iwl_w, frac_w}, fi{iwl_in, frac_in}, fg{1u, iwl_w + frac_w - 1u};
    const int act = act_id(activation);
    if (act != kActNull)
        k_act_bwd<<<cdiv(dim_out, kBlock), kBlock, 0, 0>>>(dev_out_vec, dev_grad_in, dev_grad_in, dim_out, act, f_fixed, fw);
    // w_del[o][i] += grad_in[o] . in[i]   (always float: the reference passes f_fixed = false here)
    k_prod<0, true><<<cdiv(dim_out * dim_in, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_vec, dev_w_mat_del,
                                                                    dim_out * dim_in, 1, 0, dim_in, false, fw, fi, fw);
    // grad_out[i] = sum_o w[o][i] . grad_in[o]   (float as well)
    k_prod<1, false><<<cdiv(dim_in, kBlock), kBlock, 0, 0>>>(dev_w_mat, dev_grad_in, dev_grad_out, dim_in, dim_out,
                                                           dim_in, 1, false, fw, fw, fg);
    QM_LAUNCH_CHECK();
}

static void mat_w_up(float *dev_w_mat, float *dev_w_mat_del, float *dev_grad_l2_norm, unsigned dim_in, unsigned dim_out,
                     unsigned batch_size, float lr, float lambda, float max_norm, bool f_fixed, unsigned iwl,
                     unsigned frac)
{
    if (dim_in == 0 || dim_out == 0) return;
    QM_HIP(hipMemsetAsync(dev_grad_l2_norm, 0, sizeof(float), 0));
    k_row_l2_accum<<<dim_out, 64, 0, 0>>>(dev_w_mat_del, dev_grad_l2_norm, dim_in);
    k_mat_w_up<<<cdiv(dim_out * dim_in, kBlock), kBlock, 0, 0>>>(dev_w_mat_del, dev_w_mat, dim_out * dim_in, batch_size,
                                                               lr, lambda, max_norm, dev_grad_l2_norm, f_fixed,
                                                               QFmt{iwl, frac});
    QM_LAUNCH_CHECK();
}

void cuda_dense_w_up(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                     float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm, unsigned int dim_in, unsigned int dim_out,
                     unsigned int batch_size, float *lr, float *lambda, float *max_grad_l2_norm, bool f_fixed,
                     unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_bias; (void)dev_bias_del; (void)dev_grad_bias_l2_norm; (void)f_mode; (void)verbose;
    mat_w_up(dev_w_mat, dev_w_mat_del, dev_grad_l2_norm, dim_in, dim_out, batch_size, *lr, *lambda, *max_grad_l2_norm,
             f_fixed, iwl, frac);
}

void cuda_dense_mat_bwd(float *dev_in_mat, float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                        float *dev_grad_in, float *dev_grad_out, float *dev_f_overflow, unsigned int dim_in,
                        unsigned int dim_out, unsigned int dim_len, bool f_fixed, unsigned int iwl, unsigned int frac,
                        unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_bias; (void)dev_bias_del; (void)dev_f_overflow; (void)f_mode; (void)verbose;
    if (dim_in == 0 || dim_out == 0 || dim_len == 0) return;
    const QFmt f{iwl, frac}, fg{1u, iwl + frac - 1u};
    // w_del[j][k] += sum_s grad_in[s][j] . in_mat[s][k]   (float)
    k_prod<1, true><<<cdiv(dim_out * dim_in, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_mat, dev_w_mat_del,
                                                                    dim_out * dim_in, dim_len, dim_out, dim_in, false, f, f, f);
    // grad_out[s][k] = Qg( sum_j Q(Q(grad_in[s][j]) . Q(w[j][k])) )
    k_prod<0, false><<<cdiv(dim_len * dim_in, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_w_mat, dev_grad_out,
                                                                     dim_len * dim_in, dim_out, 0, dim_in, f_fixed, f, f, fg);
    QM_LAUNCH_CHECK();
}

void cuda_dense_mat_w_up(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                         float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm, float *w_mat, float *w_mat_del,
                         unsigned int dim_in, unsigned int dim_out, unsigned int batch_size, float *lr, float *lambda,
                         float *max_grad_l2_norm, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode,
                         bool verbose)
{
    QM_SYNC_WRITES();
    (void)dev_bias; (void)dev_bias_del; (void)dev_grad_bias_l2_norm; (void)w_mat; (void)w_mat_del; (void)f_mode; (void)verbose;
    mat_w_up(dev_w_mat, dev_w_mat_del, dev_grad_l2_norm, dim_in, dim_out, batch_size, *lr, *lambda, *max_grad_l2_norm,
             f_fixed, iwl, frac);
}

void cuda_activation_bwd(float *dev_out, float *dev_grad_in, float *dev_grad_out, char *type_act, unsigned int dim,
                         bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode)
{
    QM_SYNC_WRITES();
    (void)f_mode;
    if (dim == 0) return;
    k_act_bwd<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_out, dev_grad_in, dev_grad_out, dim, act_id(type_act), f_fixed,
                                                 QFmt{1u, iwl + frac - 1u});
    QM_LAUNCH_CHECK();
}

void cuda_scale_bwd(float *dev_in, float *dev_grad_in, float *dev_w, float *dev_w_del, float *dev_grad_out,
                    unsigned int dim, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)f_fixed; (void)iwl; (void)frac; (void)f_mode; (void)verbose;
    if (dim == 0) return;
    k_scale_bwd<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in, dev_w, dev_w_del, dev_grad_out, dim);
    QM_LAUNCH_CHECK();
}

void cuda_scale_w_up(float *dev_w, float *dev_w_del, unsigned int dim, unsigned int batch_size, float *lr,
                     float *lambda, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose)
{
    QM_SYNC_WRITES();
    (void)f_fixed; (void)iwl; (void)frac; (void)f_mode; (void)verbose;
    k_scalar_w_up<<<1, 1, 0, 0>>>(dev_w, dev_w_del, batch_size * dim, *lr, *lambda);
    QM_LAUNCH_CHECK();
}

void cuda_mult_e_vec_bwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_grad_out_a, float *dev_grad_out_b,
                         float *dev_grad_in, float *grad_in, float *grad_out_a, float *grad_out_b, unsigned int dim)
{
    QM_SYNC_WRITES();
    (void)grad_in; (void)grad_out_a; (void)grad_out_b;
    if (dim == 0) return;
    k_vec_mul<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_vec_b, dev_grad_out_a, dim);
    k_vec_mul<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dev_grad_in, dev_in_vec_a, dev_grad_out_b, dim);
    QM_LAUNCH_CHECK();
}

void cuda_mult_e_mat_bwd(float *dev_in_mat_a, float *dev_in_mat_b, float *dev_grad_out_a, float *dev_grad_out_b,
                         float *dev_grad_in, float *grad_in, float *grad_out_a, float *grad_out_b, unsigned int dim_row,
                         unsigned int dim_col)
{
    QM_SYNC_WRITES();
    cuda_mult_e_vec_bwd(dev_in_mat_a, dev_in_mat_b, dev_grad_out_a, dev_grad_out_b, dev_grad_in, grad_in, grad_out_a,
                        grad_out_b, dim_row * dim_col);
}

void cuda_dup_grad_bwd(float *dev_dup_grad, float *dotmv_dev_grad_out_vec, float *sv_dev_grad_out_vec, float *dup_grad,
                       unsigned int dim, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode)
{
    QM_SYNC_WRITES();
    (void)dup_grad; (void)f_mode;
    if (dim == 0) return;
    k_vec_sum_q<<<cdiv(dim, kBlock), kBlock, 0, 0>>>(dotmv_dev_grad_out_vec, sv_dev_grad_out_vec, dev_dup_grad, dim,
                                                   f_fixed, QFmt{1u, iwl + frac - 1u});
    QM_LAUNCH_CHECK();
}

}  // extern "C"
