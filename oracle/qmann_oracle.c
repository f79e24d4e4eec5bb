/*
 * qmann_oracle.c -- TEST INFRASTRUCTURE ONLY (see qmann_oracle.h).
 *
 * Scalar CPU restatement of the Q-MANN test-phase forward arithmetic.  Written
 * from the behaviour of the reference, not from its text: each function names
 * the reference lines it restates.  Build: oracle/Makefile (gcc -O2
 * -ffp-contract=off; no fast-math, so float sums keep the reference's serial
 * order and rounding).
 */
#include "qmann_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* MemN2N/define.h:67 -- compile-time constant in the reference */
#define QO_ATTENTION_CONST_SCALE (-3)

/* ------------------------------------------------------------------------ */
/* numeric core                                                              */
/* ------------------------------------------------------------------------ */

/* lib/layer_cuda.h:207-211,233 (_CUDA_FLOAT2FIXED, truncation build):
 * two's-complement integer t = trunc(x * 2^frac), symmetric saturation at
 * +-(2^(iwl+frac)-1).  Limits are compared in float exactly as the macro does.
 * A float->int conversion that overflows saturates (the CUDA conversion does;
 * C leaves it undefined) -- only reachable for iwl+frac == 31. */
static int32_t f2f_twos(float x, unsigned iwl, unsigned frac)
{
    const int32_t M = (int32_t)((1u << (iwl + frac)) - 1u);
    const float scale = (float)(1 << frac);
    const float maxf = (float)M / scale;
    const float minf = -1 * maxf;
    if (x > maxf) return M;
    if (x < minf) return -M;
    float p = x * scale;
    if (p >= 2147483648.0f) return INT32_MAX;
    if (p <= -2147483648.0f) return INT32_MIN;
    return (int32_t)p;
}

/* lib/layer_cuda.h:246 (CUDA_FLOAT2FIXED): sign-magnitude word.  A negative x
 * whose magnitude truncates to 0 yields 0x80000000 ("minus zero"). */
int32_t qo_float2fixed(float x, unsigned iwl, unsigned frac)
{
    int32_t t = f2f_twos(x, iwl, frac);
    if (x >= 0.0f) return t & 0x7FFFFFFF;
    return (int32_t)(((uint32_t)(-(int64_t)t)) | 0x80000000u);
}

/* lib/layer_cuda.h:247 (CUDA_FIXED2FLOAT) */
float qo_fixed2float(int32_t w, unsigned frac)
{
    const float scale = (float)(1 << frac);
    if (((uint32_t)w & 0x80000000u) == 0) return (float)w / scale;
    int32_t mag = (int32_t)((uint32_t)w & 0x7FFFFFFFu);
    return (float)(-mag) / scale;
}

/* lib/layer_cuda.h:253 (CUDA_FLOAT_QUANT), host twin lib/common.h:221 */
float qo_quant(float x, unsigned iwl, unsigned frac)
{
    if (iwl + frac == 0) return (x >= 0.0f) ? 1.0f : -1.0f;
    return qo_fixed2float(qo_float2fixed(x, iwl, frac), frac);
}

/* lib/layer_cuda.h:258 (CUDA_FIXED_MUL): result in a's format */
float qo_fixed_mul(float a, float b, unsigned iwl_a, unsigned frac_a,
                   unsigned iwl_b, unsigned frac_b)
{
    float p = qo_quant(a, iwl_a, frac_a) * qo_quant(b, iwl_b, frac_b);
    return qo_quant(p, iwl_a, frac_a);
}

/* lib/layer_cuda.h:257 (CUDA_FIXED_ADD) */
float qo_fixed_add(float a, float b, unsigned iwl_a, unsigned frac_a,
                   unsigned iwl_b, unsigned frac_b)
{
    float s = qo_quant(a, iwl_a, frac_a) + qo_quant(b, iwl_b, frac_b);
    return qo_quant(s, iwl_a, frac_a);
}

int qo_code8(float x, unsigned iwl, unsigned frac)
{
    return (int)f2f_twos(x, iwl, frac);
}

/* ------------------------------------------------------------------------ */
/* Hamming family                                                            */
/* ------------------------------------------------------------------------ */

static inline bool bit_eq(int32_t a, int32_t b, unsigned i)
{
    uint32_t m = 0x80000000u >> i;
    return ((uint32_t)a & m) == ((uint32_t)b & m);
}

/* lib/common.c:223-246 */
unsigned qo_hamming_similarity(int32_t a, int32_t b, unsigned num_bit)
{
    if (num_bit > 32) return 0;
    unsigned n = 0;
    for (unsigned i = 0; i < num_bit; i++)
        if (bit_eq(a, b, i)) n++;
    return n;
}

/* lib/common.c:249-312: bits 1..num_bit-1 weighted 2^-(i+1), signed by the
 * product of the operand signs */
float qo_hamming_similarity_w(int32_t a, int32_t b, unsigned num_bit)
{
    if (num_bit > 32) return 0;
    float sa = (((uint32_t)a & 0x80000000u) == 0) ? 1.0f : -1.0f;
    float sb = (((uint32_t)b & 0x80000000u) == 0) ? 1.0f : -1.0f;
    float acc = 0;
    for (int i = 1; i < (int)num_bit; i++)
        if (bit_eq(a, b, (unsigned)i)) acc += pow(2, (int)(-i - 1));
    return sa * sb * acc;
}

/* lib/layer_cuda.cu:218-326: device variant; weights 2^-i, and the unweighted
 * form also skips bit 0 */
float qo_cuda_hamming_similarity(int32_t a, int32_t b, unsigned num_bit, bool weighted)
{
    float acc = 0.0f;
    if (weighted) {
        int sa = (((uint32_t)a & 0x80000000u) == 0) ? 1 : -1;
        int sb = (((uint32_t)b & 0x80000000u) == 0) ? 1 : -1;
        for (int i = 1; i < (int)num_bit; i++)
            if (bit_eq(a, b, (unsigned)i)) acc += powf(2, (int)(-i));
        return (sa == sb) ? acc : -1.0f * acc;
    }
    for (int i = 1; i < (int)num_bit; i++)
        if (bit_eq(a, b, (unsigned)i)) acc += 1.0f;
    return acc;
}

/* ------------------------------------------------------------------------ */
/* operators                                                                 */
/* ------------------------------------------------------------------------ */

/* lib/layer_cuda.cu:1664-1703 */
void qo_activation_fwd(const float *in, float *out, unsigned dim, const char *act,
                       bool f_fixed, unsigned iwl, unsigned frac)
{
    for (unsigned i = 0; i < dim; i++) {
        float v;
        if (!strcmp(act, "SIGMOID")) v = (float)(1.0 / (1.0 + expf(-in[i])));
        else if (!strcmp(act, "RELU")) v = (in[i] > 0.0f) ? in[i] : 0.0f;
        else v = in[i];
        out[i] = f_fixed ? qo_quant(v, iwl, frac) : v;
    }
}

/* lib/layer_cuda.cu:49-83 launched <<<dim_out,dim_in>>> from :3184 with
 * (iwl_m,frac_m)=(iwl_w,frac_w), (iwl_v,frac_v)=(iwl_in,frac_in); the bias is
 * never added; SIGMOID/RELU post-ops :3204-3208 quantise with the weight format.
 * The iwl_w+frac_w==0 mean|W| rescale (:3188-3200) is not restated: it reads an
 * uninitialised accumulator in the reference. */
void qo_dense_fwd(const float *w, const float *in, float *out,
                  unsigned dim_in, unsigned dim_out, const char *act, bool f_fixed,
                  unsigned iwl_in, unsigned frac_in, unsigned iwl_w, unsigned frac_w)
{
    for (unsigned o = 0; o < dim_out; o++) {
        float sum = 0;
        if (f_fixed) {
            for (unsigned i = 0; i < dim_in; i++)
                sum += qo_fixed_mul(w[o * dim_in + i], in[i], iwl_w, frac_w, iwl_in, frac_in);
            out[o] = qo_quant(sum, iwl_w, frac_w);
        } else {
            for (unsigned i = 0; i < dim_in; i++) {
                float t = w[o * dim_in + i] * in[i];
                sum += t;
            }
            out[o] = sum;
        }
    }
    if (act && (!strcmp(act, "SIGMOID") || !strcmp(act, "RELU")))
        qo_activation_fwd(out, out, dim_out, act, f_fixed, iwl_w, frac_w);
}

/* lib/layer_cuda.cu:105-172 launched <<<dim_len*dim_out,dim_in>>> from :3531
 * (mat_a = in_mat rows, mat_b = weight rows, every format = (iwl,frac)).
 * The live CPU branch lib/layer.c:2671-2696 computes the same expression. */
void qo_dense_mat_fwd(const float *w, const float *in_mat, float *out_mat,
                      unsigned dim_in, unsigned dim_out, unsigned dim_len,
                      bool f_fixed, unsigned iwl, unsigned frac)
{
    for (unsigned s = 0; s < dim_len; s++)
        for (unsigned j = 0; j < dim_out; j++) {
            float sum = 0;
            if (f_fixed) {
                for (unsigned k = 0; k < dim_in; k++)
                    sum += qo_fixed_mul(in_mat[s * dim_in + k], w[j * dim_in + k],
                                        iwl, frac, iwl, frac);
                out_mat[s * dim_out + j] = qo_quant(sum, iwl, frac);
            } else {
                for (unsigned k = 0; k < dim_in; k++) {
                    float t = in_mat[s * dim_in + k] * w[j * dim_in + k];
                    sum += t;
                }
                out_mat[s * dim_out + j] = sum;
            }
        }
}

/* lib/layer_cuda.cu:2429-2438.
 * non-trans: _cuda_mat_mat_trans_product<<<r,c>>>(mat, vec, out, 1, ..., m, v, out=m)
 * trans    : _cuda_mat_trans_mat_product<<<c,r>>>(vec, mat, out, 1, c, ..., m, out=m)
 *            where the product is Qm(Qm(vec[r]) * Qm(mat[r][c])) (:562). */
void qo_dot_mat_vec_fwd(const float *mat, const float *vec, float *out,
                        unsigned r, unsigned c, bool f_trans, bool f_fixed,
                        unsigned iwl_m, unsigned frac_m, unsigned iwl_v, unsigned frac_v)
{
    if (!f_trans) {
        for (unsigned i = 0; i < r; i++) {
            float sum = 0;
            if (f_fixed) {
                for (unsigned j = 0; j < c; j++)
                    sum += qo_fixed_mul(mat[i * c + j], vec[j], iwl_m, frac_m, iwl_v, frac_v);
                out[i] = qo_quant(sum, iwl_m, frac_m);
            } else {
                for (unsigned j = 0; j < c; j++) {
                    float t = mat[i * c + j] * vec[j];
                    sum += t;
                }
                out[i] = sum;
            }
        }
    } else {
        for (unsigned j = 0; j < c; j++) {
            float sum = 0;
            if (f_fixed) {
                for (unsigned i = 0; i < r; i++)
                    sum += qo_fixed_mul(vec[i], mat[i * c + j], iwl_m, frac_m, iwl_m, frac_m);
                out[j] = qo_quant(sum, iwl_m, frac_m);
            } else {
                for (unsigned i = 0; i < r; i++) {
                    float t = vec[i] * mat[i * c + j];
                    sum += t;
                }
                out[j] = sum;
            }
        }
    }
}

/* One element pair of lib/layer_cuda.cu:355-520 ("V2"): operands re-encoded
 * with frac = 31-iwl, common magnitude removed (same sign) or moved onto the
 * larger operand (opposite sign; the 32-bit add may carry into the sign bit),
 * weighted bit agreement over bits 1..num_bit-1, scaled by 2^-3, quantised. */
static float appx_pair(float a, float b, unsigned iwl, unsigned num_bit)
{
    const unsigned frac = 31 - iwl;
    uint32_t fa = (uint32_t)qo_float2fixed(a, iwl, frac);
    uint32_t fb = (uint32_t)qo_float2fixed(b, iwl, frac);
    uint32_t sa = fa & 0x80000000u, sb = fb & 0x80000000u;
    uint32_t ma = fa & 0x7FFFFFFFu, mb = fb & 0x7FFFFFFFu;
    uint32_t mn = (ma >= mb) ? mb : ma;
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
    float sim = qo_cuda_hamming_similarity((int32_t)fa, (int32_t)fb, num_bit, true);
    float t = sim * powf(2, (int)QO_ATTENTION_CONST_SCALE);
    return qo_quant(t, iwl, frac);
}

/* lib/layer_cuda.cu:2491-2517: the non-trans kernel ignores the caller's frac
 * and uses 32-1-iwl (:2515); the trans branch is the ordinary read-out. */
void qo_dot_mat_vec_fwd_appx(const float *mat, const float *vec, float *out,
                             unsigned r, unsigned c, bool f_fixed,
                             unsigned iwl, unsigned frac, unsigned num_bit, bool f_trans)
{
    if (f_trans) {
        qo_dot_mat_vec_fwd(mat, vec, out, r, c, true, f_fixed, iwl, frac, iwl, frac);
        return;
    }
    for (unsigned i = 0; i < r; i++) {
        float sum = 0;
        for (unsigned j = 0; j < c; j++)
            sum += appx_pair(mat[i * c + j], vec[j], iwl, num_bit);
        out[i] = qo_quant(sum, iwl, 31 - iwl);
    }
}

/* Row scorer over the live CPU hamming functions (shape of the dead caller
 * lib/layer.c:330-340; frac_code chooses the word alignment -- 31-iwl is the
 * CUDA alignment, 7-iwl the degenerate right-aligned one the dead caller used) */
void qo_attention_hamming(const float *mat, const float *vec, float *out,
                          unsigned r, unsigned c, unsigned iwl, unsigned frac_code,
                          unsigned num_bit, int variant)
{
    for (unsigned i = 0; i < r; i++) {
        float sum = 0;
        for (unsigned j = 0; j < c; j++) {
            int32_t a = qo_float2fixed(mat[i * c + j], iwl, frac_code);
            int32_t b = qo_float2fixed(vec[j], iwl, frac_code);
            if (variant == 0) sum += (float)qo_hamming_similarity(a, b, num_bit);
            else sum += qo_hamming_similarity_w(a, b, num_bit);
        }
        out[i] = sum;
    }
}

/* lib/common.c:51-73 with the table lib/common.h:270-286 */
static float exp_plan(float in)
{
    static const float wt[4] = { 0.597226f, 0.141642f, 0.070265f, 0.0f };
    static const float bt[4] = { 0.933989f, 0.43981f, 0.10888f, 0.0f };
    float out = wt[0] * in + bt[0];
    for (unsigned i = 1; i < 4; i++) {
        float tmp = wt[i] * in + bt[i];
        if (out < tmp) out = tmp;
    }
    return out;
}

void qo_softmax_fwd(const float *in, float *out, unsigned dim, int variant, bool f_shift_based)
{
    if (dim == 0) return;
    float max = in[0];
    for (unsigned i = 1; i < dim; i++)
        if (in[i] > max) max = in[i];

    if (variant == QO_SM_CUDA) {
        /* lib/layer_cuda.cu:1969-2060 (max from :1895-1916): both branches use
         * exp(x-max); total is a double summed serially by thread 0 */
        double total = 0.0;
        for (unsigned i = 0; i < dim; i++) {
            out[i] = expf(in[i] - max);
            total += out[i];
        }
        if (f_shift_based) {
            long long d = llrintf(log2f((float)total));
            for (unsigned i = 0; i < dim; i++) out[i] = out[i] / d;
        } else {
            for (unsigned i = 0; i < dim; i++) out[i] = (float)(out[i] / total);
        }
        return;
    }
    /* lib/layer.c:1184-1244 (live CPU branch) */
    float tot = 0.0f;
    for (unsigned i = 0; i < dim; i++) {
        if (variant == QO_SM_CPU_EXP_PLAN) out[i] = exp_plan(in[i] - max);
        else if (f_shift_based) out[i] = (float)pow(2, in[i] - max + 1.0);
        else out[i] = (float)pow(2, in[i] - max);
        tot += out[i];
    }
    for (unsigned i = 0; i < dim; i++) out[i] = out[i] / tot;
}

/* lib/layer_cuda.cu:1535-1542; CPU lib/layer.c:1502-1511 */
void qo_sum_vec_fwd(const float *a, const float *b, float *out, unsigned dim,
                    bool f_fixed, unsigned iwl, unsigned frac)
{
    for (unsigned i = 0; i < dim; i++)
        out[i] = f_fixed ? qo_fixed_add(a[i], b[i], iwl, frac, iwl, frac) : a[i] + b[i];
}

/* lib/layer_cuda.cu:1918-1939: pairwise tree over thread indices; the left
 * candidate survives only when strictly greater, so ties go to the highest index */
unsigned qo_argmax_hi(const float *in, unsigned dim)
{
    if (dim == 0) return 0;
    unsigned *idx = (unsigned *)malloc(dim * sizeof(unsigned));
    for (unsigned i = 0; i < dim; i++) idx[i] = i;
    for (unsigned step = 1; step < dim; step *= 2)
        for (unsigned t = 0; t < dim; t++)
            if ((t % (2 * step) == 0) && (t + step < dim))
                if (!(in[idx[t]] > in[idx[t + step]])) idx[t] = idx[t + step];
    unsigned r = idx[0];
    free(idx);
    return r;
}

/* lib/layer_cuda.cu:3773-3783 with :2191-2251: cost += -h[y], m_cnt += (pred==y) */
unsigned qo_cross_entropy_run(const float *h, const float *y, unsigned dim,
                              float *cost_acc, unsigned *m_cnt_acc, float *grad_out)
{
    unsigned pred = qo_argmax_hi(h, dim);
    for (unsigned i = 0; i < dim; i++) {
        if (y[i] == 1.0f) {
            if (cost_acc) *cost_acc += -1.0 * h[i];
            if (m_cnt_acc && i == pred) *m_cnt_acc += 1;
        }
        if (grad_out) grad_out[i] = (y[i] == 1.0f) ? (float)(1.0 - h[i]) : -h[i];
    }
    return pred;
}

/* ------------------------------------------------------------------------ */
/* training verbs (restated from CUDA text only)                             */
/* ------------------------------------------------------------------------ */

/* lib/layer_cuda.cu:695-738 (_cuda_mat_mat_product): out[R][C] = Qo(sum_t mul(a[R][t], b[t][C])) */
static void mat_mat(const float *a, const float *b, float *out, unsigned R, unsigned C, unsigned K, bool fixed,
                    unsigned iwl, unsigned frac, unsigned iwl_o, unsigned frac_o, bool accum)
{
    for (unsigned r = 0; r < R; r++)
        for (unsigned c = 0; c < C; c++) {
            float sum = 0;
            for (unsigned t = 0; t < K; t++) {
                float p = fixed ? qo_fixed_mul(a[r * K + t], b[c + C * t], iwl, frac, iwl, frac)
                                : a[r * K + t] * b[c + C * t];
                sum += p;
            }
            float v = fixed ? qo_quant(sum, iwl_o, frac_o) : sum;
            if (accum) out[r * C + c] += v; else out[r * C + c] = v;
        }
}

/* lib/layer_cuda.cu:547-690 (_cuda_mat_trans_mat_product[_accum]): out[R][C] = sum_t a[t][R] . b[t][C] */
static void mat_trans_mat(const float *a, const float *b, float *out, unsigned R, unsigned C, unsigned K, bool fixed,
                          unsigned iwl, unsigned frac, unsigned iwl_o, unsigned frac_o, bool accum)
{
    for (unsigned r = 0; r < R; r++)
        for (unsigned c = 0; c < C; c++) {
            float sum = 0;
            for (unsigned t = 0; t < K; t++) {
                float p = fixed ? qo_fixed_mul(a[r + R * t], b[c + C * t], iwl, frac, iwl, frac)
                                : a[r + R * t] * b[c + C * t];
                sum += p;
            }
            float v = fixed ? qo_quant(sum, iwl_o, frac_o) : sum;
            if (accum) out[r * C + c] += v; else out[r * C + c] = v;
        }
}

void qo_dot_mat_vec_bwd(const float *mat, const float *vec, const float *grad_in, float *grad_out_mat,
                        float *grad_out_vec, unsigned r, unsigned c, bool f_trans, bool f_fixed,
                        unsigned iwl_m, unsigned frac_m)
{
    const unsigned io = 1, fo = iwl_m + frac_m - 1;           /* gradients leave in Q(1.wl-2) */
    if (f_trans) {
        mat_mat(vec, grad_in, grad_out_mat, r, c, 1, f_fixed, iwl_m, frac_m, io, fo, false);      /* :2586 */
        for (unsigned i = 0; i < r; i++) {                                                        /* :2587 */
            float sum = 0;
            for (unsigned j = 0; j < c; j++) {
                float p = f_fixed ? qo_fixed_mul(mat[i * c + j], grad_in[j], iwl_m, frac_m, iwl_m, frac_m)
                                  : mat[i * c + j] * grad_in[j];
                sum += p;
            }
            grad_out_vec[i] = f_fixed ? qo_quant(sum, io, fo) : sum;
        }
    } else {
        mat_mat(grad_in, vec, grad_out_mat, r, c, 1, f_fixed, iwl_m, frac_m, io, fo, false);      /* :2595 */
        mat_trans_mat(grad_in, mat, grad_out_vec, 1, c, r, f_fixed, iwl_m, frac_m, io, fo, false); /* :2596 */
    }
}

/* the operand transform shared by forward and backward of the approximate attention (:383-420, :777-817) */
static void appx_transform(float m, float v, unsigned iwl, uint32_t *fm, uint32_t *fv, float *sm, float *sv)
{
    const unsigned frac = 31 - iwl;
    uint32_t a = (uint32_t)qo_float2fixed(m, iwl, frac), b = (uint32_t)qo_float2fixed(v, iwl, frac);
    *sm = ((int32_t)a >= 0) ? 1.0f : -1.0f;
    *sv = ((int32_t)b >= 0) ? 1.0f : -1.0f;
    uint32_t sa = a & 0x80000000u, sb = b & 0x80000000u, ma = a & 0x7FFFFFFFu, mb = b & 0x7FFFFFFFu;
    uint32_t mn = ma >= mb ? mb : ma;
    if (*sm == *sv) { a = sa | (ma - mn); b = sb | (mb - mn); }
    else if (ma >= mb) { a = sa | (ma + mn); b = sb; }
    else { a = sa; b = sb | (mb + mn); }
    *fm = a; *fv = b;
}

void qo_dot_mat_vec_bwd_appx(const float *mat, const float *vec, const float *grad_in, float *grad_out_mat,
                             float *grad_out_vec, unsigned r, unsigned c, bool f_fixed, unsigned iwl,
                             unsigned frac, unsigned num_bit, bool f_trans)
{
    if (f_trans) {
        qo_dot_mat_vec_bwd(mat, vec, grad_in, grad_out_mat, grad_out_vec, r, c, true, f_fixed, iwl, frac);
        return;
    }
    const float k = powf(2, (int)QO_ATTENTION_CONST_SCALE);
    for (unsigned i = 0; i < r; i++)                               /* _cuda_backprop_grad_out_mat */
        for (unsigned j = 0; j < c; j++) {
            uint32_t fm, fv; float sm, sv;
            appx_transform(mat[i * c + j], vec[j], iwl, &fm, &fv, &sm, &sv);
            float ta = 0.0f;
            for (unsigned b = 0; b < num_bit; b++) {
                int bm = (int)((fm >> (31 - b)) & 1u), bv = (int)((fv >> (31 - b)) & 1u);
                if (bm != bv) {
                    if (b == 0) ta += (bm - bv) * sm * k;
                    else ta += -1.0 * sv * k * (bm - bv);
                }
            }
            grad_out_mat[i * c + j] = ta * grad_in[i];
        }
    for (unsigned j = 0; j < c; j++) {                             /* _cuda_backprop_grad_out_vec */
        float sum = 0.0f;
        for (unsigned i = 0; i < r; i++) {
            uint32_t fm, fv; float sm, sv;
            appx_transform(mat[i * c + j], vec[j], iwl, &fm, &fv, &sm, &sv);
            float ta = 0.0f, ga = 0.0f;                            /* ta is NOT reset inside the bit loop */
            for (unsigned b = 0; b < num_bit; b++) {
                int bm = (int)((fm >> (31 - b)) & 1u), bv = (int)((fv >> (31 - b)) & 1u);
                if (bm != bv) {
                    if (b == 0) ta = -1.0 * (bm - bv) * sv * k;
                    else ta = 1.0 * sm * k * (bm - bv);
                }
                ga += ta;
            }
            float t = ga * grad_in[i];
            sum += t;
        }
        grad_out_vec[j] = sum;
    }
}

void qo_softmax_bwd(const float *out_vec, const float *grad_in, float *grad_out, unsigned dim, bool f_shift_based)
{
    float sum = 0.0f;
    for (unsigned i = 0; i < dim; i++) { float t = out_vec[i] * grad_in[i]; sum += t; }
    for (unsigned i = 0; i < dim; i++) {
        if (f_shift_based) grad_out[i] = 0.7 * out_vec[i] * (grad_in[i] - sum);
        else grad_out[i] = out_vec[i] * (grad_in[i] - sum);
    }
}

void qo_dense_bwd(const float *w, float *w_del, const float *in, const float *out, float *grad_in, float *grad_out,
                  unsigned dim_in, unsigned dim_out, const char *act, bool f_fixed, unsigned iwl_w, unsigned frac_w)
{
    if (act && !strcmp(act, "SIGMOID"))
        for (unsigned i = 0; i < dim_out; i++) {
            double g = grad_in[i] * out[i] * (1.0 - out[i]);
            grad_in[i] = f_fixed ? qo_quant((float)g, iwl_w, frac_w) : (float)g;
        }
    else if (act && !strcmp(act, "RELU"))
        for (unsigned i = 0; i < dim_out; i++) {
            float g = out[i] > 0.0f ? grad_in[i] : 0.0f;
            grad_in[i] = f_fixed ? qo_quant(g, iwl_w, frac_w) : g;
        }
    mat_mat(grad_in, in, w_del, dim_out, dim_in, 1, false, 0, 0, 0, 0, true);                  /* :3276 */
    mat_trans_mat(w, grad_in, grad_out, dim_in, 1, dim_out, false, 0, 0, 0, 0, false);         /* :3284 */
}

void qo_dense_mat_bwd(const float *in_mat, const float *w, float *w_del, const float *grad_in, float *grad_out,
                      unsigned dim_in, unsigned dim_out, unsigned dim_len, bool f_fixed, unsigned iwl, unsigned frac)
{
    mat_trans_mat(grad_in, in_mat, w_del, dim_out, dim_in, dim_len, false, 0, 0, 0, 0, true);  /* :3592 */
    mat_mat(grad_in, w, grad_out, dim_len, dim_in, dim_out, f_fixed, iwl, frac, 1, iwl + frac - 1, false); /* :3600 */
}

float qo_mat_w_up(float *w, float *w_del, unsigned dim_in, unsigned dim_out, unsigned batch_size, float lr,
                  float lambda, float max_grad_l2_norm, bool f_fixed, unsigned iwl, unsigned frac)
{
    float norm = 0.0f;                                           /* sum over rows of the row 2-norms (:1596-1622) */
    for (unsigned o = 0; o < dim_out; o++) {
        float sum = 0.0f;
        for (unsigned i = 0; i < dim_in; i++) { float t = w_del[o * dim_in + i] * w_del[o * dim_in + i]; sum += t; }
        norm += sqrtf(sum);
    }
    for (unsigned k = 0; k < dim_in * dim_out; k++) {
        float step = (norm > max_grad_l2_norm) ? lr / batch_size * w_del[k] * max_grad_l2_norm / norm
                                               : lr / batch_size * w_del[k];
        float decay = lr * lambda * w[k];
        if (f_fixed) {
            w[k] += qo_quant(step, iwl, frac) + qo_quant(decay, iwl, frac);
            w[k] = qo_quant(w[k], iwl, frac);
        } else {
            w[k] += step + decay;
        }
        w_del[k] = 0.0f;
    }
    return norm;
}

void qo_dup_grad_bwd(const float *a, const float *b, float *out, unsigned dim, bool f_fixed, unsigned iwl, unsigned frac)
{
    qo_sum_vec_fwd(a, b, out, dim, f_fixed, 1, iwl + frac - 1);
}

/* ------------------------------------------------------------------------ */
/* composite forward                                                         */
/* ------------------------------------------------------------------------ */

/* One hop, MemN2N/MemN2N.c:2644-2666 with the mode dispatch lib/layer.c:176-233. */
static void hop_forward(const qo_model *m, unsigned h, const float *keys, const float *vals,
                        unsigned n_sen, float *u, qo_taps *taps)
{
    const unsigned D = m->dim_emb;
    float *s = (float *)malloc((n_sen ? n_sen : 1) * sizeof(float));
    float *p = (float *)malloc((n_sen ? n_sen : 1) * sizeof(float));
    float *o = (float *)malloc(D * sizeof(float));
    float *lu = (float *)malloc(D * sizeof(float));

    /* EN_NON_LINEARITY: from the second hop on the attention reads non_lin[h-1].out = RELU(sv[h-1])
     * (MemN2N.c:2435-2437, constructor :894-896), while lin_map[h] keeps reading sv[h-1] itself (:2471-2473) */
    const float *u_sv = u;
    float *u_relu = NULL;
    if (m->en_non_lin && h > 0) {
        u_relu = (float *)malloc(D * sizeof(float));
        qo_activation_fwd(u_sv, u_relu, D, "RELU", m->f_fixed, m->iwl[h - 1], m->frac[h - 1]);
    }
    const float *u_att = u_relu ? u_relu : u_sv;
    /* attention scores: dotmv[h] (constructor MemN2N.c:846-850) */
    if (m->attention_mode == 1)
        qo_dot_mat_vec_fwd(keys, u_att, s, n_sen, D, false, false, 0, 0, 0, 0);
    else if (m->attention_mode == 2)
        qo_dot_mat_vec_fwd(keys, u_att, s, n_sen, D, false, true,
                           m->iwl_att[h], m->frac_att[h], m->iwl_bin, m->frac_bin);
    else if (m->attention_mode == 3)
        qo_dot_mat_vec_fwd_appx(keys, u_att, s, n_sen, D, m->f_fixed, m->iwl_att[h], m->frac_att[h],
                                1 + m->iwl_att[h] + m->frac_att[h], false);
    else /* 10 / 11: the CPU hamming functions on the CUDA word alignment (frac = 31 - iwl) */
        qo_attention_hamming(keys, u_att, s, n_sen, D, m->iwl_att[h], 31 - m->iwl_att[h], m->num_bit,
                             m->attention_mode == 10 ? 0 : 1);
    free(u_relu);
    if (taps && taps->scores) memcpy(taps->scores + (size_t)h * n_sen, s, n_sen * sizeof(float));
    /* optional scale layer sc_sf_in[h] (MemN2N.c:2647-2649): plain float product with one scalar */
    if (m->en_sc_att)
        for (unsigned i = 0; i < n_sen; i++) s[i] = s[i] * m->sc_att[h];
    /* softmax over slots: sf_in[h] (MemN2N.c:2651; constructor :856 carries f_shift_based) */
    qo_softmax_fwd(s, p, n_sen, m->softmax_variant, m->f_shift_based);
    /* weighted read-out: w_sum[h] (constructor MemN2N.c:863, formats (iwl[h],frac[h])) */
    if (m->attention_mode == 1)
        qo_dot_mat_vec_fwd(vals, p, o, n_sen, D, true, false, 0, 0, 0, 0);
    else if (m->attention_mode == 2 || m->attention_mode >= 10)
        qo_dot_mat_vec_fwd(vals, p, o, n_sen, D, true, true,
                           m->iwl[h], m->frac[h], m->iwl[h], m->frac[h]);
    else
        qo_dot_mat_vec_fwd_appx(vals, p, o, n_sen, D, m->f_fixed, m->iwl[h], m->frac[h],
                                1 + m->iwl[h] + m->frac[h], true);
    /* lin_map[h] (constructor MemN2N.c:873: in=(iwl_bin,frac_bin), w=(iwl_w,frac_w)) */
    if (m->en_lin_map)
        qo_dense_fwd(m->w_h[h], u, lu, D, D, "NULL", m->f_fixed,
                     m->iwl_bin, m->frac_bin, m->iwl_w[h], m->frac_w[h]);
    else
        memcpy(lu, u, D * sizeof(float));
    if (taps) {
        if (taps->probs) memcpy(taps->probs + (size_t)h * n_sen, p, n_sen * sizeof(float));
        if (taps->o) memcpy(taps->o + (size_t)h * D, o, D * sizeof(float));
        if (taps->lu) memcpy(taps->lu + (size_t)h * D, lu, D * sizeof(float));
    }
    /* sv[h] (constructor MemN2N.c:889): u' = Q(Q(lu)+Q(o)) */
    qo_sum_vec_fwd(lu, o, u, D, m->f_fixed, m->iwl[h], m->frac[h]);
    if (taps && taps->u) memcpy(taps->u + (size_t)h * D, u, D * sizeof(float));
    free(s); free(p); free(o); free(lu);
}

/* ds_ans (float, MemN2N.c:902-906) -> sf_out (:910) -> cross_entropy arg-max (:2697) */
static unsigned answer_forward(const qo_model *m, const float *u, qo_taps *taps)
{
    const unsigned D = m->dim_emb, V = m->dim_input;
    float *a = (float *)malloc(V * sizeof(float));
    float *ph = (float *)malloc(V * sizeof(float));
    float *u_in = (float *)malloc(D * sizeof(float));
    /* with EN_NON_LINEARITY the answer layer reads non_lin[NUM_HOP-1].out (MemN2N.c:2535-2537) */
    if (m->en_non_lin) qo_activation_fwd(u, u_in, D, "RELU", m->f_fixed, m->iwl[m->n_hop - 1], m->frac[m->n_hop - 1]);
    else memcpy(u_in, u, D * sizeof(float));
    qo_dense_fwd(m->w_ans, u_in, a, D, V, "NULL", false, 0, 0, 0, 0);
    free(u_in);
    qo_softmax_fwd(a, ph, V, m->softmax_variant, false);
    unsigned pred = qo_argmax_hi(ph, V);
    if (taps) {
        if (taps->logits) memcpy(taps->logits, a, V * sizeof(float));
        if (taps->out_probs) memcpy(taps->out_probs, ph, V * sizeof(float));
    }
    free(a); free(ph);
    return pred;
}

unsigned qo_memn2n_forward_mem(const qo_model *m, const float *keys, const float *vals,
                               unsigned n_sen, const float *u0, qo_taps *taps)
{
    const unsigned D = m->dim_emb;
    float *u = (float *)malloc(D * sizeof(float));
    memcpy(u, u0, D * sizeof(float));
    for (unsigned h = 0; h < m->n_hop; h++)
        hop_forward(m, h, keys + (size_t)h * n_sen * D, vals + (size_t)h * n_sen * D, n_sen, u, taps);
    unsigned pred = answer_forward(m, u, taps);
    free(u);
    return pred;
}

unsigned qo_memn2n_forward(const qo_model *m, const float *story, unsigned n_sen,
                           const float *question, qo_taps *taps)
{
    const unsigned D = m->dim_emb, V = m->dim_input;
    const size_t msz = (size_t)(n_sen ? n_sen : 1) * D;
    float *u = (float *)malloc(D * sizeof(float));
    float *keys = (float *)malloc(msz * sizeof(float));
    float *vals = (float *)malloc(msz * sizeof(float));
    /* emb_q (constructor MemN2N.c:826: in and w both (iwl_w[0],frac_w[0])) */
    qo_dense_fwd(m->w_q, question, u, V, D, "NULL", m->f_fixed,
                 m->iwl_w[0], m->frac_w[0], m->iwl_w[0], m->frac_w[0]);
    if (taps && taps->u0) memcpy(taps->u0, u, D * sizeof(float));
    for (unsigned h = 0; h < m->n_hop; h++) {
        /* emb_m[h], emb_c[h] (constructors MemN2N.c:835,838) */
        qo_dense_mat_fwd(m->w_a[h], story, keys, V, D, n_sen, m->f_fixed, m->iwl_w[h], m->frac_w[h]);
        qo_dense_mat_fwd(m->w_c[h], story, vals, V, D, n_sen, m->f_fixed, m->iwl_w[h], m->frac_w[h]);
        if (taps) {
            if (taps->keys) memcpy(taps->keys + (size_t)h * n_sen * D, keys, (size_t)n_sen * D * sizeof(float));
            if (taps->vals) memcpy(taps->vals + (size_t)h * n_sen * D, vals, (size_t)n_sen * D * sizeof(float));
        }
        hop_forward(m, h, keys, vals, n_sen, u, taps);
    }
    unsigned pred = answer_forward(m, u, taps);
    free(u); free(keys); free(vals);
    return pred;
}

/* ------------------------------------------------------------------------ */
/* batch driver for the tests: many stories given as word lists, on threads  */
/* ------------------------------------------------------------------------ */
#include <pthread.h>

/* uint16 word lists -> the float bag-of-words rows sample.c builds (MemN2N/sample.c:466-475, 544-548): word entries COUNT
 * occurrences, the last valid entry of a STORY row is its time index and is SET to 1; 0xFFFF = unused slot; words outside
 * the dictionary are ignored.  (Position encoding is not handled here: the tests that use EN_PE build their rows themselves.) */
static void words_to_row(const uint16_t *w, unsigned width, unsigned V, int with_time, float *row)
{
    memset(row, 0, V * sizeof(float));
    int last = -1;
    for (unsigned k = 0; k < width; k++)
        if (w[k] != 0xFFFFu) last = (int)k;
    for (int k = 0; k <= last; k++) {
        if (w[k] == 0xFFFFu || w[k] >= V) continue;
        if (with_time && k == last) row[w[k]] = 1.0f;
        else row[w[k]] += 1.0f;
    }
}

typedef struct {
    const qo_model *m;
    const uint16_t *sw, *qw;
    unsigned sw_width, qw_width;
    const uint32_t *row_off;
    unsigned n_query, n_threads, tid;
    uint32_t *pred;
    float *u_final, *top2_gap;
    uint8_t *near_step;
} qo_batch_job;

static void *qo_batch_worker(void *p)
{
    const qo_batch_job *j = (const qo_batch_job *)p;
    const qo_model *m = j->m;
    const unsigned V = m->dim_input, D = m->dim_emb, H = m->n_hop;
    unsigned max_sen = 1;
    for (unsigned q = 0; q < j->n_query; q++)
        if (j->row_off[q + 1] - j->row_off[q] > max_sen) max_sen = j->row_off[q + 1] - j->row_off[q];
    float *story = (float *)malloc((size_t)max_sen * V * sizeof(float));
    float *ques = (float *)malloc(V * sizeof(float));
    float *probs = (float *)malloc((size_t)H * max_sen * sizeof(float));
    float *u = (float *)malloc((size_t)H * D * sizeof(float));
    float *outp = (float *)malloc(V * sizeof(float));
    for (unsigned q = j->tid; q < j->n_query; q += j->n_threads) {
        const unsigned r0 = j->row_off[q], ns = j->row_off[q + 1] - r0;
        for (unsigned s = 0; s < ns; s++) words_to_row(j->sw + (size_t)(r0 + s) * j->sw_width, j->sw_width, V, 1, story + (size_t)s * V);
        words_to_row(j->qw + (size_t)q * j->qw_width, j->qw_width, V, 0, ques);
        qo_taps t;
        memset(&t, 0, sizeof t);
        t.probs = probs; t.u = u; t.out_probs = outp;
        j->pred[q] = qo_memn2n_forward(m, story, ns, ques, &t);
        if (j->u_final) memcpy(j->u_final + (size_t)q * D, u + (size_t)(H - 1) * D, D * sizeof(float));
        if (j->top2_gap) {
            float a = -1.0f, b = -1.0f;
            for (unsigned i = 0; i < V; i++) {
                if (outp[i] > a) { b = a; a = outp[i]; }
                else if (outp[i] > b) b = outp[i];
            }
            j->top2_gap[q] = a - b;
        }
        if (j->near_step) {
            /* does a softmax weight of some hop sit within 1e-5 (relative) of a truncation step of Q(act[h])?  There the
             * 1e-5 float tolerance of the softmax may move Q(p) by one code, and only there may a hop output differ */
            uint8_t near = 0;
            for (unsigned h = 0; h < H; h++)
                for (unsigned s = 0; s < ns; s++) {
                    const double x = (double)probs[(size_t)h * ns + s] * (double)(1u << m->frac[h]);
                    const double k = (double)(long long)(x + 0.5);
                    const double tol = 1e-5 * (x > 1.0 ? x : 1.0);
                    if (k > 0.0 && (x > k ? x - k : k - x) <= tol) near = 1;
                }
            j->near_step[q] = near;
        }
    }
    free(story); free(ques); free(probs); free(u); free(outp);
    return NULL;
}

void qo_memn2n_forward_words_batch(const qo_model *m, const uint16_t *story_words, unsigned sw_width, const uint16_t *question_words,
                                   unsigned qw_width, const uint32_t *row_off, unsigned n_query, unsigned n_threads, uint32_t *pred,
                                   float *u_final, float *top2_gap, uint8_t *near_step)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    qo_batch_job jobs[64];
    for (unsigned t = 0; t < n_threads; t++) {
        qo_batch_job j = {m, story_words, question_words, sw_width, qw_width, row_off, n_query, n_threads, t, pred, u_final, top2_gap, near_step};
        jobs[t] = j;
        pthread_create(&th[t], NULL, qo_batch_worker, &jobs[t]);
    }
    for (unsigned t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}
