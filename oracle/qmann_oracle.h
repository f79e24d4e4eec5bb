/*
 * qmann_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, scalar, single thread) of the arithmetic on the
 * test-phase forward path of seongsikpark/Q-MANN.  It is the *checker* that the
 * HIP kernels are compared against.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link, load or call anything in this
 * directory; the product library (q-mann_amd/csrc) never does.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   pinned by outputs of the reference's own live C code run in this container
 *   (oracle/_ref, fixtures in tests/golden/ref_*.npz):
 *       quantiser macros, hamming_similarity{,_w}, softmax_fwd CPU branch,
 *       sum_vec_fwd, dense_mat_fwd, cross_entropy_run CPU cost/grad,
 *       activation_fwd.
 *   restated from CUDA source text that cannot be built here (no nvcc, no
 *   NVIDIA GPU) and therefore PARITY UNPINNED beyond the shared pieces above:
 *       _cuda_softmax_fwd's exp base / double total, _cuda_approximate_attention
 *       (Hamming "V2"), _cuda_max/_cuda_max_i tie rule, _cuda_cross_entropy_cost.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef QMANN_ORACLE_H
#define QMANN_ORACLE_H

#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- numeric core: lib/layer_cuda.h:207-259, lib/common.h:178-227 ---- */
int32_t  qo_float2fixed(float x, unsigned iwl, unsigned frac);   /* sign-magnitude word */
float    qo_fixed2float(int32_t w, unsigned frac);
float    qo_quant(float x, unsigned iwl, unsigned frac);          /* FLOAT_QUANT */
float    qo_fixed_mul(float a, float b, unsigned iwl_a, unsigned frac_a,
                      unsigned iwl_b, unsigned frac_b);           /* CUDA_FIXED_MUL */
float    qo_fixed_add(float a, float b, unsigned iwl_a, unsigned frac_a,
                      unsigned iwl_b, unsigned frac_b);           /* CUDA_FIXED_ADD */
/* two's-complement int8 code of Q(x) for word length 8 (|code| <= 127) */
int      qo_code8(float x, unsigned iwl, unsigned frac);

/* ---- Hamming family: lib/common.c:223-312, lib/layer_cuda.cu:218-326 ---- */
unsigned qo_hamming_similarity(int32_t a, int32_t b, unsigned num_bit);          /* V0 */
float    qo_hamming_similarity_w(int32_t a, int32_t b, unsigned num_bit);        /* V1 (CPU weights 2^-(i+1)) */
float    qo_cuda_hamming_similarity(int32_t a, int32_t b, unsigned num_bit, bool weighted); /* CUDA weights 2^-i */

/* ---- operators (float in / float out, row-major) ---- */
/* lib/layer_cuda.cu:49-83 via cuda_dense_fwd :3163-3208.  act: "NULL"|"SIGMOID"|"RELU" */
void qo_dense_fwd(const float *w, const float *in, float *out,
                  unsigned dim_in, unsigned dim_out, const char *act, bool f_fixed,
                  unsigned iwl_in, unsigned frac_in, unsigned iwl_w, unsigned frac_w);
/* lib/layer_cuda.cu:105-172 via cuda_dense_mat_fwd :3512-3531; CPU twin lib/layer.c:2671-2696 */
void qo_dense_mat_fwd(const float *w, const float *in_mat, float *out_mat,
                      unsigned dim_in, unsigned dim_out, unsigned dim_len,
                      bool f_fixed, unsigned iwl, unsigned frac);
/* lib/layer_cuda.cu:2406-2448 (-> :105-172 non-trans, :547-635 trans) */
void qo_dot_mat_vec_fwd(const float *mat, const float *vec, float *out,
                        unsigned r, unsigned c, bool f_trans, bool f_fixed,
                        unsigned iwl_m, unsigned frac_m, unsigned iwl_v, unsigned frac_v);
/* lib/layer_cuda.cu:2491-2517 (-> :355-541 non-trans "V2", :547-635 trans) */
void qo_dot_mat_vec_fwd_appx(const float *mat, const float *vec, float *out,
                             unsigned r, unsigned c, bool f_fixed,
                             unsigned iwl, unsigned frac, unsigned num_bit, bool f_trans);
/* Row scorers built on the CPU hamming functions (dead caller lib/layer.c:322-356):
 * out[i] = sum_j sim(F2F(mat[i][j],iwl,frac_code), F2F(vec[j],iwl,frac_code), num_bit).
 * variant 0 -> hamming_similarity (V0, unsigned count), 1 -> hamming_similarity_w (V1). */
void qo_attention_hamming(const float *mat, const float *vec, float *out,
                          unsigned r, unsigned c, unsigned iwl, unsigned frac_code,
                          unsigned num_bit, int variant);

enum { QO_SM_CUDA = 0,      /* lib/layer_cuda.cu:1969-2060: __expf, double total  */
       QO_SM_CPU_POW2 = 1,  /* lib/layer.c:1184-1258: pow(2,.), float total        */
       QO_SM_CPU_EXP_PLAN = 2, /* lib/layer.c:1196-1198 + lib/common.c:51-73       */
};
void qo_softmax_fwd(const float *in, float *out, unsigned dim, int variant, bool f_shift_based);
/* lib/layer_cuda.cu:1535-1542; CPU lib/layer.c:1502-1511 */
void qo_sum_vec_fwd(const float *a, const float *b, float *out, unsigned dim,
                    bool f_fixed, unsigned iwl, unsigned frac);
/* lib/layer_cuda.cu:1664-1703 */
void qo_activation_fwd(const float *in, float *out, unsigned dim, const char *act,
                       bool f_fixed, unsigned iwl, unsigned frac);
/* lib/layer_cuda.cu:1895-1939: tree arg-max, ties -> highest index */
unsigned qo_argmax_hi(const float *in, unsigned dim);
/* lib/layer_cuda.cu:3750-3781, 2191-2251: returns pred index; accumulates cost/m_cnt */
unsigned qo_cross_entropy_run(const float *h, const float *y, unsigned dim,
                              float *cost_acc, unsigned *m_cnt_acc, float *grad_out);

/* ---- training verbs; restated from CUDA text, parity unpinned ---- */
/* lib/layer_cuda.cu:2560-2665 */
void qo_dot_mat_vec_bwd(const float *mat, const float *vec, const float *grad_in, float *grad_out_mat,
                        float *grad_out_vec, unsigned r, unsigned c, bool f_trans, bool f_fixed,
                        unsigned iwl_m, unsigned frac_m);
/* lib/layer_cuda.cu:2666-2767 with the surrogate-gradient kernels :742-1463 */
void qo_dot_mat_vec_bwd_appx(const float *mat, const float *vec, const float *grad_in, float *grad_out_mat,
                             float *grad_out_vec, unsigned r, unsigned c, bool f_fixed, unsigned iwl,
                             unsigned frac, unsigned num_bit, bool f_trans);
/* lib/layer_cuda.cu:2062-2135 */
void qo_softmax_bwd(const float *out_vec, const float *grad_in, float *grad_out, unsigned dim, bool f_shift_based);
/* lib/layer_cuda.cu:3232-3315: w_del accumulates, grad_out is overwritten */
void qo_dense_bwd(const float *w, float *w_del, const float *in, const float *out, float *grad_in, float *grad_out,
                  unsigned dim_in, unsigned dim_out, const char *act, bool f_fixed, unsigned iwl_w, unsigned frac_w);
/* lib/layer_cuda.cu:3570-3610 */
void qo_dense_mat_bwd(const float *in_mat, const float *w, float *w_del, const float *grad_in, float *grad_out,
                      unsigned dim_in, unsigned dim_out, unsigned dim_len, bool f_fixed, unsigned iwl, unsigned frac);
/* lib/layer_cuda.cu:3317-3363 / :3612-3651 with :1596-1622, :1783-1830; returns the norm it used */
float qo_mat_w_up(float *w, float *w_del, unsigned dim_in, unsigned dim_out, unsigned batch_size, float lr,
                  float lambda, float max_grad_l2_norm, bool f_fixed, unsigned iwl, unsigned frac);
/* lib/layer_cuda.cu:3908-3946 */
void qo_dup_grad_bwd(const float *a, const float *b, float *out, unsigned dim, bool f_fixed, unsigned iwl, unsigned frac);

/* ---- composite: one query through the test-phase forward, MemN2N/MemN2N.c:2626-2697 ---- */
#define QO_MAX_HOP 8
typedef struct {
    unsigned n_hop, dim_emb, dim_input;
    unsigned attention_mode;      /* 1 float, 2 fixed dot, 3 appx (V2); define.h:10-15;
                                   * 10 / 11: hamming_similarity / hamming_similarity_w row scorers */
    unsigned num_bit;             /* bits compared in modes 10 / 11 */
    int      softmax_variant;     /* QO_SM_* for the in-hop softmax and the output softmax */
    bool     f_fixed;             /* EN_FIXED_POINT, define.h:31 */
    bool     en_lin_map;          /* define.h:291 */
    unsigned iwl[QO_MAX_HOP],     frac[QO_MAX_HOP];      /* activations  MemN2N.c:715-716 */
    unsigned iwl_w[QO_MAX_HOP],   frac_w[QO_MAX_HOP];    /* weights      :718-719,748-754 */
    unsigned iwl_att[QO_MAX_HOP], frac_att[QO_MAX_HOP];  /* attention    :721-722 */
    unsigned iwl_bin, frac_bin;                          /* :769-775 */
    const float *w_q;                 /* [D][dim_input]          emb_q   :826 */
    const float *w_a[QO_MAX_HOP];     /* [D][dim_input]          emb_m   :835 */
    const float *w_c[QO_MAX_HOP];     /* [D][dim_input]          emb_c   :838 */
    const float *w_h[QO_MAX_HOP];     /* [D][D]                  lin_map :873 */
    const float *w_ans;               /* [dim_input][D], float   ds_ans  :902-906 */
    /* optional in-hop softmax variants (all zero = off) */
    bool     f_shift_based;           /* EN_SHIFT_BASED_SM: sf_in only, MemN2N.c:856 (sf_out never, :910) */
    bool     en_sc_att;               /* EN_SC_ATT: scale layer between dotmv and sf_in, MemN2N.c:2647-2651 */
    float    sc_att[QO_MAX_HOP];      /* its scalar weight per hop (out = in * w, lib/layer_cuda.cu:1551-1558) */
    bool     en_non_lin;              /* EN_NON_LINEARITY: RELU layer after sv[h], formats (iwl[h],frac[h]); MemN2N.c:894-896, 2668-2671 */
} qo_model;

typedef struct {                 /* optional taps; any pointer may be NULL */
    float *u0;                   /* [D] */
    float *keys, *vals;          /* [n_hop][n_sen][D] */
    float *scores, *probs;       /* [n_hop][n_sen] */
    float *o, *lu, *u;           /* [n_hop][D] */
    float *logits, *out_probs;   /* [dim_input] */
} qo_taps;

/* story [n_sen][dim_input], question [dim_input]; returns arg-max prediction */
unsigned qo_memn2n_forward(const qo_model *m, const float *story, unsigned n_sen,
                           const float *question, qo_taps *taps);

/* Same forward, but starting from already-embedded memories (the synthetic
 * |mem|=10000 configs): keys/vals [n_hop][n_sen][D] float-on-grid, u0 [D]. */
unsigned qo_memn2n_forward_mem(const qo_model *m, const float *keys, const float *vals,
                               unsigned n_sen, const float *u0, qo_taps *taps);

/* Test driver: the forward above for MANY stories given as uint16 word lists (the product's wire format; rows are expanded
 * to the bag-of-words floats of sample.c per story), on n_threads threads.  pred [n], u_final [n][D] = sv[n_hop-1],
 * top2_gap [n] = difference of the two largest output probabilities, near_step [n] = 1 when some in-hop softmax weight lies
 * within 1e-5 of a truncation step of its format (the only place a hop output may legitimately differ by a code). */
void qo_memn2n_forward_words_batch(const qo_model *m, const uint16_t *story_words, unsigned sw_width, const uint16_t *question_words,
                                   unsigned qw_width, const uint32_t *row_off, unsigned n_query, unsigned n_threads, uint32_t *pred,
                                   float *u_final, float *top2_gap, uint8_t *near_step);

#ifdef __cplusplus
}
#endif
#endif
