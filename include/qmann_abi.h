/*
 * qmann_abi.h -- drop-in boundary B of the Q-MANN test-phase forward path.
 *
 * libqmann_hip.so exports, unmangled and with the C calling convention, the 66
 * `cuda_*` entry points that the reference's unmodified `lib/layer.o` (56
 * imports) and `MemN2N/MemN2N.o` (10 direct imports) bind.  Each prototype
 * below names the reference definition it replaces
 * (lib/layer_cuda.cu:<line>, paths relative to the reference tree).  All
 * pointers named dev_* are device addresses (plain HBM allocations: pointer
 * arithmetic on them is legal, as MemN2N/MemN2N.c:2416 does); everything else
 * is host memory.  No torch / HIP types appear in any signature.
 *
 * Conventions kept from the reference:
 *   - no return codes: a failure prints "[*E] ..." to stderr and exit()s
 *     (lib/layer_cuda.h:13-22);
 *   - single host thread, one in-order stream (the HIP null stream), no
 *     synchronisation between ops; D2H copies are the only sync points;
 *   - `bool` and `unsigned int` scalars: the reference calls these functions
 *     without prototypes, so they arrive as promoted ints -- the definitions
 *     take them exactly as declared here, which is ABI-compatible on x86-64.
 *     `cuda_set_value`'s float `value` arrives as a promoted double from such
 *     a caller; like the reference's own nvcc-built callee this library reads
 *     it as float, which is only meaningful for 0.0 (the one value the
 *     reference ever passes, MemN2N/MemN2N.c:1824-1831).  Prototyped callers
 *     (this header) are exact.
 *
 * Status of each group is given in its comment: FORWARD = real HIP kernels
 * (hot path, parity-tested); UTIL = allocation / copy helpers, real;
 * TRAIN = backward / weight-update verbs: functional
 * HIP, one thread per output element with the reference's serial reduction
 * order (bit-identical gradients, parity-tested), not tuned -- enough for the
 * unmodified MemN2N.c to train and test end to end on the GPU.
 */
#ifndef QMANN_ABI_H
#define QMANN_ABI_H

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dot_mat_vec: attention scores and weighted read-out ------------------ */
/* UTIL     lib/layer_cuda.cu:2296 */
void cuda_dot_mat_vec_constructor(float **dev_out_vec, float **dev_grad_out_vec, float **dev_grad_out_mat,
                                  float **dev_f_overflow, float **dev_cliff_marker,
                                  unsigned int dim_mat_r, unsigned int dim_mat_c, bool f_trans);
/* UTIL     lib/layer_cuda.cu:2364 */
void cuda_dot_mat_vec_init(float *dev_out_vec, float *dev_grad_out_vec, float *dev_grad_out_mat,
                           float *dev_f_overflow, float *dev_cliff_marker,
                           unsigned int dim_mat_r, unsigned int dim_mat_c, bool f_trans);
/* FORWARD  lib/layer_cuda.cu:2405 (kernels :105-172, :547-635) */
void cuda_dot_mat_vec_fwd(float *dev_in_mat, float *dev_in_vec, float *dev_out_vec, float *dev_f_overflow,
                          unsigned int dim_mat_r, unsigned int dim_mat_c, bool f_trans, bool f_fixed,
                          unsigned int iwl_m, unsigned int frac_m, unsigned int iwl_v, unsigned int frac_v,
                          unsigned int f_mode, bool verbose);
/* FORWARD  lib/layer_cuda.cu:2490 (kernels :355-541 with :218-326, :547-635) */
void cuda_dot_mat_vec_fwd_appx(float *dev_in_mat, float *dev_in_vec, float *dev_out_vec, float *dev_f_overflow,
                               float *dev_cliff_marker, unsigned int dim_mat_r, unsigned int dim_mat_c,
                               bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode,
                               unsigned int num_bit_attention, bool f_trans, bool verbose);
/* TRAIN    lib/layer_cuda.cu:2560 */
void cuda_dot_mat_vec_bwd(float *dev_in_mat, float *dev_in_vec, float *dev_grad_in, float *dev_grad_out_mat,
                          float *dev_grad_out_vec, float *dev_f_overflow, unsigned int dim_mat_r,
                          unsigned int dim_mat_c, bool f_trans, bool f_fixed, unsigned int iwl_m,
                          unsigned int frac_m, unsigned int iwl_v, unsigned int frac_v, unsigned int f_mode,
                          bool verbose);
/* TRAIN    lib/layer_cuda.cu:2666 */
void cuda_dot_mat_vec_bwd_appx(float *dev_in_mat, float *dev_in_vec, float *dev_grad_in, float *dev_grad_out_mat,
                               float *dev_grad_out_vec, float *dev_f_overflow, float *dev_cliff_marker,
                               unsigned int dim_mat_r, unsigned int dim_mat_c, bool f_fixed, unsigned int iwl,
                               unsigned int frac, unsigned int f_mode, unsigned int num_bit_attention,
                               bool f_trans, bool verbose, unsigned int hop);
/* UTIL     lib/layer_cuda.cu:2769 */
void cuda_dot_mat_vec_destructor(float *dev_out_vec, float *dev_grad_out_vec, float *dev_grad_out_mat,
                                 float *dev_f_overflow, float *dev_cliff_marker);

/* ---- softmax --------------------------------------------------------------- */
/* UTIL     lib/layer_cuda.cu:2791 */
void cuda_softmax_constructor(float **dev_out_vec, float **dev_grad_out, float **dev_max, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:2819 */
void cuda_softmax_init(float *dev_out_vec, float *dev_grad_out, float *dev_max, unsigned int dim);
/* FORWARD  lib/layer_cuda.cu:2844 (kernels :1895-1916, :1969-2060); any dim (the reference stops at 1024) */
void cuda_softmax_fwd(float *dev_out_vec, float *dev_in_vec, float *out_vec, float *in_vec, float *dev_max,
                      unsigned int dim, bool f_shift_based, bool verbose);
/* TRAIN    lib/layer_cuda.cu:2885 */
void cuda_softmax_bwd(float *dev_grad_in, float *dev_out_vec, float *dev_grad_out, float *dev_in_vec,
                      unsigned int dim, bool f_shift_based, bool verbose);
/* UTIL     lib/layer_cuda.cu:2923 */
void cuda_softmax_destructor(float *dev_out_vec, float *dev_grad_out, float *dev_max);

/* ---- sum_vec ----------------------------------------------------------------- */
/* UTIL     lib/layer_cuda.cu:2941 */
void cuda_sum_vec_constructor(float **dev_out_vec, float **dev_grad_out, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:2968 */
void cuda_sum_vec_init(float *dev_out_vec, float *dev_grad_out, unsigned int dim);
/* FORWARD  lib/layer_cuda.cu:2991 (kernel :1535-1542) */
void cuda_sum_vec_fwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_out_vec, unsigned int dim,
                      bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:3031 */
void cuda_sum_vec_bwd(float *dev_grad_out, float *dev_grad_in, float *grad_in, float *grad_out, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:3063 */
void cuda_sum_vec_destructor(float *dev_out_vec, float *dev_grad_out);

/* ---- dense (question embedding, linear map, answer projection) ------------- */
/* UTIL     lib/layer_cuda.cu:3079 */
void cuda_dense_constructor(float **dev_w_mat, float **dev_w_mat_del, float **dev_w_mat_best, float **dev_bias,
                            float **dev_bias_del, float **dev_out_vec, float **dev_grad_out,
                            float **dev_grad_l2_norm, float **dev_grad_bias_l2_norm, float **dev_f_overflow,
                            unsigned int dim_in, unsigned int dim_out);
/* UTIL     lib/layer_cuda.cu:3125 (uploads w_mat / bias) */
void cuda_dense_init(float *dev_out_vec, float *dev_grad_out, float *dev_w_mat_del, float *dev_w_mat,
                     float *dev_bias, float *dev_bias_del, float *w_mat, float *bias, float *dev_f_overflow,
                     unsigned int dim_in, unsigned int dim_out);
/* FORWARD  lib/layer_cuda.cu:3162 (kernel :49-83, post-ops :1664-1703) */
void cuda_dense_fwd(float *dev_w_mat, float *dev_bias, float *dev_in_vec, float *dev_out_vec,
                    float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out, char *activation,
                    bool f_fixed, unsigned int iwl_in, unsigned int frac_in, unsigned int iwl_w,
                    unsigned int frac_w, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:3232 */
void cuda_dense_bwd(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                    float *dev_in_vec, float *dev_out_vec, float *dev_grad_in, float *dev_grad_out,
                    float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out, char *activation,
                    bool f_fixed, unsigned int iwl_in, unsigned int frac_in, unsigned int iwl_w,
                    unsigned int frac_w, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:3317 */
void cuda_dense_w_up(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                     float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm, unsigned int dim_in,
                     unsigned int dim_out, unsigned int batch_size, float *lr, float *lambda,
                     float *max_grad_l2_norm, bool f_fixed, unsigned int iwl, unsigned int frac,
                     unsigned int f_mode, bool verbose);
/* UTIL     lib/layer_cuda.cu:3365 */
void cuda_dense_destructor(float *dev_w_mat, float *dev_w_mat_del, float *dev_w_mat_best, float *dev_out_vec,
                           float *dev_grad_out, float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm,
                           float *dev_f_overflow);

/* ---- dense_mat (story embedding) ------------------------------------------ */
/* UTIL     lib/layer_cuda.cu:3417 */
void cuda_dense_mat_constructor(float **dev_w_mat, float **dev_w_mat_del, float **dev_w_mat_best,
                                float **dev_bias, float **dev_bias_del, float **dev_out_mat,
                                float **dev_grad_out, float **dev_grad_l2_norm, float **dev_grad_bias_l2_norm,
                                float **dev_f_overflow, unsigned int dim_in, unsigned int dim_out,
                                unsigned int dim_len);
/* UTIL     lib/layer_cuda.cu:3468 (uploads w_mat / bias) */
void cuda_dense_mat_init(float *dev_out_mat, float *dev_grad_out, float *dev_w_mat, float *dev_w_mat_del,
                         float *dev_bias, float *dev_bias_del, float *w_mat, float *bias,
                         float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out,
                         unsigned int dim_len);
/* FORWARD  lib/layer_cuda.cu:3511 (kernel :105-172) */
void cuda_dense_mat_fwd(float *dev_w_mat, float *dev_bias, float *dev_in_mat, float *dev_out_mat,
                        float *dev_f_overflow, unsigned int dim_in, unsigned int dim_out, unsigned int dim_len,
                        bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:3570 */
void cuda_dense_mat_bwd(float *dev_in_mat, float *dev_w_mat, float *dev_w_mat_del, float *dev_bias,
                        float *dev_bias_del, float *dev_grad_in, float *dev_grad_out, float *dev_f_overflow,
                        unsigned int dim_in, unsigned int dim_out, unsigned int dim_len, bool f_fixed,
                        unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:3612 */
void cuda_dense_mat_w_up(float *dev_w_mat, float *dev_w_mat_del, float *dev_bias, float *dev_bias_del,
                         float *dev_grad_l2_norm, float *dev_grad_bias_l2_norm, float *w_mat, float *w_mat_del,
                         unsigned int dim_in, unsigned int dim_out, unsigned int batch_size, float *lr,
                         float *lambda, float *max_grad_l2_norm, bool f_fixed, unsigned int iwl,
                         unsigned int frac, unsigned int f_mode, bool verbose);
/* UTIL     lib/layer_cuda.cu:3653 */
void cuda_dense_mat_destructor(float *dev_w_mat, float *dev_w_mat_del, float *dev_w_mat_best,
                               float *dev_out_mat, float *dev_grad_out, float *dev_grad_l2_norm,
                               float *dev_f_overflow);

/* ---- cross entropy (prediction / match counting in the test phase) ------- */
/* UTIL     lib/layer_cuda.cu:3679 */
void cuda_cross_entropy_constructor(float **dev_cost_train, float **dev_cost_valid, float **dev_cost_test,
                                    unsigned int **dev_m_cnt_train, unsigned int **dev_m_cnt_valid,
                                    unsigned int **dev_m_cnt_test, unsigned int **dev_pred_i,
                                    float **dev_grad_out, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:3716 */
void cuda_cross_entropy_init(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                             unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                             unsigned int *dev_m_cnt_test, float *dev_grad_out, unsigned int dim);
/* FORWARD  lib/layer_cuda.cu:3749 (kernels :1918-1939, :2191-2251); mode 1 train / 2 valid / 3 test */
void cuda_cross_entropy_run(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                            unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                            unsigned int *dev_m_cnt_test, unsigned int *dev_pred_i, float *cost, float *dev_h,
                            float *dev_y, float *h, float *y, float *dev_grad_out, float *grad_out,
                            unsigned int dim, unsigned int mode);
/* FORWARD  lib/layer_cuda.cu:3812 (D2H of the three accumulators, then reset) */
void cuda_cross_entropy_cost_load(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                                  float *cost_train, float *cost_valid, float *cost_test);
/* FORWARD  lib/layer_cuda.cu:3832 */
void cuda_cross_entropy_m_cnt_load(unsigned int *dev_m_cnt_train, unsigned int *dev_m_cnt_valid,
                                   unsigned int *dev_m_cnt_test, unsigned int *m_cnt_train,
                                   unsigned int *m_cnt_valid, unsigned int *m_cnt_test);
/* UTIL     lib/layer_cuda.cu:3853 */
void cuda_cross_entropy_destructor(float *dev_cost_train, float *dev_cost_valid, float *dev_cost_test,
                                   float *dev_m_cnt_train, float *dev_m_cnt_valid, float *dev_m_cnt_test,
                                   float *dev_pred_i, float *dev_grad_out);

/* ---- activation / scale (optional layers, define.h:59,294) ---------------- */
/* UTIL     lib/layer_cuda.cu:4503 */
void cuda_activation_constructor(float **dev_out, float **dev_grad_out, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:4527 */
void cuda_activation_init(float *dev_out, float *dev_grad_out, unsigned int dim);
/* FORWARD  lib/layer_cuda.cu:4547 (kernels :1664-1703) */
void cuda_activation_fwd(float *dev_in, float *dev_out, char *type_act, unsigned int dim, bool f_fixed,
                         unsigned int iwl, unsigned int frac, unsigned int f_mode);
/* TRAIN    lib/layer_cuda.cu:4586 */
void cuda_activation_bwd(float *dev_out, float *dev_grad_in, float *dev_grad_out, char *type_act,
                         unsigned int dim, bool f_fixed, unsigned int iwl, unsigned int frac,
                         unsigned int f_mode);
/* UTIL     lib/layer_cuda.cu:4616 */
void cuda_activation_destructor(float *dev_out, float *dev_grad_out);
/* UTIL     lib/layer_cuda.cu:4747 */
void cuda_scale_constructor(float **dev_w, float **dev_w_del, float **dev_w_best, float **dev_out,
                            float **dev_grad_out, unsigned int dim);
/* UTIL     lib/layer_cuda.cu:4777 */
void cuda_scale_init(float *dev_w, float *dev_w_del, float *dev_out, float *dev_grad_out, float *w,
                     unsigned int dim);
/* FORWARD  lib/layer_cuda.cu:4804 (kernel :1551-1558) */
void cuda_scale_fwd(float *dev_in, float *dev_w, float *dev_out, unsigned int dim, bool f_fixed,
                    unsigned int iwl, unsigned int frac, unsigned int f_mode, bool verbose);
/* TRAIN    lib/layer_cuda.cu:4829 */
void cuda_scale_bwd(float *dev_in, float *dev_grad_in, float *dev_w, float *dev_w_del, float *dev_grad_out,
                    unsigned int dim, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode,
                    bool verbose);
/* TRAIN    lib/layer_cuda.cu:4860 */
void cuda_scale_w_up(float *dev_w, float *dev_w_del, unsigned int dim, unsigned int batch_size, float *lr,
                     float *lambda, bool f_fixed, unsigned int iwl, unsigned int frac, unsigned int f_mode,
                     bool verbose);
/* UTIL     lib/layer_cuda.cu:4910 */
void cuda_scale_destructor(float *dev_w, float *dev_w_del, float *dev_w_best, float *dev_out,
                           float *dev_grad_out);

/* ---- mult_e_vec / mult_e_mat: layers no program instantiates;
 *      imported by layer.o, so the symbols exist.  constructor/init/destructor
 *      are UTIL, fwd is FORWARD (element-wise product), bwd is TRAIN. --------- */
void cuda_mult_e_vec_constructor(float **dev_out_vec, float **dev_grad_out_a, float **dev_grad_out_b,
                                 unsigned int dim);                                   /* :4175 */
void cuda_mult_e_vec_init(float *dev_out_vec, float *dev_grad_out_a, float *dev_grad_out_b,
                          unsigned int dim);                                          /* :4204 */
void cuda_mult_e_vec_fwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_out_vec, float *in_vec_a,
                         float *in_vec_b, float *out_vec, unsigned int dim);          /* :4233 */
void cuda_mult_e_vec_bwd(float *dev_in_vec_a, float *dev_in_vec_b, float *dev_grad_out_a,
                         float *dev_grad_out_b, float *dev_grad_in, float *grad_in, float *grad_out_a,
                         float *grad_out_b, unsigned int dim);                        /* :4265 */
void cuda_mult_e_vec_destructor(void);                                                /* :4301 */
void cuda_mult_e_mat_constructor(float **dev_out_mat, float **dev_grad_out_a, float **dev_grad_out_b,
                                 unsigned int dim_row, unsigned int dim_col);         /* :4312 */
void cuda_mult_e_mat_init(float *dev_out_mat, float *dev_grad_out_a, float *dev_grad_out_b,
                          unsigned int dim_row, unsigned int dim_col);                /* :4342 */
void cuda_mult_e_mat_fwd(float *dev_in_mat_a, float *dev_in_mat_b, float *dev_out_mat, float *in_mat_a,
                         float *in_mat_b, float *out_mat, unsigned int dim_row,
                         unsigned int dim_col);                                       /* :4372 */
void cuda_mult_e_mat_bwd(float *dev_in_mat_a, float *dev_in_mat_b, float *dev_grad_out_a,
                         float *dev_grad_out_b, float *dev_grad_in, float *grad_in, float *grad_out_a,
                         float *grad_out_b, unsigned int dim_row, unsigned int dim_col); /* :4396 */
void cuda_mult_e_mat_destructor(void);                                                /* :4427 */

/* ---- the 10 helpers MemN2N.o imports directly ------------------------------- */
/* UTIL     lib/layer_cuda.cu:3959  ragged pools: stories [dim_len][dim_in], questions/answers [num_sample][dim_in] */
void cuda_data_constructor(float **dev_m, float **dev_q, float **dev_a, unsigned int dim_len,
                           unsigned int dim_in, unsigned int num_sample);
/* UTIL     lib/layer_cuda.cu:3990  one bulk H2D per phase (MemN2N/MemN2N.c:2337-2349) */
void cuda_data_in(float *dev_m, float *dev_q, float *dev_a, float *m, float *q, float *a,
                  unsigned int dim_len, unsigned int dim_in, unsigned int num_sample);
/* UTIL     lib/layer_cuda.cu:4022 */
void cuda_data_destructor(float *dev_m, float *dev_q, float *dev_a);
/* UTIL     lib/layer_cuda.cu:3884 */
void cuda_dup_grad_constructor(float **dev_dup_grad, unsigned int num_hop, unsigned int dim);
/* TRAIN    lib/layer_cuda.cu:3908 */
void cuda_dup_grad_bwd(float *dev_dup_grad, float *dotmv_dev_grad_out_vec, float *sv_dev_grad_out_vec,
                       float *dup_grad, unsigned int dim, bool f_fixed, unsigned int iwl, unsigned int frac,
                       unsigned int f_mode);
/* UTIL     lib/layer_cuda.cu:3948 */
void cuda_dup_grad_destructor(float *dev_dup_grad);
/* UTIL     lib/layer_cuda.cu:4116  src is [dim_row][dim_col]; f_trans writes dest as [dim_col][dim_row] */
void cuda_copy_mat(float *dev_src, float *dev_dest, unsigned int dim_col, unsigned int dim_row, bool f_trans);
/* UTIL     lib/layer_cuda.cu:4152  dest += src (same indexing as cuda_copy_mat) */
void cuda_accum_mat(float *dev_src, float *dev_dest, unsigned int dim_col, unsigned int dim_row, bool f_trans);
/* UTIL     lib/layer_cuda.cu:4645  dest[i] = value for i < dim with i % stride == start_idx
 *          (the reference writes past dim up to the next multiple of 1024; this one does not) */
void cuda_set_value(float *dest, float value, unsigned int dim, unsigned int start_idx, unsigned int stride);
/* UTIL     lib/layer_cuda.cu:4931  synchronous D2H of `size` floats */
void cuda_copy_dev2host(float *host, float *dev, unsigned int size);

/* ---- library-level switches that have no reference counterpart ------------ */
/* Exponential used by cuda_softmax_fwd: 0 = e^x as _cuda_softmax_fwd (lib/layer_cuda.cu:2006, default),
 * 1 = 2^x as the live CPU branch (lib/layer.c:1225), 2 = its piece-wise linear exp_plan (lib/layer.c:1196-1199). */
void qmann_abi_set_softmax_base(int base);
/* Number of `cuda_*` symbols of boundary B this build exports (66). */
unsigned int qmann_abi_symbol_count(void);

/* Deferred execution of the forward verbs.  The host drives the test and validation phases one query at a time, 31 verbs
 * per query, and reads nothing back until the accumulators are fetched after the loop (MemN2N/MemN2N.c:2378-2702;
 * lib/layer_cuda.cu:3813-3851).  By default the nine forward verbs (cuda_dense_fwd, cuda_dense_mat_fwd,
 * cuda_dot_mat_vec_fwd, cuda_dot_mat_vec_fwd_appx, cuda_softmax_fwd, cuda_sum_vec_fwd, cuda_scale_fwd, cuda_activation_fwd,
 * cuda_cross_entropy_run) therefore only RECORD their call; every other cuda_* verb first drains the record.  Whole queries
 * are recognised by their pointer wiring and a run of them becomes one call of the batched forward (qmann_model.h) on the
 * host's own device pools, the match count and cost going into the accumulators the verbs named; anything else (training
 * steps, unusual layer orders, options the batched kernels refuse) is executed verb by verb as before.  After a batched run
 * the last query is replayed verb by verb so that every layer's device buffer holds what the serial loop leaves there.
 * CONTRACT: device buffers written by forward verbs are up to date after the next non-forward cuda_* verb (as for the
 * reference host, which reads only through cuda_cross_entropy_*_load / cuda_copy_dev2host) or after qmann_abi_flush();
 * a host that reads them with its own hipMemcpy must call qmann_abi_flush() first, or switch the queue off.  That call
 * is a pure read barrier: it keeps the cached batched model.  WRITES behind the library's back have their own call: the
 * batched model built for a run of queries is cached, keyed on the weight pointers and formats (not on the values), and
 * dropped by every cuda_* verb that can change a weight or an input pool; a host that overwrites weights or pools with its
 * own hipMemcpy between two forward phases must call qmann_abi_invalidate_model(), which drains the record and forgets the
 * cached model (the next run rebuilds it from the device matrices).
 * A forward verb called with verbose = true is a synchronisation point too: the record is drained, the verb runs at once and
 * prints its operands as the reference does (lib/layer_cuda.cu:13-47, 2450-2484).
 * THREADS: the record is process-wide; the verbs serialise on one lock around it (recording, draining, these switches).
 *   mode 0: off -- every verb launches at once;  1: on (default);  2: verify -- every recognised run is computed BOTH ways,
 *   the verb-by-verb result goes into the accumulators and a line comparing the two match counts is printed on stderr.
 * Environment: QMANN_DEFER=0|1|verify (QMANN_NO_DEFER=1 = 0), QMANN_DEFER_STATS=1 prints the counters below at exit,
 * QMANN_SAVE_WEIGHTS_DIR=<dir> writes, when a TEST-phase run is dispatched, the matrices it is tested with as weight files
 * (include/qmann_weights.h: the reference's disabled EN_WRITE_WEIGHT layout, MemN2N.c:2853-2978) and the model's quantised
 * parameter blob (<dir>/qmann_params.bin, qmann_model_create_from_params). */
typedef struct qmann_defer_stats {
    unsigned long long ops_queued, ops_replayed;        /* forward verbs recorded / executed one by one */
    unsigned long long queries_batched, batches;        /* queries that went through the batched forward, in how many calls */
    unsigned long long models_built, verify_runs, verify_mismatch;
    double ms_batched, ms_replayed, ms_model;           /* wall time (device-synchronised only with QMANN_DEFER_STATS / verify) */
} qmann_defer_stats;
void qmann_abi_set_defer(int mode);
void qmann_abi_flush(void);               /* drain (read barrier); the cached model stays */
void qmann_abi_invalidate_model(void);    /* drain + forget the cached model (weights / pools rewritten behind the library) */
void qmann_abi_defer_stats(qmann_defer_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* QMANN_ABI_H */
