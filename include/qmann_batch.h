/*
 * qmann_batch.h -- batched, int8-native entry points of libqmann_hip.so.
 *
 * These have no counterpart in the reference (which runs one query at a time
 * through ~31 kernel launches, MemN2N/MemN2N.c:2378-2697); they are the
 * throughput face of the same arithmetic: many independent queries per launch,
 * memories stored as true int8 codes of the reference's Q(iwl.frac) grid
 * values (lossless for word length 8), the multi-hop loop fused
 * into one kernel.  Plain C ABI: device pointers, sizes, an opaque stream
 * handle (hipStream_t passed as void*; NULL = the null stream).  Every call is
 * asynchronous and, to its caller, ONE ordered step on that stream; a hop launch
 * over >= 32 768 stories of mixed length puts one of its two kernels on a second
 * stream of the library's own, forked from and joined back into the caller's
 * stream by events inside the call (qmann_model.h; QMANN_NO_CORUN turns it off).
 *
 * Return value of every function: 0 on success, a negative QMANN_E* code on a
 * caller error (nothing is launched) or QMANN_EHIP when the HIP runtime fails
 * (message on stderr; unlike the drop-in cuda_* verbs, which exit() as the
 * reference does, these calls return).
 *
 * Data layout ("memory" = the per-query story slots):
 *   keys, vals : int8 [n_hop][rows_total][Dp]    SIGN-MAGNITUDE codes: bit 7 = sign, bits 6..0 =
 *                |code| -- the top byte of the reference's own FLOAT2FIXED word
 *                (lib/common.h:210), so "minus zero" (0x80) is representable as it is there.
 *                Dp = dim_emb_pad (multiple of 16, >= D); columns D..Dp-1 are 0.
 *                keys[h] carry Q(att[h]) codes, vals[h] carry Q(act[h]) codes (valid codes of those
 *                formats: |code| <= 2^(iwl+frac) - 1, which matters for word lengths below 8) --
 *                i.e. what the reference's dot_mat_vec layers see after their
 *                own operand quantisation (lib/layer_cuda.cu:120, :562).
 *   row_off    : uint32 [n_query + 1]            first row of each query's
 *                slots; query q owns rows row_off[q] .. row_off[q+1]-1 in every
 *                hop plane (ragged, like the reference's pools
 *                MemN2N/MemN2N.c:2294-2333).
 *   u vectors  : float [n_query][D]              values on the producer's grid,
 *                exactly the floats the reference passes between layers.
 */
#ifndef QMANN_BATCH_H
#define QMANN_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMANN_MAX_HOP 8

enum { QMANN_OK = 0, QMANN_EINVAL = -1, QMANN_ERANGE = -2, QMANN_EUNSUPPORTED = -3, QMANN_EIO = -4,
       QMANN_EHIP = -5 /* a HIP runtime call failed (message on stderr); the process goes on */,
       QMANN_ECOMM = -6 /* RCCL: the library could not be loaded or a collective call failed (qmann_dist.h) */ };

/* attention_mode: MemN2N/define.h:10-15 (1..3) plus the packed-code Hamming forms */
enum {
    QMANN_ATT_FLOAT = 1,      /* float dot product over the grid values, float read-out           */
    QMANN_ATT_FIXED = 2,      /* per-product quantised dot product, quantised read-out            */
    QMANN_ATT_APPX = 3,       /* CUDA "approximate" Hamming attention (lib/layer_cuda.cu:355-541) */
    QMANN_ATT_HAMMING_V0 = 10,/* bit-agreement count over the top n bits (lib/common.c:223-246)   */
    QMANN_ATT_HAMMING_V1 = 11 /* signed, weighted bit agreement (lib/common.c:249-312)            */
};
/* BINARY_MODE (define.h:87-88: the query binarised to +-1 in the scores and the linear map) is not a mode
 * of its own: set qmann_net.bin = {0, 0}, exactly what the reference does (MemN2N.c:769-775). */

/* byte layout of an int8 code */
enum { QMANN_CODE_TWOS = 0 /* two's complement (weights) */, QMANN_CODE_SIGNMAG = 1 /* memories */ };

enum { QMANN_SOFTMAX_EXP = 0 /* lib/layer_cuda.cu:2006 */, QMANN_SOFTMAX_POW2 = 1 /* lib/layer.c:1225 */,
       QMANN_SOFTMAX_EXP_PLAN = 2 /* piece-wise linear exp, lib/layer.c:1196-1199 + lib/common.c:51-73 */ };

typedef struct qmann_fmt {
    uint32_t iwl;
    uint32_t frac;
} qmann_fmt;

typedef struct qmann_net {
    uint32_t n_hop;
    uint32_t dim_emb;        /* D */
    uint32_t dim_emb_pad;    /* Dp: 64, 128 or 256 */
    uint32_t dim_input;      /* V: dictionary + time slots */
    uint32_t attention_mode;
    uint32_t softmax_base;
    uint32_t en_lin_map;     /* MemN2N/define.h:291 */
    uint32_t num_bit;        /* Hamming forms: bits compared (<= 8) */
    qmann_fmt act[QMANN_MAX_HOP];   /* (iwl[h], frac[h])          MemN2N/MemN2N.c:715-716 */
    qmann_fmt w[QMANN_MAX_HOP];     /* (iwl_w[h], frac_w[h])      :718-719, 748-754       */
    qmann_fmt att[QMANN_MAX_HOP];   /* (iwl_att[h], frac_att[h])  :721-722                */
    qmann_fmt bin;                  /* (iwl_bin, frac_bin)        :769-775; {0,0} binarises u   */
    const int8_t *lin_map[QMANN_MAX_HOP]; /* device, [D][Dp] sign-magnitude codes (QMANN_CODE_SIGNMAG) in format w[h]; NULL when !en_lin_map */
    /* in-hop softmax variants; all zero = the stock configuration */
    uint32_t softmax_shift_based;   /* EN_SHIFT_BASED_SM (define.h:54-55): power-of-two style normaliser, sf_in only (MemN2N.c:856) */
    uint32_t en_att_scale;          /* EN_SC_ATT (define.h:58-59): scores times one learnt scalar before the softmax (MemN2N.c:2647-2649) */
    float att_scale[QMANN_MAX_HOP]; /* that scalar per hop (scale.w, lib/layer.h:786-810) */
    uint32_t en_non_linearity;      /* EN_NON_LINEARITY (define.h): RELU layers non_lin[h] (MemN2N.c:894-896, 2668-2671): the attention of
                                     * hop h >= 1 and the answer layer read RELU(sv), lin_map keeps reading sv (:2435-2437, 2471-2473, 2535-2537);
                                     * u_out is then RELU(sv[n_hop-1]) */
    uint32_t en_pe;                 /* EN_PE (define.h:298): position encoding.  In the reference it changes the QUESTION's bag-of-words
                                     * row only (the story lines are commented out, sample.c:529-541): entry of word w at slot j is SET to
                                     * pe_w[w][j] = 1 + 4 (w / dim_input - 0.5)(j / dim_word - 0.5) (MemN2N.c:606-616, sample.c:559-560).
                                     * qmann_embed_query_idx applies it to word-index questions; bag-of-words callers pass such rows themselves */
    uint32_t pe_dim_word;           /* dim_word of that formula: longest sentence + 1 with the time entry (MemN2N.c:575-580) */
} qmann_net;

/* optional per-query taps for parity tests; any pointer may be NULL */
typedef struct qmann_taps {
    int32_t *score_codes;   /* [n_hop][rows_total]  integer score in units of 2^-frac_att (fixed) or 2^-10 (appx) */
    float   *scores;        /* [n_hop][rows_total]  the same as float, as the reference's dev_out_vec */
    float   *probs;         /* [n_hop][rows_total] */
    float   *o;             /* [n_query][n_hop][D] */
    float   *u;             /* [n_query][n_hop][D] */
} qmann_taps;

/* float -> int8 code of Q(iwl.frac)(x) in the given byte layout; rows of `cols` values are written
 * with pitch `pitch` (>= cols, padding zeroed).  Rejects formats with iwl + frac > 7. */
int qmann_quantize_i8(const float *src, int8_t *dst, size_t rows, uint32_t cols, uint32_t pitch,
                      qmann_fmt fmt, int layout, void *stream);

/* The hot path: all hops of all queries in one launch (one workgroup per query).
 *   max_slots                  -- bound on row_off[q+1] - row_off[q]; sizes the per-query LDS.  A story longer than the bound
 *                                 is CUT to max_slots by every kernel: the call still returns 0 and
 *                                 the result is that of the shortened story -- pass the true maximum (qmann_check_slots below
 *                                 verifies a bound on the device)
 *   u0    [n_query][D] float   -- question embedding (emb_q output)
 *   u_out [n_query][D] float   -- sv[n_hop-1] output, input of the answer layer
 * Replaces, per query and hop, the reference sequence dot_mat_vec_fwd -> softmax_fwd ->
 * dot_mat_vec_fwd(trans) -> dense_fwd(lin_map) -> sum_vec_fwd (MemN2N/MemN2N.c:2644-2666).
 * attention_mode QMANN_ATT_FIXED (define.h mode 2), QMANN_ATT_APPX (mode 3; needs att formats with
 * iwl + frac = 7; the key bytes are the ones the embedding entry points store -- sign of the value | magnitude
 * on the attention grid, or the key's own code where w[h] is one bit finer than att[h], exactly -2^iwl_att as
 * 0x80 -- and carry the reference's 32-bit operand words for these grids: u and the keys inside the attention
 * grid; both at least one fractional bit coarser (wider range: they saturate); u inside and the keys one bit
 * finer.  EN_MQ's formats, MemN2N/MemN2N.c:748-754, are the second, first and third case at hops 0, 1, 2.
 * Any other combination: QMANN_EUNSUPPORTED) or
 * QMANN_ATT_FLOAT (mode 1: float scores / softmax / read-out over the same int8 memories, which
 * then carry Q(w[h]) codes -- the embedding outputs, not re-quantised), or QMANN_ATT_HAMMING_V0 / _V1
 * computed straight from the int8 keys (the top num_bit bits of a sign-magnitude byte are its bit planes):
 * same scores as qmann_hops_packed without a packing pass -- the choice for short memories and for
 * num_bit = 8, where planes are no smaller than bytes.
 * hop_stride may be 0: every hop then reads the same key / value plane -- layer-wise weight tying with equal
 * formats on every hop (TYPE_WEIGHT_TYING 2, MemN2N/define.h:287; MemN2N.c:1770-1773 copies hop 0's embedding
 * matrices over the others), where the per-hop memories are the same bytes.  `taps` need distinct planes. */
int qmann_hops_i8(const qmann_net *net, const int8_t *keys, const int8_t *vals, size_t hop_stride,
                  const uint32_t *row_off, uint32_t max_slots, const float *u0, float *u_out,
                  const qmann_taps *taps, uint32_t n_query, void *stream);

/* Packed binary codes for the Hamming forms: sign-magnitude bytes [rows][Dp] -> bit planes
 * uint64 [rows][Dp/64][num_bit] (plane 0 = sign bits, plane i = magnitude bit 7-i; bit b of a word
 * is column 64.g + b).  num_bit in 1..8.  sm_codes 16-byte aligned, planes 8-byte aligned (QMANN_EINVAL otherwise). */
int qmann_pack_bitplanes(const int8_t *sm_codes, uint64_t *planes, size_t rows, uint32_t dim_emb_pad,
                         uint32_t num_bit, void *stream);

/* qmann_hops_i8 for QMANN_ATT_HAMMING_V0 / _V1: keys are packed planes (layout above, per hop
 * key_hop_stride bytes apart, num_bit = net->num_bit in {1,2,4,8}), values stay int8.  Scores:
 * V0 = number of agreeing bits over the top num_bit bits of every column (lib/common.c:223-246),
 * V1 = sum over columns of sgn.sgn.sum_{i>=1} 2^-(i+1) [bit i agrees] (lib/common.c:249-312),
 * both on the left-aligned words the CUDA path builds (frac = 31 - iwl, lib/layer_cuda.cu:2515). */
int qmann_hops_packed(const qmann_net *net, const uint64_t *key_planes, size_t key_hop_stride,
                      const int8_t *vals, size_t val_hop_stride, const uint32_t *row_off,
                      uint32_t max_slots, const float *u0, float *u_out, const qmann_taps *taps,
                      uint32_t n_query, void *stream);

/* Answer layer for a batch: logits = W_ans . u (float, ds_ans is always float: MemN2N.c:902-906),
 * softmax over V, arg-max with ties to the highest index, and -- when `answer` is given -- the
 * test-phase bookkeeping of cross_entropy_run mode 3 (cost += -p[answer], match += pred==answer).
 *   w_ans [V][D] float, answer [n_query] uint32 or NULL, pred [n_query] uint32,
 *   probs [n_query][V] float or NULL, cost/match: single device words (accumulated) or NULL.
 * Arithmetic.  qmann_answer_f32_serial sums every logit over the embedding axis serially in float, separate multiply and add:
 * the reference's own order (lib/layer_cuda.cu:70-80), bit-equal logits; e^x through expf, a double total, the quotient of the
 * double division (:2006-2042).  qmann_answer_f32 takes, at the bAbI shapes (dim_emb <= 64, dim_input <= 256, e^x / 2^x base),
 * the FUSED form on the bf16 matrix cores instead (csrc/batch_io.hip::k_answer_mfma): u -- which must lie on an 8-bit grid, as
 * the hop kernels write it -- times W split exactly into three bf16 parts, float accumulation, hardware 2^x; closer to the
 * exact sum than the serial float loop, and within north_star's 1e-5 of it on the probabilities (absolute; relative 1e-5 while
 * |logit| <= 16, where a unit in the last place of a logit is still below 1e-6); predictions agree wherever the two best
 * probabilities differ by more than 1e-6, exact ties go to the highest index in both.  Nine times fewer vector instructions
 * per query at the joint dictionary.  QMANN_ANSWER_EXACT (environment) makes qmann_answer_f32 the serial form everywhere;
 * the drop-in queue behind the cuda_* verbs always runs the serial form. */
int qmann_answer_f32(const qmann_net *net, const float *w_ans, const float *u, const uint32_t *answer,
                     uint32_t *pred, float *probs, float *cost, uint32_t *match, uint32_t n_query,
                     void *stream);
int qmann_answer_f32_serial(const qmann_net *net, const float *w_ans, const float *u, const uint32_t *answer,
                            uint32_t *pred, float *probs, float *cost, uint32_t *match, uint32_t n_query,
                            void *stream);

/* The same answer layer when the answer matrix is on an int8 grid: w_ans_i8 [V][Dp] two's-complement
 * codes of Q(w_fmt); the projection runs on the int8 matrix cores (v_mfma_i32_16x16x64_i8).
 *   probs == NULL: one pass -- projection, running-maximum softmax statistics and arg-max in the accumulator registers, W
 *       tiles shared through LDS, no [n_query][V] logits round trip.  Predictions (ties to the highest index) and the match
 *       count equal the float path exactly (logits are exact integers); the cost -p[answer] agrees within the 1e-5 softmax
 *       tolerance (normaliser accumulated against the running maximum, hardware exp as the reference's __expf).
 *   probs != NULL: two kernels -- logits to the workspace, then the float path's softmax on them: probabilities bit-identical
 *       to qmann_answer_f32_serial on the same grid values.
 * logits_ws: caller-provided workspace, float [n_query][V] (the one-pass form keeps its per-slice records there). */
int qmann_answer_i8(const qmann_net *net, const int8_t *w_ans_i8, qmann_fmt w_fmt, const float *u,
                    float *logits_ws, const uint32_t *answer, uint32_t *pred, float *probs, float *cost,
                    uint32_t *match, uint32_t n_query, void *stream);

/* Story / question embedding for a batch of bag-of-words inputs (dense_mat_fwd / dense_fwd,
 * lib/layer_cuda.cu:3511, :3162), writing the int8 memories directly.
 *   story [rows_total][V] float (word counts + time bit), w_a[h], w_c[h], w_q : [D][V] float. */
int qmann_embed_story(const qmann_net *net, const float *story, uint32_t rows_total,
                      const float *const *w_a, const float *const *w_c, int8_t *keys, int8_t *vals,
                      size_t hop_stride, void *stream);
int qmann_embed_query(const qmann_net *net, const float *question, const float *w_q, float *u0,
                      uint32_t n_query, void *stream);

/* Bag-of-words rows -> word lists, for callers that hold the reference's float pools (dev_m_test / dev_q_test,
 * MemN2N.c:2337-2349) and want the word-index kernels' speed: a row whose non-zero entries are integers 1..16 (counts;
 * the time entry is a 1) becomes words[row][0..15] (ascending indices, an index repeated by its count, 0xFFFF = unused) --
 * feed them to qmann_embed_story_idx with max_words = 16 and time_last = 0 (every entry counts), or to
 * qmann_embed_query_idx.  Any other row (fractional or negative entries such as EN_PE position weights, more than 16 words)
 * gets an empty list and its index is appended to irregular_rows[(*n_irregular)++] (device memory, capacity `rows`;
 * *n_irregular is NOT reset here): redo exactly those rows with the float kernels through qmann_embed_story_rows /
 * qmann_embed_query_rows, which take the list and read its length on the device -- no host round trip. */
int qmann_bow_to_words(const float *bow, uint32_t rows, uint32_t dim_input, uint16_t *words, uint32_t *irregular_rows,
                       uint32_t *n_irregular, void *stream);
int qmann_embed_story_rows(const qmann_net *net, const float *story, uint32_t rows_total, const uint32_t *row_list,
                           const uint32_t *n_list, const float *const *w_a, const float *const *w_c, int8_t *keys,
                           int8_t *vals, size_t hop_stride, void *stream);
int qmann_embed_query_rows(const qmann_net *net, const float *question, const uint32_t *row_list, const uint32_t *n_list,
                           const float *w_q, float *u0, uint32_t n_query, void *stream);

/* Compact wire format for stories: word indices instead of float bag-of-words
 * rows.  words: uint16 [rows][max_words] (max_words <= 16, unused entries 0xFFFF); with time_last the
 * last valid entry of a row is its time-encoding index (bag-of-words entry SET to 1; word entries
 * COUNT occurrences -- MemN2N/sample.c:466-475, 544-548).  Tables: int8 [V][Dp] two's-complement
 * codes of Q(w[h]) made by qmann_quantize_table_i8 from the float [D][V] matrices.  Results are
 * bit-identical to qmann_embed_story / qmann_embed_query on the equivalent bag-of-words input (with net->en_pe: on the
 * question row that carries the position weights). */
int qmann_quantize_table_i8(const float *w, int8_t *table, uint32_t dim_emb, uint32_t dim_emb_pad,
                            uint32_t dim_input, qmann_fmt fmt, void *stream);
/* The way back: a table's codes as the float matrix [D][V] of grid values (code . 2^-frac, exact).  Quantising that matrix
 * again gives the same table, and the float embedding kernels (which quantise their weights on entry, as dense_mat_fwd does)
 * give the same memories from it as from the original matrix -- so a model can live on its int8 tables alone. */
int qmann_dequantize_table_f32(const int8_t *table, float *w, uint32_t dim_emb, uint32_t dim_emb_pad, uint32_t dim_input,
                               qmann_fmt fmt, void *stream);
int qmann_embed_story_idx(const qmann_net *net, const uint16_t *words, uint32_t rows_total, uint32_t max_words,
                          int time_last, const int8_t *const *t_a, const int8_t *const *t_c, int8_t *keys,
                          int8_t *vals, size_t hop_stride, void *stream);
int qmann_embed_query_idx(const qmann_net *net, const uint16_t *words, uint32_t max_words, const int8_t *t_q,
                          float *u0, uint32_t n_query, void *stream);

/* Validation helper: *n_over (a device word, accumulated into) += the number of queries whose story is longer than
 * max_slots, i.e. would be cut by the hop kernels. */
int qmann_check_slots(const uint32_t *row_off, uint32_t n_query, uint32_t max_slots, uint32_t *n_over, void *stream);

/* bytes of LDS one workgroup of qmann_hops_i8 needs for `max_slots` slots (for sizing checks) */
size_t qmann_hops_lds_bytes(uint32_t max_slots);

/* The library's A/B switches (environment variables QMANN_NO_LEAN, QMANN_NO_MID, QMANN_NO_W7, QMANN_NO_TIED, QMANN_NO_TIGHT,
 * QMANN_LEAN_SPARSE, QMANN_EMBED_VALU, QMANN_EMBED_GENERAL_EPILOGUE, QMANN_ANSWER_TWO_PASS: INTEGRATION.md) are read ONCE,
 * at the first call that needs one; no launch touches the environment afterwards.  A host that changes one later calls
 * this to have them read again -- from one thread, while no other thread is inside the library. */
void qmann_tuning_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* QMANN_BATCH_H */
