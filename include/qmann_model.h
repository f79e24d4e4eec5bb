/*
 * qmann_model.h -- the test-phase forward of a whole model as ONE host call per batch.
 *
 * What MemN2N/MemN2N.c does per query in its test loop (:2378-2699: pointer wiring :2410-2548,
 * 31 layer calls :2626-2697, bookkeeping :2701-2702) this does per batch: story embedding ->
 * question embedding -> every hop -> answer layer -> prediction / match / cost, through the batched
 * kernels of qmann_batch.h.  The object owns the device copies of the parameters in the layouts the
 * kernels want (int8 gather tables, sign-magnitude lin_map codes; packed bit planes are made per
 * batch, and only where one plane serves several hops -- tied matrices, long memories, num_bit < 8 --
 * since packing re-reads the key bytes: otherwise the Hamming scores come from the int8 keys
 * themselves) and a workspace for the int8 memories that grows on demand.  Every attention mode of
 * qmann_batch.h, the Hamming ones also under EN_MQ's per-hop weight formats.  Host language is C++ inside
 * the library; the interface is plain C.
 *
 * Inputs are DEVICE pointers (the reference's cuda_data_in pools, or word-index arrays); every call
 * takes a stream and returns without synchronising.  One object per host thread / stream.
 *
 * Batches of >= 32 768 queries use a SECOND stream of the library's own beside the caller's (one per device and
 * caller stream, created on first use): the question embedding and the split of a mixed batch by story length run
 * there while the stories are embedded, and the long stories' hop kernel runs there beside the short stories'.
 * Every such branch is forked from and joined back into the caller's stream by events inside the call, so the
 * caller still sees ONE ordered step on its stream, and a stream capture of a forward records both branches
 * (tests/test_gpu_graph.py).  QMANN_NO_CORUN keeps everything on the caller's stream.
 */
#ifndef QMANN_MODEL_H
#define QMANN_MODEL_H

#include "qmann_batch.h"
#include "qmann_weights.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmann_model qmann_model;

/* net: dimensions, attention mode, softmax variant and every Q-format (lin_map pointers are ignored:
 * the object builds its own); w: HOST float matrices as the layer structs hold them (e.g. straight
 * from qmann_weights_load, or the reference's emb_q.w_mat[0] ... after training).  The weights are
 * uploaded and converted on `stream`; the host arrays may be freed when the call returns.
 * Layer-wise weight tying (what the reference trains: TYPE_WEIGHT_TYING 2, MemN2N/define.h:287, MemN2N.c:1770-1773) is
 * detected here: when every hop's embedding matrices equal hop 0's and the hops share their formats (EN_MQ off), the
 * memories are embedded once and all hops read the one plane -- same results, a third of the embedding work. */
int qmann_model_create(qmann_model **out, const qmann_net *net, const qmann_weights *w, void *stream);
void qmann_model_destroy(qmann_model *m);

/* The same on a named GPU (hipSetDevice index; -1 = the calling thread's current device, which is what
 * qmann_model_create uses).  A model is BOUND to its device: parameters and workspace live there, and every later call on
 * the object runs there whatever the calling thread's current device is (it is switched for the length of the call and put
 * back) -- `stream` and the data pointers of those calls must belong to that device.  One object per host thread / stream;
 * different objects (on the same or on different GPUs) may be driven from different threads concurrently: this is how a
 * host shards the reference's serial test loop (MemN2N/MemN2N.c:2378) over the GPUs of a node, one thread per GPU
 * (examples/forward_sharded.c, qmann_dist.h). */
int qmann_model_create_on(qmann_model **out, int device, const qmann_net *net, const qmann_weights *w, void *stream);
int qmann_model_device(const qmann_model *m);

/* The model's QUANTISED parameters as one position-independent blob in device memory (owned by the model, valid until it is
 * destroyed): a header with the qmann_net (dimensions, modes, every Q-format, scale factors), the int8 embedding tables
 * [V][Dp] of emb_q / emb_m[h] / emb_c[h], the sign-magnitude linear-map codes [D][Dp] and the float answer matrix [V][D]
 * (ds_ans is a float layer, MemN2N.c:902-906).  About (1 + 2 H) V Dp + H D Dp + 4 V D bytes: 40 KB for bAbI task 1, 0.5 MB
 * for D = V = 256.  This is what a multi-GPU host broadcasts once (qmann_comm_broadcast_params). */
int qmann_model_params(const qmann_model *m, const void **blob, size_t *bytes);

/* A replica from such a blob -- `blob` may be device memory of this or of a peer GPU, or host memory (a file read back);
 * it is copied, the caller keeps ownership.  No float matrix and no quantisation step is involved: the replica computes
 * from the very bytes the source model computes from, so its results are identical.  (The float embedding kernels that
 * qmann_model_forward_bow uses for irregular rows get the tables' grid values, qmann_dequantize_table_f32.)
 * Returns QMANN_EINVAL for a blob that is not one (magic, version, size or section offsets do not match). */
int qmann_model_create_from_params(qmann_model **out, int device, const void *blob, size_t bytes, void *stream);

/* Is this HOST buffer a parameter blob qmann_model_create_from_params would accept?  Pure host code, no GPU needed (a
 * receiver can vet bytes that came over a wire or out of a file before any device is touched): magic and version, the size
 * field against `bytes`, the net's dimensions (check of qmann_model_create), every Q-format (iwl + frac <= 7, the int8 word),
 * enumerated fields in range, null lin_map pointers, and the section offsets against the canonical layout of those
 * dimensions.  QMANN_OK, or QMANN_EINVAL / QMANN_EUNSUPPORTED exactly as create_from_params would answer.  When `net` is
 * not NULL it receives the header's net (dimensions and formats of the model the blob holds). */
int qmann_params_validate(const void *host_blob, size_t bytes, qmann_net *net);

/* The net as the model uses it -- lin_map[h] filled with the device pointers into the blob -- and the answer matrix: for
 * hosts that drive qmann_hops_i8 / qmann_answer_f32 themselves on memories of their own but take the parameters from a
 * (broadcast) model.  Pointers are valid until the model is destroyed. */
int qmann_model_net(const qmann_model *m, qmann_net *net, const float **w_ans);

/* Forward from word indices (qmann_embed_story_idx's wire format).
 *   story_words [rows_total][max_words] uint16, question_words [n_query][max_q_words] uint16,
 *   row_off [n_query + 1], max_slots >= every story's slot count,
 *   answer [n_query] uint32 or NULL; pred [n_query] uint32;
 *   cost / match: single device words that are ACCUMULATED into (cross_entropy_run mode 3), or NULL. */
int qmann_model_forward_words(qmann_model *m, const uint16_t *story_words, uint32_t rows_total, uint32_t max_words,
                              const uint16_t *question_words, uint32_t max_q_words, const uint32_t *row_off,
                              uint32_t max_slots, uint32_t n_query, const uint32_t *answer, uint32_t *pred,
                              float *cost, uint32_t *match, void *stream);

/* Forward from the reference's float bag-of-words pools (dev_m_test / dev_q_test, MemN2N.c:2337-2349):
 *   story [rows_total][dim_input] float, question [n_query][dim_input] float. */
int qmann_model_forward_bow(qmann_model *m, const float *story, uint32_t rows_total, const float *question,
                            const uint32_t *row_off, uint32_t max_slots, uint32_t n_query, const uint32_t *answer,
                            uint32_t *pred, float *cost, uint32_t *match, void *stream);

/* device pointer to the last batch's final hop state u [n_query][D] (valid until the next call) */
const float *qmann_model_last_u(const qmann_model *m);

#ifdef __cplusplus
}
#endif
#endif /* QMANN_MODEL_H */
