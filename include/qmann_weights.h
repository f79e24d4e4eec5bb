/*
 * qmann_weights.h -- weight files of a MemN2N model.
 *
 * Revives the file layout of the reference's disabled EN_LOAD_WEIGHT / EN_WRITE_WEIGHT blocks
 * (MemN2N/MemN2N.c:2553-2618 load, :2853-2978 write): raw little-endian files in the working
 * directory, every matrix written COLUMN-major -- for (j < dim_in) for (i < dim_out) w_mat[i][j] --
 * with the per-hop matrices of a file back to back:
 *
 *   w_emb_a_float.bin   n_hop x [V][D] float32      emb_m[h].w_mat  (:2562-2574, :2861-2871)
 *   w_emb_c_float.bin   n_hop x [V][D] float32      emb_c[h].w_mat  (:2576-2590, :2873-2884)
 *   w_emb_q_float.bin           [V][D] float32      emb_q.w_mat     (:2592-2602, :2886-2896)
 *   w_float.bin                 [D][V] float32      ds_ans.w_mat    (:2604-2614, :2898-2909)
 *   w_emb_{a,c,q}_fixed.bin     the same matrices as sign-magnitude int32 words FLOAT2FIXED(w)
 *                               (bit 31 sign, low bits |trunc(w . 2^frac)|; :2912-2975)
 *   w_lin_map_float.bin / w_lin_map_fixed.bin   n_hop x [D][D]: lin_map[h].w_mat -- NOT in the reference's
 *                               block (it predates the linear map, define.h:291); same layout, our addition
 *
 * The answer matrix has no fixed file of its own here: ds_ans is a float layer (MemN2N.c:902-906), so
 * loading "from fixed" reads w_float.bin for it.  Host memory only; no device work.
 */
#ifndef QMANN_WEIGHTS_H
#define QMANN_WEIGHTS_H

#include "qmann_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host matrices, row-major as the layer structs hold them (dense.w_mat[dim_out][dim_in], lib/layer.c:1638-1642). */
typedef struct qmann_weights {
    uint32_t n_hop, dim_emb, dim_input;
    float *w_q;                   /* [D][V]                 */
    float *w_a[QMANN_MAX_HOP];    /* [D][V]                 */
    float *w_c[QMANN_MAX_HOP];    /* [D][V]                 */
    float *w_h[QMANN_MAX_HOP];    /* [D][D], NULL = no linear map */
    float *w_ans;                 /* [V][D]                 */
} qmann_weights;

/* Writes the float files; when `fmt_w` is given (per-hop weight formats, net->w) also the *_fixed files
 * (emb_q uses fmt_w[0], MemN2N.c:826).  Returns QMANN_OK or QMANN_EINVAL / QMANN_EIO. */
int qmann_weights_save(const char *dir, const qmann_weights *w, const qmann_fmt *fmt_w);

/* Fills the caller-allocated matrices of `w` (dims set by the caller).  from_fixed = 0: the float files;
 * from_fixed = 1: the *_fixed files, decoded on the grids `fmt_w` (required then).  A file whose size
 * does not match the dimensions is an error (QMANN_EIO), never a partial load. */
int qmann_weights_load(const char *dir, qmann_weights *w, int from_fixed, const qmann_fmt *fmt_w);

#ifdef __cplusplus
}
#endif
#endif /* QMANN_WEIGHTS_H */
