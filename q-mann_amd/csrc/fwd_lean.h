// fwd_lean.h -- internal interface of the one-kernel forward for bAbI-sized queries (fwd_lean.hip), used by the
// host model (model_host.hip).
#pragma once
#include "qfmt.h"
#include "../../include/qmann_batch.h"

namespace qmann {

struct FwdArgs {
    const uint16_t *story_words;      // [rows_total][max_words]
    const uint16_t *question_words;   // [n_query][max_q_words]
    const uint32_t *row_off;          // [n_query + 1]
    const int8_t *t_q;                // int8 [V][64] two's complement gather tables (qmann_quantize_table_i8)
    const int8_t *t_a[QMANN_MAX_HOP];
    const int8_t *t_c[QMANN_MAX_HOP];
    const float *w_ans;               // [V][D], or NULL: the answer layer stays with the caller
    const uint32_t *answer;           // [n_query] or NULL
    uint32_t *pred;                   // [n_query]
    float *cost;                      // accumulated, or NULL
    uint32_t *match;
    uint32_t n_query, max_words, max_q_words, time_last;
    uint32_t V;                       // filled by fwd_lean()
    QFmt emb_w[QMANN_MAX_HOP], emb_att[QMANN_MAX_HOP], emb_act[QMANN_MAX_HOP];   // filled by fwd_lean() from emb_net
};

// Runs embed + hops (+ answer layer when it fits: *answer_done = 1) for a batch in one launch.  `net`: the model;
// `emb_net`: the formats the memories are quantised to (differs from `net` in attention mode 1 only, which this path
// does not take).  u_out [n_query][D] receives the final hop state either way.  QMANN_EUNSUPPORTED: the caller runs the
// staged pipeline instead (wide embeddings, stories longer than 64 sentences, float attention, packed planes).
int fwd_lean(const qmann_net *net, const qmann_net *emb_net, FwdArgs f, uint32_t max_slots, float *u_out, int *answer_done, void *stream);

}  // namespace qmann
