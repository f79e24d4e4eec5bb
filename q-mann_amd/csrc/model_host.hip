// model_host.hip -- include/qmann_model.h: host-side C++ that owns a model's device parameters and
// runs the whole test-phase forward of a batch through the batched kernels.  No kernels here.
#include "qfmt.h"
#include "rt.h"
#include "../../include/qmann_model.h"

#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct qmann_model {
    qmann_net net{};
    qmann_net emb_net{};     // formats the embedding kernels quantise the memories to (see qmann_model_create)
    uint32_t H = 0, D = 0, Dp = 0, V = 0;
    // Layer-wise weight tying (TYPE_WEIGHT_TYING 2, MemN2N/define.h:287: after every update the reference copies hop 0's
    // embedding matrices over the other hops', MemN2N.c:1770-1773): with equal formats on every hop the hops' memories
    // are the same bytes, so they are embedded ONCE and every hop reads the one plane (hop stride 0)
    bool tied = false;
    // parameters on the device
    float *w_q = nullptr, *w_ans = nullptr;
    float *w_a[QMANN_MAX_HOP] = {}, *w_c[QMANN_MAX_HOP] = {};
    int8_t *lin_map[QMANN_MAX_HOP] = {};
    int8_t *t_q = nullptr, *t_a[QMANN_MAX_HOP] = {}, *t_c[QMANN_MAX_HOP] = {};
    // workspace, grown on demand
    int8_t *keys = nullptr, *vals = nullptr;
    uint64_t *planes = nullptr;
    float *u0 = nullptr, *u = nullptr;
    // bag-of-words input: word lists made from the rows, the rows that are no plain bags of words, two counters
    uint16_t *bow_words = nullptr;
    uint32_t *bow_irr = nullptr;
    size_t cap_rows = 0, cap_plane_words = 0, cap_query = 0, cap_bow = 0;
};

namespace {

bool packed_mode(const qmann_net &n) { return n.attention_mode == QMANN_ATT_HAMMING_V0 || n.attention_mode == QMANN_ATT_HAMMING_V1; }

template <typename T>
void regrow(T **p, size_t n)
{
    if (*p) QM_HIP(hipFree(*p));
    *p = nullptr;
    QM_HIP(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
}

float *upload(const float *host, size_t n, hipStream_t st)
{
    float *d = nullptr;
    QM_HIP(hipMalloc((void **)&d, n * sizeof(float)));
    QM_HIP(hipMemcpyAsync(d, host, n * sizeof(float), hipMemcpyHostToDevice, st));
    return d;
}

// packed planes pay off when they are smaller than the bytes (num_bit < 8) and the memory is long enough for
// bandwidth to matter; otherwise the same scores come straight from the int8 keys.  A plane row must fill a 16-byte load
// (qmann_hops_packed refuses Dp = 64 with a single plane: 8 bytes per row): that case takes the byte form too.
bool use_planes(const qmann_model *m, uint32_t max_slots)
{
    return packed_mode(m->net) && m->net.num_bit < 8 && max_slots > 64 && (m->Dp / 64) * m->net.num_bit * 8 >= 16;
}

int ensure(qmann_model *m, size_t rows, uint32_t n_query, bool planes)
{
    const size_t n_plane = m->tied ? 1 : m->H;          // hop planes held
    if (rows > m->cap_rows || !m->keys) {               // (a first batch may hold no rows at all: the planes still exist)
        const size_t cap = rows + rows / 4 + 1;
        regrow(&m->keys, n_plane * cap * m->Dp);
        regrow(&m->vals, n_plane * cap * m->Dp);
        m->cap_rows = cap;
    }
    if (planes) {
        const size_t words = n_plane * m->cap_rows * (m->Dp / 64) * m->net.num_bit;
        if (words > m->cap_plane_words || !m->planes) { regrow(&m->planes, words); m->cap_plane_words = words; }
    }
    if (n_query > m->cap_query || !m->u0) {
        const size_t cap = (size_t)n_query + n_query / 4;
        regrow(&m->u0, cap * m->D);
        regrow(&m->u, cap * m->D);
        m->cap_query = cap;
    }
    return QMANN_OK;
}

// hops + answer on the memories sitting in the workspace (hop planes rows_total . Dp apart; one shared plane when tied)
int hops_and_answer(qmann_model *m, uint32_t rows_total, const uint32_t *row_off, uint32_t max_slots, uint32_t n_query,
                    const uint32_t *answer, uint32_t *pred, float *cost, uint32_t *match, void *stream)
{
    const size_t hop_stride = m->tied ? 0 : (size_t)rows_total * m->Dp;
    int rc;
    if (use_planes(m, max_slots)) {
        const size_t key_hop_stride = m->tied ? 0 : (size_t)rows_total * (m->Dp / 64) * m->net.num_bit * 8;
        rc = qmann_pack_bitplanes(m->keys, m->planes, (size_t)(m->tied ? 1 : m->H) * rows_total, m->Dp, m->net.num_bit, stream);
        if (rc) return rc;
        rc = qmann_hops_packed(&m->net, m->planes, key_hop_stride, m->vals, hop_stride, row_off, max_slots, m->u0, m->u,
                               nullptr, n_query, stream);
    } else {
        rc = qmann_hops_i8(&m->net, m->keys, m->vals, hop_stride, row_off, max_slots, m->u0, m->u, nullptr, n_query, stream);
    }
    if (rc) return rc;
    return qmann_answer_f32(&m->net, m->w_ans, m->u, answer, pred, nullptr, cost, match, n_query, stream);
}

}  // namespace

extern "C" {

int qmann_model_create(qmann_model **out, const qmann_net *net, const qmann_weights *w, void *stream)
{
    QmBatched qm_scope;
    if (!out || !net || !w) return QMANN_EINVAL;
    *out = nullptr;
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP || net->dim_emb == 0 || net->dim_emb > net->dim_emb_pad) return QMANN_EINVAL;
    if (net->dim_emb_pad != 64 && net->dim_emb_pad != 128 && net->dim_emb_pad != 256) return QMANN_EUNSUPPORTED;
    if (w->n_hop != net->n_hop || w->dim_emb != net->dim_emb || w->dim_input != net->dim_input) return QMANN_EINVAL;
    if (!w->w_q || !w->w_ans) return QMANN_EINVAL;
    for (uint32_t h = 0; h < net->n_hop; h++)
        if (!w->w_a[h] || !w->w_c[h] || (net->en_lin_map && !w->w_h[h])) return QMANN_EINVAL;

    qmann_model *m = new (std::nothrow) qmann_model();
    if (!m) return QMANN_ERANGE;
    m->net = *net;
    m->emb_net = *net;
    // mode 1 runs its attention on the embedding outputs as they are (weight grid, no attention
    // re-quantisation: lib/layer.c:177-195 forces f_fixed = false on both dot_mat_vec layers)
    if (net->attention_mode == QMANN_ATT_FLOAT)
        for (uint32_t h = 0; h < net->n_hop; h++) m->emb_net.att[h] = m->emb_net.act[h] = net->w[h];
    m->H = net->n_hop; m->D = net->dim_emb; m->Dp = net->dim_emb_pad; m->V = net->dim_input;
    hipStream_t st = (hipStream_t)stream;
    const size_t DV = (size_t)m->D * m->V, DD = (size_t)m->D * m->D;
    m->tied = net->n_hop > 1 && !getenv("QMANN_NO_TIED");
    for (uint32_t h = 1; h < net->n_hop && m->tied; h++) {
        const qmann_net &e = m->emb_net;
        m->tied = e.w[h].iwl == e.w[0].iwl && e.w[h].frac == e.w[0].frac && e.att[h].iwl == e.att[0].iwl && e.att[h].frac == e.att[0].frac &&
                  e.act[h].iwl == e.act[0].iwl && e.act[h].frac == e.act[0].frac &&
                  memcmp(w->w_a[h], w->w_a[0], DV * sizeof(float)) == 0 && memcmp(w->w_c[h], w->w_c[0], DV * sizeof(float)) == 0;
    }
    if (m->tied) m->emb_net.n_hop = 1;       // the embedding kernels fill one plane
    std::vector<float *> staged;             // float copies needed only for the conversion below
    m->w_q = upload(w->w_q, DV, st);
    m->w_ans = upload(w->w_ans, DV, st);
    QM_HIP(hipMalloc((void **)&m->t_q, (size_t)m->V * m->Dp));
    int rc = qmann_quantize_table_i8(m->w_q, m->t_q, m->D, m->Dp, m->V, net->w[0], stream);
    for (uint32_t h = 0; h < m->H && rc == QMANN_OK; h++) {
        m->w_a[h] = upload(w->w_a[h], DV, st);
        m->w_c[h] = upload(w->w_c[h], DV, st);
        QM_HIP(hipMalloc((void **)&m->t_a[h], (size_t)m->V * m->Dp));
        QM_HIP(hipMalloc((void **)&m->t_c[h], (size_t)m->V * m->Dp));
        rc = qmann_quantize_table_i8(m->w_a[h], m->t_a[h], m->D, m->Dp, m->V, net->w[h], stream);
        if (rc == QMANN_OK) rc = qmann_quantize_table_i8(m->w_c[h], m->t_c[h], m->D, m->Dp, m->V, net->w[h], stream);
        m->net.lin_map[h] = nullptr;
        if (rc == QMANN_OK && net->en_lin_map) {
            float *wh = upload(w->w_h[h], DD, st);
            staged.push_back(wh);
            QM_HIP(hipMalloc((void **)&m->lin_map[h], (size_t)m->D * m->Dp));
            rc = qmann_quantize_i8(wh, m->lin_map[h], m->D, m->D, m->Dp, net->w[h], QMANN_CODE_SIGNMAG, stream);
            m->net.lin_map[h] = m->lin_map[h];
        }
    }
    QM_HIP(hipStreamSynchronize(st));        // the host arrays and the staged floats are free again
    for (float *p : staged) QM_HIP(hipFree(p));
    if (rc == QMANN_OK) rc = qm_scope.rc();
    if (rc != QMANN_OK) { qmann_model_destroy(m); return rc; }
    *out = m;
    return QMANN_OK;
}

void qmann_model_destroy(qmann_model *m)
{
    if (!m) return;
    auto drop = [](void *p) { if (p) QM_HIP(hipFree(p)); };
    drop(m->w_q); drop(m->w_ans); drop(m->t_q);
    for (uint32_t h = 0; h < QMANN_MAX_HOP; h++) { drop(m->w_a[h]); drop(m->w_c[h]); drop(m->lin_map[h]); drop(m->t_a[h]); drop(m->t_c[h]); }
    drop(m->keys); drop(m->vals); drop(m->planes); drop(m->u0); drop(m->u); drop(m->bow_words); drop(m->bow_irr);
    delete m;
}

int qmann_model_forward_words(qmann_model *m, const uint16_t *story_words, uint32_t rows_total, uint32_t max_words,
                              const uint16_t *question_words, uint32_t max_q_words, const uint32_t *row_off,
                              uint32_t max_slots, uint32_t n_query, const uint32_t *answer, uint32_t *pred,
                              float *cost, uint32_t *match, void *stream)
{
    QmBatched qm_scope;
    if (!m) return QMANN_EINVAL;
    if (n_query == 0) return QMANN_OK;                  // (an empty batch needs no arrays)
    if ((!story_words && rows_total) || !question_words || !row_off || !pred) return QMANN_EINVAL;   // (every story may be empty)
    int rc = ensure(m, rows_total, n_query, use_planes(m, max_slots));
    if (rc) return rc;
    rc = qmann_embed_story_idx(&m->emb_net, story_words, rows_total, max_words, 1, m->t_a, m->t_c, m->keys, m->vals,
                               (size_t)rows_total * m->Dp, stream);
    if (rc) return rc;
    rc = qmann_embed_query_idx(&m->net, question_words, max_q_words, m->t_q, m->u0, n_query, stream);
    if (rc) return rc;
    rc = hops_and_answer(m, rows_total, row_off, max_slots, n_query, answer, pred, cost, match, stream);
    return rc ? rc : qm_scope.rc();
}

int qmann_model_forward_bow(qmann_model *m, const float *story, uint32_t rows_total, const float *question,
                            const uint32_t *row_off, uint32_t max_slots, uint32_t n_query, const uint32_t *answer,
                            uint32_t *pred, float *cost, uint32_t *match, void *stream)
{
    QmBatched qm_scope;
    if (!m) return QMANN_EINVAL;
    if (n_query == 0) return QMANN_OK;
    if ((!story && rows_total) || !question || !row_off || !pred) return QMANN_EINVAL;
    int rc = ensure(m, rows_total, n_query, use_planes(m, max_slots));
    if (rc) return rc;
    // Rows that are plain bags of words (integer counts) go through the word-index kernels -- the int8 gather / matrix-core
    // path, bit-identical to the float path and several times faster; the others (fractional entries: position encoding,
    // long rows) are listed on the device and redone by the float kernels.  No host round trip.
    const size_t n_all = (size_t)rows_total + n_query;
    if (n_all > m->cap_bow || !m->bow_words) {
        const size_t cap = n_all + n_all / 4 + 1;
        regrow(&m->bow_words, cap * 16);
        regrow(&m->bow_irr, cap + 2);
        m->cap_bow = cap;
    }
    hipStream_t st = (hipStream_t)stream;
    uint32_t *n_irr = m->bow_irr, *irr_s = m->bow_irr + 2, *irr_q = irr_s + rows_total;
    uint16_t *sw = m->bow_words, *qw = m->bow_words + (size_t)rows_total * 16;
    QM_HIP(hipMemsetAsync(n_irr, 0, 2 * sizeof(uint32_t), st));
    rc = qmann_bow_to_words(story, rows_total, m->V, sw, irr_s, n_irr, stream);
    if (rc) return rc;
    rc = qmann_bow_to_words(question, n_query, m->V, qw, irr_q, n_irr + 1, stream);
    if (rc) return rc;
    const size_t hop_stride = (size_t)rows_total * m->Dp;
    rc = qmann_embed_story_idx(&m->emb_net, sw, rows_total, 16, /*time_last=*/0, m->t_a, m->t_c, m->keys, m->vals, hop_stride, stream);
    if (rc) return rc;
    rc = qmann_embed_story_rows(&m->emb_net, story, rows_total, irr_s, n_irr, m->w_a, m->w_c, m->keys, m->vals, hop_stride, stream);
    if (rc) return rc;
    qmann_net qnet = m->net;
    qnet.en_pe = 0;                                     // (position weights, if any, are IN the rows: such rows take the float kernel)
    rc = qmann_embed_query_idx(&qnet, qw, 16, m->t_q, m->u0, n_query, stream);
    if (rc) return rc;
    rc = qmann_embed_query_rows(&m->net, question, irr_q, n_irr + 1, m->w_q, m->u0, n_query, stream);
    if (rc) return rc;
    rc = hops_and_answer(m, rows_total, row_off, max_slots, n_query, answer, pred, cost, match, stream);
    return rc ? rc : qm_scope.rc();
}

const float *qmann_model_last_u(const qmann_model *m) { return m ? m->u : nullptr; }

}  // extern "C"
