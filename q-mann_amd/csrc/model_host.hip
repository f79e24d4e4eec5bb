// model_host.hip -- include/qmann_model.h: host-side C++ that owns a model's device parameters and
// runs the whole test-phase forward of a batch through the batched kernels.  No kernels here.
//
// The parameters live in ONE device allocation, the "parameter blob": a header (dimensions, formats, options: the
// qmann_net), the int8 embedding tables, the sign-magnitude linear-map codes and the float answer matrix -- i.e. the
// QUANTISED model, position independent (offsets, no pointers).  It is what a multi-GPU host broadcasts
// (qmann_dist.h: one ncclBroadcast of these bytes) and what qmann_model_create_from_params builds a replica from,
// with no float embedding matrix and no re-quantisation on the receiving side.
#include "qfmt.h"
#include "rt.h"
#include "../../include/qmann_model.h"

#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr uint32_t kBlobMagic = 0x42504D51u;        // "QMPB"
constexpr uint32_t kBlobVersion = 1u;
constexpr size_t kBlobAlign = 256;

struct BlobHeader {
    uint32_t magic, version;
    uint64_t bytes;                                  // the whole blob, header included
    uint32_t tied, reserved;
    qmann_net net;                                   // lin_map pointers are null in a blob (offsets below)
    uint64_t off_tq, off_ta[QMANN_MAX_HOP], off_tc[QMANN_MAX_HOP], off_lm[QMANN_MAX_HOP], off_wans;
};

size_t align_up(size_t x) { return (x + kBlobAlign - 1) & ~(kBlobAlign - 1); }

// section offsets of a model of these dimensions; returns the blob size
size_t blob_layout(const qmann_net &n, BlobHeader *h)
{
    const size_t tab = align_up((size_t)n.dim_input * n.dim_emb_pad), lm = align_up((size_t)n.dim_emb * n.dim_emb_pad);
    size_t o = align_up(sizeof(BlobHeader));
    auto take = [&](size_t bytes) { const size_t at = o; o += bytes; return (uint64_t)at; };
    const uint64_t tq = take(tab);
    uint64_t ta[QMANN_MAX_HOP] = {}, tc[QMANN_MAX_HOP] = {}, l[QMANN_MAX_HOP] = {};
    for (uint32_t i = 0; i < n.n_hop; i++) { ta[i] = take(tab); tc[i] = take(tab); }
    for (uint32_t i = 0; i < n.n_hop; i++) l[i] = n.en_lin_map ? take(lm) : 0;
    const uint64_t wa = take(align_up((size_t)n.dim_input * n.dim_emb * sizeof(float)));
    if (h) {
        h->off_tq = tq; h->off_wans = wa;
        for (uint32_t i = 0; i < QMANN_MAX_HOP; i++) { h->off_ta[i] = ta[i]; h->off_tc[i] = tc[i]; h->off_lm[i] = l[i]; }
        h->bytes = o;
    }
    return o;
}

// the calling thread's current device for the length of a call into a device-bound model
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    explicit DeviceScope(int device)
    {
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) { QM_HIP(hipSetDevice(device)); changed = true; }
    }
    ~DeviceScope() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
};

}  // namespace

struct qmann_model {
    qmann_net net{};
    qmann_net emb_net{};     // formats the embedding kernels quantise the memories to (see bind_sections)
    uint32_t H = 0, D = 0, Dp = 0, V = 0;
    int device = 0;          // the GPU the parameters and the workspace live on; every call runs there
    // Layer-wise weight tying (TYPE_WEIGHT_TYING 2, MemN2N/define.h:287: after every update the reference copies hop 0's
    // embedding matrices over the other hops', MemN2N.c:1770-1773): with equal formats on every hop the hops' memories
    // are the same bytes, so they are embedded ONCE and every hop reads the one plane (hop stride 0)
    bool tied = false;
    // the parameter blob (one allocation) and the sections inside it
    uint8_t *blob = nullptr;
    size_t blob_bytes = 0;
    const float *w_ans = nullptr;
    const int8_t *t_q = nullptr, *t_a[QMANN_MAX_HOP] = {}, *t_c[QMANN_MAX_HOP] = {};
    // float matrices [D][V] for the float embedding kernels (bag-of-words rows that are no plain bags of words): the grid
    // values of the tables, made on the first qmann_model_forward_bow
    float *w_q = nullptr, *w_a[QMANN_MAX_HOP] = {}, *w_c[QMANN_MAX_HOP] = {};
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

// false when the allocation failed (the pointer is then null and the caller must not record the new capacity)
template <typename T>
bool regrow(T **p, size_t n)
{
    if (*p) QM_HIP(hipFree(*p));
    *p = nullptr;
    const hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
        fprintf(stderr, "[*E] HIP : qmann_model workspace of %zu bytes : %s\n", (n ? n : 1) * sizeof(T), hipGetErrorString(e));
        (void)hipGetLastError();
        *p = nullptr;
        return false;
    }
    return true;
}

float *upload(const float *host, size_t n, hipStream_t st)
{
    float *d = nullptr;
    QM_HIP(hipMalloc((void **)&d, n * sizeof(float)));
    if (!d) return nullptr;                  // (recorded by QM_HIP: the entry point returns QMANN_EHIP)
    QM_HIP(hipMemcpyAsync(d, host, n * sizeof(float), hipMemcpyHostToDevice, st));
    return d;
}

// Everything a net must satisfy to be created -- and therefore to be replicated: qmann_model_create_on() and the blob receivers
// (qmann_params_validate, qmann_model_create_from_params) run this SAME check, so a net the root can build is never refused on
// a receiver after the broadcast has run, and a bad one fails on the root before any collective.
int check_net(const qmann_net *net)
{
    if (net->n_hop == 0 || net->n_hop > QMANN_MAX_HOP || net->dim_emb == 0 || net->dim_emb > net->dim_emb_pad || net->dim_input == 0)
        return QMANN_EINVAL;
    if (net->dim_emb_pad != 64 && net->dim_emb_pad != 128 && net->dim_emb_pad != 256) return QMANN_EUNSUPPORTED;
    const qmann_net &n = *net;
    switch (n.attention_mode) {
    case QMANN_ATT_FLOAT: case QMANN_ATT_FIXED: case QMANN_ATT_APPX: case QMANN_ATT_HAMMING_V0: case QMANN_ATT_HAMMING_V1: break;
    default: return QMANN_EINVAL;
    }
    if (n.softmax_base > QMANN_SOFTMAX_EXP_PLAN || n.en_lin_map > 1u || n.num_bit > 8u || n.softmax_shift_based > 1u ||
        n.en_att_scale > 1u || n.en_non_linearity > 1u || n.en_pe > 1u)
        return QMANN_EINVAL;
    auto fmt_ok = [](qmann_fmt f) { return f.iwl + f.frac <= 7u && f.iwl <= 7u; };
    for (uint32_t h = 0; h < n.n_hop; h++)
        if (!fmt_ok(n.act[h]) || !fmt_ok(n.w[h]) || !fmt_ok(n.att[h])) return QMANN_EINVAL;
    if (!fmt_ok(n.bin)) return QMANN_EINVAL;
    return QMANN_OK;
}

// a blob header as it must be: what qmann_params_validate answers and qmann_model_create_from_params relies on
int check_header(const BlobHeader &hd, size_t bytes)
{
    if (hd.magic != kBlobMagic || hd.version != kBlobVersion || hd.bytes != bytes || hd.tied > 1u || hd.reserved != 0u) return QMANN_EINVAL;
    const int rc = check_net(&hd.net);
    if (rc != QMANN_OK) return rc;
    const qmann_net &n = hd.net;
    for (uint32_t h = 0; h < n.n_hop; h++)
        if (n.lin_map[h] != nullptr) return QMANN_EINVAL;             // a blob carries offsets, never pointers
    BlobHeader want = hd;
    if (blob_layout(n, &want) != bytes || memcmp(&want, &hd, sizeof hd) != 0) return QMANN_EINVAL;   // offsets must be the canonical ones
    return QMANN_OK;
}

// dimensions, the embedding formats and the section pointers of a model whose blob is in place
void bind_sections(qmann_model *m, const BlobHeader &h)
{
    m->net = h.net;
    m->emb_net = h.net;
    // mode 1 runs its attention on the embedding outputs as they are (weight grid, no attention
    // re-quantisation: lib/layer.c:177-195 forces f_fixed = false on both dot_mat_vec layers)
    if (h.net.attention_mode == QMANN_ATT_FLOAT)
        for (uint32_t i = 0; i < h.net.n_hop; i++) m->emb_net.att[i] = m->emb_net.act[i] = h.net.w[i];
    m->H = h.net.n_hop; m->D = h.net.dim_emb; m->Dp = h.net.dim_emb_pad; m->V = h.net.dim_input;
    m->tied = h.tied != 0;
    if (m->tied) m->emb_net.n_hop = 1;       // the embedding kernels fill one plane
    m->blob_bytes = (size_t)h.bytes;
    m->t_q = (const int8_t *)(m->blob + h.off_tq);
    m->w_ans = (const float *)(m->blob + h.off_wans);
    for (uint32_t i = 0; i < m->H; i++) {
        m->t_a[i] = (const int8_t *)(m->blob + h.off_ta[i]);
        m->t_c[i] = (const int8_t *)(m->blob + h.off_tc[i]);
        m->net.lin_map[i] = h.net.en_lin_map ? (const int8_t *)(m->blob + h.off_lm[i]) : nullptr;
    }
    for (uint32_t i = m->H; i < QMANN_MAX_HOP; i++) m->net.lin_map[i] = nullptr;
}

// Packed planes pay off when they are smaller than the bytes (num_bit < 8), the memory is long enough for bandwidth to
// matter, AND a plane is read more than once: packing reads every key byte once more and writes the planes, so with a plane
// per hop (each read by its own hop only) the packed path moves 2 + num_bit/4 bytes per key byte where the byte form moves 2
// (embedding store + one scan).  Only tied hops -- one plane scanned by every hop -- win: 2 + (1 + H) num_bit/8 against 1 + H.
// Otherwise the same scores come straight from the int8 keys (the V0 / V1 byte forms).  A plane row must fill a 16-byte load
// (qmann_hops_packed refuses Dp = 64 with a single plane: 8 bytes per row): that case takes the byte form too.
bool use_planes(const qmann_model *m, uint32_t max_slots)
{
    return packed_mode(m->net) && m->net.num_bit < 8 && max_slots > 64 && (m->Dp / 64) * m->net.num_bit * 8 >= 16 && m->tied && m->H > 1;
}

int ensure(qmann_model *m, size_t rows, uint32_t n_query, bool planes)
{
    // A capacity is recorded only when every buffer it describes exists: after a failed allocation the next call tries again
    // (the test is on every pointer, not only the first of a group) and this one returns QMANN_EHIP.
    const size_t n_plane = m->tied ? 1 : m->H;          // hop planes held
    if (rows > m->cap_rows || !m->keys || !m->vals) {   // (a first batch may hold no rows at all: the planes still exist)
        const size_t cap = rows + rows / 4 + 1;
        m->cap_rows = 0;
        const bool ok_k = regrow(&m->keys, n_plane * cap * m->Dp), ok_v = regrow(&m->vals, n_plane * cap * m->Dp);
        if (!ok_k || !ok_v) return QMANN_EHIP;
        m->cap_rows = cap;
    }
    if (planes) {
        const size_t words = n_plane * m->cap_rows * (m->Dp / 64) * m->net.num_bit;
        if (words > m->cap_plane_words || !m->planes) {
            m->cap_plane_words = 0;
            if (!regrow(&m->planes, words)) return QMANN_EHIP;
            m->cap_plane_words = words;
        }
    }
    if (n_query > m->cap_query || !m->u0 || !m->u) {
        const size_t cap = (size_t)n_query + n_query / 4;
        m->cap_query = 0;
        const bool ok_0 = regrow(&m->u0, cap * m->D), ok_u = regrow(&m->u, cap * m->D);
        if (!ok_0 || !ok_u) return QMANN_EHIP;
        m->cap_query = cap;
    }
    return QMANN_OK;
}

// the float matrices of the float embedding kernels: the tables' grid values (qmann_dequantize_table_f32)
int ensure_float_matrices(qmann_model *m, void *stream)
{
    if (m->w_q) return QMANN_OK;
    const size_t DV = (size_t)m->D * m->V;
    float *all = nullptr;
    if (!regrow(&all, DV * (1 + 2 * (size_t)m->H))) return QMANN_EHIP;
    int rc = qmann_dequantize_table_f32(m->t_q, all, m->D, m->Dp, m->V, m->net.w[0], stream);
    for (uint32_t h = 0; h < m->H && rc == QMANN_OK; h++) {
        float *a = all + DV * (1 + 2 * (size_t)h), *c = a + DV;
        rc = qmann_dequantize_table_f32(m->t_a[h], a, m->D, m->Dp, m->V, m->net.w[h], stream);
        if (rc == QMANN_OK) rc = qmann_dequantize_table_f32(m->t_c[h], c, m->D, m->Dp, m->V, m->net.w[h], stream);
        m->w_a[h] = a; m->w_c[h] = c;
    }
    if (rc != QMANN_OK) { QM_HIP(hipFree(all)); for (uint32_t h = 0; h < m->H; h++) m->w_a[h] = m->w_c[h] = nullptr; return rc; }
    m->w_q = all;                            // (owns the one allocation)
    return QMANN_OK;
}

// hops + answer on the memories sitting in the workspace (hop planes rows_total . Dp apart; one shared plane when tied)
int hops_and_answer(qmann_model *m, uint32_t rows_total, const uint32_t *row_off, uint32_t max_slots, uint32_t n_query,
                    const uint32_t *answer, uint32_t *pred, float *cost, uint32_t *match, void *stream)
{
    const size_t hop_stride = m->tied ? 0 : (size_t)rows_total * m->Dp;
    QmRowsHint plane_rows(rows_total);               // (a tied model passes hop_stride = 0: the hop kernels still learn the plane's size)
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

int resolve_device(int device, int *out)
{
    if (device < 0) return hipGetDevice(out) == hipSuccess ? QMANN_OK : QMANN_EHIP;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return QMANN_EHIP;
    if (device >= n) return QMANN_EINVAL;
    *out = device;
    return QMANN_OK;
}

}  // namespace

extern "C" {

int qmann_model_create_on(qmann_model **out, int device, const qmann_net *net, const qmann_weights *w, void *stream)
{
    QmBatched qm_scope;
    if (!out || !net || !w) return QMANN_EINVAL;
    *out = nullptr;
    int rc = check_net(net);
    if (rc) return rc;
    if (w->n_hop != net->n_hop || w->dim_emb != net->dim_emb || w->dim_input != net->dim_input) return QMANN_EINVAL;
    if (!w->w_q || !w->w_ans) return QMANN_EINVAL;
    for (uint32_t h = 0; h < net->n_hop; h++)
        if (!w->w_a[h] || !w->w_c[h] || (net->en_lin_map && !w->w_h[h])) return QMANN_EINVAL;
    int dev = 0;
    if ((rc = resolve_device(device, &dev)) != QMANN_OK) return rc;
    DeviceScope on(dev);

    qmann_model *m = new (std::nothrow) qmann_model();
    if (!m) return QMANN_ERANGE;
    m->device = dev;
    BlobHeader hd{};
    hd.magic = kBlobMagic; hd.version = kBlobVersion;
    hd.net = *net;
    for (uint32_t h = 0; h < QMANN_MAX_HOP; h++) hd.net.lin_map[h] = nullptr;
    blob_layout(*net, &hd);
    const uint32_t H = net->n_hop, D = net->dim_emb, Dp = net->dim_emb_pad, V = net->dim_input;
    const size_t DV = (size_t)D * V, DD = (size_t)D * D;
    // tied embedding matrices (see the struct); mode 1 compares the formats its embedding really uses
    bool tied = H > 1 && !qm_tuning().no_tied;
    for (uint32_t h = 1; h < H && tied; h++) {
        const bool fl = net->attention_mode == QMANN_ATT_FLOAT;
        const qmann_fmt a0 = fl ? net->w[0] : net->att[0], ah = fl ? net->w[h] : net->att[h];
        const qmann_fmt c0 = fl ? net->w[0] : net->act[0], ch = fl ? net->w[h] : net->act[h];
        tied = net->w[h].iwl == net->w[0].iwl && net->w[h].frac == net->w[0].frac && ah.iwl == a0.iwl && ah.frac == a0.frac &&
               ch.iwl == c0.iwl && ch.frac == c0.frac &&
               memcmp(w->w_a[h], w->w_a[0], DV * sizeof(float)) == 0 && memcmp(w->w_c[h], w->w_c[0], DV * sizeof(float)) == 0;
    }
    hd.tied = tied ? 1u : 0u;

    hipStream_t st = (hipStream_t)stream;
    QM_HIP(hipMalloc((void **)&m->blob, (size_t)hd.bytes));
    if (!m->blob) { delete m; return QMANN_EHIP; }
    QM_HIP(hipMemsetAsync(m->blob, 0, (size_t)hd.bytes, st));                 // (padding bytes are part of what is broadcast)
    QM_HIP(hipMemcpyAsync(m->blob, &hd, sizeof hd, hipMemcpyHostToDevice, st));
    bind_sections(m, hd);
    std::vector<float *> staged;             // float copies needed only for the conversions below
    auto table = [&](const float *host, const int8_t *dst, qmann_fmt f) {
        float *d = upload(host, DV, st);
        if (!d) return (int)QMANN_EHIP;
        staged.push_back(d);
        return qmann_quantize_table_i8(d, (int8_t *)dst, D, Dp, V, f, stream);
    };
    rc = table(w->w_q, m->t_q, net->w[0]);                                    // emb_q uses the first hop's format (MemN2N.c:826)
    for (uint32_t h = 0; h < H && rc == QMANN_OK; h++) {
        rc = table(w->w_a[h], m->t_a[h], net->w[h]);
        if (rc == QMANN_OK) rc = table(w->w_c[h], m->t_c[h], net->w[h]);
        if (rc == QMANN_OK && net->en_lin_map) {
            float *wh = upload(w->w_h[h], DD, st);
            if (!wh) { rc = QMANN_EHIP; break; }
            staged.push_back(wh);
            rc = qmann_quantize_i8(wh, (int8_t *)m->net.lin_map[h], D, D, Dp, net->w[h], QMANN_CODE_SIGNMAG, stream);
        }
    }
    if (rc == QMANN_OK) QM_HIP(hipMemcpyAsync((void *)m->w_ans, w->w_ans, DV * sizeof(float), hipMemcpyHostToDevice, st));
    QM_HIP(hipStreamSynchronize(st));        // the host arrays and the staged floats are free again
    for (float *p : staged) QM_HIP(hipFree(p));
    if (qm_scope.rc() != QMANN_OK) rc = qm_scope.rc();     // (a failed allocation shows up as EHIP, not as the EINVAL of the null it left)
    if (rc != QMANN_OK) { qmann_model_destroy(m); return rc; }
    *out = m;
    return QMANN_OK;
}

int qmann_model_create(qmann_model **out, const qmann_net *net, const qmann_weights *w, void *stream)
{
    return qmann_model_create_on(out, -1, net, w, stream);
}

int qmann_model_create_from_params(qmann_model **out, int device, const void *blob, size_t bytes, void *stream)
{
    QmBatched qm_scope;
    if (!out || !blob || bytes < sizeof(BlobHeader)) return QMANN_EINVAL;
    *out = nullptr;
    int dev = 0, rc;
    if ((rc = resolve_device(device, &dev)) != QMANN_OK) return rc;
    DeviceScope on(dev);
    hipStream_t st = (hipStream_t)stream;
    BlobHeader hd;
    QM_HIP(hipMemcpyAsync(&hd, blob, sizeof hd, hipMemcpyDefault, st));
    QM_HIP(hipStreamSynchronize(st));
    if (qm_scope.rc()) return qm_scope.rc();
    if ((rc = check_header(hd, bytes)) != QMANN_OK) return rc;

    qmann_model *m = new (std::nothrow) qmann_model();
    if (!m) return QMANN_ERANGE;
    m->device = dev;
    QM_HIP(hipMalloc((void **)&m->blob, bytes));
    if (!m->blob) { delete m; return QMANN_EHIP; }
    QM_HIP(hipMemcpyAsync(m->blob, blob, bytes, hipMemcpyDefault, st));       // (same device, a peer device, or host memory)
    QM_HIP(hipStreamSynchronize(st));
    bind_sections(m, hd);
    if (qm_scope.rc()) { qmann_model_destroy(m); return qm_scope.rc(); }
    *out = m;
    return QMANN_OK;
}

int qmann_params_validate(const void *host_blob, size_t bytes, qmann_net *net)
{
    if (!host_blob || bytes < sizeof(BlobHeader)) return QMANN_EINVAL;
    BlobHeader hd;
    memcpy(&hd, host_blob, sizeof hd);
    const int rc = check_header(hd, bytes);
    if (rc == QMANN_OK && net) *net = hd.net;
    return rc;
}

void qmann_model_destroy(qmann_model *m)
{
    if (!m) return;
    DeviceScope on(m->device);
    auto drop = [](void *p) { if (p) QM_HIP(hipFree(p)); };
    drop(m->blob); drop(m->w_q);             // (w_a / w_c point into w_q's allocation)
    drop(m->keys); drop(m->vals); drop(m->planes); drop(m->u0); drop(m->u); drop(m->bow_words); drop(m->bow_irr);
    delete m;
}

int qmann_model_device(const qmann_model *m) { return m ? m->device : -1; }

int qmann_model_params(const qmann_model *m, const void **blob, size_t *bytes)
{
    if (!m || !blob || !bytes) return QMANN_EINVAL;
    *blob = m->blob; *bytes = m->blob_bytes;
    return QMANN_OK;
}

int qmann_model_net(const qmann_model *m, qmann_net *net, const float **w_ans)
{
    if (!m || !net) return QMANN_EINVAL;
    *net = m->net;
    if (w_ans) *w_ans = m->w_ans;
    return QMANN_OK;
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
    DeviceScope on(m->device);
    int rc = ensure(m, rows_total, n_query, use_planes(m, max_slots));
    if (rc) return rc;
    // Large batches: the question embedding on a second stream BESIDE the story embedding (forked and joined by events: one ordered
    // step to the caller's stream, both branches to a stream capture).  The story kernels are latency-bound persistent grids that
    // leave 32 .. 80 registers per SIMD lane unallocated; the question kernel is built to fit 32 (batch_io.hip) and is bound by its
    // own stores, so its workgroups run in that room.  Launched AFTER the story kernel, so that one keeps its full residency.
    QmSide *sd = (n_query >= kQmCorunMinQueries && !qm_tuning().no_corun) ? qm_side_stream((hipStream_t)stream) : nullptr;
    if (sd) {
        QM_HIP(hipEventRecord(sd->fork, (hipStream_t)stream));
        QM_HIP(hipStreamWaitEvent(sd->side, sd->fork, 0));
    }
    rc = qmann_embed_story_idx(&m->emb_net, story_words, rows_total, max_words, 1, m->t_a, m->t_c, m->keys, m->vals,
                               (size_t)rows_total * m->Dp, stream);
    const int rc_q = qmann_embed_query_idx(&m->net, question_words, max_q_words, m->t_q, m->u0, n_query, sd ? (void *)sd->side : stream);
    // ... and behind it, still beside the story embedding, the two index lists of a batch that the hop launch will split by
    // story length (they depend on row_off alone; 12 us of memset + kernel + launch gaps off the caller's stream)
    struct SplitScope { ~SplitScope() { qm_split_ready = QmSplitReady{nullptr, 0, 0, nullptr}; } } split_scope;
    if (sd && !rc_q && qm_split_applies(rows_total, n_query, max_slots))
        qm_split_ready = QmSplitReady{row_off, n_query, max_slots, qm_split_early(row_off, n_query, max_slots, (hipStream_t)stream, sd->side)};
    if (sd) {                                            // (joined on every path: a capture must not end with an open branch)
        QM_HIP(hipEventRecord(sd->join, sd->side));
        QM_HIP(hipStreamWaitEvent((hipStream_t)stream, sd->join, 0));
    }
    if (rc) return rc;
    if (rc_q) return rc_q;
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
    DeviceScope on(m->device);
    int rc = ensure(m, rows_total, n_query, use_planes(m, max_slots));
    if (rc) return rc;
    if ((rc = ensure_float_matrices(m, stream)) != QMANN_OK) return rc;
    // Rows that are plain bags of words (integer counts) go through the word-index kernels -- the int8 gather / matrix-core
    // path, bit-identical to the float path and several times faster; the others (fractional entries: position encoding,
    // long rows) are listed on the device and redone by the float kernels.  No host round trip.
    const size_t n_all = (size_t)rows_total + n_query;
    if (n_all > m->cap_bow || !m->bow_words || !m->bow_irr) {
        const size_t cap = n_all + n_all / 4 + 1;
        m->cap_bow = 0;
        const bool ok_w = regrow(&m->bow_words, cap * 16), ok_i = regrow(&m->bow_irr, cap + 2);
        if (!ok_w || !ok_i) return QMANN_EHIP;
        m->cap_bow = cap;
    }
    hipStream_t st = (hipStream_t)stream;
    uint32_t *n_irr = m->bow_irr, *irr_s = m->bow_irr + 2, *irr_q = irr_s + rows_total;
    uint16_t *sw = m->bow_words, *qw = m->bow_words + (size_t)rows_total * 16;
    QM_HIP(hipMemsetAsync(n_irr, 0, 2 * sizeof(uint32_t), st));
    // Both conversions first, then -- as in qmann_model_forward_words -- the question embedding and the length split of a mixed
    // batch on the second stream beside the story embedding.  (The question's CONVERSION beside the story's was measured too:
    // 0.735 against 0.714 ms on task 1 -- its workgroups sit on the CUs when the story embedding's persistent grid arrives.)
    const size_t hop_stride = (size_t)rows_total * m->Dp;
    rc = qmann_bow_to_words(story, rows_total, m->V, sw, irr_s, n_irr, stream);
    if (rc) return rc;
    rc = qmann_bow_to_words(question, n_query, m->V, qw, irr_q, n_irr + 1, stream);
    if (rc) return rc;
    QmSide *sd = (n_query >= kQmCorunMinQueries && !qm_tuning().no_corun) ? qm_side_stream(st) : nullptr;
    void *qs = sd ? (void *)sd->side : stream;
    if (sd) {
        QM_HIP(hipEventRecord(sd->fork, st));
        QM_HIP(hipStreamWaitEvent(sd->side, sd->fork, 0));
    }
    rc = qmann_embed_story_idx(&m->emb_net, sw, rows_total, 16, /*time_last=*/0, m->t_a, m->t_c, m->keys, m->vals, hop_stride, stream);
    if (!rc) rc = qmann_embed_story_rows(&m->emb_net, story, rows_total, irr_s, n_irr, m->w_a, m->w_c, m->keys, m->vals, hop_stride, stream);
    qmann_net qnet = m->net;
    qnet.en_pe = 0;                                     // (position weights, if any, are IN the rows: such rows take the float kernel)
    int rc_q = qmann_embed_query_idx(&qnet, qw, 16, m->t_q, m->u0, n_query, qs);
    if (!rc_q) rc_q = qmann_embed_query_rows(&m->net, question, irr_q, n_irr + 1, m->w_q, m->u0, n_query, qs);
    struct SplitScope { ~SplitScope() { qm_split_ready = QmSplitReady{nullptr, 0, 0, nullptr}; } } split_scope;
    if (sd && !rc_q && qm_split_applies(rows_total, n_query, max_slots))
        qm_split_ready = QmSplitReady{row_off, n_query, max_slots, qm_split_early(row_off, n_query, max_slots, st, sd->side)};
    if (sd) {                                            // (joined on every path: a capture must not end with an open branch)
        QM_HIP(hipEventRecord(sd->join, sd->side));
        QM_HIP(hipStreamWaitEvent(st, sd->join, 0));
    }
    if (rc) return rc;
    if (rc_q) return rc_q;
    rc = hops_and_answer(m, rows_total, row_off, max_slots, n_query, answer, pred, cost, match, stream);
    return rc ? rc : qm_scope.rc();
}

const float *qmann_model_last_u(const qmann_model *m) { return m ? m->u : nullptr; }

}  // extern "C"
