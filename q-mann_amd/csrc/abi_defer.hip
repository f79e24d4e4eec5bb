// abi_defer.hip -- the deferred queue behind the drop-in verbs (see defer.h for the idea).
//
// What the reference's unmodified host does in its test and validation phases (MemN2N/MemN2N.c:2378-2702, :1950-2163) is
// a fixed sequence of layer verbs per query whose pointer wiring names every role: the dense layer whose output feeds the
// first attention is emb_q, the two dense_mat layers reading the same story rows are emb_m[h] / emb_c[h], and so on down to
// cross_entropy_run.  match_query() follows that wiring; a run of queries with the same parameters, formats and
// accumulators goes through ONE qmann_model_forward_bow call on the host's own device pools.  Results are those of the
// batched path (bit-exact integer stages, float stages within the 1e-5 the north star allows); the match count and
// accumulated cost land in the reference's own accumulators, every layer buffer ends up as the serial loop leaves it.
#include "defer.h"
#include "rt.h"
#include "../../include/qmann_abi.h"
#include "../../include/qmann_model.h"

#include <chrono>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace qmdefer {
namespace {

// ops; a longer phase is drained at the next query boundary.  About 4 000 queries of 31 verbs: batching gains nothing beyond
// that, and the record (136 bytes per op) stays under 20 MB of host memory however long the host's test phase is.
constexpr size_t kQueueCap = 1u << 17;

// The verbs of boundary B are stateless launches in the reference; the record behind them is process-wide state, so every
// entry into it (a forward verb recording itself, a synchronisation point draining, the switches below) takes this lock:
// a host that calls cuda_* verbs from several threads gets them serialised, not a corrupted queue.  Recursive: a drain runs
// library calls that are synchronisation points themselves.
std::recursive_mutex g_lock;

int g_mode = -1;                                    // Mode; -1 = not yet read from the environment
std::vector<Op> g_queue;
bool g_draining = false;
qmann_defer_stats g_stats{};
bool g_timing = false;                              // QMANN_DEFER_STATS / verify: bracket the phases with device syncs
const char *g_save_dir = nullptr;                   // QMANN_SAVE_WEIGHTS_DIR
std::string g_verify_log;                           // verify mode: one line per run, printed when the process ends

bool feq(QFmt a, QFmt b) { return a.iwl == b.iwl && a.frac == b.frac; }

// everything the queries of one batched run must share
struct Sig {
    uint32_t H, D, V, att_mode, num_bit, ce_mode, has_lin, has_scale, has_relu, shift, softmax_base, pad;
    QFmt f_w[QMANN_MAX_HOP], f_att[QMANN_MAX_HOP], f_act[QMANN_MAX_HOP], f_bin;
    const float *Wq, *Wa[QMANN_MAX_HOP], *Wc[QMANN_MAX_HOP], *Wh[QMANN_MAX_HOP], *Wans, *sc[QMANN_MAX_HOP];
    float *cost;
    unsigned *cnt;
    uint64_t wiring;                                // hash of every buffer of the query's ops but its three inputs
};
struct Query {
    const float *q, *m, *a;
    uint32_t n_sen;
    size_t first, n_ops;
};

void exit_report()
{
    static bool registered = false;
    if (registered) return;
    registered = true;
    atexit([] {
        std::lock_guard<std::recursive_mutex> hold(g_lock);      // (a verb may still be running on another thread)
        const qmann_defer_stats s = g_stats;
        fflush(stdout);
        fprintf(stderr, "\n%s", g_verify_log.c_str());
        if (!getenv("QMANN_DEFER_STATS")) return;
        fprintf(stderr, "[qmann defer] verbs queued %llu, replayed one by one %llu; queries batched %llu in %llu runs (%.3f ms), "
                        "op-by-op %.3f ms, models built %llu (%.3f ms); verify: %llu runs, %llu mismatches\n",
                (unsigned long long)s.ops_queued, (unsigned long long)s.ops_replayed, (unsigned long long)s.queries_batched,
                (unsigned long long)s.batches, s.ms_batched, s.ms_replayed, (unsigned long long)s.models_built, s.ms_model,
                (unsigned long long)s.verify_runs, (unsigned long long)s.verify_mismatch);
    });
}

void init_mode()
{
    if (g_mode >= 0) return;
    const char *e = getenv("QMANN_DEFER");
    g_mode = kOn;
    if (e && (!strcmp(e, "0") || !strcmp(e, "off"))) g_mode = kOff;
    else if (e && !strcmp(e, "verify")) g_mode = kVerify;
    if (getenv("QMANN_NO_DEFER")) g_mode = kOff;
    g_save_dir = getenv("QMANN_SAVE_WEIGHTS_DIR");
    g_timing = getenv("QMANN_DEFER_STATS") != nullptr || g_mode == kVerify;
    if (getenv("QMANN_DEFER_STATS") || g_mode == kVerify) exit_report();
}

double now_ms()
{
    if (g_timing) QM_HIP(hipStreamSynchronize(0));
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- pattern ------------------------------------------------------------------------------------------------------------
// ops[i ...] as one query of the test-phase forward (MemN2N.c:2626-2697); returns the number of ops it spans or 0
size_t match_query(const std::vector<Op> &ops, size_t i, Sig &s, Query &qr)
{
    memset(&s, 0, sizeof s);
    uint64_t wh = 1469598103934665603ull;
    auto mix = [&](const void *p) { wh = (wh ^ (uint64_t)(uintptr_t)p) * 1099511628211ull; };
    auto at = [&](size_t j) -> const Op * { return j < ops.size() ? &ops[j] : nullptr; };
    size_t j = i;
    const Op *o = at(j);
    // emb_q: a fixed-point dense layer whose input and weight formats agree (MemN2N.c:826)
    if (!o || o->kind != kDense || !o->fixed || o->act != 0 || !feq(o->fa, o->fb)) return 0;
    s.V = o->c; s.D = o->r; s.Wq = o->w;
    qr.q = o->in;
    const QFmt f_q = o->fb;
    const float *u_att = o->out, *u_lin = o->out;   // what the attention / the linear map of the next hop read
    mix(o->out);
    j++;
    uint32_t h = 0;
    bool float_att = false;
    for (;; h++) {
        o = at(j);
        if (!o || o->kind != kDenseMat) break;
        if (h >= QMANN_MAX_HOP) return 0;
        // emb_m[h], emb_c[h]: two fixed-point dense_mat layers over the SAME story rows
        if (!o->fixed || o->c != s.V || o->k != s.D) return 0;
        if (h == 0) { qr.m = o->in; qr.n_sen = o->r; }
        else if (o->in != qr.m || o->r != qr.n_sen) return 0;
        s.Wa[h] = o->w; s.f_w[h] = o->fa;
        const float *KM = o->out;
        mix(KM); j++;
        o = at(j);
        if (!o || o->kind != kDenseMat || !o->fixed || o->c != s.V || o->k != s.D || o->in != qr.m || o->r != qr.n_sen ||
            !feq(o->fa, s.f_w[h]))
            return 0;
        s.Wc[h] = o->w;
        const float *CM = o->out;
        mix(CM); j++;
        // scores over the keys with the current hop state
        o = at(j);
        if (!o || (o->kind != kDot && o->kind != kDotAppx) || o->trans || o->in != KM || o->in2 != u_att || o->r != qr.n_sen ||
            o->c != s.D)
            return 0;
        uint32_t mode;
        if (o->kind == kDotAppx) {
            mode = QMANN_ATT_APPX;
            if (o->k != 1u + o->fa.iwl + o->fa.frac) return 0;               // lib/layer.c:230
            s.f_att[h] = o->fa; s.num_bit = o->k;
        } else if (o->fixed) {
            mode = QMANN_ATT_FIXED;
            s.f_att[h] = o->fa;
            if (h == 0) s.f_bin = o->fb; else if (!feq(o->fb, s.f_bin)) return 0;
        } else {
            mode = QMANN_ATT_FLOAT;
            float_att = true;
        }
        if (h == 0) s.att_mode = mode; else if (mode != s.att_mode) return 0;
        const float *SC = o->out;
        mix(SC); j++;
        // [scale] softmax
        o = at(j);
        if (o && o->kind == kScale) {
            if (o->in != SC || o->r != qr.n_sen) return 0;
            if (h == 0) s.has_scale = 1; else if (!s.has_scale) return 0;
            s.sc[h] = o->w; SC = o->out;
            mix(SC); j++;
            o = at(j);
        } else if (h > 0 && s.has_scale) return 0;
        if (!o || o->kind != kSoftmax || o->in != SC || o->r != qr.n_sen) return 0;
        if (h == 0) s.shift = o->shift ? 1 : 0; else if ((o->shift ? 1u : 0u) != s.shift) return 0;
        const float *P = o->out;
        mix(P); mix(o->aux); j++;
        // read-out
        o = at(j);
        if (!o || o->kind != kDot || !o->trans || o->in != CM || o->in2 != P || o->r != qr.n_sen || o->c != s.D) return 0;
        if (o->fixed == float_att) return 0;                                 // float attention <=> float read-out (lib/layer.c:188, :207)
        s.f_act[h] = o->fa;
        const float *O = o->out;
        mix(O); j++;
        // [linear map] sum
        o = at(j);
        const float *A = u_lin;
        if (o && o->kind == kDense) {
            if (!o->fixed || o->act != 0 || o->in != u_lin || o->r != s.D || o->c != s.D || !feq(o->fb, s.f_w[h])) return 0;
            if (s.att_mode == QMANN_ATT_FIXED) { if (!feq(o->fa, s.f_bin)) return 0; }
            else if (h == 0) s.f_bin = o->fa; else if (!feq(o->fa, s.f_bin)) return 0;
            if (h == 0) s.has_lin = 1; else if (!s.has_lin) return 0;
            s.Wh[h] = o->w; A = o->out;
            mix(A); j++;
            o = at(j);
        } else if (h > 0 && s.has_lin) return 0;
        if (!o || o->kind != kSumVec || !o->fixed || o->in != A || o->in2 != O || o->r != s.D) return 0;
        if (float_att) s.f_act[h] = o->fa; else if (!feq(o->fa, s.f_act[h])) return 0;
        const float *SV = o->out;
        mix(SV); j++;
        u_att = u_lin = SV;
        // [RELU]: the attention and the answer layer read the activation, the linear map keeps reading sv (MemN2N.c:2435-2473)
        o = at(j);
        if (o && o->kind == kAct) {
            if (o->act != 2 || o->in != SV || o->r != s.D) return 0;
            if (h == 0) s.has_relu = 1; else if (!s.has_relu) return 0;
            u_att = o->out;
            mix(o->out); j++;
        } else if (h > 0 && s.has_relu) return 0;
    }
    if (h == 0 || !feq(f_q, s.f_w[0])) return 0;
    s.H = h;
    if (s.att_mode != QMANN_ATT_FIXED && !s.has_lin) s.f_bin = s.f_att[0];   // (nothing reads it then)
    if (s.att_mode == QMANN_ATT_FLOAT)                                        // mode 1 memories carry the embedding grid
        for (uint32_t k = 0; k < h; k++) s.f_att[k] = s.f_w[k];
    // the byte arithmetic of the batched mode-3 kernels carries the reference's operand words only for these grid combinations
    // (EN_MQ's among them: qfmt.h::ham_hop_kind); anything else stays with the verbs, which work on the words themselves
    if (s.att_mode == QMANN_ATT_APPX)
        for (uint32_t k = 0; k < h; k++)
            if (ham_hop_kind(k == 0 ? s.f_w[0] : s.f_act[k - 1], s.f_w[k], s.f_att[k]) == kHamNone) return 0;
    // ds_ans (float), output softmax, cross entropy of the valid / test phase
    o = at(j);
    if (!o || o->kind != kDense || o->fixed || o->act != 0 || o->in != u_att || o->c != s.D || o->r != s.V) return 0;
    s.Wans = o->w;
    const float *L = o->out;
    mix(L); j++;
    o = at(j);
    if (!o || o->kind != kSoftmax || o->in != L || o->r != s.V || o->shift) return 0;
    const float *PO = o->out;
    mix(PO); mix(o->aux); j++;
    o = at(j);
    if (!o || o->kind != kCrossEntropy || o->in != PO || o->r != s.V || (o->mode != 2 && o->mode != 3)) return 0;
    qr.a = o->in2;
    s.ce_mode = o->mode; s.cost = o->cost[o->mode - 1]; s.cnt = o->cnt[o->mode - 1];
    mix(o->out); mix(o->pred); j++;
    s.wiring = wh;
    s.softmax_base = (uint32_t)softmax_base();
    qr.first = i; qr.n_ops = j - i;
    return j - i;
}

// ---- the model of a signature ------------------------------------------------------------------------------------------
qmann_model *g_model = nullptr;
Sig g_model_sig;
bool g_model_valid = false;

void drop_model()
{
    if (g_model) qmann_model_destroy(g_model);
    g_model = nullptr; g_model_valid = false;
}

bool d2h(std::vector<float> &dst, const float *src, size_t n)
{
    dst.resize(n);
    return hipMemcpy(dst.data(), src, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
}

// build (or reuse) the batched model for `s` from the layers' DEVICE weight matrices; false = this configuration stays op by op
bool model_for(const Sig &s)
{
    if (g_model_valid && memcmp(&s, &g_model_sig, sizeof s) == 0) return g_model != nullptr;
    const double t0 = now_ms();
    drop_model();
    g_model_sig = s; g_model_valid = true;              // (a refusal is cached too: the next flush does not try again)
    if (s.D > 256) return false;
    qmann_net net{};
    net.n_hop = s.H; net.dim_emb = s.D; net.dim_emb_pad = s.D <= 64 ? 64 : s.D <= 128 ? 128 : 256; net.dim_input = s.V;
    net.attention_mode = s.att_mode; net.softmax_base = s.softmax_base; net.en_lin_map = s.has_lin; net.num_bit = s.num_bit ? s.num_bit : 8;
    net.softmax_shift_based = s.shift; net.en_att_scale = s.has_scale; net.en_non_linearity = s.has_relu;
    net.bin = qmann_fmt{s.f_bin.iwl, s.f_bin.frac};
    std::vector<float> wq, wans, wa[QMANN_MAX_HOP], wc[QMANN_MAX_HOP], whh[QMANN_MAX_HOP];
    qmann_weights w{};
    w.n_hop = s.H; w.dim_emb = s.D; w.dim_input = s.V;
    const size_t DV = (size_t)s.D * s.V, DD = (size_t)s.D * s.D;
    bool ok = d2h(wq, s.Wq, DV) && d2h(wans, s.Wans, DV);
    for (uint32_t h = 0; h < s.H && ok; h++) {
        net.act[h] = qmann_fmt{s.f_act[h].iwl, s.f_act[h].frac};
        net.w[h] = qmann_fmt{s.f_w[h].iwl, s.f_w[h].frac};
        net.att[h] = qmann_fmt{s.f_att[h].iwl, s.f_att[h].frac};
        ok = d2h(wa[h], s.Wa[h], DV) && d2h(wc[h], s.Wc[h], DV) && (!s.has_lin || d2h(whh[h], s.Wh[h], DD));
        if (ok && s.has_scale) ok = hipMemcpy(&net.att_scale[h], s.sc[h], sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
        w.w_a[h] = wa[h].data(); w.w_c[h] = wc[h].data(); w.w_h[h] = s.has_lin ? whh[h].data() : nullptr;
    }
    w.w_q = wq.data(); w.w_ans = wans.data();
    if (!ok) { (void)hipGetLastError(); return false; }
    const int rc = qmann_model_create_on(&g_model, -1, &net, &w, nullptr);
    if (rc != QMANN_OK) g_model = nullptr;
    g_stats.models_built += g_model ? 1 : 0;
    g_stats.ms_model += now_ms() - t0;
    // QMANN_SAVE_WEIGHTS_DIR: the matrices the host program is testing with, in the reference's (disabled) weight-file layout
    // (MemN2N.c:2853-2978 = include/qmann_weights.h), plus the model's quantised parameter blob
    if (g_model && g_save_dir && s.ce_mode == 3) {
        if (qmann_weights_save(g_save_dir, &w, net.w) != QMANN_OK) fprintf(stderr, "[*E] qmann : cannot write the weight files to %s\n", g_save_dir);
        const void *blob = nullptr;
        size_t bytes = 0;
        if (qmann_model_params(g_model, &blob, &bytes) == QMANN_OK) {
            std::vector<unsigned char> host(bytes);
            if (hipMemcpy(host.data(), blob, bytes, hipMemcpyDeviceToHost) == hipSuccess) {
                const std::string path = std::string(g_save_dir) + "/qmann_params.bin";
                if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(host.data(), 1, bytes, f); fclose(f); }
            }
        }
    }
    return g_model != nullptr;
}

// ---- scratch ------------------------------------------------------------------------------------------------------------
struct Scratch {
    uint32_t *row_off = nullptr, *label = nullptr, *pred = nullptr, *counters = nullptr;    // counters: [0] irregular answers, [1] match
    float *cost = nullptr;
    size_t cap = 0;
    bool ensure(size_t n)
    {
        if (n <= cap && row_off) return true;
        for (void *p : {(void *)row_off, (void *)label, (void *)pred, (void *)counters, (void *)cost}) if (p) (void)hipFree(p);
        row_off = label = pred = counters = nullptr; cost = nullptr; cap = 0;
        const size_t c = n + n / 2 + 16;
        if (hipMalloc((void **)&row_off, (c + 1) * 4) != hipSuccess || hipMalloc((void **)&label, c * 4) != hipSuccess ||
            hipMalloc((void **)&pred, c * 4) != hipSuccess || hipMalloc((void **)&counters, 16) != hipSuccess ||
            hipMalloc((void **)&cost, 16) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        cap = c;
        return true;
    }
} g_scr;

// one-hot answer rows -> label indices; rows that are not exactly one-hot (in the sense of cross_entropy_run: entries == 1)
// are counted, and the run then stays op by op
__global__ void k_onehot_labels(const float *__restrict__ y, uint32_t n, uint32_t V, uint32_t *__restrict__ label, uint32_t *n_irregular)
{
    const uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63u;
    if (row >= n) return;
    uint32_t cnt = 0, idx = 0;
    for (uint32_t i = lane; i < V; i += 64)
        if (y[(size_t)row * V + i] == 1.0f) { cnt++; idx = i; }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t c2 = (uint32_t)__shfl_xor((int)cnt, o), i2 = (uint32_t)__shfl_xor((int)idx, o);
        idx = c2 ? (cnt ? (idx > i2 ? idx : i2) : i2) : idx;
        cnt += c2;
    }
    if (lane == 0) {
        label[row] = idx;
        if (cnt != 1) atomicAdd(n_irregular, 1u);
    }
}

// verb-by-verb execution of ops [a, b); timed (device-synchronised when statistics are on) only for whole runs of queries,
// not for the single verbs of a training step
void replay(const std::vector<Op> &ops, size_t a, size_t b, bool timed = true)
{
    const double t0 = timed ? now_ms() : 0.0;
    for (size_t i = a; i < b; i++) run_now(ops[i]);
    g_stats.ops_replayed += b - a;
    if (timed) g_stats.ms_replayed += now_ms() - t0;
}

// queries [a, b) of `qs` share `s` and lie back to back in the host's pools: one batched forward.  false = nothing was done
bool run_batched(const std::vector<Op> &ops, const Sig &s, const std::vector<Query> &qs, size_t a, size_t b, float *cost, unsigned *cnt)
{
    const size_t n = b - a;
    if (!g_scr.ensure(n)) return false;
    std::vector<uint32_t> ro(n + 1);
    uint32_t max_slots = 1;
    ro[0] = 0;
    for (size_t i = 0; i < n; i++) {
        ro[i + 1] = ro[i] + qs[a + i].n_sen;
        if (qs[a + i].n_sen > max_slots) max_slots = qs[a + i].n_sen;
    }
    bool ok = hipMemcpyAsync(g_scr.row_off, ro.data(), (n + 1) * 4, hipMemcpyHostToDevice, 0) == hipSuccess;
    ok = ok && hipMemsetAsync(g_scr.counters, 0, 16, 0) == hipSuccess;
    if (ok) k_onehot_labels<<<(unsigned)((n * 64 + 255) / 256), 256, 0, 0>>>(qs[a].a, (uint32_t)n, s.V, g_scr.label, g_scr.counters);
    uint32_t irregular = 1;
    ok = ok && hipMemcpy(&irregular, g_scr.counters, 4, hipMemcpyDeviceToHost) == hipSuccess;      // (also waits for the row_off copy)
    if (!ok) { (void)hipGetLastError(); return false; }
    if (irregular) return false;
    QmAnswerExact exact_answer_layer;                 // the queue promises the verb-by-verb loop's results: serial-order float sums
    const int rc = qmann_model_forward_bow(g_model, qs[a].m, ro[n], qs[a].q, g_scr.row_off, max_slots, (uint32_t)n, g_scr.label, g_scr.pred,
                                           cost, cnt, nullptr);
    if (rc != QMANN_OK) {
        // e.g. QMANN_EUNSUPPORTED from the hop kernels (operand grids the byte arithmetic cannot carry): nothing has
        // touched the accumulators yet (the answer layer comes last), the run goes op by op, and so will the next ones
        drop_model();
        g_model_valid = true;                       // (keeps g_model_sig: model_for() answers "no" for this signature)
        return false;
    }
    (void)ops;
    return true;
}

// a maximal run of matched queries with one signature: split where the host's pools are not contiguous, batch each piece
void dispatch_run(const std::vector<Op> &ops, const Sig &s, const std::vector<Query> &qs)
{
    if (qs.empty()) return;
    const size_t first_op = qs.front().first, end_op = qs.back().first + qs.back().n_ops;
    if (!model_for(s) || !g_scr.ensure(qs.size())) { replay(ops, first_op, end_op); return; }
    size_t a = 0;
    while (a < qs.size()) {
        size_t b = a + 1;
        while (b < qs.size() && qs[b].q == qs[b - 1].q + s.V && qs[b].a == qs[b - 1].a + s.V &&
               qs[b].m == qs[b - 1].m + (size_t)qs[b - 1].n_sen * s.V)
            b++;
        const size_t op_a = qs[a].first, op_b = qs[b - 1].first + qs[b - 1].n_ops;
        if (g_mode == kVerify) {
            // both ways on the same data: the batched result into scratch accumulators, the op-by-op one into the real ones
            float c0 = 0, c1 = 0, cb = 0;
            unsigned m0 = 0, m1 = 0, mb = 0;
            QM_HIP(hipMemsetAsync(g_scr.cost, 0, 16, 0));      // (scratch sized above: run_batched does not reallocate it)
            const double t0 = now_ms();
            const bool did = run_batched(ops, s, qs, a, b, g_scr.cost, g_scr.counters + 2);
            const double t1 = now_ms();
            QM_HIP(hipMemcpy(&c0, s.cost, 4, hipMemcpyDeviceToHost)); QM_HIP(hipMemcpy(&m0, s.cnt, 4, hipMemcpyDeviceToHost));
            replay(ops, op_a, op_b);
            QM_HIP(hipMemcpy(&c1, s.cost, 4, hipMemcpyDeviceToHost)); QM_HIP(hipMemcpy(&m1, s.cnt, 4, hipMemcpyDeviceToHost));
            if (did) {
                QM_HIP(hipMemcpy(&cb, g_scr.cost, 4, hipMemcpyDeviceToHost)); QM_HIP(hipMemcpy(&mb, g_scr.counters + 2, 4, hipMemcpyDeviceToHost));
                g_stats.verify_runs++;
                g_stats.queries_batched += b - a; g_stats.batches++; g_stats.ms_batched += t1 - t0;
                const bool same = mb == m1 - m0;
                if (!same) g_stats.verify_mismatch++;
                // (kept for the end of the process: a line on stderr in the middle of the run would cut the host's own
                // half-printed stdout lines in two when both go to one file)
                char line[320];
                snprintf(line, sizeof line, "[qmann defer verify] %zu queries (cross_entropy mode %u): batched match %u cost %.6f | op-by-op match %u cost %.6f | %s\n",
                         b - a, s.ce_mode, mb, cb, m1 - m0, c1 - c0, same ? "equal" : "MISMATCH");
                g_verify_log += line;
            }
        } else {
            const double t0 = now_ms();
            if (run_batched(ops, s, qs, a, b, s.cost, s.cnt)) {
                g_stats.queries_batched += b - a; g_stats.batches++;
                g_stats.ms_batched += now_ms() - t0;
                // the serial loop leaves every layer buffer with the LAST query's values: replay that one query, its
                // accumulators pointed at scratch (the batch has counted it already)
                if (b == qs.size()) {
                    for (size_t i = qs[b - 1].first; i < op_b; i++) {
                        Op o = ops[i];
                        if (o.kind == kCrossEntropy) { o.cost[o.mode - 1] = g_scr.cost + 1; o.cnt[o.mode - 1] = g_scr.counters + 1; }
                        run_now(o);
                        g_stats.ops_replayed++;
                    }
                }
            } else {
                replay(ops, op_a, op_b);
            }
        }
        a = b;
    }
}

void drain()
{
    if (g_queue.empty() || g_draining) return;
    g_draining = true;
    std::vector<Op> ops;
    ops.swap(g_queue);
    size_t i = 0;
    Sig run_sig{}, s;
    std::vector<Query> run;
    auto close_run = [&] { dispatch_run(ops, run_sig, run); run.clear(); };
    while (i < ops.size()) {
        Query qr{};
        const size_t n = match_query(ops, i, s, qr);
        if (n) {
            if (!run.empty() && memcmp(&s, &run_sig, sizeof s) != 0) close_run();
            if (run.empty()) run_sig = s;
            run.push_back(qr);
            i += n;
        } else {
            close_run();
            replay(ops, i, i + 1, /*timed=*/false);
            i++;
        }
    }
    close_run();
    g_draining = false;
}

}  // namespace

bool submit(const Op &op)
{
    std::lock_guard<std::recursive_mutex> hold(g_lock);
    init_mode();
    if (g_mode == kOff || g_draining) return false;
    g_queue.push_back(op);
    g_stats.ops_queued++;
    // a training step never batches: its cross entropy (mode 1) is followed by backward verbs; draining here keeps the queue short
    if (op.kind == kCrossEntropy && (op.mode == 1 || g_queue.size() >= kQueueCap)) drain();
    return true;
}

void sync_point(bool writes)
{
    std::lock_guard<std::recursive_mutex> hold(g_lock);
    init_mode();
    if (g_mode != kOff) drain();
    if (writes && g_model_valid) drop_model();      // (also while the queue is switched off: the cache outlives the switch)
}

}  // namespace qmdefer

extern "C" {

void qmann_abi_set_defer(int mode)
{
    std::lock_guard<std::recursive_mutex> hold(qmdefer::g_lock);
    qmdefer::init_mode();
    qmdefer::drain();
    qmdefer::g_mode = mode == 0 ? qmdefer::kOff : mode == 2 ? qmdefer::kVerify : qmdefer::kOn;
    if (mode == 2) { qmdefer::g_timing = true; qmdefer::exit_report(); }
}

// a READ barrier: drains the record and keeps the cached batched model (a host that flushes between forward phases only to read
// layer buffers does not pay a re-quantisation of its weights per flush)
void qmann_abi_flush(void) { qmdefer::sync_point(false); }

// drains AND forgets the cached batched model: the cache is keyed on the weight POINTERS and formats, so a host that changes
// weight values or input pools behind the library's back (its own hipMemcpy) comes through here (qmann_abi.h: CONTRACT)
void qmann_abi_invalidate_model(void) { qmdefer::sync_point(true); }

void qmann_abi_defer_stats(qmann_defer_stats *out)
{
    std::lock_guard<std::recursive_mutex> hold(qmdefer::g_lock);
    if (out) *out = qmdefer::g_stats;
}

}  // extern "C"
