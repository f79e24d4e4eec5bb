/*
 * ref_forward.c -- TEST INFRASTRUCTURE ONLY (the "reference+port" CPU baseline of bench.py).
 *
 * The 3-hop test-phase forward of MemN2N/MemN2N.c:2626-2697 composed, as SURVEY.md 8(d) and
 * BASELINE.md section 2 prescribe, from
 *   (i)  the reference's own LIVE C functions, compiled unmodified from /root/reference/lib
 *        {common.c, layer.c} where they lie (oracle/Makefile, target `refcpu`):
 *            dense_mat_fwd  lib/layer.c:2671-2696   story embedding A / C of every hop
 *            softmax_fwd    lib/layer.c:1184-1258   sf_in[h] and sf_out (the CPU form: 2^(x-max), float total)
 *            sum_vec_fwd    lib/layer.c:1502-1511   sv[h]
 *            hamming_similarity, hamming_similarity_w  lib/common.c:223-312  (attention modes 10 / 11)
 *   (ii) our restatement (oracle/qmann_oracle.c, the "port") for the ops whose CPU bodies are dead
 *        in the reference (lib/layer.c:254-437, 1854-1929 are commented out / print "NOT YET FIX CPU
 *        MODE"): dot_mat_vec_fwd both ways, dense_fwd (emb_q, lin_map, ds_ans), the mode-3 Hamming
 *        attention, arg-max.
 * The library is built twice, every file at the same flags: `gcc -w` with no optimisation flag (the
 * reference's own, MemN2N/Makefile:15-16) and `gcc -w -O2`.  Its cuda_* imports bind to
 * oracle/cuda_stubs.c (abort-if-called); en_gpu_model = false.
 *
 * Results: identical to qo_memn2n_forward{,_mem} with softmax_variant = QO_SM_CPU_POW2
 * (tests/test_ref_forward.py) -- the live reference functions and their restatements agree bit for bit.
 *
 * Also here: a pthread timing loop, so that the baseline is timed without any Python in the loop.
 */
#define _GNU_SOURCE
#include "layer.h"          /* /root/reference/lib */
#include "qmann_oracle.h"

#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>

/* the two globals every program using layer.h must define (lib/layer.h:8-9) */
bool en_gpu_model = false;
bool en_cpu = true;

typedef struct rf_ctx {
    qo_model m;
    unsigned max_sen;
    dense_mat emb_m[QO_MAX_HOP], emb_c[QO_MAX_HOP];
    softmax sf_in[QO_MAX_HOP], sf_out;
    sum_vec sv[QO_MAX_HOP];
    float **rows;                  /* story row pointers for dense_mat_in */
    float *keys, *vals;            /* [max_sen][D] one hop's memories (bag-of-words entry) */
    float *s, *o, *lu, *u, *a, *words_a, *words_b;
} rf_ctx;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* The reference's constructors printf a banner line each (e.g. lib/layer.c:75): stdout is pointed at
 * /dev/null while they run, so a caller's own stdout protocol (bench.py prints one JSON line) stays clean. */
static int quiet_begin(void)
{
    fflush(stdout);
    int saved = dup(1);
    int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) { dup2(nul, 1); close(nul); }
    return saved;
}
static void quiet_end(int saved)
{
    fflush(stdout);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
}

rf_ctx *rf_create(const qo_model *m, unsigned max_sen)
{
    rf_ctx *c = (rf_ctx *)calloc(1, sizeof(rf_ctx));
    const unsigned D = m->dim_emb, V = m->dim_input, H = m->n_hop;
    FILE *nul = fopen("/dev/null", "w");
    int saved = quiet_begin();
    c->m = *m;
    c->max_sen = max_sen ? max_sen : 1;
    for (unsigned h = 0; h < H; h++) {
        if (m->w_a[h]) {       /* bag-of-words entry: emb_m[h], emb_c[h] (constructors MemN2N.c:835,838) */
            dense_mat_constructor(&c->emb_m[h], c->max_sen, V, D, false, 0.0f, m->f_fixed, m->iwl_w[h], m->frac_w[h], 3, nul);
            dense_mat_constructor(&c->emb_c[h], c->max_sen, V, D, false, 0.0f, m->f_fixed, m->iwl_w[h], m->frac_w[h], 3, nul);
            memcpy(c->emb_m[h].w_mat[0], m->w_a[h], (size_t)V * D * sizeof(float));
            memcpy(c->emb_c[h].w_mat[0], m->w_c[h], (size_t)V * D * sizeof(float));
        }
        /* sf_in[h] (MemN2N.c:856), sv[h] (:889) */
        softmax_constructor(&c->sf_in[h], c->max_sen, m->softmax_variant == QO_SM_CPU_EXP_PLAN, m->f_shift_based, nul);
        sum_vec_constructor(&c->sv[h], D, m->f_fixed, m->iwl[h], m->frac[h], 3, nul);
    }
    softmax_constructor(&c->sf_out, V, m->softmax_variant == QO_SM_CPU_EXP_PLAN, false, nul);   /* sf_out, MemN2N.c:910 */
    quiet_end(saved);
    fclose(nul);
    c->rows = (float **)malloc(c->max_sen * sizeof(float *));
    c->keys = (float *)malloc((size_t)c->max_sen * D * sizeof(float));
    c->vals = (float *)malloc((size_t)c->max_sen * D * sizeof(float));
    c->s = (float *)malloc(c->max_sen * sizeof(float));
    c->o = (float *)malloc(D * sizeof(float));
    c->lu = (float *)malloc(D * sizeof(float));
    c->u = (float *)malloc(D * sizeof(float));
    c->a = (float *)malloc(V * sizeof(float));
    return c;
}

void rf_destroy(rf_ctx *c)
{
    if (!c) return;
    /* (the reference's destructors assume device buffers; host buffers of the few structs are left to the process) */
    free(c->rows); free(c->keys); free(c->vals); free(c->s); free(c->o); free(c->lu); free(c->u); free(c->a);
    free(c);
}

/* attention modes 10 / 11: the row scorer of the dead CPU caller lib/layer.c:322-356 around the reference's
 * LIVE hamming_similarity{,_w} (lib/common.c:223-312), on the CUDA word alignment frac = 31 - iwl
 * (lib/layer_cuda.cu:2515); words built by the port's quantiser (pinned against FLOAT2FIXED, ref_quant.npz) */
static void ham_scores(const rf_ctx *c, unsigned h, const float *keys, const float *u, float *s, unsigned n_sen)
{
    const qo_model *m = &c->m;
    const unsigned D = m->dim_emb, iwl = m->iwl_att[h], frac = 31 - iwl;
    int32_t uw[512];
    for (unsigned j = 0; j < D; j++) uw[j] = qo_float2fixed(u[j], iwl, frac);
    for (unsigned r = 0; r < n_sen; r++) {
        float acc = 0.0f;
        for (unsigned j = 0; j < D; j++) {
            const int kw = qo_float2fixed(keys[(size_t)r * D + j], iwl, frac);
            if (m->attention_mode == 10) acc += (float)hamming_similarity(kw, uw[j], m->num_bit);
            else acc += hamming_similarity_w(kw, uw[j], m->num_bit, false);
        }
        s[r] = acc;
    }
}

/* one hop on memories already embedded: MemN2N.c:2644-2666 */
static void rf_hop(rf_ctx *c, unsigned h, const float *keys, const float *vals, unsigned n_sen)
{
    const qo_model *m = &c->m;
    const unsigned D = m->dim_emb;
    float *u = c->u;
    float u_relu[512];
    const float *u_att = u;
    if (m->en_non_lin && h > 0) {                         /* non_lin[h-1], MemN2N.c:2435-2437 (port: CUDA-only fixed-point RELU) */
        qo_activation_fwd(u, u_relu, D, "RELU", m->f_fixed, m->iwl[h - 1], m->frac[h - 1]);
        u_att = u_relu;
    }
    /* dotmv[h]: dead CPU body -> port */
    if (m->attention_mode == 1) qo_dot_mat_vec_fwd(keys, u_att, c->s, n_sen, D, false, false, 0, 0, 0, 0);
    else if (m->attention_mode == 2)
        qo_dot_mat_vec_fwd(keys, u_att, c->s, n_sen, D, false, true, m->iwl_att[h], m->frac_att[h], m->iwl_bin, m->frac_bin);
    else if (m->attention_mode == 3)
        qo_dot_mat_vec_fwd_appx(keys, u_att, c->s, n_sen, D, m->f_fixed, m->iwl_att[h], m->frac_att[h],
                                1 + m->iwl_att[h] + m->frac_att[h], false);
    else ham_scores(c, h, keys, u_att, c->s, n_sen);
    if (m->en_sc_att)
        for (unsigned i = 0; i < n_sen; i++) c->s[i] = c->s[i] * m->sc_att[h];
    /* sf_in[h]: LIVE reference code */
    softmax_in(&c->sf_in[h], n_sen, c->s, NULL, NULL, NULL);
    softmax_fwd(&c->sf_in[h], false);
    const float *p = c->sf_in[h].out_vec;
    /* w_sum[h]: dead CPU body -> port */
    if (m->attention_mode == 1) qo_dot_mat_vec_fwd(vals, p, c->o, n_sen, D, true, false, 0, 0, 0, 0);
    else if (m->attention_mode == 2 || m->attention_mode >= 10)
        qo_dot_mat_vec_fwd(vals, p, c->o, n_sen, D, true, true, m->iwl[h], m->frac[h], m->iwl[h], m->frac[h]);
    else
        qo_dot_mat_vec_fwd_appx(vals, p, c->o, n_sen, D, m->f_fixed, m->iwl[h], m->frac[h], 1 + m->iwl[h] + m->frac[h], true);
    /* lin_map[h]: dead CPU body -> port */
    if (m->en_lin_map)
        qo_dense_fwd(m->w_h[h], u, c->lu, D, D, "NULL", m->f_fixed, m->iwl_bin, m->frac_bin, m->iwl_w[h], m->frac_w[h]);
    else memcpy(c->lu, u, D * sizeof(float));
    /* sv[h]: LIVE reference code */
    sum_vec_in(&c->sv[h], c->lu, c->o, NULL, NULL, NULL, NULL);
    sum_vec_fwd(&c->sv[h], false);
    memcpy(u, c->sv[h].out_vec, D * sizeof(float));
}

/* ds_ans (port) -> sf_out (LIVE) -> arg-max (port) */
static unsigned rf_answer(rf_ctx *c)
{
    const qo_model *m = &c->m;
    const unsigned D = m->dim_emb, V = m->dim_input;
    float u_in[512];
    if (m->en_non_lin) qo_activation_fwd(c->u, u_in, D, "RELU", m->f_fixed, m->iwl[m->n_hop - 1], m->frac[m->n_hop - 1]);
    else memcpy(u_in, c->u, D * sizeof(float));
    qo_dense_fwd(m->w_ans, u_in, c->a, D, V, "NULL", false, 0, 0, 0, 0);
    softmax_in(&c->sf_out, V, c->a, NULL, NULL, NULL);
    softmax_fwd(&c->sf_out, false);
    return qo_argmax_hi(c->sf_out.out_vec, V);
}

/* memories already embedded (the synthetic |mem| = 10 000 workloads): keys/vals [n_hop][n_sen][D], u0 [D] */
unsigned rf_forward_mem(rf_ctx *c, const float *keys, const float *vals, unsigned n_sen, const float *u0, float *u_out)
{
    const unsigned D = c->m.dim_emb;
    if (n_sen > c->max_sen || D > 512) return 0xFFFFFFFFu;
    memcpy(c->u, u0, D * sizeof(float));
    for (unsigned h = 0; h < c->m.n_hop; h++)
        rf_hop(c, h, keys + (size_t)h * n_sen * D, vals + (size_t)h * n_sen * D, n_sen);
    if (u_out) memcpy(u_out, c->u, D * sizeof(float));
    return rf_answer(c);
}

/* from bag-of-words rows: story [n_sen][V], question [V] */
unsigned rf_forward(rf_ctx *c, const float *story, unsigned n_sen, const float *question, float *u_out)
{
    const qo_model *m = &c->m;
    const unsigned D = m->dim_emb, V = m->dim_input;
    if (n_sen > c->max_sen || D > 512 || !m->w_q) return 0xFFFFFFFFu;
    /* emb_q: dead CPU body -> port (constructor MemN2N.c:826) */
    qo_dense_fwd(m->w_q, question, c->u, V, D, "NULL", m->f_fixed, m->iwl_w[0], m->frac_w[0], m->iwl_w[0], m->frac_w[0]);
    for (unsigned i = 0; i < n_sen; i++) c->rows[i] = (float *)story + (size_t)i * V;
    for (unsigned h = 0; h < m->n_hop; h++) {
        /* emb_m[h], emb_c[h]: LIVE reference code */
        dense_mat_in(&c->emb_m[h], n_sen, c->rows, NULL, NULL, NULL);
        dense_mat_fwd(&c->emb_m[h], false);
        dense_mat_in(&c->emb_c[h], n_sen, c->rows, NULL, NULL, NULL);
        dense_mat_fwd(&c->emb_c[h], false);
        for (unsigned i = 0; i < n_sen; i++) {
            memcpy(c->keys + (size_t)i * D, c->emb_m[h].out_mat[i], D * sizeof(float));
            memcpy(c->vals + (size_t)i * D, c->emb_c[h].out_mat[i], D * sizeof(float));
        }
        rf_hop(c, h, c->keys, c->vals, n_sen);
    }
    if (u_out) memcpy(u_out, c->u, D * sizeof(float));
    return rf_answer(c);
}

/* ---- timing: n_threads workers walk a pool of queries until the deadline; no Python in the loop ---- */
typedef struct {
    rf_ctx *c;
    /* pool of n_pool queries; bag-of-words: story[i] [n_sen[i]][V] + question[i]; mem: keys[i], vals[i], u0[i] */
    const float *const *a;        /* story rows or keys */
    const float *const *b;        /* question or vals */
    const float *const *u0;       /* mem form only (NULL = bag-of-words form) */
    const unsigned *n_sen;
    unsigned n_pool, first, stride;
    double deadline;
    unsigned long count;
    unsigned *preds;              /* [n_pool] or NULL: prediction of each pool entry (last write wins) */
} rf_job;

static void *rf_worker(void *arg)
{
    rf_job *j = (rf_job *)arg;
    unsigned i = j->first;
    do {
        const unsigned k = i % j->n_pool;
        const unsigned p = j->u0 ? rf_forward_mem(j->c, j->a[k], j->b[k], j->n_sen[k], j->u0[k], NULL)
                                 : rf_forward(j->c, j->a[k], j->n_sen[k], j->b[k], NULL);
        if (j->preds) j->preds[k] = p;
        j->count++;
        i += j->stride;
    } while (now_s() < j->deadline);
    return NULL;
}

/* Runs for about `seconds` on `n_threads` threads (contexts made up front, outside the timed region).
 * Returns forwards per second; *n_done = forwards completed, *wall = seconds measured. */
double rf_time(const qo_model *m, unsigned max_sen, const float *const *a, const float *const *b, const float *const *u0,
               const unsigned *n_sen, unsigned n_pool, unsigned n_threads, double seconds, unsigned long *n_done,
               double *wall, unsigned *preds)
{
    if (n_threads == 0 || n_pool == 0) return 0.0;
    rf_job *jobs = (rf_job *)calloc(n_threads, sizeof(rf_job));
    pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    for (unsigned t = 0; t < n_threads; t++) {
        jobs[t] = (rf_job){rf_create(m, max_sen), a, b, u0, n_sen, n_pool, t, n_threads, 0.0, 0, preds};
    }
    const double t0 = now_s();
    for (unsigned t = 0; t < n_threads; t++) jobs[t].deadline = t0 + seconds;
    if (n_threads == 1) rf_worker(&jobs[0]);
    else {
        for (unsigned t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, rf_worker, &jobs[t]);
        for (unsigned t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    const double dt = now_s() - t0;
    unsigned long total = 0;
    for (unsigned t = 0; t < n_threads; t++) { total += jobs[t].count; rf_destroy(jobs[t].c); }
    free(jobs); free(th);
    if (n_done) *n_done = total;
    if (wall) *wall = dt;
    return (double)total / dt;
}

/* which flags this copy was built with (the Makefile passes -DRF_FLAGS=...) */
#ifndef RF_FLAGS
#define RF_FLAGS "unknown"
#endif
const char *rf_build_flags(void) { return RF_FLAGS; }
