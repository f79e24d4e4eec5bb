/*
 * forward_sharded.c -- the reference's serial test loop (MemN2N/MemN2N.c:2378-2702: one query after the other on one
 * device) as a sharded run over the GPUs of a node, in plain C with one host thread per shard:
 *
 *   parsed bAbI record files -> word lists (qmann_dataset.h)
 *   -> N contiguous shards of the test queries (qmann_shard_range)
 *   -> shard s runs on GPU devices[s mod G]; thread 0 builds the model from the weight files (qmann_weights.h,
 *      qmann_model_create_on), its QUANTISED parameter blob is broadcast once over RCCL (qmann_comm_broadcast_params: one
 *      rank per GPU), every other shard builds a replica from the blob on its GPU (qmann_model_create_from_params)
 *   -> one qmann_model_forward_words call per shard on its own stream
 *   -> predictions concatenated in query order, match counts and costs added up.
 *
 *   gcc -std=c99 -I include examples/forward_sharded.c -L q-mann_amd/lib -lqmann_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/q-mann_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread -o forward_sharded
 *   ./forward_sharded <train_set> <test_set> <weight dir> <iwl> <n_shards> [pred.bin] [devices, e.g. 0,1,2,3] [rccl: auto|on|off]
 *
 * More shards than GPUs is allowed (several threads then share a GPU, each with its own model object and stream): with
 * `n_shards 2` on a one-GPU machine this exercises exactly the thread-safety and the shard / concatenate logic of the
 * multi-GPU run.  rccl "auto" uses the collective when there are at least two GPUs in play; "on" also with one (a
 * one-rank communicator: the same calls, no peer); "off" copies the blob GPU to GPU instead (hipMemcpy peer copy).
 * Formats as in forward_dataset.c (run.sh / MemN2N.c:714-775 for BW_WL 8, EN_MQ, ATTENTION_MODE 2, 3 hops, DIM_EMB 60).
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qmann_dataset.h"
#include "qmann_dist.h"

#define MAX_SHARDS 64

typedef struct shared {
    const qmann_dataset *ds;
    const qmann_net *net;
    const qmann_weights *w;
    uint32_t n_shards, n_gpu, max_slots;
    int devices[MAX_SHARDS];
    int use_rccl;
    unsigned char comm_id[QMANN_COMM_ID_BYTES];
    const void *root_blob;                 /* thread 0's model blob (rccl off: the others copy from it) */
    size_t blob_bytes;
    void *gpu_blob[MAX_SHARDS];            /* per GPU rank: the broadcast copy on that GPU */
    pthread_barrier_t bar;
    uint32_t *pred;                        /* [n_query], every shard writes its range */
    uint32_t match[MAX_SHARDS];
    float cost[MAX_SHARDS];
    int rc[MAX_SHARDS];
    int rccl_version;
} shared;

typedef struct shard_arg { shared *sh; uint32_t s; } shard_arg;

static void die(const char *m) { fprintf(stderr, "forward_sharded: %s\n", m); exit(2); }

#define TRY(call, what)                                                                              \
    do { if (ok && (call) != 0) { fprintf(stderr, "forward_sharded: shard %u: %s failed\n", s, what); ok = 0; } } while (0)

static void *to_dev(const void *host, size_t n, int *ok)
{
    void *d = NULL;
    if (hipMalloc(&d, n ? n : 1) != hipSuccess || (n && hipMemcpy(d, host, n, hipMemcpyHostToDevice) != hipSuccess)) *ok = 0;
    return d;
}

static void *shard_main(void *p)
{
    shard_arg *a = (shard_arg *)p;
    shared *sh = a->sh;
    const uint32_t s = a->s, g = s % sh->n_gpu;            /* GPU rank this shard runs on */
    const int dev = sh->devices[g], leader = s < sh->n_gpu; /* the first shard of every GPU joins the communicator */
    const qmann_dataset *ds = sh->ds;
    int ok = 1;
    hipStream_t st = NULL;
    qmann_model *m = NULL;
    qmann_comm *comm = NULL;

    if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&st) != hipSuccess) { fprintf(stderr, "forward_sharded: shard %u: GPU %d\n", s, dev); ok = 0; }
    if (s == 0) {
        TRY(qmann_model_create_on(&m, dev, sh->net, sh->w, st), "qmann_model_create_on");
        if (ok) TRY(qmann_model_params(m, &sh->root_blob, &sh->blob_bytes), "qmann_model_params");
        if (ok && sh->use_rccl) TRY(qmann_comm_get_id(sh->comm_id), "qmann_comm_get_id");
    }
    /* every rank that will join the communicator checks -- locally, without blocking -- that it CAN (librccl loads, the GPU exists);
     * the verdicts are agreed on at barrier A, before anybody enters the blocking qmann_comm_init_rank (qmann_dist.h) */
    if (ok && leader && sh->use_rccl > 0) TRY(qmann_comm_probe(dev), "qmann_comm_probe");
    sh->rc[s] = ok ? 0 : 1;
    pthread_barrier_wait(&sh->bar);                          /* A: root model and communicator id are there */
    for (uint32_t i = 0; i < sh->n_shards; i++) if (sh->rc[i]) ok = 0;   /* (nobody enters a rendezvous that cannot complete) */
    if (ok && leader && sh->use_rccl > 0) {
        size_t bytes = 0;
        TRY(qmann_comm_init_rank(&comm, (int)sh->n_gpu, (int)g, sh->comm_id, dev), "qmann_comm_init_rank");
        if (ok) TRY(qmann_comm_broadcast_params(comm, 0, s == 0 ? m : NULL, &sh->gpu_blob[g], &bytes, st), "qmann_comm_broadcast_params");
        if (ok && s == 0) qmann_comm_info(comm, NULL, NULL, NULL, &sh->rccl_version);
        if (ok && bytes != sh->blob_bytes) { fprintf(stderr, "forward_sharded: shard %u: blob size\n", s); ok = 0; }
    }
    sh->rc[s] = ok ? 0 : 1;
    pthread_barrier_wait(&sh->bar);                          /* B: every GPU holds the blob */
    for (uint32_t i = 0; i < sh->n_shards; i++) if (sh->rc[i]) ok = 0;
    if (ok && s != 0) {
        const void *src = sh->use_rccl > 0 ? sh->gpu_blob[g] : sh->root_blob;
        TRY(qmann_model_create_from_params(&m, dev, src, sh->blob_bytes, st), "qmann_model_create_from_params");
    }

    /* this shard's queries and their sentences */
    uint32_t lo = 0, hi = 0;
    qmann_shard_range(ds->n_query, s, sh->n_shards, &lo, &hi);
    const uint32_t nq = hi - lo, r0 = ds->row_off[lo], rows = ds->row_off[hi] - r0;
    uint32_t *ro = malloc((nq + 1) * sizeof *ro);
    for (uint32_t q = 0; q <= nq; q++) ro[q] = ds->row_off[lo + q] - r0;
    uint16_t *d_sw = to_dev(ds->story_words + (size_t)r0 * ds->max_words, (size_t)rows * ds->max_words * 2, &ok);
    uint16_t *d_qw = to_dev(ds->question_words + (size_t)lo * ds->max_q_words, (size_t)nq * ds->max_q_words * 2, &ok);
    uint32_t *d_ro = to_dev(ro, (nq + 1) * 4, &ok), *d_ans = to_dev(ds->answer + lo, (size_t)nq * 4, &ok), *d_pred = NULL;
    const uint32_t zero = 0; const float fzero = 0.0f;
    uint32_t *d_match = to_dev(&zero, 4, &ok);
    float *d_cost = to_dev(&fzero, 4, &ok);
    if (hipMalloc((void **)&d_pred, (size_t)nq * 4 + 4) != hipSuccess) ok = 0;
    if (ok) TRY(qmann_model_forward_words(m, d_sw, rows, ds->max_words, d_qw, ds->max_q_words, d_ro, sh->max_slots, nq, d_ans, d_pred,
                                          d_cost, d_match, st), "qmann_model_forward_words");
    if (ok && hipStreamSynchronize(st) != hipSuccess) ok = 0;
    if (ok && nq && hipMemcpy(sh->pred + lo, d_pred, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess) ok = 0;
    if (ok && (hipMemcpy(&sh->match[s], d_match, 4, hipMemcpyDeviceToHost) != hipSuccess ||
               hipMemcpy(&sh->cost[s], d_cost, 4, hipMemcpyDeviceToHost) != hipSuccess)) ok = 0;
    sh->rc[s] = ok ? 0 : 1;
    pthread_barrier_wait(&sh->bar);                          /* C: nobody reads a blob any more */
    hipFree(d_sw); hipFree(d_qw); hipFree(d_ro); hipFree(d_ans); hipFree(d_pred); hipFree(d_match); hipFree(d_cost);
    free(ro);
    if (leader && sh->gpu_blob[g]) qmann_params_free(sh->gpu_blob[g]);
    qmann_comm_destroy(comm);
    qmann_model_destroy(m);
    if (st) hipStreamDestroy(st);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 6) die("usage: forward_sharded <train_set> <test_set> <weight dir> <iwl> <n_shards> [pred.bin] [devices a,b,..] [rccl auto|on|off]");
    const uint32_t iwl = (uint32_t)atoi(argv[4]), H = 3, D = 60;
    const int n_shards = atoi(argv[5]);
    if (iwl < 1 || iwl > 6) die("iwl must be 1..6");
    if (n_shards < 1 || n_shards > MAX_SHARDS) die("n_shards must be 1..64");

    qmann_dataset ds;
    if (qmann_dataset_load(argv[1], argv[2], 50, 0, 0, &ds) != QMANN_OK) die("cannot read the record files");
    const uint32_t V = ds.dim_input, nq = ds.n_query;

    qmann_net net = {0};
    net.n_hop = H; net.dim_emb = D; net.dim_emb_pad = 64; net.dim_input = V;
    net.attention_mode = QMANN_ATT_FIXED; net.softmax_base = QMANN_SOFTMAX_EXP; net.en_lin_map = 1; net.num_bit = 8;
    for (uint32_t h = 0; h < H; h++) {
        net.act[h].iwl = net.att[h].iwl = net.w[h].iwl = iwl;
        net.act[h].frac = net.att[h].frac = net.w[h].frac = 7 - iwl;
    }
    net.w[0].iwl += 1; net.w[0].frac -= 1; net.w[2].iwl -= 1; net.w[2].frac += 1;          /* EN_MQ */
    net.bin.iwl = iwl; net.bin.frac = 7 - iwl;

    qmann_weights w = {0};
    w.n_hop = H; w.dim_emb = D; w.dim_input = V;
    w.w_q = malloc((size_t)D * V * sizeof(float)); w.w_ans = malloc((size_t)D * V * sizeof(float));
    for (uint32_t h = 0; h < H; h++) {
        w.w_a[h] = malloc((size_t)D * V * sizeof(float)); w.w_c[h] = malloc((size_t)D * V * sizeof(float));
        w.w_h[h] = malloc((size_t)D * D * sizeof(float));
    }
    if (qmann_weights_load(argv[3], &w, 0, net.w) != QMANN_OK) die("weight files missing or of the wrong size for this dictionary");

    shared sh;
    memset(&sh, 0, sizeof sh);
    sh.ds = &ds; sh.net = &net; sh.w = &w; sh.n_shards = (uint32_t)n_shards;
    sh.max_slots = 1;
    for (uint32_t q = 0; q < nq; q++)
        if (ds.row_off[q + 1] - ds.row_off[q] > sh.max_slots) sh.max_slots = ds.row_off[q + 1] - ds.row_off[q];
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) die("no GPU");
    uint32_t n_list = 0;
    if (argc > 7 && strcmp(argv[7], "all") != 0) {          /* explicit device list */
        char *t = strtok(argv[7], ",");
        while (t && n_list < MAX_SHARDS) {
            const int d = atoi(t);
            if (d < 0 || d >= n_dev) die("device index out of range");
            for (uint32_t i = 0; i < n_list; i++) if (sh.devices[i] == d) die("a GPU is listed twice (use more shards than GPUs to share one)");
            sh.devices[n_list++] = d;
            t = strtok(NULL, ",");
        }
    } else {
        for (int d = 0; d < n_dev && n_list < MAX_SHARDS; d++) sh.devices[n_list++] = d;
    }
    sh.n_gpu = n_list < sh.n_shards ? n_list : sh.n_shards;
    const char *mode = argc > 8 ? argv[8] : "auto";
    sh.use_rccl = !strcmp(mode, "on") ? 1 : !strcmp(mode, "off") ? 0 : (sh.n_gpu >= 2);
    sh.pred = calloc(nq + 1, sizeof *sh.pred);
    if (pthread_barrier_init(&sh.bar, NULL, sh.n_shards) != 0) die("pthread_barrier_init");

    pthread_t th[MAX_SHARDS];
    shard_arg args[MAX_SHARDS];
    for (uint32_t s = 0; s < sh.n_shards; s++) {
        args[s].sh = &sh; args[s].s = s;
        if (pthread_create(&th[s], NULL, shard_main, &args[s]) != 0) die("pthread_create");
    }
    for (uint32_t s = 0; s < sh.n_shards; s++) pthread_join(th[s], NULL);
    pthread_barrier_destroy(&sh.bar);
    uint32_t match = 0;
    float cost = 0.0f;
    for (uint32_t s = 0; s < sh.n_shards; s++) {
        if (sh.rc[s]) die("a shard failed");
        match += sh.match[s]; cost += sh.cost[s];
    }
    printf("forward_sharded: %u test stories in %u shards on %u GPU(s), parameters by %s%s: %u correct (error %.4f), cost %.4f\n", nq,
           sh.n_shards, sh.n_gpu, sh.use_rccl > 0 ? "RCCL broadcast of the quantised blob" : "device-to-device copy of the quantised blob",
           sh.use_rccl > 0 ? (sh.n_gpu >= 2 ? " over xGMI" : " (one-rank communicator)") : "", match,
           nq ? 1.0 - (double)match / nq : 0.0, cost);
    if (sh.use_rccl > 0) printf("forward_sharded: RCCL version %d, blob %zu bytes\n", sh.rccl_version, sh.blob_bytes);
    if (argc > 6 && strcmp(argv[6], "-") != 0) {
        FILE *f = fopen(argv[6], "wb");
        if (!f) die("cannot open output");
        fwrite(sh.pred, 4, nq, f); fwrite(&match, 4, 1, f); fwrite(&cost, 4, 1, f);
        fclose(f);
    }
    qmann_dataset_free(&ds);
    return 0;
}
