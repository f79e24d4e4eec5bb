/*
 * forward_dataset.c -- test-phase inference from the reference's own files, in plain C:
 *   parsed bAbI record files (MemN2N/dataset/...) -> word lists (qmann_dataset.h)
 *   weight files of the reference's layout (qmann_weights.h) -> model object (qmann_model.h)
 *   one forward call for the whole test set -> accuracy, as MemN2N.c's test loop reports it (MemN2N.c:2378-2702).
 *
 *   gcc -std=c99 -I include examples/forward_dataset.c -L q-mann_amd/lib -lqmann_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/q-mann_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o forward_dataset
 *   ./forward_dataset <train_set> <test_set> <weight dir> <iwl> [pred.bin]
 *
 * Formats follow run.sh / MemN2N.c:714-775 for BW_WL 8: activations and attention Q(iwl.7-iwl), weights shifted by
 * EN_MQ (hop 0 one integer bit more, hop 2 one less), dot-product attention (ATTENTION_MODE 2), 3 hops, DIM_EMB 60.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "qmann_dataset.h"
#include "qmann_model.h"

static void die(const char *m) { fprintf(stderr, "forward_dataset: %s\n", m); exit(2); }
static void *to_dev(const void *host, size_t n)
{
    void *d = NULL;
    if (hipMalloc(&d, n ? n : 1) != hipSuccess || hipMemcpy(d, host, n, hipMemcpyHostToDevice) != hipSuccess) die("device copy");
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 5) die("usage: forward_dataset <train_set> <test_set> <weight dir> <iwl> [pred.bin]");
    const uint32_t iwl = (uint32_t)atoi(argv[4]), H = 3, D = 60;
    if (iwl < 1 || iwl > 6) die("iwl must be 1..6");

    qmann_dataset ds;
    if (qmann_dataset_load(argv[1], argv[2], 50, 0, 0, &ds) != QMANN_OK) die("cannot read the record files");
    const uint32_t V = ds.dim_input, nq = ds.n_query;
    uint32_t max_slots = 1;
    for (uint32_t q = 0; q < nq; q++)
        if (ds.row_off[q + 1] - ds.row_off[q] > max_slots) max_slots = ds.row_off[q + 1] - ds.row_off[q];

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

    qmann_model *m = NULL;
    if (qmann_model_create(&m, &net, &w, NULL) != QMANN_OK) die("qmann_model_create");
    uint32_t *d_ro = to_dev(ds.row_off, (nq + 1) * 4), *d_ans = to_dev(ds.answer, nq * 4), *d_pred = NULL;
    uint16_t *d_sw = to_dev(ds.story_words, (size_t)ds.rows_total * ds.max_words * 2);
    uint16_t *d_qw = to_dev(ds.question_words, (size_t)nq * ds.max_q_words * 2);
    const uint32_t zero = 0; const float fzero = 0.0f;
    uint32_t *d_match = to_dev(&zero, 4);
    float *d_cost = to_dev(&fzero, 4);
    if (hipMalloc((void **)&d_pred, nq * 4 + 4) != hipSuccess) die("hipMalloc");
    if (qmann_model_forward_words(m, d_sw, ds.rows_total, ds.max_words, d_qw, ds.max_q_words, d_ro, max_slots, nq, d_ans, d_pred,
                                  d_cost, d_match, NULL) != QMANN_OK)
        die("qmann_model_forward_words");
    uint32_t *pred = malloc(nq * 4 + 4), match = 0;
    float cost = 0;
    if (hipMemcpy(pred, d_pred, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) die("copy back");
    hipMemcpy(&match, d_match, 4, hipMemcpyDeviceToHost); hipMemcpy(&cost, d_cost, 4, hipMemcpyDeviceToHost);
    printf("forward_dataset: %u test stories (%u sentences, dictionary %u, dim_input %u), %u correct (error %.4f), cost %.4f\n",
           nq, ds.rows_total, ds.dim_dict, V, match, nq ? 1.0 - (double)match / nq : 0.0, cost);
    if (argc > 5) {
        FILE *f = fopen(argv[5], "wb");
        if (!f) die("cannot open output");
        fwrite(pred, 4, nq, f); fwrite(&match, 4, 1, f); fwrite(&cost, 4, 1, f);
        fclose(f);
    }
    qmann_model_destroy(m);
    qmann_dataset_free(&ds);
    return 0;
}
