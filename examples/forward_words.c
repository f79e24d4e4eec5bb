/*
 * forward_words.c -- a plain C host using the library the way a maintainer of the reference would:
 * weight files -> qmann_model -> one call per batch of stories given as word indices.
 *
 *   gcc -std=c99 -I include examples/forward_words.c -L q-mann_amd/lib -lqmann_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/q-mann_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o forward_words
 *   ./forward_words <weight dir> <batch.bin> <pred.bin>
 *
 * batch.bin (little endian): u32 {dim_input, dim_emb, n_hop, iwl, n_query, rows_total, max_words, max_q_words,
 * max_slots}, u32 row_off[n_query + 1], u16 story_words[rows_total][max_words],
 * u16 question_words[n_query][max_q_words], u32 answer[n_query].
 * pred.bin: u32 pred[n_query], u32 match, f32 cost.
 * Formats follow run.sh / MemN2N.c:714-775 for BW_WL 8: activations and attention Q(iwl.7-iwl), weights
 * shifted by EN_MQ (hop 0 one integer bit more, hop 2 one less), dot-product attention (mode 2).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "qmann_model.h"

static void die(const char *m) { fprintf(stderr, "forward_words: %s\n", m); exit(2); }
static void rd(void *p, size_t n, FILE *f) { if (fread(p, 1, n, f) != n) die("short input"); }
static void *to_dev(const void *host, size_t n)
{
    void *d = NULL;
    if (hipMalloc(&d, n ? n : 1) != hipSuccess || hipMemcpy(d, host, n, hipMemcpyHostToDevice) != hipSuccess) die("device copy");
    return d;
}

int main(int argc, char **argv)
{
    if (argc != 4) die("usage: forward_words <weight dir> <batch.bin> <pred.bin>");
    FILE *f = fopen(argv[2], "rb");
    if (!f) die("cannot open batch");
    uint32_t hd[9];
    rd(hd, sizeof hd, f);
    const uint32_t V = hd[0], D = hd[1], H = hd[2], iwl = hd[3], nq = hd[4], rows = hd[5], mw = hd[6], mqw = hd[7], max_slots = hd[8];
    if (H == 0 || H > QMANN_MAX_HOP || iwl < 1 || iwl > 6) die("bad header");
    uint32_t *row_off = malloc((nq + 1) * sizeof *row_off), *answer = malloc(nq * sizeof *answer);
    uint16_t *sw = malloc((size_t)rows * mw * 2), *qw = malloc((size_t)nq * mqw * 2);
    rd(row_off, (nq + 1) * sizeof *row_off, f); rd(sw, (size_t)rows * mw * 2, f); rd(qw, (size_t)nq * mqw * 2, f);
    rd(answer, nq * sizeof *answer, f);
    fclose(f);

    qmann_net net = {0};
    net.n_hop = H; net.dim_emb = D; net.dim_emb_pad = (D + 63) / 64 * 64 > 128 ? 256 : (D + 63) / 64 * 64; net.dim_input = V;
    net.attention_mode = QMANN_ATT_FIXED; net.softmax_base = QMANN_SOFTMAX_EXP; net.en_lin_map = 1; net.num_bit = 8;
    for (uint32_t h = 0; h < H; h++) {
        net.act[h].iwl = net.att[h].iwl = net.w[h].iwl = iwl;
        net.act[h].frac = net.att[h].frac = net.w[h].frac = 7 - iwl;
    }
    if (H >= 3) { net.w[0].iwl += 1; net.w[0].frac -= 1; net.w[2].iwl -= 1; net.w[2].frac += 1; }
    net.bin.iwl = iwl; net.bin.frac = 7 - iwl;

    qmann_weights w = {0};
    w.n_hop = H; w.dim_emb = D; w.dim_input = V;
    w.w_q = malloc((size_t)D * V * sizeof(float)); w.w_ans = malloc((size_t)D * V * sizeof(float));
    for (uint32_t h = 0; h < H; h++) {
        w.w_a[h] = malloc((size_t)D * V * sizeof(float)); w.w_c[h] = malloc((size_t)D * V * sizeof(float));
        w.w_h[h] = malloc((size_t)D * D * sizeof(float));
    }
    if (qmann_weights_load(argv[1], &w, 0, net.w) != QMANN_OK) die("weight files missing or of the wrong size");

    qmann_model *m = NULL;
    if (qmann_model_create(&m, &net, &w, NULL) != QMANN_OK) die("qmann_model_create");
    uint32_t *d_ro = to_dev(row_off, (nq + 1) * 4), *d_ans = to_dev(answer, nq * 4), *d_pred = NULL, *d_match = NULL;
    uint16_t *d_sw = to_dev(sw, (size_t)rows * mw * 2), *d_qw = to_dev(qw, (size_t)nq * mqw * 2);
    float *d_cost = NULL;
    const uint32_t zero = 0; const float fzero = 0.0f;
    hipMalloc((void **)&d_pred, nq * 4 + 4);
    d_match = to_dev(&zero, 4); d_cost = to_dev(&fzero, 4);
    if (qmann_model_forward_words(m, d_sw, rows, mw, d_qw, mqw, d_ro, max_slots, nq, d_ans, d_pred, d_cost, d_match, NULL) != QMANN_OK)
        die("qmann_model_forward_words");
    uint32_t *pred = malloc(nq * 4 + 4), match = 0; float cost = 0;
    if (hipMemcpy(pred, d_pred, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) die("copy back");
    hipMemcpy(&match, d_match, 4, hipMemcpyDeviceToHost); hipMemcpy(&cost, d_cost, 4, hipMemcpyDeviceToHost);
    f = fopen(argv[3], "wb");
    if (!f) die("cannot open output");
    fwrite(pred, 4, nq, f); fwrite(&match, 4, 1, f); fwrite(&cost, 4, 1, f);
    fclose(f);
    printf("forward_words: %u queries, %u match the given answers, cost %.4f\n", nq, match, cost);
    qmann_model_destroy(m);
    return 0;
}
