/*
 * ref_host_infer.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A small C host that drives the REFERENCE'S OWN operator API (lib/layer.h, compiled unmodified
 * from /root/reference/lib/layer.c + common.c where they lie) with en_gpu_model = true, so every
 * layer verb goes layer.c -> cuda_* -> libqmann_hip.so on the MI355X.  It repeats the wiring and
 * call order of the reference's test phase (MemN2N/MemN2N.c:826-912 constructors, :2410-2548
 * pointer wiring, :2626-2697 forward, :2701-2702 accumulator loads); MemN2N.c itself cannot be used
 * for this because it uploads weights only inside its training branch (:993-1031).
 *
 * This file is ours.  Built by `make -C oracle ref` into oracle/_ref/ (binary travels to the GPU
 * box); run by tests/test_gpu_ref_host.py, which checks its outputs against the oracle.
 *
 *   ref_host_infer <in.bin> <out.bin>
 * in : u32 {dim_input, dim_emb, n_hop, n_query, iwl, attention_mode, en_mq, seed}, u32 n_sen[n_query],
 *      f32 story[sum n_sen][dim_input], f32 question[n_query][dim_input], f32 answer[n_query][dim_input]
 * out: f32 w_q[D][V], per hop {w_a, w_c [D][V], w_h [D][D]}, w_ans [V][D], then per query
 *      {u32 pred, f32 u_final[D]}, then u32 match_count_test, f32 cost_test
 */
#include "layer.h"

bool en_gpu_model = true;
bool en_cpu = false;

#define MAXH 8

static void die(const char *m) { fprintf(stderr, "ref_host_infer: %s\n", m); exit(2); }

static void rd(void *p, size_t n, FILE *f) { if (fread(p, 1, n, f) != n) die("short input"); }

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 4) die("usage: ref_host_infer in.bin out.bin [deferred]");
    const int deferred = argc == 4 && !strcmp(argv[3], "deferred");
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) die("cannot open input");
    unsigned hdr[8];
    rd(hdr, sizeof hdr, fi);
    const unsigned V = hdr[0], D = hdr[1], H = hdr[2], NQ = hdr[3], iwl_argv = hdr[4], att_mode = hdr[5];
    const unsigned en_mq = hdr[6], seed = hdr[7];
    if (H == 0 || H > MAXH) die("bad n_hop");
    const unsigned frac_argv = 8 - 1 - iwl_argv;                       /* BW_WL 8, MemN2N.c:213-216 */
    unsigned *n_sen = (unsigned *)malloc(NQ * sizeof(unsigned));
    rd(n_sen, NQ * sizeof(unsigned), fi);
    unsigned total = 0, max_line = 0, i, h, q;
    for (q = 0; q < NQ; q++) { total += n_sen[q]; if (n_sen[q] > max_line) max_line = n_sen[q]; }
    float *m_test = (float *)malloc((size_t)total * V * sizeof(float));
    float *q_test = (float *)malloc((size_t)NQ * V * sizeof(float));
    float *a_test = (float *)malloc((size_t)NQ * V * sizeof(float));
    rd(m_test, (size_t)total * V * sizeof(float), fi);
    rd(q_test, (size_t)NQ * V * sizeof(float), fi);
    rd(a_test, (size_t)NQ * V * sizeof(float), fi);
    fclose(fi);
    FILE *fp = fopen("/dev/null", "w");

    /* formats, MemN2N.c:714-775 */
    unsigned iwl[MAXH], frac[MAXH], iwl_w[MAXH], frac_w[MAXH], iwl_att[MAXH], frac_att[MAXH];
    for (h = 0; h < H; h++) {
        iwl[h] = iwl_w[h] = iwl_att[h] = iwl_argv;
        frac[h] = frac_w[h] = frac_att[h] = frac_argv;
    }
    if (en_mq && H >= 3) { iwl_w[0] += 1; frac_w[0] -= 1; iwl_w[2] -= 1; frac_w[2] += 1; }
    const unsigned iwl_bin = iwl_argv, frac_bin = frac_argv, f_mode = 3;
    const bool f_fixed = true;

    /* constructors, MemN2N.c:826-912 */
    dense emb_q, ds_ans;
    dense_mat emb_m[MAXH], emb_c[MAXH];
    dot_mat_vec dotmv[MAXH], w_sum[MAXH];
    softmax sf_in[MAXH], sf_out;
    dense lin_map[MAXH];
    sum_vec sv[MAXH];
    cross_entropy ce;
    dense_constructor(&emb_q, V, D, true, 40.0f, "NULL", f_fixed, iwl_w[0], frac_w[0], iwl_w[0], frac_w[0], f_mode, fp);
    for (h = 0; h < H; h++) {
        dense_mat_constructor(&emb_m[h], max_line, V, D, true, 40.0f, f_fixed, iwl_w[h], frac_w[h], f_mode, fp);
        dense_mat_constructor(&emb_c[h], max_line, V, D, true, 40.0f, f_fixed, iwl_w[h], frac_w[h], f_mode, fp);
        if (att_mode == 2)
            dot_mat_vec_constructor(&dotmv[h], max_line, D, D, false, f_fixed, iwl_att[h], frac_att[h], iwl_bin,
                                    frac_bin, f_mode, att_mode, fp);
        else
            dot_mat_vec_constructor(&dotmv[h], max_line, D, D, false, f_fixed, iwl_att[h], frac_att[h], iwl_att[h],
                                    frac_att[h], f_mode, att_mode, fp);
        softmax_constructor(&sf_in[h], max_line, false, false, fp);
        dot_mat_vec_constructor(&w_sum[h], max_line, D, max_line, true, f_fixed, iwl[h], frac[h], iwl[h], frac[h],
                                f_mode, att_mode, fp);
        dense_constructor(&lin_map[h], D, D, true, 20.0f, "NULL", f_fixed, iwl_bin, frac_bin, iwl_w[h], frac_w[h],
                          f_mode, fp);
        sum_vec_constructor(&sv[h], D, f_fixed, iwl[h], frac[h], f_mode, fp);
    }
    dense_constructor(&ds_ans, D, V, true, 40.0f, "NULL", false, 8, 7, 8, 7, f_mode, fp);
    softmax_constructor(&sf_out, V, false, false, fp);
    cross_entropy_constructor(&ce, V, fp);

    /* data pools, MemN2N.c:938-987 and :2337-2349 */
    float *dev_m, *dev_q, *dev_a, *dev_dup_grad;
    cuda_data_constructor(&dev_m, &dev_q, &dev_a, total, V, NQ);
    cuda_dup_grad_constructor(&dev_dup_grad, H, D);
    cuda_data_in(dev_m, dev_q, dev_a, m_test, q_test, a_test, total, V, NQ);

    /* init = random weights + upload (the reference's only upload path, MemN2N.c:993-1031) */
    srand(seed);
    dense_init(&emb_q);
    for (h = 0; h < H; h++) {
        dense_mat_init(&emb_m[h]);
        dense_mat_init(&emb_c[h]);
        dot_mat_vec_init(&dotmv[h]);
        softmax_init(&sf_in[h]);
        dot_mat_vec_init(&w_sum[h]);
        dense_init(&lin_map[h]);
        sum_vec_init(&sv[h]);
    }
    dense_init(&ds_ans);
    softmax_init(&sf_out);
    cross_entropy_init(&ce);

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) die("cannot open output");
    fwrite(emb_q.w_mat[0], sizeof(float), (size_t)D * V, fo);
    for (h = 0; h < H; h++) {
        fwrite(emb_m[h].w_mat[0], sizeof(float), (size_t)D * V, fo);
        fwrite(emb_c[h].w_mat[0], sizeof(float), (size_t)D * V, fo);
        fwrite(lin_map[h].w_mat[0], sizeof(float), (size_t)D * D, fo);
    }
    fwrite(ds_ans.w_mat[0], sizeof(float), (size_t)V * D, fo);

    /* host-side twins of the device operands: the reference's verbs read some of them even in
     * GPU mode (softmax_fwd loads in_vec[0] before looking at en_cpu, lib/layer.c:1163), so they
     * must be valid memory, as they are in MemN2N.c */
    const unsigned big = (V > D ? V : D) > max_line ? (V > D ? V : D) : max_line;
    float *hv = (float *)calloc(big, sizeof(float));
    float **hm = (float **)malloc((max_line ? max_line : 1) * sizeof(float *));
    for (i = 0; i < max_line; i++) hm[i] = hv;

    /* test loop, MemN2N.c:2378-2697 */
    unsigned addr_m = 0;
    float *u_host = (float *)malloc(D * sizeof(float));
    for (q = 0; q < NQ; q++) {
        const unsigned ns = n_sen[q];
        dense_in(&emb_q, hv, hv, &dev_q[(size_t)q * V], NULL);
        for (h = 0; h < H; h++) {
            dense_mat_in(&emb_m[h], ns, hm, hm, &dev_m[addr_m], NULL);
            dense_mat_in(&emb_c[h], ns, hm, hm, &dev_m[addr_m], NULL);
            float *dev_u = (h == 0) ? emb_q.dev_out_vec : sv[h - 1].dev_out_vec;
            dot_mat_vec_in(&dotmv[h], ns, hm, hv, hv, emb_m[h].dev_out_mat, dev_u, NULL);
            softmax_in(&sf_in[h], ns, hv, hv, dotmv[h].dev_out_vec, NULL);
            dot_mat_vec_in(&w_sum[h], ns, hm, hv, hv, emb_c[h].dev_out_mat, sf_in[h].dev_out_vec, NULL);
            dense_in(&lin_map[h], hv, hv, dev_u, NULL);
            sum_vec_in(&sv[h], hv, hv, hv, lin_map[h].dev_out_vec, w_sum[h].dev_out_vec, NULL);
        }
        dense_in(&ds_ans, hv, hv, sv[H - 1].dev_out_vec, NULL);
        softmax_in(&sf_out, V, hv, hv, ds_ans.dev_out_vec, NULL);
        cross_entropy_in(&ce, hv, hv, sf_out.dev_out_vec, &dev_a[(size_t)q * V]);

        dense_fwd(&emb_q, false);
        for (h = 0; h < H; h++) {
            dense_mat_fwd(&emb_m[h], false);
            dense_mat_fwd(&emb_c[h], false);
            dot_mat_vec_fwd(&dotmv[h], false);
            softmax_fwd(&sf_in[h], false);
            dot_mat_vec_fwd(&w_sum[h], false);
            dense_fwd(&lin_map[h], false);
            sum_vec_fwd(&sv[h], false);
        }
        dense_fwd(&ds_ans, false);
        softmax_fwd(&sf_out, false);
        cross_entropy_run(&ce, 3);

        unsigned pred = 0;
        /* argv[3] = "deferred": read nothing back between the queries, as MemN2N.c's own test loop (:2378-2702) -- the
         * library may then run the whole loop as one batch; only the last query's buffers are looked at (they must hold what
         * the serial loop leaves there), the other records are written as zeros */
        if (!deferred || q == NQ - 1) {
            cuda_copy_dev2host((float *)&pred, (float *)ce.dev_pred_i, 1);
            cuda_copy_dev2host(u_host, sv[H - 1].dev_out_vec, D);
        } else {
            memset(u_host, 0, D * sizeof(float));
        }
        fwrite(&pred, sizeof pred, 1, fo);
        fwrite(u_host, sizeof(float), D, fo);
        addr_m += ns * V;
    }
    unsigned m_tr = 0, m_va = 0, m_te = 0;
    float c_tr = 0, c_va = 0, c_te = 0;
    cross_entropy_m_cnt_load(&ce, &m_tr, &m_va, &m_te);
    cross_entropy_cost_load(&ce, &c_tr, &c_va, &c_te);
    fwrite(&m_te, sizeof m_te, 1, fo);
    fwrite(&c_te, sizeof c_te, 1, fo);
    fclose(fo);

    cuda_data_destructor(dev_m, dev_q, dev_a);
    cuda_dup_grad_destructor(dev_dup_grad);
    printf("ref_host_infer: %u queries, match %u\n", NQ, m_te);
    return 0;
}
