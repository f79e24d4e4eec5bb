/*
 * ref_glue.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points over the reference's own, unmodified C sources so that
 * Python (ctypes) can run the *live* reference code on chosen inputs.  This
 * file is ours; it is compiled together with /root/reference/lib/common.c,
 * /root/reference/lib/layer.c and /root/reference/MemN2N/sample.c where those
 * lie (see oracle/Makefile, target `ref`), outputs go to oracle/_ref/ only.
 * The 56 cuda_* imports of layer.o resolve against our own drop-in library
 * (libqmann_hip.so) -- with en_gpu_model = false none of them is ever called,
 * the reference's CPU branches run.
 *
 * Used by oracle/gen_golden.py to pin the restated oracle (fixtures under
 * tests/golden/) and, on the GPU box, by bench.py's optional "reference"-kind
 * CPU baseline.  Never linked into the product.
 */
#include "layer.h"   /* /root/reference/lib */
#include "sample.h"  /* /root/reference/MemN2N */

/* the two globals every program using layer.h must define (lib/layer.h:8-9) */
bool en_gpu_model = false;
bool en_cpu = true;

static FILE *devnull(void)
{
    static FILE *f = NULL;
    if (!f) f = fopen("/dev/null", "w");
    return f;
}

/* ---- quantiser macros, lib/common.h:178-227 ---- */
float ref_float_quant(float x, unsigned iwl, unsigned frac) { return FLOAT_QUANT(x, iwl, frac); }
int ref_float2fixed(float x, unsigned iwl, unsigned frac) { return (int)(FLOAT2FIXED(x, iwl, frac)); }
float ref_fixed2float(int w, unsigned iwl, unsigned frac) { return FIXED2FLOAT(w, iwl, frac); }
float ref_fixed_mul(float a, float b, unsigned iwl, unsigned frac)
{
    float out;
    FIXED_MUL(out, a, b, iwl, frac);
    return out;
}
float ref_fixed_add(float a, float b, unsigned iwl, unsigned frac)
{
    float out;
    FIXED_ADD(out, a, b, iwl, frac);
    return out;
}

/* ---- softmax_fwd CPU branch, lib/layer.c:1184-1258 ---- */
void ref_softmax_fwd(const float *in, float *out, unsigned dim, int f_exp_plan, int f_shift_based)
{
    softmax sf;
    softmax_constructor(&sf, dim, f_exp_plan != 0, f_shift_based != 0, devnull());
    softmax_in(&sf, dim, (float *)in, NULL, NULL, NULL);
    softmax_fwd(&sf, false);
    memcpy(out, sf.out_vec, dim * sizeof(float));
    free(sf.out_vec);
    free(sf.grad_out);
}

/* ---- sum_vec_fwd CPU branch, lib/layer.c:1502-1511 ---- */
void ref_sum_vec_fwd(const float *a, const float *b, float *out, unsigned dim, int f_fixed, unsigned iwl,
                     unsigned frac)
{
    sum_vec sv;
    sum_vec_constructor(&sv, dim, f_fixed != 0, iwl, frac, 3, devnull());
    sum_vec_in(&sv, (float *)a, (float *)b, NULL, NULL, NULL, NULL);
    sum_vec_fwd(&sv, false);
    memcpy(out, sv.out_vec, dim * sizeof(float));
}

/* ---- dense_mat_fwd CPU branch, lib/layer.c:2671-2696 ---- */
void ref_dense_mat_fwd(const float *w, const float *in_mat, float *out_mat, unsigned dim_in, unsigned dim_out,
                       unsigned dim_len, int f_fixed, unsigned iwl, unsigned frac)
{
    dense_mat ds;
    unsigned i;
    float **rows = (float **)malloc(dim_len * sizeof(float *));
    dense_mat_constructor(&ds, dim_len, dim_in, dim_out, false, 0.0f, f_fixed != 0, iwl, frac, 3, devnull());
    memcpy(ds.w_mat[0], w, (size_t)dim_in * dim_out * sizeof(float));
    for (i = 0; i < dim_len; i++) rows[i] = (float *)in_mat + (size_t)i * dim_in;
    dense_mat_in(&ds, dim_len, rows, NULL, NULL, NULL);
    dense_mat_fwd(&ds, false);
    for (i = 0; i < dim_len; i++) memcpy(out_mat + (size_t)i * dim_out, ds.out_mat[i], dim_out * sizeof(float));
    free(rows);
}

/* ---- cross_entropy_run CPU branch, lib/layer.c:3190-3208 ---- */
float ref_cross_entropy_run(const float *h, const float *y, float *grad_out, unsigned dim)
{
    cross_entropy ce;
    cross_entropy_constructor(&ce, dim, devnull());
    cross_entropy_in(&ce, (float *)h, (float *)y, NULL, NULL);
    ce.cost = 0.0f;
    cross_entropy_run(&ce, 3);
    memcpy(grad_out, ce.grad_out, dim * sizeof(float));
    return ce.cost;
}

/* ---- activation_fwd CPU branch (float only), lib/layer.c:4226-4244 ---- */
void ref_activation_fwd(const float *in, float *out, unsigned dim, const char *type_act)
{
    activation act;
    activation_constructor(&act, dim, (char *)type_act, false, 0, 0, 3, devnull());
    activation_in(&act, (float *)in, NULL, NULL, NULL);
    activation_fwd(&act, false);
    memcpy(out, act.out, dim * sizeof(float));
}

/* ---- dataset -> bag-of-words, MemN2N/sample.c + the set-up MemN2N/MemN2N.c:535-620 ---- */
static sample *g_test = NULL;
static unsigned g_n_test = 0, g_dim_input = 0;

static int babi_load(const char *train_path, const char *test_path, unsigned max_sen_len, unsigned n_train_cap,
                     unsigned n_test_cap, unsigned *dim_input, unsigned *dim_dict, unsigned *max_line_out, int en_pe, unsigned *dim_word_out);

int ref_babi_load(const char *train_path, const char *test_path, unsigned max_sen_len, unsigned n_train_cap,
                  unsigned n_test_cap, unsigned *dim_input, unsigned *dim_dict, unsigned *max_line_out)
{
    return babi_load(train_path, test_path, max_sen_len, n_train_cap, n_test_cap, dim_input, dim_dict, max_line_out, 0, NULL);
}

/* the same with EN_PE (define.h:298): sample_vectorization then SETS a question word's bag-of-words entry to the position
 * weight pe_w[word][position] (sample.c:559-560); pe_w as the program builds it (MemN2N.c:606-616) */
int ref_babi_load_pe(const char *train_path, const char *test_path, unsigned max_sen_len, unsigned n_train_cap,
                     unsigned n_test_cap, unsigned *dim_input, unsigned *dim_dict, unsigned *max_line_out, unsigned *dim_word_out)
{
    return babi_load(train_path, test_path, max_sen_len, n_train_cap, n_test_cap, dim_input, dim_dict, max_line_out, 1, dim_word_out);
}

static int babi_load(const char *train_path, const char *test_path, unsigned max_sen_len, unsigned n_train_cap,
                     unsigned n_test_cap, unsigned *dim_input, unsigned *dim_dict, unsigned *max_line_out, int en_pe, unsigned *dim_word_out)
{
    static dictionary dict;
    unsigned n_train = 0, i, j, max_line = 0, max_word = 0;
    unsigned *idx;
    sample *train = sample_constructor((char *)train_path, max_sen_len, &n_train, n_train_cap);
    if (!train) return -1;
    dictionary_constructor(&dict, train, n_train);
    for (i = 0; i < n_train; i++) {
        if (train[i].n_sen > max_line) max_line = train[i].n_sen;
        for (j = 0; j < train[i].n_sen; j++)
            if (train[i].sentences[j].n > max_word) max_word = train[i].sentences[j].n;
    }
    g_dim_input = dict.n + max_line;           /* EN_TIME true: MemN2N.c:574-578 */
    g_test = sample_constructor((char *)test_path, max_line, &g_n_test, n_test_cap);
    if (!g_test) return -2;
    for (i = 0; i < g_n_test; i++) {
        g_test[i].dim_input = g_dim_input;
        g_test[i].dim_word = max_word + 1;
        g_test[i].dim_dict = dict.n;
    }
    sample_init(g_test, g_n_test, 0, true);
    idx = (unsigned *)malloc(g_n_test * sizeof(unsigned));
    for (i = 0; i < g_n_test; i++) idx[i] = i;
    if (en_pe) {
        const unsigned dim_word = max_word + 1;
        float **pe_w = (float **)malloc(g_dim_input * sizeof(float *));
        pe_w[0] = (float *)malloc((size_t)g_dim_input * dim_word * sizeof(float));
        for (i = 1; i < g_dim_input; i++) pe_w[i] = pe_w[i - 1] + dim_word;
        for (i = 0; i < g_dim_input; i++)
            for (j = 0; j < dim_word; j++)
                pe_w[i][j] = 1.0 + 4.0 * ((float)i / (float)g_dim_input - 0.5) * ((float)j / (float)dim_word - 0.5);   /* MemN2N.c:615 */
        sample_vectorization(g_test, &dict, idx, g_n_test, 0, true, 0, true, pe_w, 0.0f);
        if (dim_word_out) *dim_word_out = dim_word;
        free(pe_w[0]); free(pe_w);
    } else {
        sample_vectorization(g_test, &dict, idx, g_n_test, 0, true, 0, false, NULL, 0.0f);
    }
    free(idx);
    *dim_input = g_dim_input;
    *dim_dict = dict.n;
    *max_line_out = max_line;
    return (int)g_n_test;
}

unsigned ref_babi_nsen(unsigned i) { return g_test[i].n_sen; }

void ref_babi_get(unsigned i, float *story, float *question, float *answer)
{
    unsigned s;
    for (s = 0; s < g_test[i].n_sen; s++)
        memcpy(story + (size_t)s * g_dim_input, g_test[i].sentences_b[s], g_dim_input * sizeof(float));
    memcpy(question, g_test[i].question_b, g_dim_input * sizeof(float));
    memcpy(answer, g_test[i].answer_b, g_dim_input * sizeof(float));
}
