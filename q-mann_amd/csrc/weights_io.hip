// weights_io.hip -- weight files of a model (include/qmann_weights.h): host-only code, no kernels.
// Layout follows the reference's disabled load / write blocks (MemN2N/MemN2N.c:2553-2618, :2853-2978):
// column-major matrices, per-hop matrices back to back, float32 or sign-magnitude int32 words.
#include "qfmt.h"
#include "../../include/qmann_weights.h"

#include <stdio.h>
#include <string>
#include <vector>

namespace {

struct Mat {
    float *w;            // row-major [dim_out][dim_in]
    uint32_t dim_out, dim_in;
    qmann_fmt fmt;
};

std::string path_of(const char *dir, const char *name) { return std::string(dir && *dir ? dir : ".") + "/" + name; }

// one file = the listed matrices back to back, each column-major
int write_file(const std::string &path, const std::vector<Mat> &mats, bool fixed)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return QMANN_EIO;
    std::vector<uint32_t> col;
    for (const Mat &m : mats) {
        col.resize(m.dim_out);
        for (uint32_t j = 0; j < m.dim_in; j++) {
            for (uint32_t i = 0; i < m.dim_out; i++) {
                const float x = m.w[(size_t)i * m.dim_in + j];
                col[i] = fixed ? qm_signmag(x, m.fmt.iwl, m.fmt.frac) : __builtin_bit_cast(uint32_t, x);
            }
            if (fwrite(col.data(), sizeof(uint32_t), m.dim_out, f) != m.dim_out) { fclose(f); return QMANN_EIO; }
        }
    }
    return fclose(f) == 0 ? QMANN_OK : QMANN_EIO;
}

int read_file(const std::string &path, const std::vector<Mat> &mats, bool fixed)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return QMANN_EIO;
    size_t want = 0;
    for (const Mat &m : mats) want += (size_t)m.dim_out * m.dim_in * sizeof(uint32_t);
    if (fseek(f, 0, SEEK_END) != 0 || (size_t)ftell(f) != want) { fclose(f); return QMANN_EIO; }   // never a partial load
    rewind(f);
    std::vector<uint32_t> col;
    for (const Mat &m : mats) {
        col.resize(m.dim_out);
        for (uint32_t j = 0; j < m.dim_in; j++) {
            if (fread(col.data(), sizeof(uint32_t), m.dim_out, f) != m.dim_out) { fclose(f); return QMANN_EIO; }
            for (uint32_t i = 0; i < m.dim_out; i++) {
                float x;
                if (fixed) {
                    const float mag = (float)(col[i] & 0x7FFFFFFFu) / (float)(1u << m.fmt.frac);
                    x = (col[i] & 0x80000000u) ? -mag : mag;        // "minus zero" words decode to -0.0f
                } else {
                    x = __builtin_bit_cast(float, col[i]);
                }
                m.w[(size_t)i * m.dim_in + j] = x;
            }
        }
    }
    fclose(f);
    return QMANN_OK;
}

bool shape_ok(const qmann_weights *w)
{
    if (!w || w->n_hop == 0 || w->n_hop > QMANN_MAX_HOP || w->dim_emb == 0 || w->dim_input == 0) return false;
    if (!w->w_q || !w->w_ans) return false;
    const bool lin = w->w_h[0] != nullptr;
    for (uint32_t h = 0; h < w->n_hop; h++)
        if (!w->w_a[h] || !w->w_c[h] || (w->w_h[h] != nullptr) != lin) return false;
    return true;
}

// the files and the matrices each holds
struct Plan {
    const char *name_float, *name_fixed;
    std::vector<Mat> mats;
};

std::vector<Plan> plan_of(const qmann_weights *w, const qmann_fmt *fmt_w)
{
    const uint32_t D = w->dim_emb, V = w->dim_input;
    const qmann_fmt none{0, 0};
    std::vector<Plan> p(5);
    p[0] = {"w_emb_a_float.bin", "w_emb_a_fixed.bin", {}};
    p[1] = {"w_emb_c_float.bin", "w_emb_c_fixed.bin", {}};
    p[2] = {"w_emb_q_float.bin", "w_emb_q_fixed.bin", {Mat{w->w_q, D, V, fmt_w ? fmt_w[0] : none}}};
    p[3] = {"w_float.bin", nullptr, {Mat{w->w_ans, V, D, none}}};
    p[4] = {"w_lin_map_float.bin", "w_lin_map_fixed.bin", {}};
    for (uint32_t h = 0; h < w->n_hop; h++) {
        const qmann_fmt f = fmt_w ? fmt_w[h] : none;
        p[0].mats.push_back(Mat{w->w_a[h], D, V, f});
        p[1].mats.push_back(Mat{w->w_c[h], D, V, f});
        if (w->w_h[h]) p[4].mats.push_back(Mat{w->w_h[h], D, D, f});
    }
    if (p[4].mats.empty()) p.pop_back();
    return p;
}

bool fmts_ok(const qmann_weights *w, const qmann_fmt *fmt_w)
{
    for (uint32_t h = 0; h < w->n_hop; h++)
        if (fmt_w[h].iwl + fmt_w[h].frac < 1 || fmt_w[h].iwl + fmt_w[h].frac > 31) return false;
    return true;
}

}  // namespace

extern "C" {

int qmann_weights_save(const char *dir, const qmann_weights *w, const qmann_fmt *fmt_w)
{
    if (!shape_ok(w) || (fmt_w && !fmts_ok(w, fmt_w))) return QMANN_EINVAL;
    for (const Plan &p : plan_of(w, fmt_w)) {
        int rc = write_file(path_of(dir, p.name_float), p.mats, false);
        if (rc == QMANN_OK && fmt_w && p.name_fixed) rc = write_file(path_of(dir, p.name_fixed), p.mats, true);
        if (rc != QMANN_OK) return rc;
    }
    return QMANN_OK;
}

int qmann_weights_load(const char *dir, qmann_weights *w, int from_fixed, const qmann_fmt *fmt_w)
{
    if (!shape_ok(w) || (from_fixed && (!fmt_w || !fmts_ok(w, fmt_w)))) return QMANN_EINVAL;
    for (const Plan &p : plan_of(w, fmt_w)) {
        const bool fixed = from_fixed && p.name_fixed;
        const int rc = read_file(path_of(dir, fixed ? p.name_fixed : p.name_float), p.mats, fixed);
        if (rc != QMANN_OK) return rc;
    }
    return QMANN_OK;
}

}  // extern "C"
