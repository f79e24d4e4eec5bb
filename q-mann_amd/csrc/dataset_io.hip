// dataset_io.hip -- include/qmann_dataset.h: host-side C++ (no kernels, no HIP calls); see the header for the rules
// of MemN2N/sample.c it restates.
#include "../../include/qmann_batch.h"
#include "../../include/qmann_dataset.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Record {
    std::vector<std::vector<std::string>> sentences;
    std::vector<std::string> question, answer;
};

std::vector<std::string> split_words(const std::string &line)              // strtok(" "): runs of blanks separate
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && line[i] == ' ') i++;
        size_t j = i;
        while (j < line.size() && line[j] != ' ') j++;
        if (j > i) out.emplace_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

bool next_line(FILE *f, std::string &line)
{
    line.clear();
    int c;
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        if (c != '\r') line.push_back((char)c);
    }
    return any;
}

// a count line: decimal digits (blanks around them allowed), nothing else -- "abc" or "" is a format error, not a 0
bool parse_count(const std::string &line, uint32_t &out)
{
    size_t i = 0, n = line.size();
    while (i < n && line[i] == ' ') i++;
    while (n > i && line[n - 1] == ' ') n--;
    if (i == n || n - i > 9) return false;
    uint32_t v = 0;
    for (; i < n; i++) {
        if (line[i] < '0' || line[i] > '9') return false;
        v = v * 10u + (uint32_t)(line[i] - '0');
    }
    out = v;
    return true;
}

// record file -> samples; stories longer than max_len keep their LAST max_len sentences (sample.c:152-166)
int read_records(const char *path, uint32_t max_len, uint32_t cap, std::vector<Record> &out)
{
    FILE *f = fopen(path, "r");
    if (!f) return QMANN_EIO;
    std::string line;
    uint32_t declared = 0;
    bool have_count = false;
    while (next_line(f, line)) {                                            // header: blank line, "+NS+", count
        if (line == "+NS+") {
            if (!next_line(f, line)) break;
            if (!parse_count(line, declared)) { fclose(f); return QMANN_EIO; }
            have_count = true;
            break;
        }
    }
    if (!have_count) { fclose(f); return QMANN_EIO; }
    const uint32_t limit = cap && cap < declared ? cap : declared;
    int rc = QMANN_OK;
    while (out.size() < limit && next_line(f, line)) {
        if (line != "+I+") continue;                                        // (blank separators)
        Record r;
        if (!next_line(f, line)) { rc = QMANN_EIO; break; }                 // the id: records are numbered by position (sample.c:145)
        if (!next_line(f, line) || line != "+S+") { rc = QMANN_EIO; break; }
        if (!next_line(f, line)) { rc = QMANN_EIO; break; }
        uint32_t n_ori = 0;
        if (!parse_count(line, n_ori)) { rc = QMANN_EIO; break; }
        const uint32_t skip = n_ori > max_len ? n_ori - max_len : 0;
        for (uint32_t i = 0; i < n_ori && rc == QMANN_OK; i++) {
            if (!next_line(f, line)) { rc = QMANN_EIO; break; }
            if (i >= skip) r.sentences.emplace_back(split_words(line));
        }
        if (rc != QMANN_OK) break;
        if (!next_line(f, line) || line != "+Q+" || !next_line(f, line)) { rc = QMANN_EIO; break; }
        r.question = split_words(line);
        if (!next_line(f, line) || line != "+A+" || !next_line(f, line)) { rc = QMANN_EIO; break; }
        r.answer = split_words(line);
        out.emplace_back(std::move(r));
    }
    fclose(f);
    return rc;
}

std::string lower(const std::string &s)
{
    std::string t(s);
    for (char &c : t) c = (char)tolower((unsigned char)c);
    return t;
}

}  // namespace

extern "C" {

int qmann_dataset_load(const char *train_path, const char *test_path, uint32_t max_sen_len, uint32_t n_train_cap,
                       uint32_t n_test_cap, qmann_dataset *out)
{
    if (!train_path || !test_path || !out || max_sen_len == 0) return QMANN_EINVAL;
    memset(out, 0, sizeof *out);
    std::vector<Record> train, test;
    int rc = read_records(train_path, max_sen_len, n_train_cap, train);
    if (rc != QMANN_OK) return rc;

    // dictionary: "NULL" first, then first appearances over the training samples (sample.c:852-940), without case
    std::unordered_map<std::string, uint32_t> dict;
    dict.emplace(lower("NULL"), 0u);
    auto learn = [&](const std::vector<std::string> &ws) {
        for (const std::string &w : ws) dict.emplace(lower(w), (uint32_t)dict.size());
    };
    uint32_t max_line = 0, max_word = 0;
    for (const Record &r : train) {
        for (const auto &s : r.sentences) { learn(s); max_word = std::max<uint32_t>(max_word, (uint32_t)s.size()); }
        learn(r.question);
        learn(r.answer);
        max_line = std::max<uint32_t>(max_line, (uint32_t)r.sentences.size());
    }
    const uint32_t dim_dict = (uint32_t)dict.size(), dim_word = max_word + 1, dim_input = dim_dict + max_line;
    if (dim_input >= 0xFFFFu) return QMANN_ERANGE;
    const uint32_t pitch = (dim_word + 3u) & ~3u;
    if (pitch > 16u) return QMANN_ERANGE;                                   // kMaxWords of the embedding kernels

    rc = read_records(test_path, max_line ? max_line : 1u, n_test_cap, test);   // the test set is cut to max_line (MemN2N.c:585)
    if (rc != QMANN_OK) return rc;
    size_t rows = 0;
    for (const Record &r : test) rows += r.sentences.size();
    if (rows >= 0xFFFFFFFFull) return QMANN_ERANGE;

    out->n_query = (uint32_t)test.size(); out->rows_total = (uint32_t)rows;
    out->max_words = pitch; out->max_q_words = pitch;
    out->dim_dict = dim_dict; out->dim_input = dim_input; out->max_line = max_line; out->dim_word = dim_word;
    out->row_off = (uint32_t *)malloc((test.size() + 1) * sizeof(uint32_t));
    out->story_words = (uint16_t *)malloc((rows ? rows : 1) * pitch * sizeof(uint16_t));
    out->question_words = (uint16_t *)malloc((test.size() ? test.size() : 1) * pitch * sizeof(uint16_t));
    out->answer = (uint32_t *)malloc((test.size() ? test.size() : 1) * sizeof(uint32_t));
    if (!out->row_off || !out->story_words || !out->question_words || !out->answer) { qmann_dataset_free(out); return QMANN_ERANGE; }
    memset(out->story_words, 0xFF, (rows ? rows : 1) * pitch * sizeof(uint16_t));
    memset(out->question_words, 0xFF, (test.size() ? test.size() : 1) * pitch * sizeof(uint16_t));

    auto index_of = [&](const std::string &w) -> uint32_t {
        const auto it = dict.find(lower(w));
        return it == dict.end() ? 0xFFFFu : it->second;                      // (the reference writes row[-1] here)
    };
    size_t row = 0;
    for (size_t q = 0; q < test.size(); q++) {
        const Record &r = test[q];
        out->row_off[q] = (uint32_t)row;
        const uint32_t n = (uint32_t)r.sentences.size();
        for (uint32_t j = 0; j < n; j++, row++) {
            uint16_t *dst = out->story_words + row * pitch;
            const uint32_t keep = std::min<uint32_t>((uint32_t)r.sentences[j].size(), dim_word - 1);   // sample.c:340-346
            uint32_t k = 0, slot = 0;
            for (; k < keep; k++) {
                const uint32_t w = index_of(r.sentences[j][k]);
                if (w != 0xFFFFu) dst[slot++] = (uint16_t)w;
            }
            dst[slot] = (uint16_t)(dim_dict + n - 1 - j);                    // time entry, most recent sentence first (sample.c:474)
        }
        uint16_t *qd = out->question_words + q * pitch;
        const uint32_t qkeep = std::min<uint32_t>((uint32_t)r.question.size(), dim_word - 1);           // sample.c:365-371
        // a question word keeps its ORIGINAL slot (an unknown word leaves a 0xFFFF hole, which the kernels skip): with
        // EN_PE the weight of a word is pe_w[word][k] with k its position in the question (sample.c:559)
        for (uint32_t k = 0; k < qkeep; k++) qd[k] = (uint16_t)index_of(r.question[k]);
        out->answer[q] = 0xFFFFFFFFu;
        if (!r.answer.empty()) {
            const uint32_t w = index_of(r.answer[0]);
            if (w != 0xFFFFu) out->answer[q] = w;
        }
    }
    out->row_off[test.size()] = (uint32_t)row;
    return QMANN_OK;
}

void qmann_dataset_free(qmann_dataset *ds)
{
    if (!ds) return;
    free(ds->row_off); free(ds->story_words); free(ds->question_words); free(ds->answer);
    memset(ds, 0, sizeof *ds);
}

}  // extern "C"
