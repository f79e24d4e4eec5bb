/*
 * qmann_dataset.h -- the reference's parsed bAbI files straight into the word-index wire format.
 *
 * MemN2N/sample.c reads a record file (blank line, "+NS+", count; then per record "+I+" id "+S+" n, n sentences,
 * "+Q+" question, "+A+" answer: sample.c:118-234), builds a dictionary over the TRAINING samples and expands every
 * sentence to a float bag-of-words row of dim_input entries.  This loader keeps the first half and drops the second:
 * the same dictionary, the same clipping rules, and the sentences as uint16 word lists -- the input of
 * qmann_embed_story_idx / qmann_model_forward_words (16 bytes per sentence instead of 4 . dim_input).
 *
 * What it reproduces, with the lines it follows:
 *   dictionary    index 0 = "NULL" (define.h:232), then the words of the training samples in order of first appearance
 *                 -- sentences, question, answer of every sample --, compared without case (sample.c:852-940);
 *   max_line      the longest training story after cutting to max_sen_len sentences (the LAST ones are kept,
 *                 sample.c:152-166); the test file is then read with max_line as its cut (MemN2N.c:585);
 *   dim_word      longest training sentence + 1 (the time entry); a sentence keeps its first dim_word - 1 words
 *                 (sample.c:340-346), a question / answer likewise (:365-399);
 *   dim_input     dictionary size + max_line (EN_TIME, MemN2N.c:574-578);
 *   time entry    sentence j of n gets dim_dict + n - 1 - j, the most recent sentence index dim_dict (sample.c:474);
 *   answer        the index of the answer word (the rows are one-hot in every bAbI task).
 * One deliberate difference: a test word that is not in the training dictionary is dropped (no label; in a question it
 * leaves a 0xFFFF hole in ITS slot so that the other words keep their positions, which EN_PE weighs: sample.c:559); the
 * reference's word_idx returns -1 there and the row write goes out of bounds (sample.c:838-848, :544).  A count line that is
 * not a decimal number is a format error (QMANN_EIO), not a count of 0.
 * Host-only code (no GPU call); pinned against the fixtures the reference's own sample.c produced (tests/test_dataset_io.py).
 */
#ifndef QMANN_DATASET_H
#define QMANN_DATASET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmann_dataset {
    uint32_t n_query;          /* test samples read */
    uint32_t rows_total;       /* sentences of all of them */
    uint32_t max_words;        /* pitch of story_words: dim_word rounded up to a multiple of 4 slots (<= 16) */
    uint32_t max_q_words;      /* pitch of question_words, likewise */
    uint32_t dim_dict, dim_input, max_line, dim_word;
    uint32_t *row_off;         /* [n_query + 1] first sentence of every story */
    uint16_t *story_words;     /* [rows_total][max_words]: the words in sentence order, then the time index; 0xFFFF unused */
    uint16_t *question_words;  /* [n_query][max_q_words]: slot k = word k of the question (EN_PE needs the positions), 0xFFFF = unknown word / unused */
    uint32_t *answer;          /* [n_query], 0xFFFFFFFF when the answer word is not in the dictionary */
} qmann_dataset;

/* train_path: the file the dictionary, max_line and dim_word come from; test_path: the stories to return.
 * max_sen_len: MAX_SEN_LEN of the configuration (50 single task, 64 joint); n_*_cap: read at most that many samples
 * (0 = all).  Returns QMANN_OK, QMANN_EIO (file / record format) or QMANN_ERANGE (a sentence needs more than 16 slots,
 * dim_input >= 65535).  The arrays are malloc'ed; release them with qmann_dataset_free. */
int qmann_dataset_load(const char *train_path, const char *test_path, uint32_t max_sen_len, uint32_t n_train_cap,
                       uint32_t n_test_cap, qmann_dataset *out);
void qmann_dataset_free(qmann_dataset *ds);

#ifdef __cplusplus
}
#endif
#endif
