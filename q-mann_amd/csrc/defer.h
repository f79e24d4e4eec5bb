// defer.h -- deferred execution behind boundary B (the drop-in cuda_* verbs).
//
// The reference's host drives the test phase one query at a time: 31 layer verbs per query (MemN2N/MemN2N.c:2626-2697),
// no device-to-host traffic in between -- the accumulators are fetched once after the loop (:2701-2702,
// lib/layer_cuda.cu:3813-3851).  Launching those verbs one by one is what makes the reference, and a verb-by-verb
// replacement, launch bound (31 000 launches for 1 000 queries).  SURVEY.md 8(b) "Threading": a replacement may batch
// internally as long as results are visible in order through the raw pointers.
//
// So the nine FORWARD verbs do not launch: they append an Op to a queue.  Every other verb is a synchronisation point and
// drains the queue first (abi_defer.hip): the queued ops are parsed into whole queries by following the pointer wiring
// (this op's input is that op's output), runs of queries that share weights, formats and accumulators become ONE call of
// the batched model forward (qmann_model_forward_bow on the very device pools the host filled), and whatever does not fit
// the pattern -- training steps, partial sequences, unsupported options -- is executed op by op exactly as before.  After a
// batched run the last query is replayed op by op (accumulators excepted) so that every layer buffer holds what the serial
// loop would have left in it.
#pragma once
#include "qfmt.h"

#include <stddef.h>
#include <stdint.h>

namespace qmdefer {

enum OpKind : uint8_t { kDense, kDenseMat, kDot, kDotAppx, kSoftmax, kSumVec, kScale, kAct, kCrossEntropy };

// One forward verb with its arguments as passed (device pointers are not dereferenced before the drain).
struct Op {
    uint8_t kind;
    bool fixed, trans, shift;
    int act;                              // kDense / kAct: 0 NULL, 1 SIGMOID, 2 RELU
    unsigned r, c, k, mode;               // kDense: r = dim_out, c = dim_in | kDenseMat: r = dim_len, c = dim_in, k = dim_out
                                          // kDot / kDotAppx: r, c (k = num_bit_attention) | kSoftmax, kSumVec, kScale, kAct, kCrossEntropy: r = dim
    QFmt fa, fb;                          // kDense: fa = input, fb = weights | kDenseMat, kSumVec, kAct: fa | kDot: fa = matrix, fb = vector
    const float *w, *in, *in2;            // kDense, kDenseMat: w, in | kDot*: in = matrix, in2 = vector | kSumVec: in, in2 | kScale: in, w
                                          // kCrossEntropy: in = h, in2 = y
    float *out, *aux;                     // kSoftmax: aux = dev_max | kCrossEntropy: out = dev_grad_out
    float *cost[3];                       // kCrossEntropy: train / valid / test accumulators
    unsigned *cnt[3], *pred;
};

enum Mode { kOff = 0, kOn = 1, kVerify = 2 };

// appends (returns true) or, when deferral is off, returns false and the caller launches at once
bool submit(const Op &op);
// drain the queue.  writes = the calling verb may change weights or inputs afterwards: the cached model is dropped
void sync_point(bool writes);
// the immediate implementation of an op (abi_ops.hip)
void run_now(const Op &op);
int softmax_base();                        // abi_ops.hip: base selected by qmann_abi_set_softmax_base

}  // namespace qmdefer

#define QM_SYNC_WRITES() qmdefer::sync_point(true)
#define QM_SYNC_READS() qmdefer::sync_point(false)
