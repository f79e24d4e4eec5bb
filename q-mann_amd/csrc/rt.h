// rt.h -- tiny runtime layer: error convention, allocation, launch helpers.
// Error convention of the boundary (lib/layer_cuda.h:13-22 in the reference):
// message on stderr, then exit(code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// The batched entry points (qmann_*: they return QMANN_E* codes) do not end the host process on a HIP failure: inside a
// QmBatched scope the failure is reported on stderr, remembered, and the entry point returns QMANN_EHIP.  The drop-in
// cuda_* verbs keep the reference's convention (exit), they have no way to return an error.
inline thread_local int qm_batched_depth = 0;
inline thread_local int qm_batched_err = 0;
struct QmBatched {
    // The outermost scope starts clean: hipGetLastError() reads AND clears the thread's last error, so a failure an earlier
    // call reported (or one the host application left behind) is not reported again by this call's launch checks.
    QmBatched() { if (qm_batched_depth++ == 0) { qm_batched_err = 0; (void)hipGetLastError(); } }
    ~QmBatched() { --qm_batched_depth; }
    QmBatched(const QmBatched &) = delete;
    int rc(int ok = 0) const { return qm_batched_err ? -5 /* QMANN_EHIP */ : ok; }
};

#define QM_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "[*E] HIP : %s : %s (%s:%d)\n", __func__,                 \
                    hipGetErrorString(e_), __FILE__, __LINE__);                       \
            if (qm_batched_depth > 0) qm_batched_err = (int)e_;                       \
            else exit((int)e_ ? (int)e_ : 1);                                         \
        }                                                                             \
    } while (0)

// (hipGetLastError, not hipPeekAtLastError: the error is consumed when it is reported, a later valid call is not blamed for it)
#define QM_LAUNCH_CHECK() QM_HIP(hipGetLastError())

static inline void qm_fail(const char *fn, const char *msg)
{
    fprintf(stderr, "[*E] qmann : %s : %s\n", fn, msg);
    exit(1);
}

template <typename T>
static inline void qm_alloc(T **p, size_t n)
{
    // zero-sized requests still hand back a valid pointer, like cudaMalloc(0) would not:
    // the reference never frees a null, so keep every slot non-null
    QM_HIP(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
}

// Calls inside a QmAnswerExact scope take the serial-order float answer kernels (bit-equal to lib/layer_cuda.cu:70-80's loop):
// the drop-in queue (abi_defer.hip), whose results are promised to be those of the verb-by-verb loop.
inline thread_local int qm_answer_exact_depth = 0;
struct QmAnswerExact {
    QmAnswerExact() { ++qm_answer_exact_depth; }
    ~QmAnswerExact() { --qm_answer_exact_depth; }
    QmAnswerExact(const QmAnswerExact &) = delete;
};

// A hop-plane's size in rows, from a caller that knows it, for calls that pass hop_stride = 0 (tied hops: one plane for every
// hop, qmann_model's hops_and_answer): hops_quad.h addresses rows by 32-bit byte offsets from the plane's start and must know that
// the plane stays below 2 GiB; without the hint such a call keeps the one-wavefront-per-query kernel.
inline thread_local size_t qm_rows_hint = 0;
struct QmRowsHint {
    size_t prev;
    explicit QmRowsHint(size_t rows) : prev(qm_rows_hint) { qm_rows_hint = rows; }
    ~QmRowsHint() { qm_rows_hint = prev; }
    QmRowsHint(const QmRowsHint &) = delete;
};

// The library's A/B and tuning switches (INTEGRATION.md lists them) are read from the environment ONCE per process -- on
// first use, through a thread-safe function-local static -- and then live in this struct: no launch calls getenv(), which
// is not safe against a setenv() running in another thread of a threaded host (examples/forward_sharded.c is one).
// qmann_tuning_reload() (include/qmann_batch.h) reads the environment again, for hosts and tests that change a switch
// after the first launch; it must not run while another thread launches.
struct QmTuning {
    bool no_w7, no_mid, no_lean, no_tied, embed_general_epilogue, embed_valu, answer_two_pass;
    bool embed_per_hop;                               // QMANN_EMBED_PER_HOP: the joint-dictionary story embedding keeps one workgroup per hop (k_embed_story_mfma<4, 16>)
    bool answer_exact;                                // QMANN_ANSWER_EXACT: the float answer layer keeps the reference's serial order of additions (no bf16 MFMA form)
    bool no_quad_long;                                // QMANN_NO_QUAD_LONG: stories of 17 .. 64 rows keep the one-wavefront-per-query kernel (hops_quad.h's four-chunk form off)
    bool no_corun;                                    // QMANN_NO_CORUN: no second stream (the two hop kernels of a split batch, the question embedding beside the story embedding: in sequence)
    bool no_quad;                                     // QMANN_NO_QUAD: short stories keep the one-wavefront-per-query kernel (hops_quad.h off)
    bool no_tight;                                    // QMANN_NO_TIGHT (set, any value): the lean kernels keep their four-wave (128-register) builds
    int lean_sparse;                                  // -1 = the launcher chooses, 0 / 1 forced
    uint32_t quad_min_queries;                        // QMANN_QUAD_MIN_QUERIES (default 8192): batches of at most this many stories keep one story per wavefront (hops_lean.h)
};
const QmTuning &qm_tuning();                          // (tuning.hip)

// compute units of the CURRENT device (256 on an MI355X in SPX mode; fewer in a partitioned mode), asked once per device ordinal
unsigned qm_cu_count();                               // (tuning.hip)

// Persistent kernels: how many workgroups of `waves` wavefronts are resident AT ONCE on the device, for a kernel compiled for
// `waves_per_simd` wavefronts per SIMD (its __launch_bounds__) that takes `lds` bytes of LDS per workgroup.  A persistent grid
// must not exceed this: a workgroup that has to wait for a slot runs after the others, alone on a mostly idle chip -- two extra
// workgroups on a one-per-CU kernel double its run time (round 4 found three launchers sized by LDS alone).
static inline unsigned qm_resident_groups(unsigned waves, unsigned waves_per_simd, size_t lds)
{
    const unsigned by_regs = waves_per_simd * 4u / waves, by_lds = (unsigned)(160u * 1024u / (lds + 256u)), by_hw = 32u / waves;
    unsigned per_cu = by_regs < by_lds ? by_regs : by_lds;
    per_cu = per_cu < by_hw ? per_cu : by_hw;
    return qm_cu_count() * (per_cu ? per_cu : 1u);
}

// Device scratch of at least `words` 32-bit words for the launch being enqueued on `stream` (index lists of a batch split by
// story length): one buffer per (device, stream), grown when a launch needs more (the only time this allocates -- like a model's
// workspace it reaches its size on the first batches), kept for the life of the process.  nullptr when the allocation fails.
uint32_t *qm_scratch_u32(size_t words, hipStream_t stream);

// A second stream beside `stream` with the two events that fork work onto it and join it back (cached per (device, stream)).
struct QmSide {
    hipStream_t side;
    hipEvent_t fork, join;
    // pinned host words: the two list lengths (short, long) of the LAST batch split by length on this stream, stored by its
    // short-story kernel (hops_quad.h: QuadArgs::publish) -- no copy, no synchronisation; 0xFFFFFFFF until one has landed.  A hint for the next launch's choice between
    // "side by side" and "in sequence" only -- never a result.
    volatile uint32_t *last_counts;
};
// A batch's two index lists (stories of <= 16 rows / longer ones: hops_quad.h::k_split_by_length) prepared AHEAD of the hop launch:
// the host model computes them on the second stream while the stories are embedded and names them here for the next hop
// launch of this thread, which takes them if (row_off, n_query, max_slots) are the ones it was asked for.
struct QmSplitReady { const uint32_t *row_off; uint32_t n_query, max_slots; uint32_t *ws; };
extern thread_local QmSplitReady qm_split_ready;
bool qm_split_applies(size_t rows_total, uint32_t n_query, uint32_t max_slots);          // (batch_hops.hip)
uint32_t *qm_split_early(const uint32_t *row_off, uint32_t n_query, uint32_t max_slots, hipStream_t owner, hipStream_t run_on);
constexpr uint32_t kQmCorunMinQueries = 32768;        // batches below run their kernels in sequence (measured equal at 32 768 and 65 536 queries, +4 % at 262 000)
QmSide *qm_side_stream(hipStream_t stream);       // (tuning.hip)

static inline unsigned qm_cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }
