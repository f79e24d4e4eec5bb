// tuning.hip -- the process-wide switches of rt.h's QmTuning: read from the environment once, re-read on request.
#include "rt.h"
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include "../../include/qmann_batch.h"

namespace {

QmTuning g_tuning;

void read_env(QmTuning &t)
{
    auto on = [](const char *n) { return getenv(n) != nullptr; };
    auto tri = [](const char *n) { const char *e = getenv(n); return e ? (e[0] == '1' ? 1 : 0) : -1; };
    t.no_w7 = on("QMANN_NO_W7");
    t.no_mid = on("QMANN_NO_MID");
    t.no_lean = on("QMANN_NO_LEAN");
    t.no_quad = on("QMANN_NO_QUAD");
    t.no_corun = on("QMANN_NO_CORUN");
    t.no_quad_long = on("QMANN_NO_QUAD_LONG");
    t.answer_exact = on("QMANN_ANSWER_EXACT");
    t.no_tied = on("QMANN_NO_TIED");
    t.embed_general_epilogue = on("QMANN_EMBED_GENERAL_EPILOGUE");
    t.embed_valu = on("QMANN_EMBED_VALU");
    t.embed_per_hop = on("QMANN_EMBED_PER_HOP");
    t.answer_two_pass = on("QMANN_ANSWER_TWO_PASS");
    t.lean_sparse = tri("QMANN_LEAN_SPARSE");
    {
        const char *e = getenv("QMANN_QUAD_MIN_QUERIES");
        t.quad_min_queries = (e && *e) ? (uint32_t)strtoul(e, nullptr, 10) : 8192u;
    }
    t.no_tight = on("QMANN_NO_TIGHT");              // presence-only, like its siblings
}

}  // namespace

const QmTuning &qm_tuning()
{
    static const bool once = (read_env(g_tuning), true);      // thread-safe first use
    (void)once;
    return g_tuning;
}

unsigned qm_cu_count()
{
    // per device ordinal: a one-thread-per-GPU host (examples/forward_sharded.c) may drive differently partitioned devices, and
    // a persistent grid sized with another device's count runs late workgroups alone (rt.h::qm_resident_groups)
    constexpr int kMaxDev = 64;
    static std::atomic<unsigned> cache[kMaxDev];                  // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const bool slot = dev >= 0 && dev < kMaxDev;
    if (slot) {
        const unsigned c = cache[dev].load(std::memory_order_relaxed);
        if (c) return c;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    if (slot) cache[dev].store((unsigned)cus, std::memory_order_relaxed);
    return (unsigned)cus;
}

uint32_t *qm_scratch_u32(size_t words, hipStream_t stream)
{
    struct Buf { uint32_t *p = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> hold(mu);
    Buf &b = bufs[{dev, stream}];
    if (words > b.cap) {
        // work enqueued earlier on this stream may still read the old buffer: let it finish before the buffer goes
        if (b.p) { (void)hipStreamSynchronize(stream); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
        const size_t cap = words + words / 4 + 64;
        if (hipMalloc((void **)&b.p, cap * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return nullptr; }
        b.cap = cap;
    }
    return b.p;
}

thread_local QmSplitReady qm_split_ready{nullptr, 0, 0, nullptr};

QmSide *qm_side_stream(hipStream_t stream)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, QmSide> sides;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> hold(mu);
    auto it = sides.find({dev, stream});
    if (it != sides.end()) return &it->second;
    QmSide sd{};
    if (hipStreamCreateWithFlags(&sd.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&sd.last_counts, 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    sd.last_counts[0] = sd.last_counts[1] = 0xFFFFFFFFu;
    return &(sides[{dev, stream}] = sd);
}

extern "C" void qmann_tuning_reload(void)
{
    (void)qm_tuning();
    read_env(g_tuning);
}
