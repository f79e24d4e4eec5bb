// tuning.hip -- the process-wide switches of rt.h's QmTuning: read from the environment once, re-read on request.
#include "rt.h"
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
    t.no_tied = on("QMANN_NO_TIED");
    t.embed_general_epilogue = on("QMANN_EMBED_GENERAL_EPILOGUE");
    t.embed_valu = on("QMANN_EMBED_VALU");
    t.answer_two_pass = on("QMANN_ANSWER_TWO_PASS");
    t.lean_sparse = tri("QMANN_LEAN_SPARSE");
    t.no_tight = tri("QMANN_NO_TIGHT") == 1;
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
    static const unsigned n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        return (unsigned)cus;
    }();
    return n;
}

extern "C" void qmann_tuning_reload(void)
{
    (void)qm_tuning();
    read_env(g_tuning);
}
