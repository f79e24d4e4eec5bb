// dist.hip -- include/qmann_dist.h: shards, the RCCL rendezvous and the one-time broadcast of the quantised parameter
// blob.  Host-side C++, no kernels.  librccl is loaded with dlopen on first use (see the header for why).
#include "rt.h"
#include "../../include/qmann_dist.h"

#include <dlfcn.h>
#include <mutex>
#include <new>
#include <rccl/rccl.h>          // types and prototypes only: no symbol of it is linked
#include <string.h>
#include <vector>

struct qmann_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1, device = 0;
};

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    // QMANN_RCCL_PATH, when set, is the ONLY name tried: an override that cannot be loaded must not end in some other copy
    // of the library being picked up silently (a process may hold one RCCL only, see the header)
    const char *forced = getenv("QMANN_RCCL_PATH");
    const char *names[4] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const int n_names = (forced && *forced) ? 1 : 4;
    for (int i = 0; i < n_names; i++) {
        const char *n = names[i];
        if (!n || !*n) continue;
        // RTLD_GLOBAL is not wanted: nothing else should bind to it through us.  A process that already holds an RCCL
        // (PyTorch's, say) gets that same copy back when the soname matches.
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        fprintf(stderr, "[*E] qmann : RCCL : cannot load librccl (%s); set QMANN_RCCL_PATH\n", dlerror());
        return;
    }
#define QM_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.handle, name)
    QM_SYM(GetUniqueId, "ncclGetUniqueId"); QM_SYM(CommInitRank, "ncclCommInitRank"); QM_SYM(CommDestroy, "ncclCommDestroy");
    QM_SYM(Broadcast, "ncclBroadcast"); QM_SYM(AllGather, "ncclAllGather"); QM_SYM(GetErrorString, "ncclGetErrorString");
    QM_SYM(GetVersion, "ncclGetVersion");
#undef QM_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Broadcast && g_rccl.AllGather &&
                g_rccl.GetErrorString;
    if (!g_rccl.ok) fprintf(stderr, "[*E] qmann : RCCL : the loaded library lacks an entry point\n");
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int nccl_rc(const Rccl *r, ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return QMANN_OK;
    fprintf(stderr, "[*E] qmann : RCCL : %s : %s\n", what, r->GetErrorString(e));
    return QMANN_ECOMM;
}

struct OnDevice {
    int prev = -1;
    bool changed = false;
    explicit OnDevice(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) { QM_HIP(hipSetDevice(device)); changed = true; }
    }
    ~OnDevice() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

void qmann_shard_range(uint32_t n_query, uint32_t rank, uint32_t world, uint32_t *lo, uint32_t *hi)
{
    if (world == 0) world = 1;
    if (rank >= world) rank = world - 1;
    const uint32_t base = n_query / world, rem = n_query % world;
    const uint32_t a = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < rem ? 1u : 0u);
}

int qmann_comm_probe(int device)
{
    if (!rccl()) return QMANN_ECOMM;
    if (device < 0) return QMANN_OK;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); return QMANN_EHIP; }
    return device < n_dev ? QMANN_OK : QMANN_EINVAL;
}

int qmann_comm_get_id(void *id)
{
    if (!id) return QMANN_EINVAL;
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;
    static_assert(sizeof(ncclUniqueId) == QMANN_COMM_ID_BYTES, "qmann_dist.h: id size");
    ncclUniqueId u;
    const int rc = nccl_rc(r, r->GetUniqueId(&u), "ncclGetUniqueId");
    if (rc == QMANN_OK) memcpy(id, &u, sizeof u);
    return rc;
}

int qmann_comm_init_rank(qmann_comm **out, int n_ranks, int rank, const void *id, int device)
{
    QmBatched qm_scope;
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return QMANN_EINVAL;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) return QMANN_EHIP;
    if (device < 0 || device >= n_dev) return QMANN_EINVAL;
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;
    qmann_comm *c = new (std::nothrow) qmann_comm();
    if (!c) return QMANN_ERANGE;
    c->rank = rank; c->n_ranks = n_ranks; c->device = device;
    OnDevice on(device);                        // ncclCommInitRank binds the communicator to the current device
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    int rc = nccl_rc(r, r->CommInitRank(&c->comm, n_ranks, u, rank), "ncclCommInitRank");
    if (rc == QMANN_OK) rc = qm_scope.rc();
    if (rc != QMANN_OK) { delete c; return rc; }
    *out = c;
    return QMANN_OK;
}

void qmann_comm_destroy(qmann_comm *c)
{
    if (!c) return;
    const Rccl *r = rccl();
    if (r && c->comm) {
        OnDevice on(c->device);
        (void)nccl_rc(r, r->CommDestroy(c->comm), "ncclCommDestroy");
    }
    delete c;
}

int qmann_comm_info(const qmann_comm *c, int *rank, int *n_ranks, int *device, int *rccl_version)
{
    if (!c) return QMANN_EINVAL;
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    if (device) *device = c->device;
    if (rccl_version) {
        const Rccl *r = rccl();
        *rccl_version = 0;
        if (r && r->GetVersion) (void)r->GetVersion(rccl_version);
    }
    return QMANN_OK;
}

int qmann_comm_broadcast(qmann_comm *c, int root, void *buf, size_t bytes, void *stream)
{
    QmBatched qm_scope;
    if (!c || root < 0 || root >= c->n_ranks || (!buf && bytes)) return QMANN_EINVAL;
    if (bytes == 0) return QMANN_OK;
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;
    OnDevice on(c->device);
    const int rc = nccl_rc(r, r->Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
    return rc ? rc : qm_scope.rc();
}

int qmann_comm_allgather_u32(qmann_comm *c, const uint32_t *send, uint32_t *recv, size_t count, void *stream)
{
    QmBatched qm_scope;
    if (!c || ((!send || !recv) && count)) return QMANN_EINVAL;
    if (count == 0) return QMANN_OK;
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;
    OnDevice on(c->device);
    const int rc = nccl_rc(r, r->AllGather(send, recv, count, ncclUint32, c->comm, (hipStream_t)stream), "ncclAllGather");
    return rc ? rc : qm_scope.rc();
}

int qmann_comm_broadcast_params(qmann_comm *c, int root, const qmann_model *root_model, void **blob, size_t *bytes, void *stream)
{
    QmBatched qm_scope;
    // arguments every rank passes alike (the contract in the header): an error here is the same error on every rank
    if (!c || !blob || !bytes || root < 0 || root >= c->n_ranks) return QMANN_EINVAL;
    *blob = nullptr; *bytes = 0;
    const bool is_root = c->rank == root;
    // what only the root can get wrong: it still JOINS the size broadcast (with size 0, which every receiver maps to
    // QMANN_ECOMM) -- the other ranks are already waiting in it
    const void *src = nullptr;
    size_t n = 0;
    int root_err = QMANN_OK;
    if (is_root) {
        if (!root_model || qmann_model_params(root_model, &src, &n) != QMANN_OK || !src || n == 0) root_err = QMANN_EINVAL;
        else if (qmann_model_device(root_model) != c->device) root_err = QMANN_EINVAL;      // the blob must sit on this rank's GPU
        if (root_err) { src = nullptr; n = 0; }
    }
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;                 // (unreachable with a live communicator: it was made through this library)
    OnDevice on(c->device);
    hipStream_t st = (hipStream_t)stream;
    // one small device scratch: [0] the size, [1] this rank's status, [2 ..] every rank's status
    const size_t n_scr = 2 + (size_t)c->n_ranks;
    unsigned long long *d_scr = nullptr;
    QM_HIP(hipMalloc((void **)&d_scr, n_scr * sizeof *d_scr));
    if (!d_scr) return QMANN_EHIP;              // (a 100-byte allocation failing means the device is gone: nothing to agree on)
    // 1. the size (the other ranks need not know the model's dimensions beforehand)
    unsigned long long h_n = (unsigned long long)n;
    QM_HIP(hipMemcpyAsync(d_scr, &h_n, sizeof h_n, hipMemcpyHostToDevice, st));
    int rc = nccl_rc(r, r->Broadcast(d_scr, d_scr, 1, ncclUint64, root, c->comm, st), "ncclBroadcast(size)");
    if (rc == QMANN_OK) {
        QM_HIP(hipMemcpyAsync(&h_n, d_scr, sizeof h_n, hipMemcpyDeviceToHost, st));
        QM_HIP(hipStreamSynchronize(st));
        rc = qm_scope.rc();
    }
    if (rc != QMANN_OK) { QM_HIP(hipFree(d_scr)); return rc; }
    if (h_n == 0 || h_n > (1ull << 34)) {       // the root had nothing valid to send: every rank leaves here, in step
        QM_HIP(hipFree(d_scr));
        return root_err ? root_err : QMANN_ECOMM;
    }
    // 2. a fresh buffer on every rank (the root's model keeps its own) -- and an agreement that every rank HAS one, so that
    //    no rank waits in the bytes broadcast for one that left with an allocation failure
    void *dst = nullptr;
    const hipError_t e_alloc = hipMalloc(&dst, (size_t)h_n);
    if (e_alloc != hipSuccess) {
        fprintf(stderr, "[*E] HIP : qmann_comm_broadcast_params : %zu bytes : %s\n", (size_t)h_n, hipGetErrorString(e_alloc));
        (void)hipGetLastError();
        dst = nullptr;
    }
    std::vector<unsigned long long> stat(n_scr, 0);
    stat[1] = dst ? 1ull : 0ull;
    QM_HIP(hipMemcpyAsync(d_scr + 1, &stat[1], sizeof stat[1], hipMemcpyHostToDevice, st));
    rc = nccl_rc(r, r->AllGather(d_scr + 1, d_scr + 2, 1, ncclUint64, c->comm, st), "ncclAllGather(status)");
    if (rc == QMANN_OK) {
        QM_HIP(hipMemcpyAsync(stat.data() + 2, d_scr + 2, (size_t)c->n_ranks * sizeof stat[0], hipMemcpyDeviceToHost, st));
        QM_HIP(hipStreamSynchronize(st));
        rc = qm_scope.rc();
    }
    QM_HIP(hipFree(d_scr));
    bool all_ok = rc == QMANN_OK;
    for (int i = 0; i < c->n_ranks && all_ok; i++) all_ok = stat[2 + (size_t)i] == 1ull;
    if (!all_ok) {
        if (dst) QM_HIP(hipFree(dst));
        return rc != QMANN_OK ? rc : (stat[1] ? QMANN_ECOMM /* a peer could not allocate */ : QMANN_EHIP);
    }
    // 3. the bytes
    rc = nccl_rc(r, r->Broadcast(is_root ? src : dst, dst, (size_t)h_n, ncclUint8, root, c->comm, st), "ncclBroadcast(params)");
    QM_HIP(hipStreamSynchronize(st));
    if (rc == QMANN_OK) rc = qm_scope.rc();
    if (rc != QMANN_OK) { QM_HIP(hipFree(dst)); return rc; }
    *blob = dst; *bytes = (size_t)h_n;
    return QMANN_OK;
}

void qmann_params_free(void *blob)
{
    if (blob) (void)hipFree(blob);
}

}  // extern "C"
