// dist.hip -- include/qmann_dist.h: shards, the RCCL rendezvous and the one-time broadcast of the quantised parameter
// blob.  Host-side C++, no kernels.  librccl is loaded with dlopen on first use (see the header for why).
#include "rt.h"
#include "../../include/qmann_dist.h"

#include <dlfcn.h>
#include <mutex>
#include <new>
#include <rccl/rccl.h>          // types and prototypes only: no symbol of it is linked
#include <string.h>

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
    const char *names[4] = {getenv("QMANN_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
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
    if (!c || !blob || !bytes || root < 0 || root >= c->n_ranks) return QMANN_EINVAL;
    *blob = nullptr; *bytes = 0;
    const bool is_root = c->rank == root;
    const void *src = nullptr;
    size_t n = 0;
    if (is_root) {
        if (!root_model || qmann_model_params(root_model, &src, &n) != QMANN_OK || !src || n == 0) return QMANN_EINVAL;
        if (qmann_model_device(root_model) != c->device) return QMANN_EINVAL;      // the blob must sit on this rank's GPU
    }
    const Rccl *r = rccl();
    if (!r) return QMANN_ECOMM;
    OnDevice on(c->device);
    hipStream_t st = (hipStream_t)stream;
    // 1. the size (the other ranks need not know the model's dimensions beforehand)
    unsigned long long *d_n = nullptr, h_n = (unsigned long long)n;
    QM_HIP(hipMalloc((void **)&d_n, sizeof *d_n));
    if (!d_n) return QMANN_EHIP;
    QM_HIP(hipMemcpyAsync(d_n, &h_n, sizeof h_n, hipMemcpyHostToDevice, st));
    int rc = nccl_rc(r, r->Broadcast(d_n, d_n, 1, ncclUint64, root, c->comm, st), "ncclBroadcast(size)");
    if (rc == QMANN_OK) {
        QM_HIP(hipMemcpyAsync(&h_n, d_n, sizeof h_n, hipMemcpyDeviceToHost, st));
        QM_HIP(hipStreamSynchronize(st));
        rc = qm_scope.rc();
    }
    QM_HIP(hipFree(d_n));
    if (rc != QMANN_OK) return rc;
    if (h_n == 0 || h_n > (1ull << 34)) return QMANN_ECOMM;
    // 2. the bytes, into a fresh buffer on every rank (the root's model keeps its own)
    void *dst = nullptr;
    QM_HIP(hipMalloc(&dst, (size_t)h_n));
    if (!dst) return QMANN_EHIP;
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
