/*
 * qmann_dist.h -- the test-phase forward on several GPUs of one node, from a C host.
 *
 * The reference has no counterpart: MemN2N/MemN2N.c:2378 walks the test samples one by one on one device, and
 * lib/layer_cuda.cu holds no cudaSetDevice, no stream and no collective.  Queries are independent, so the batch is cut into
 * contiguous shards, one per GPU (qmann_shard_range), each GPU holds a replica of the model (qmann_model_create_from_params)
 * and its own queries' memories; nothing crosses GPUs during the forward pass.  The ONE collective is start-up: the
 * quantised parameter blob of the root's model (qmann_model_params: int8 tables, linear-map codes, formats, float answer
 * matrix) is broadcast with ncclBroadcast -- RCCL over xGMI.  Optionally the per-rank predictions are all-gathered.
 *
 * One rank = one GPU, driven by one host thread (threads of one process, as in examples/forward_sharded.c) or by one
 * process (as bench.py does under torch.distributed.run); the two set-ups use the same calls.
 *
 * librccl.so is loaded on first use (dlopen), NOT linked: a host that never calls a qmann_comm_* function -- the
 * reference's unmodified single-GPU program among them -- does not load the 0.5 GB library.  QMANN_RCCL_PATH, when set,
 * is the only file tried (a process must hold ONE RCCL: point it at the copy the process already has).
 * Errors: QMANN_ECOMM when the library cannot be loaded or an RCCL call fails (message on stderr), QMANN_EHIP / QMANN_EINVAL
 * as elsewhere.
 */
#ifndef QMANN_DIST_H
#define QMANN_DIST_H

#include "qmann_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmann_comm qmann_comm;

#define QMANN_COMM_ID_BYTES 128          /* = NCCL_UNIQUE_ID_BYTES */

/* Contiguous range [*lo, *hi) of `n_query` queries owned by `rank` of `world`; sizes differ by at most one and the ranges
 * tile [0, n_query) in rank order, so concatenating the ranks' predictions restores query order. */
void qmann_shard_range(uint32_t n_query, uint32_t rank, uint32_t world, uint32_t *lo, uint32_t *hi);

/* Rendezvous.  ONE rank calls qmann_comm_get_id (ncclGetUniqueId) and hands the 128 bytes to the others by whatever means
 * the host has (shared memory between threads, the launcher's store between processes); then EVERY rank calls
 * qmann_comm_init_rank with its own GPU (hipSetDevice index; ranks of one communicator need distinct GPUs -- RCCL refuses
 * two ranks on one device).  The call blocks until all n_ranks have joined. */
int qmann_comm_get_id(void *id /* QMANN_COMM_ID_BYTES */);
/* Local and non-blocking: can this rank take part?  QMANN_OK when librccl loads with every entry point and `device` is a
 * GPU of this process (device < 0: the library only); QMANN_ECOMM / QMANN_EHIP / QMANN_EINVAL otherwise.  A launcher has
 * every rank call this and AGREE on the result (over whatever channel carried the id) BEFORE any rank enters
 * qmann_comm_init_rank, which blocks until all n_ranks have joined: a rank that cannot join must keep the others out. */
int qmann_comm_probe(int device);
int qmann_comm_init_rank(qmann_comm **out, int n_ranks, int rank, const void *id, int device);
void qmann_comm_destroy(qmann_comm *c);
/* rank, size, GPU of this communicator and the RCCL version number the loaded library reports (any pointer may be NULL) */
int qmann_comm_info(const qmann_comm *c, int *rank, int *n_ranks, int *device, int *rccl_version);

/* The parameter broadcast.  On `root`, `root_model` is the model to replicate (NULL on the other ranks).  Every rank gets
 * *blob / *bytes: a device buffer on ITS GPU holding the root's parameter blob (the root gets a copy too), to be passed to
 * qmann_model_create_from_params and then released with qmann_params_free.  Collectives on `stream`: ncclBroadcast of the
 * size, a one-word ncclAllGather (has every rank got its buffer?), ncclBroadcast of the bytes; the call returns after
 * synchronising the stream.
 * Errors keep the ranks in step: `c`, `root`, `blob`, `bytes` are checked first and must be valid on EVERY rank (same
 * arguments, same verdict, no collective entered).  Whatever only ONE rank can get wrong is agreed on inside the call: a root
 * without a valid model (NULL, empty, or on another GPU than the communicator's) still joins the size broadcast and sends 0
 * -- it returns QMANN_EINVAL, the others QMANN_ECOMM; a rank whose buffer allocation fails still joins the status exchange --
 * it returns QMANN_EHIP, the others QMANN_ECOMM, nobody enters the bytes broadcast.  Every rank has then returned and the
 * communicator stays usable.  (An RCCL call that itself fails leaves the communicator undefined: destroy it on all ranks.) */
int qmann_comm_broadcast_params(qmann_comm *c, int root, const qmann_model *root_model, void **blob, size_t *bytes, void *stream);
void qmann_params_free(void *blob);

/* Plain collectives on device buffers of this rank's GPU, for hosts that need them (asynchronous on `stream`):
 * broadcast of `bytes` bytes in place, and an all-gather of `count` uint32 per rank into recv[n_ranks * count] in rank
 * order (predictions: pad the shards to a common count). */
int qmann_comm_broadcast(qmann_comm *c, int root, void *buf, size_t bytes, void *stream);
int qmann_comm_allgather_u32(qmann_comm *c, const uint32_t *send, uint32_t *recv, size_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QMANN_DIST_H */
