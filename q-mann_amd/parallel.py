"""Multi-GPU plumbing: queries are independent, so ranks never exchange activations.  The one
collective is a broadcast of the parameters from rank 0 at start-up; shards are contiguous query ranges.

On GPUs the broadcast is the LIBRARY's (include/qmann_dist.h, csrc/dist.hip): rank 0's model hands out its QUANTISED
parameter blob (int8 tables, linear-map codes, formats, float answer matrix), `qmann_comm_broadcast_params` moves it with
ncclBroadcast (RCCL over xGMI) and every other rank builds its replica from the bytes (`replicate_model`).  torch.distributed
only carries the 128-byte rendezvous id.  The float-blob functions below (`broadcast_params`) remain for the CPU-only plumbing
test over gloo, where no GPU model can exist."""
from __future__ import annotations

import time

import numpy as np
import torch


def shard_range(n_query: int, rank: int, world: int):
    """Contiguous range [lo, hi) of queries owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_query, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def param_layout(cfg, with_emb: bool = False):
    """(name, shape) list of the parameter blob, in order.  with_emb adds the embedding matrices
    (models read from weight files; synthetic-memory runs need only the linear maps and the answer matrix)."""
    H, D, V = cfg["n_hop"], cfg["dim_emb"], cfg["dim_input"]
    lay = [(f"w_h{h}", (D, D)) for h in range(H)] + [("w_ans", (V, D))]
    if with_emb:
        lay += [("w_q", (D, V))] + [(f"w_a{h}", (D, V)) for h in range(H)] + [(f"w_c{h}", (D, V)) for h in range(H)]
    return lay


def pack_params(wts, cfg, with_emb: bool = False) -> np.ndarray:
    def get(name):
        if name in ("w_ans", "w_q"):
            return wts[name]
        return wts[name[:3]][int(name[3:])]
    return np.concatenate([np.ascontiguousarray(get(n), np.float32).ravel() for n, _ in param_layout(cfg, with_emb)])


def unpack_params(flat: np.ndarray, cfg, with_emb: bool = False) -> dict:
    out, o = {"w_h": [], "w_a": [], "w_c": []}, 0
    for name, shape in param_layout(cfg, with_emb):
        n = int(np.prod(shape))
        a = flat[o:o + n].reshape(shape).copy()
        o += n
        if name in ("w_ans", "w_q"):
            out[name] = a
        else:
            out[name[:3]].append(a)
    if not with_emb:
        del out["w_a"], out["w_c"]
    return out


def broadcast_params(wts, cfg, dev, rank: int, world: int, with_emb: bool = False):
    """Rank 0 holds `wts` (made in place or read with model.load_weights); every rank returns the same dict.
    One broadcast of one flat float32 blob.  Returns (wts, milliseconds or None)."""
    if world == 1:
        return wts, None
    import torch.distributed as dist
    n = sum(int(np.prod(s)) for _, s in param_layout(cfg, with_emb))
    blob = (torch.from_numpy(pack_params(wts, cfg, with_emb)).to(dev) if rank == 0
            else torch.zeros(n, dtype=torch.float32, device=dev))
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.broadcast(blob, src=0)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    return unpack_params(blob.cpu().numpy(), cfg, with_emb), ms


def gather_predictions(pred_local: torch.Tensor, n_query: int, rank: int, world: int) -> torch.Tensor:
    """Concatenate per-rank predictions in query order (optional; uneven shards are padded)."""
    if world == 1:
        return pred_local
    import torch.distributed as dist
    sizes = [shard_range(n_query, r, world) for r in range(world)]
    m = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros(m, dtype=pred_local.dtype, device=pred_local.device)
    buf[: pred_local.numel()] = pred_local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([parts[r][: sizes[r][1] - sizes[r][0]] for r in range(world)])


class Comm:
    """include/qmann_dist.h communicator of this rank: the id comes from rank 0 through the process group (any backend),
    the communicator itself is the library's (ncclCommInitRank inside libqmann_hip.so)."""

    def __init__(self, rank: int, world: int, device_index: int):
        import ctypes as C
        import os
        import torch.distributed as dist
        from . import abi
        # one RCCL per process: the copy PyTorch ships and has already loaded, unless the caller chose another
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(tl):
            os.environ.setdefault("QMANN_RCCL_PATH", tl)
        self.abi, self.rank, self.world, self.device_index = abi, rank, world, device_index
        ident = np.zeros(abi.COMM_ID_BYTES, np.uint8)
        if rank == 0:
            abi.check(abi.lib.qmann_comm_get_id(ident.ctypes.data_as(C.c_void_p)), "qmann_comm_get_id")
        on_gpu = dist.get_backend() == "nccl"
        t = torch.from_numpy(ident)
        if on_gpu:
            t = t.to(torch.device("cuda", device_index))
        dist.broadcast(t, src=0)
        ident = t.cpu().numpy().copy()
        h = C.c_void_p()
        abi.check(abi.lib.qmann_comm_init_rank(C.byref(h), world, rank, ident.ctypes.data_as(C.c_void_p), device_index),
                  "qmann_comm_init_rank")
        self.h = h

    def info(self):
        import ctypes as C
        r, n, d, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.abi.lib.qmann_comm_info(self.h, C.byref(r), C.byref(n), C.byref(d), C.byref(v))
        return dict(rank=r.value, n_ranks=n.value, device=d.value, rccl_version=v.value)

    def close(self):
        if getattr(self, "h", None):
            self.abi.lib.qmann_comm_destroy(self.h)
            self.h = None


def replicate_model(hm, cfg, dev, rank: int, world: int, comm: "Comm | None", model_mod):
    """Rank 0 holds `hm` (a model.HostModel); every rank returns a HostModel computing from the same quantised bytes.
    comm given: the library's RCCL broadcast of the blob; comm None with world > 1 (one-GPU rehearsal over gloo): the blob
    bytes travel through the process group and the replica is built from host memory.  Returns (model, ms, how)."""
    if world == 1:
        return hm, None, "single rank"
    import ctypes as C
    import torch.distributed as dist
    from . import abi
    if comm is not None:
        blob, n = C.c_void_p(), C.c_size_t()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        abi.check(abi.lib.qmann_comm_broadcast_params(comm.h, 0, hm.h if rank == 0 else None, C.byref(blob), C.byref(n), None),
                  "qmann_comm_broadcast_params")
        ms = (time.perf_counter() - t0) * 1e3
        try:
            if rank != 0:
                hm = model_mod.HostModel.from_params(cfg, blob.value, n.value, device=str(dev))
        finally:
            abi.lib.qmann_params_free(blob)
        return hm, ms, f"qmann_comm_broadcast_params: ncclBroadcast of the quantised blob ({n.value} bytes)"
    # no library communicator (the one-GPU rehearsal over gloo, or the C-level rendezvous failed): the same blob, carried by
    # the process group -- on device tensors when the group's backend is RCCL, on host tensors otherwise
    on_gpu = dist.get_backend() == "nccl"
    pg_dev = dev if on_gpu else torch.device("cpu")
    size = torch.zeros(1, dtype=torch.int64, device=pg_dev)
    raw = hm.params_bytes() if rank == 0 else b""
    if rank == 0:
        size[0] = len(raw)
    dist.broadcast(size, src=0)
    n = int(size.item())
    buf = (torch.frombuffer(bytearray(raw), dtype=torch.uint8) if rank == 0 else torch.zeros(n, dtype=torch.uint8)).to(pg_dev)
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(buf, src=0)
    if on_gpu:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        host = buf.cpu().numpy()
        hm = model_mod.HostModel.from_params(cfg, host.ctypes.data, host.size, device=str(dev))
    return hm, ms, (f"process-group broadcast of the quantised blob ({n} bytes, backend {dist.get_backend()}; "
                    "not the library's own communicator)")
