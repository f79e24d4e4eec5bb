"""Multi-GPU plumbing: queries are independent, so ranks never exchange activations.  The one
collective is a broadcast of the parameters from rank 0 at start-up; shards are contiguous query ranges.

On GPUs the broadcast is the LIBRARY's (include/qmann_dist.h, csrc/dist.hip): rank 0's model hands out its QUANTISED
parameter blob (int8 tables, linear-map codes, formats, float answer matrix), `qmann_comm_broadcast_params` moves it with
ncclBroadcast (RCCL over xGMI) and every other rank builds its replica from the bytes (`replicate_model`).  torch.distributed
only carries the 128-byte rendezvous id.  Where no library communicator exists (gloo: the CPU tests, the one-GPU rehearsal)
the SAME blob travels through the process group (`broadcast_blob`) and is vetted by the library's host-side
`qmann_params_validate` on arrival; no float weights ever cross ranks."""
from __future__ import annotations

import time

import numpy as np
import torch


def shard_range(n_query: int, rank: int, world: int):
    """Contiguous range [lo, hi) of queries owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_query, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(raw: "bytes | None", rank: int, world: int, dev=None) -> "tuple[bytes, float]":
    """The quantised parameter blob through the PROCESS GROUP (any backend): rank 0 passes the bytes, every rank returns
    them.  Size first (receivers need not know the model), then the bytes; on device tensors when the group's backend is
    RCCL, on host tensors otherwise.  Receivers vet what arrived with the library's host-side qmann_params_validate before
    any model is built from it.  Returns (bytes, milliseconds of the bytes broadcast)."""
    import ctypes as C
    import torch.distributed as dist
    from . import abi
    on_gpu = dist.get_backend() == "nccl"
    pg_dev = dev if on_gpu else torch.device("cpu")
    size = torch.zeros(1, dtype=torch.int64, device=pg_dev)
    if rank == 0:
        # a root with nothing valid still joins both broadcasts (size 0): the ranks stay in step and all of them raise
        ok = raw is not None and abi.lib.qmann_params_validate(raw, len(raw), None) == abi.QMANN_OK
        size[0] = len(raw) if ok else 0
    dist.broadcast(size, src=0)
    n = int(size.item())
    if n == 0:
        raise RuntimeError("broadcast_blob: rank 0 holds no valid parameter blob")
    buf = (torch.frombuffer(bytearray(raw), dtype=torch.uint8) if rank == 0 else torch.zeros(n, dtype=torch.uint8)).to(pg_dev)
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(buf, src=0)
    if on_gpu:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out = raw if rank == 0 else buf.cpu().numpy().tobytes()
    abi.check(abi.lib.qmann_params_validate(out, len(out), None), "qmann_params_validate (received blob)")
    return out, ms


def blob_net(raw: bytes) -> dict:
    """Dimensions and formats of the model a parameter blob holds (qmann_params_validate's `net`)."""
    import ctypes as C
    from . import abi
    net = abi.Net()
    abi.check(abi.lib.qmann_params_validate(raw, len(raw), C.byref(net)), "qmann_params_validate")
    H = net.n_hop
    return dict(n_hop=H, dim_emb=net.dim_emb, dim_emb_pad=net.dim_emb_pad, dim_input=net.dim_input,
                attention_mode=net.attention_mode, softmax_base=net.softmax_base, en_lin_map=bool(net.en_lin_map),
                fmt=[(net.act[h].iwl, net.act[h].frac) for h in range(H)], fmt_w=[(net.w[h].iwl, net.w[h].frac) for h in range(H)],
                fmt_att=[(net.att[h].iwl, net.att[h].frac) for h in range(H)])


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


class CommUnavailable(RuntimeError):
    """The ranks AGREED that the library's communicator cannot be made (every rank raises this, in step)."""


class Comm:
    """include/qmann_dist.h communicator of this rank: the id comes from rank 0 through the process group (any backend),
    the communicator itself is the library's (ncclCommInitRank inside libqmann_hip.so).

    The rendezvous keeps the ranks in step whatever fails where: (1) every rank probes locally, without blocking
    (qmann_comm_probe: librccl loads, the GPU index exists) and rank 0 also draws the id; (2) the id is broadcast ALWAYS
    (zeros when rank 0 failed); (3) the ranks all_reduce(MIN) their flags; (4) only when all agree do they enter the
    blocking qmann_comm_init_rank -- otherwise every rank raises CommUnavailable and the caller's ranks all take the
    process-group road together (bench.py / replicate_model)."""

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
        self.h = None
        on_gpu = dist.get_backend() == "nccl"
        pg_dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")
        why = []
        rc = abi.lib.qmann_comm_probe(device_index)
        if rc != abi.QMANN_OK:
            why.append(f"qmann_comm_probe({device_index}) = {rc}")
        ident = np.zeros(abi.COMM_ID_BYTES, np.uint8)
        if rank == 0 and not why:
            rc = abi.lib.qmann_comm_get_id(ident.ctypes.data_as(C.c_void_p))
            if rc != abi.QMANN_OK:
                why.append(f"qmann_comm_get_id = {rc}")
                ident[:] = 0
        t = torch.from_numpy(ident).to(pg_dev)
        dist.broadcast(t, src=0)                                     # always: nobody is left waiting in it
        ident = t.cpu().numpy().copy()
        ok = torch.tensor([0 if why else 1], dtype=torch.int32, device=pg_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise CommUnavailable("; ".join(why) if why else "another rank cannot join the library's communicator")
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
    # no library communicator (the one-GPU rehearsal over gloo, or the ranks agreed that the C-level rendezvous cannot be
    # made): the same blob, carried by the process group and vetted on arrival
    raw, ms = broadcast_blob(hm.params_bytes() if rank == 0 else None, rank, world, dev)
    if rank != 0:
        host = np.frombuffer(raw, np.uint8)
        hm = model_mod.HostModel.from_params(cfg, host.ctypes.data, host.size, device=str(dev))
    return hm, ms, (f"process-group broadcast of the quantised blob ({len(raw)} bytes, backend {dist.get_backend()}; "
                    "not the library's own communicator)")
