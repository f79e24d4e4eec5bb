"""Multi-GPU plumbing: queries are independent, so ranks never exchange activations.  The one
collective is a broadcast of the parameter blob from rank 0 at start-up (RCCL over xGMI on the
GPU box, gloo in the CPU tests); shards are contiguous query ranges."""
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
