"""N > 1 path on CPU: two gloo ranks shard a query batch, broadcast the parameters from rank 0 and
gather predictions -- the same host logic bench.py runs over RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    load_pkg()
    from qmann_amd.parallel import broadcast_params, gather_predictions, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(n_hop=3, dim_emb=20, dim_input=11)
    wts = None
    if rank == 0:
        rng = np.random.default_rng(7)
        wts = {"w_h": [rng.normal(0, 1, (20, 20)).astype(np.float32) for _ in range(3)],
               "w_ans": rng.normal(0, 1, (11, 20)).astype(np.float32)}
    wts, ms = broadcast_params(wts, cfg, torch.device("cpu"), rank, world)
    n_query = 37
    lo, hi = shard_range(n_query, rank, world)
    pred_local = torch.arange(lo, hi, dtype=torch.int32) * 3          # stands in for this rank's predictions
    allp = gather_predictions(pred_local, n_query, rank, world)
    q.put((rank, float(sum(w.sum() for w in wts["w_h"]) + wts["w_ans"].sum()), lo, hi, allp.tolist(), ms))
    dist.destroy_process_group()


def _worker_files(rank, world, port, q, directory):
    """Rank 0 reads a model from weight files (fixed-point words), every rank ends with the same matrices."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import hashlib
    import torch.distributed as dist
    load_pkg()
    import qmann_amd.model as model
    from qmann_amd.parallel import broadcast_params
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = model.babi_cfg(30, 2, 0)
    wts = model.load_weights(directory, cfg, from_fixed=True) if rank == 0 else None
    wts, _ = broadcast_params(wts, cfg, torch.device("cpu"), rank, world, with_emb=True)
    h = hashlib.sha256()
    for k in ("w_q", "w_ans"):
        h.update(wts[k].tobytes())
    for k in ("w_a", "w_c", "w_h"):
        for m in wts[k]:
            h.update(m.tobytes())
    q.put((rank, h.hexdigest(), wts["w_a"][2].shape))
    dist.destroy_process_group()


def test_two_rank_broadcast_of_a_file_backed_model(tmp_path):
    load_pkg()
    import qmann_amd.model as model
    cfg = model.babi_cfg(30, 2, 0)
    rng = np.random.default_rng(11)
    H, D, V = 3, 60, 30
    w = {"w_q": rng.normal(0, 1, (D, V)).astype(np.float32), "w_ans": rng.normal(0, 1, (V, D)).astype(np.float32),
         "w_a": [rng.normal(0, 1, (D, V)).astype(np.float32) for _ in range(H)],
         "w_c": [rng.normal(0, 1, (D, V)).astype(np.float32) for _ in range(H)],
         "w_h": [rng.normal(0, 1, (D, D)).astype(np.float32) for _ in range(H)]}
    model.save_weights(tmp_path, w, cfg, fixed=True)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_files, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == (60, 30)


def test_two_rank_broadcast_shard_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                      # identical parameters on both ranks
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 19, 19, 37)
    assert res[0][4] == res[1][4] == [3 * i for i in range(37)]
    assert res[0][5] is not None


def test_shard_range_covers_everything():
    load_pkg()
    from qmann_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
