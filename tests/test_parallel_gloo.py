"""N > 1 path on CPU: two gloo ranks carry the QUANTISED parameter blob (the committed one of the trained bAbI task-1
model, tests/golden/trained_qa1/params_q.blob -- the very bytes qmann_comm_broadcast_params moves over RCCL on GPUs),
vet it with the library's host-side qmann_params_validate, shard a query batch and gather predictions -- the host logic
bench.py and replicate_model run on the GPU box.  Also: the rendezvous of the library's communicator when one rank
cannot take part (every rank must leave it in step)."""
import hashlib
import os
import socket
import struct
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLD, ROOT, load_pkg

BLOB = GOLD / "trained_qa1" / "params_q.blob"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(target, world, *extra):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    load_pkg()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from qmann_amd.parallel import blob_net, broadcast_blob, gather_predictions, shard_range
    raw, ms = broadcast_blob(BLOB.read_bytes() if rank == 0 else None, rank, world, torch.device("cpu"))
    net = blob_net(raw)
    n_query = 37
    lo, hi = shard_range(n_query, rank, world)
    pred_local = torch.arange(lo, hi, dtype=torch.int32) * 3          # stands in for this rank's predictions
    allp = gather_predictions(pred_local, n_query, rank, world)
    q.put((rank, hashlib.sha256(raw).hexdigest(), len(raw), lo, hi, allp.tolist(), ms, net))
    dist.destroy_process_group()


def test_two_ranks_broadcast_the_quantised_blob_shard_and_gather():
    res = _spawn(_worker, 2)
    want = BLOB.read_bytes()
    assert res[0][1] == res[1][1] == hashlib.sha256(want).hexdigest() and res[0][2] == res[1][2] == len(want)
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 19, 19, 37)
    assert res[0][5] == res[1][5] == [3 * i for i in range(37)]
    assert res[1][6] is not None
    net = res[1][7]                                                     # read out of the RECEIVED bytes
    assert (net["n_hop"], net["dim_emb"], net["dim_emb_pad"], net["dim_input"], net["attention_mode"]) == (3, 60, 64, 30, 2)
    assert net["fmt"] == [(5, 2)] * 3 and net["fmt_w"] == [(6, 1), (5, 2), (4, 3)]       # EN_MQ (MemN2N.c:748-754)


def _worker_bad_root(rank, world, port, q):
    dist = _init(rank, world, port)
    from qmann_amd.parallel import broadcast_blob
    raw = None
    if rank == 0:
        raw = bytearray(BLOB.read_bytes()); raw[0] ^= 0xFF; raw = bytes(raw)     # magic broken: nothing valid to send
    try:
        broadcast_blob(raw, rank, world, torch.device("cpu"))
        got = "no error"
    except RuntimeError as e:
        got = str(e)
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)                                                    # the ranks are still in step afterwards
    q.put((rank, got, int(t.item())))
    dist.destroy_process_group()


def test_a_root_without_a_valid_blob_fails_on_every_rank_in_step():
    res = _spawn(_worker_bad_root, 2)
    assert all("no valid parameter blob" in r[1] for r in res) and all(r[2] == 3 for r in res)


def _worker_rendezvous(rank, world, port, q, missing_on):
    if rank in missing_on:
        os.environ["QMANN_RCCL_PATH"] = "/nonexistent/librccl.so"        # this rank cannot load RCCL
    dist = _init(rank, world, port)
    from qmann_amd import abi
    from qmann_amd.parallel import Comm, CommUnavailable
    probe = abi.lib.qmann_comm_probe(-1)
    try:
        Comm(rank, world, -1)                                            # device -1: the probe checks the library only (no GPU here)
        got = "made"
    except CommUnavailable as e:
        got = f"unavailable: {e}"
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)                                                    # a mismatched collective would hang or garble this
    q.put((rank, got, int(t.item()), probe))
    dist.destroy_process_group()


@pytest.mark.parametrize("missing_on", [(0,), (1,)])
def test_rendezvous_when_one_rank_cannot_load_rccl(missing_on):
    """QMANN_RCCL_PATH pointing at a missing file on ONE rank.  No rank may enter the blocking
    qmann_comm_init_rank; every rank raises CommUnavailable, and the next collective of the process group still matches."""
    load_pkg()
    from qmann_amd import abi
    res = _spawn(_worker_rendezvous, 2, missing_on)
    for rank, got, total, probe in res:
        assert got.startswith("unavailable"), got
        assert total == 3
        assert probe == (abi.QMANN_ECOMM if rank in missing_on else abi.QMANN_OK)
    bad = res[missing_on[0]][1]
    assert "qmann_comm_probe" in bad                                    # the failing rank says why
    assert "another rank" in res[1 - missing_on[0]][1]


def test_params_validate_accepts_the_committed_blob_and_refuses_damage():
    """qmann_params_validate is pure host code: header magic / version / size, dimensions, formats, enumerations, canonical
    section offsets.  (tests/test_gpu_dist.py checks that the library still PRODUCES these bytes from the float weights.)"""
    import ctypes as C
    load_pkg()
    from qmann_amd import abi
    raw = BLOB.read_bytes()
    net = abi.Net()
    assert abi.lib.qmann_params_validate(raw, len(raw), C.byref(net)) == abi.QMANN_OK
    assert (net.n_hop, net.dim_emb, net.dim_input) == (3, 60, 30) and not any(net.lin_map[h] for h in range(8))
    assert abi.lib.qmann_params_validate(None, 0, None) == abi.QMANN_EINVAL
    assert abi.lib.qmann_params_validate(raw, len(raw) - 1, None) == abi.QMANN_EINVAL          # size field != bytes
    assert abi.lib.qmann_params_validate(raw[:64], 64, None) == abi.QMANN_EINVAL                # shorter than a header

    def damaged(off, fmt, value):
        b = bytearray(raw)
        struct.pack_into(fmt, b, off, value)
        return abi.lib.qmann_params_validate(bytes(b), len(b), None)
    o_net = 24                                                          # magic, version, bytes (u64), tied, reserved
    f = {n: getattr(abi.Net, n).offset + o_net for n, _ in abi.Net._fields_}
    assert damaged(0, "<I", 0x12345678) == abi.QMANN_EINVAL             # magic
    assert damaged(4, "<I", 99) == abi.QMANN_EINVAL                     # version
    assert damaged(8, "<Q", len(raw) + 256) == abi.QMANN_EINVAL         # size field
    assert damaged(f["n_hop"], "<I", 0) == abi.QMANN_EINVAL
    assert damaged(f["n_hop"], "<I", 9) == abi.QMANN_EINVAL
    assert damaged(f["n_hop"], "<I", 2) == abi.QMANN_EINVAL             # consistent header, but the layout no longer fits the size
    assert damaged(f["dim_emb_pad"], "<I", 96) == abi.QMANN_EUNSUPPORTED
    assert damaged(f["dim_emb"], "<I", 65) == abi.QMANN_EINVAL          # D > Dp
    assert damaged(f["attention_mode"], "<I", 7) == abi.QMANN_EINVAL
    assert damaged(f["softmax_base"], "<I", 3) == abi.QMANN_EINVAL
    assert damaged(f["act"], "<I", 6) == abi.QMANN_EINVAL               # act[0].iwl = 6 with frac 2: nine bits
    assert damaged(f["lin_map"], "<Q", 0xDEAD0000) == abi.QMANN_EINVAL  # a pointer where a blob carries none
    assert damaged(len(raw) - 8 - 0, "<Q", 0) in (abi.QMANN_OK,)        # payload bytes are not the validator's business
    off_tq = o_net + C.sizeof(abi.Net)
    assert damaged(off_tq, "<Q", 512) == abi.QMANN_EINVAL               # a section offset off the canonical layout


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
