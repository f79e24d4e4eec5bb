"""bench.py --gpus N without a launcher: the parent starts N ranks itself (before anything touches the GPU),
rank 0 prints one line with n_gpus = N.  Runs here on CPU: QMANN_BENCH_PLUMBING=1 keeps the ranks off the GPU
(gloo, CPU tensors) while exercising the same launcher, rendezvous, broadcast and max-over-ranks timing code."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("qmann_bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_command_shape():
    b = _bench()
    assert b.torch is None                       # importing bench.py pulls in neither torch nor the GPU stack
    cmd, env = b.launch_command(4, ["--gpus", "4", "--steps", "7"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == str(ROOT / "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_ADDR"] == "127.0.0.1"


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(QMANN_BENCH_PLUMBING="1", **env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=e, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["collective"]["world_size_seen"] == 2 and out["collective"]["self_launched"] is True
    assert out["param_broadcast_ms"] >= 0.0
    rk = out["ranks"]
    assert (rk["rank"]["min"], rk["rank"]["max"]) == (0.0, 1.0)
    # every rank ended up with rank 0's QUANTISED blob (the committed one), and the shards tile the batch
    import zlib
    blob = (ROOT / "tests" / "golden" / "trained_qa1" / "params_q.blob").read_bytes()
    assert rk["param_crc32"]["min"] == rk["param_crc32"]["max"] == float(zlib.crc32(blob))
    assert out["param_broadcast"]["bytes"] == len(blob) == out["config"]["blob"]["bytes"]
    assert out["config"]["blob"]["dim_emb"] == 60 and out["config"]["blob"]["n_hop"] == 3
    assert rk["shard_size"]["min"] == rk["shard_size"]["max"] == 64
    assert (rk["shard_lo"]["min"], rk["shard_hi"]["min"], rk["shard_lo"]["max"], rk["shard_hi"]["max"]) == (0.0, 64.0, 64.0, 128.0)


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_single_rank_unchanged():
    r = _run(["--gpus", "1", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and "collective" not in out and "ranks" not in out
