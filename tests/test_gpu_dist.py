"""The C-level multi-GPU path (include/qmann_dist.h, the parameter blob of include/qmann_model.h, examples/forward_sharded.c):
  * a replica built from a model's QUANTISED parameter blob computes exactly what the model computes (every attention mode,
    word-index and bag-of-words inputs, blob taken from device memory or carried through host memory);
  * a blob that is not one is refused;
  * the RCCL leg: rendezvous, broadcast of the blob and all-gather through the library's own communicator -- with one rank on
    this one-GPU box (the same calls as with eight; RCCL refuses two ranks on one device), and with one rank per GPU when the
    box has at least two;
  * the sharded C host: N threads, contiguous shards, predictions concatenated -- identical to the single-shard run, with
    two and three threads sharing GPU 0 (thread safety of the library + the shard / concatenate logic).
The reference has no counterpart to compare with (one device, one query at a time: MemN2N.c:2378); the oracle check of the
predictions themselves is in test_c_host.py / test_gpu_words.py, this file checks that sharding and replication change nothing."""
import ctypes as C
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu

INC = ["-I", str(ROOT / "include")]
LINK = ["-L", str(ROOT / "q-mann_amd" / "lib"), "-lqmann_hip", "-L/opt/rocm/lib", "-lamdhip64",
        f"-Wl,-rpath,{ROOT / 'q-mann_amd' / 'lib'}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread"]


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    load_pkg()
    import qmann_amd.abi as abi
    import qmann_amd.model as model

    class Env:
        pass
    e = Env()
    e.torch, e.model, e.abi, e.dev = torch, model, abi, torch.device("cuda:0")
    return e


def weights(seed, H, D, V):
    rng = np.random.default_rng(seed)
    return {"w_q": rng.normal(0, 1.0, (D, V)).astype(np.float32),
            "w_a": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
            "w_c": [rng.normal(0, 1.0, (D, V)).astype(np.float32) for _ in range(H)],
            "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)],
            "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


def random_batch(rng, B, V, max_sen, W=8):
    n_sen = rng.integers(0, max_sen + 1, B)
    rows = int(n_sen.sum())
    dd = V - max_sen                                         # dictionary words, then the time slots
    sw = np.full((max(rows, 1), W), 0xFFFF, np.uint16)
    r = 0
    for q in range(B):
        for j in range(n_sen[q]):
            k = int(rng.integers(1, W - 1))
            sw[r, :k] = rng.integers(1, dd, k)
            sw[r, k] = dd + n_sen[q] - 1 - j
            r += 1
    qw = np.full((B, W), 0xFFFF, np.uint16)
    for q in range(B):
        k = int(rng.integers(1, 5))
        qw[q, :k] = rng.integers(1, dd, k)
    ans = rng.integers(0, dd, B).astype(np.int32)
    return sw[:rows], qw, n_sen, ans


def to_bow(words, V, with_time):
    out = np.zeros((max(words.shape[0], 1), V), np.float32)
    for r, row in enumerate(words):
        ent = [int(w) for w in row if w != 0xFFFF]
        t = ent.pop() if (with_time and ent) else None
        for w in ent:
            out[r, w] += 1.0
        if t is not None:
            out[r, t] = 1.0
    return out


@pytest.mark.parametrize("mode,nb,en_mq", [(1, 8, True), (2, 8, True), (2, 8, False), (3, 8, False), (10, 4, False), (11, 8, False)])
@pytest.mark.parametrize("via_host", [False, True])
def test_replica_from_the_parameter_blob_computes_the_same(env, mode, nb, en_mq, via_host):
    torch, model, abi = env.torch, env.model, env.abi
    V, D, H, B = 46, 60, 3, 257
    cfg = model.babi_cfg(V, attention_mode=mode, D=D, en_mq=en_mq)
    cfg["num_bit"] = nb
    wts = weights(100 + mode, H, D, V)
    if not en_mq and mode == 2:                              # tied matrices: the flag travels in the blob
        wts["w_a"] = [wts["w_a"][0]] * H; wts["w_c"] = [wts["w_c"][0]] * H
    rng = np.random.default_rng(mode * 31 + nb)
    sw, qw, n_sen, ans = random_batch(rng, B, V, 12)
    row_off = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int32)
    d_sw = torch.from_numpy(np.ascontiguousarray(sw).view(np.int16)).to(env.dev)
    d_qw = torch.from_numpy(qw.view(np.int16)).to(env.dev)
    d_ro = torch.from_numpy(row_off).to(env.dev); d_ans = torch.from_numpy(ans).to(env.dev)
    # bag-of-words rows of the same batch, some made irregular (a fractional entry: the float kernels take those rows)
    story = to_bow(sw, V, True); ques = to_bow(qw, V, False)
    story[::7, 3] = 0.5; ques[::5, 2] = 1.25
    d_st = torch.from_numpy(story).to(env.dev); d_qu = torch.from_numpy(ques).to(env.dev)

    src = model.HostModel(cfg, wts)
    p0, c0, m0 = src.forward_words(d_sw, d_qw, d_ro, 12, d_ans); torch.cuda.synchronize()
    u0 = src.last_u(B).cpu().numpy()
    pb0, _, mb0 = src.forward_bow(d_st, d_qu, d_ro, 12, d_ans); torch.cuda.synchronize()
    ub0 = src.last_u(B).cpu().numpy()
    ptr, n = src.params()
    assert n > 0 and abi.lib.qmann_model_device(src.h) == 0
    if via_host:
        raw = np.frombuffer(src.params_bytes(), np.uint8).copy()
        src.close()                                          # the replica must not depend on the source model
        rep = model.HostModel.from_params(cfg, raw.ctypes.data, raw.size)
    else:
        rep = model.HostModel.from_params(cfg, ptr, n)
        src.close()
    p1, c1, m1 = rep.forward_words(d_sw, d_qw, d_ro, 12, d_ans); torch.cuda.synchronize()
    np.testing.assert_array_equal(rep.last_u(B).cpu().numpy(), u0)
    np.testing.assert_array_equal(p1.cpu().numpy(), p0.cpu().numpy())
    assert int(m1.item()) == int(m0.item()) and float(c1.item()) == pytest.approx(float(c0.item()), rel=1e-4, abs=1e-3)
    pb1, _, mb1 = rep.forward_bow(d_st, d_qu, d_ro, 12, d_ans); torch.cuda.synchronize()
    np.testing.assert_array_equal(rep.last_u(B).cpu().numpy(), ub0)
    np.testing.assert_array_equal(pb1.cpu().numpy(), pb0.cpu().numpy())
    assert int(mb1.item()) == int(mb0.item()) and np.abs(u0).sum() > 0
    rep.close()


def test_bag_of_words_forward_equals_the_float_chain_on_irregular_rows(env):
    """the float embedding kernels of a model now read the TABLES' grid values (qmann_dequantize_table_f32) instead of the
    original float matrices: same memories, because dense_mat_fwd quantises its weights on entry anyway
    (lib/layer_cuda.cu:120) -- checked against the op-by-op chain that gets the original matrices"""
    torch, model = env.torch, env.model
    V, D, H, B = 40, 60, 3, 120
    cfg = model.babi_cfg(V, attention_mode=2, D=D)
    wts = weights(7, H, D, V)
    rng = np.random.default_rng(9)
    sw, qw, n_sen, _ = random_batch(rng, B, V, 10)
    row_off = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.int32)
    story = to_bow(sw, V, True); ques = to_bow(qw, V, False)
    story[::3, 5] = -0.75; story[1::4, 2] = 2.5; ques[::2, 7] = 0.25
    d_st = torch.from_numpy(story).to(env.dev); d_qu = torch.from_numpy(ques).to(env.dev)
    d_ro = torch.from_numpy(row_off).to(env.dev)
    hm = model.HostModel(cfg, wts)
    pred, _, _ = hm.forward_bow(d_st, d_qu, d_ro, 10); torch.cuda.synchronize()
    u = hm.last_u(B).cpu().numpy()
    net = model.QNet(cfg, wts)
    keys, vals, u0 = net.embed(d_st, d_qu)
    uc = net.hops(keys, vals, d_ro, 10, u0, taps=True)[0]
    pc = net.answer(uc)[0]; torch.cuda.synchronize()
    np.testing.assert_array_equal(u, uc.cpu().numpy())
    np.testing.assert_array_equal(pred.cpu().numpy(), pc.cpu().numpy())
    hm.close()


def test_a_blob_that_is_not_one_is_refused(env):
    torch, model, abi = env.torch, env.model, env.abi
    cfg = model.babi_cfg(30, attention_mode=2)
    hm = model.HostModel(cfg, weights(1, 3, 60, 30))
    raw = np.frombuffer(hm.params_bytes(), np.uint8).copy()
    h = C.c_void_p()

    def create(buf, n):
        return abi.lib.qmann_model_create_from_params(C.byref(h), 0, buf.ctypes.data_as(C.c_void_p), n, None)
    bad = raw.copy(); bad[0] ^= 0xFF                         # magic
    assert create(bad, bad.size) == abi.QMANN_EINVAL
    assert create(raw, raw.size - 256) == abi.QMANN_EINVAL  # size does not match the header
    bad = raw.copy(); bad[24] = 200                          # n_hop beyond QMANN_MAX_HOP (first field of the net)
    assert create(bad, bad.size) in (abi.QMANN_EINVAL, abi.QMANN_EUNSUPPORTED)
    assert create(raw, 16) == abi.QMANN_EINVAL
    assert abi.lib.qmann_model_create_from_params(C.byref(h), 99, raw.ctypes.data_as(C.c_void_p), raw.size, None) == abi.QMANN_EINVAL
    assert create(raw, raw.size) == 0                        # and the untouched bytes still make a model
    abi.lib.qmann_model_destroy(h)
    hm.close()


def _device_count():
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int()
    assert hip.hipGetDeviceCount(C.byref(n)) == 0
    return n.value


def test_rccl_one_rank_communicator_broadcasts_the_blob(env):
    """the library's own RCCL path on this box: librccl loaded by dlopen, ncclGetUniqueId, ncclCommInitRank (one rank),
    ncclBroadcast of the size and of the blob, ncclAllGather -- the calls an 8-GPU host makes, with no peer"""
    torch, model, abi = env.torch, env.model, env.abi
    cfg = model.babi_cfg(30, attention_mode=2)
    hm = model.HostModel(cfg, weights(2, 3, 60, 30))
    ident = (C.c_ubyte * abi.COMM_ID_BYTES)()
    assert abi.lib.qmann_comm_get_id(ident) == 0
    assert any(ident)
    comm = C.c_void_p()
    assert abi.lib.qmann_comm_init_rank(C.byref(comm), 1, 0, ident, 0) == 0
    r, n, d, v = C.c_int(-1), C.c_int(-1), C.c_int(-1), C.c_int(0)
    assert abi.lib.qmann_comm_info(comm, C.byref(r), C.byref(n), C.byref(d), C.byref(v)) == 0
    assert (r.value, n.value, d.value) == (0, 1, 0) and v.value > 20000, v.value
    blob, nbytes = C.c_void_p(), C.c_size_t()
    assert abi.lib.qmann_comm_broadcast_params(comm, 0, hm.h, C.byref(blob), C.byref(nbytes), None) == 0
    assert nbytes.value == hm.params()[1] and blob.value and blob.value != hm.params()[0]
    got = torch.empty(nbytes.value, dtype=torch.uint8, device=env.dev)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(got.data_ptr()), blob, C.c_size_t(nbytes.value), 3) == 0          # device to device
    assert got.cpu().numpy().tobytes() == hm.params_bytes()
    rep = model.HostModel.from_params(cfg, blob.value, nbytes.value)
    abi.lib.qmann_params_free(blob)
    rep.close()
    # wrong arguments come back as codes
    assert abi.lib.qmann_comm_broadcast_params(comm, 1, hm.h, C.byref(blob), C.byref(nbytes), None) == abi.QMANN_EINVAL
    assert abi.lib.qmann_comm_broadcast_params(comm, 0, None, C.byref(blob), C.byref(nbytes), None) == abi.QMANN_EINVAL
    send = torch.arange(5, dtype=torch.int32, device=env.dev); recv = torch.zeros(5, dtype=torch.int32, device=env.dev)
    assert abi.lib.qmann_comm_allgather_u32(comm, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 5, None) == 0
    torch.cuda.synchronize()
    assert recv.cpu().tolist() == [0, 1, 2, 3, 4]
    buf = torch.full((64,), 7, dtype=torch.uint8, device=env.dev)
    assert abi.lib.qmann_comm_broadcast(comm, 0, C.c_void_p(buf.data_ptr()), 64, None) == 0
    torch.cuda.synchronize()
    assert int(buf.sum()) == 7 * 64
    abi.lib.qmann_comm_destroy(comm)
    hm.close()


# ---- the sharded C host -------------------------------------------------------------------------------------------------

def write_set(path, records):
    out = ["", "+NS+", str(len(records)), ""]
    for i, (sens, q, a) in enumerate(records):
        out += ["+I+", str(i), "+S+", str(len(sens))] + [x + " " for x in sens] + ["+Q+", q + " ", "+A+", a, ""]
    path.write_text("\n".join(out) + "\n")


@pytest.fixture(scope="module")
def sharded_setup(tmp_path_factory, env):
    tmp = tmp_path_factory.mktemp("sharded")
    exes = {}
    for name in ("forward_dataset", "forward_sharded"):
        exe = tmp / name
        r = subprocess.run(["gcc", "-std=c99", "-Wall", *INC, "-I/opt/rocm/include", str(ROOT / "examples" / f"{name}.c"), *LINK,
                            "-o", str(exe)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        exes[name] = exe
    rng = np.random.default_rng(11)
    vocab = [f"w{i}" for i in range(24)]
    recs = lambda n: [([" ".join(rng.choice(vocab, rng.integers(1, 6))) for _ in range(rng.integers(1, 11))],
                       " ".join(rng.choice(vocab, 2)), str(rng.choice(vocab))) for _ in range(n)]
    write_set(tmp / "train", recs(300)); write_set(tmp / "test", recs(1001))       # (1001: uneven shards)
    ds = env.abi.load_dataset(tmp / "train", tmp / "test", 50)
    V, D, H, iwl = ds["dim_input"], 60, 3, 5
    cfg = env.model.babi_cfg(V, 2, 0, iwl=iwl)
    wts = {"w_q": rng.normal(0, 0.8, (D, V)).astype(np.float32), "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32),
           "w_a": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_c": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)]}
    (tmp / "weights").mkdir()
    env.model.save_weights(tmp / "weights", wts, cfg, fixed=False)
    r = subprocess.run([str(exes["forward_dataset"]), str(tmp / "train"), str(tmp / "test"), str(tmp / "weights"), str(iwl),
                        str(tmp / "single.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return tmp, exes, ds["n_query"], iwl


def read_out(path, nq):
    raw = path.read_bytes()
    return (np.frombuffer(raw[:4 * nq], np.uint32), struct.unpack("<I", raw[4 * nq:4 * nq + 4])[0],
            struct.unpack("<f", raw[4 * nq + 4:])[0])


@pytest.mark.parametrize("shards,rccl", [(2, "off"), (3, "off"), (2, "on"), (5, "auto")])
def test_sharded_host_threads_on_one_gpu_equal_the_single_shard_run(sharded_setup, shards, rccl):
    """2, 3 and 5 host threads share GPU 0, each with its own model object and stream; the parameters reach the replicas as
    the quantised blob (copied device to device, or through a one-rank RCCL communicator)"""
    tmp, exes, nq, iwl = sharded_setup
    out = tmp / f"sharded_{shards}_{rccl}.bin"
    r = subprocess.run([str(exes["forward_sharded"]), str(tmp / "train"), str(tmp / "test"), str(tmp / "weights"), str(iwl),
                        str(shards), str(out), "0", rccl], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout)
    assert ("RCCL broadcast" in r.stdout) == (rccl == "on")
    p1, m1, c1 = read_out(tmp / "single.bin", nq)
    p2, m2, c2 = read_out(out, nq)
    np.testing.assert_array_equal(p2, p1)
    assert m2 == m1 and c2 == pytest.approx(c1, rel=1e-4, abs=1e-3)


def test_sharded_host_one_rank_per_gpu_over_rccl(sharded_setup):
    """the real thing: one thread and one RCCL rank per GPU, the blob broadcast over xGMI -- needs at least two GPUs"""
    n_dev = _device_count()
    if n_dev < 2:
        pytest.skip(f"RCCL broadcast between GPUs needs >= 2 devices, this box has {n_dev} (the one-rank communicator and the "
                    "threaded shards are covered above; the driver's 8-GPU node runs this leg)")
    tmp, exes, nq, iwl = sharded_setup
    n = min(n_dev, 6)
    out = tmp / "sharded_multi.bin"
    r = subprocess.run([str(exes["forward_sharded"]), str(tmp / "train"), str(tmp / "test"), str(tmp / "weights"), str(iwl),
                        str(n), str(out), ",".join(str(i) for i in range(n)), "auto"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL broadcast of the quantised blob over xGMI" in r.stdout
    p1, m1, _ = read_out(tmp / "single.bin", nq)
    p2, m2, _ = read_out(out, nq)
    np.testing.assert_array_equal(p2, p1)
    assert m2 == m1


# ---- the N > 1 host logic of bench.py, executed -------------------------------------------------------------------------
def _bench(args, **env_extra):
    import json
    import os
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QMANN_BENCH_AS_RANK")}
    e.update(env_extra)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 only
    return json.loads(lines[0])


def test_committed_blob_is_what_the_library_produces(env):
    """tests/golden/trained_qa1/params_q.blob (what the CPU-only two-rank tests broadcast and vet) = the parameter blob the
    library makes today from the float matrices beside it (tools/make_params_blob.py)"""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    from make_params_blob import trained_cfg
    cfg = trained_cfg()
    hm = env.model.HostModel(cfg, env.model.load_weights(ROOT / "tests" / "golden" / "trained_qa1", cfg), device="cuda:0")
    raw = hm.params_bytes()
    want = (ROOT / "tests" / "golden" / "trained_qa1" / "params_q.blob").read_bytes()
    assert len(raw) == len(want) and raw == want
    assert env.abi.lib.qmann_params_validate(raw, len(raw), None) == 0
    hm.close()


@pytest.mark.parametrize("workload,queries", [("synth10k_d128", 192), ("babi_task1_trained", 2000)])
def test_bench_two_ranks_rehearsal_equals_the_single_rank_runs(workload, queries):
    """`bench.py --gpus 2` started as a fresh child process tree (the launcher, torch.distributed.run, two ranks), both ranks
    on this box's one GPU over gloo (QMANN_BENCH_REHEARSE=1: RCCL refuses two ranks on one device; everything but the
    transport of the one broadcast is the code an 8-GPU node runs): n_gpus = 2, the shards are disjoint and tile the
    global batch, the broadcast carried exactly the model's quantised blob, and each rank's predictions -- computed by a
    replica built from the RECEIVED bytes -- equal those of a single-rank run that built its model from the float weights."""
    common = ["--steps", "3", "--warmup", "1", "--workload", workload, "--queries", str(queries), "--no-sustained", "--no-cpu-baseline"]
    two = _bench(["--gpus", "2", *common], QMANN_BENCH_REHEARSE="1")
    assert two["n_gpus"] == 2 and two["steps"] == 3
    assert two["collective"]["world_size_seen"] == 2 and two["collective"]["backend"] == "gloo" and two["collective"]["self_launched"] is True
    B = two["config"]["queries_per_gpu"]
    assert two["shards"] == [[0, B], [B, 2 * B]]
    ones = [_bench(["--gpus", "1", *common], QMANN_BENCH_AS_RANK=str(r)) for r in range(2)]
    assert [o["shards"][0] for o in ones] == two["shards"]
    assert two["pred_crc32"] == [o["pred_crc32"][0] for o in ones]
    if workload == "synth10k_d128":
        assert two["pred_crc32"][0] != two["pred_crc32"][1]           # (different queries on the two ranks)
    # the one collective: the quantised blob, once
    pb = two["param_broadcast"]
    assert pb["bytes"] > 0 and "quantised blob" in pb["how"] and two["param_broadcast_ms"] >= 0.0
    if workload == "babi_task1_trained":
        want = (ROOT / "tests" / "golden" / "trained_qa1" / "params_q.blob").read_bytes()
        assert pb["bytes"] == len(want)                                # the committed blob's size: the same model
        assert two["accuracy"]["equals_reference_program"] is True     # rank 0's shard (the 1 000 test stories, replicated) scores the program's own error
    else:
        assert two["ranks"]["queries_per_s"]["min"] > 0
    # weak scaling: the whole-job figure counts both ranks' queries
    assert abs(two["value"] - 2 * B * 3 / (two["ms_per_step"] * 3e-3)) / two["value"] < 1e-6


def test_bench_default_line_with_two_ranks_rehearsal():
    """the WHOLE default line (headline + every BASELINE config under `configs`) with two ranks on this box's one GPU
    (QMANN_BENCH_REHEARSE=1, small batches): every workload builds its model on rank 0, broadcasts the quantised blob and runs on
    both ranks; the line keeps its shape -- what the driver's N > 1 runs will print"""
    two = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "2048", "--no-sustained"], QMANN_BENCH_REHEARSE="1")
    assert two["n_gpus"] == 2 and two["config"]["workload"] == "synth10k_d128_q25"   # config 4 as SURVEY 8(d) words it
    assert sorted(two["configs"]) == ["cfg2", "cfg3", "cfg4_q52", "cfg5", "mem50"]
    assert "1 000" in two["configs"]["cfg2"]["data"]                  # config 2 on the whole qa1 test set
    for k, c in two["configs"].items():
        assert c["value"] > 0 and c["roofline"]["frac"] > 0, k
    assert two["mem50_queries_per_s"] == two["configs"]["mem50"]["value"]
    assert two["param_broadcast"]["bytes"] > 0 and len(two["shards"]) == 2
