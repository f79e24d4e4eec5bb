"""The headers are C, and a plain C host can drive the library: compile checks without a GPU, and the
example host (examples/forward_words.c: weight files -> qmann_model -> one call per batch) on the GPU."""
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg

INC = ["-I", str(ROOT / "include")]
LINK = ["-L", str(ROOT / "q-mann_amd" / "lib"), "-lqmann_hip", "-L/opt/rocm/lib", "-lamdhip64",
        f"-Wl,-rpath,{ROOT / 'q-mann_amd' / 'lib'}", "-Wl,-rpath,/opt/rocm/lib", "-lm"]


def test_headers_are_strict_c99(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "qmann_abi.h"\n#include "qmann_batch.h"\n#include "qmann_weights.h"\n#include "qmann_model.h"\n#include "qmann_dataset.h"\n#include "qmann_dist.h"\n'
                   "int main(void) { qmann_net n = {0}; qmann_weights w = {0}; (void)n; (void)w; return (int)sizeof(qmann_taps) * 0; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", *INC, "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def build_example(tmp_path):
    load_pkg()                                               # (the library must exist)
    exe = tmp_path / "forward_words"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", *INC, "-I/opt/rocm/include", str(ROOT / "examples" / "forward_words.c"),
                        *LINK, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_example_host_compiles_and_links(tmp_path):
    exe = build_example(tmp_path)
    und = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    used = sorted({l.split()[-1] for l in und.splitlines() if " U qmann_" in l})
    assert used == ["qmann_model_create", "qmann_model_destroy", "qmann_model_forward_words", "qmann_weights_load"]


@pytest.mark.gpu
def test_example_host_runs_and_matches_the_oracle(tmp_path, gold, oracle):
    load_pkg()
    import qmann_amd.model as model
    exe = build_example(tmp_path)
    b = gold("babi_qa1_test64.npz")
    V, dd, D, H, iwl = int(b["dim_input"]), int(b["dim_dict"]), 60, 3, 2
    cfg = model.babi_cfg(V, 2, 0, iwl=iwl)
    rng = np.random.default_rng(77)
    wts = {"w_q": rng.normal(0, 0.8, (D, V)).astype(np.float32), "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32),
           "w_a": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_c": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)]}
    wdir = tmp_path / "weights"; wdir.mkdir()
    model.save_weights(wdir, wts, cfg, fixed=False)
    story, ques = b["story"].astype(np.float32), b["question"].astype(np.float32)
    n_sen = b["n_sen"].astype(np.int64); ans = b["answer"].argmax(1).astype(np.uint32)

    def words(bow, nd, width, with_time):
        out = np.full((bow.shape[0], width), 0xFFFF, np.uint16)
        for r, row in enumerate(bow):
            ent = [k for k in np.flatnonzero(row[:nd]) for _ in range(int(row[k]))]
            if with_time:
                ent.append(nd + int(np.flatnonzero(row[nd:])[0]))
            out[r, :len(ent)] = ent
        return out
    sw, qw = words(story, dd, 12, True), words(ques, V, 8, False)
    row_off = np.concatenate([[0], np.cumsum(n_sen)]).astype(np.uint32)
    nq = len(n_sen)
    with open(tmp_path / "batch.bin", "wb") as f:
        f.write(struct.pack("<9I", V, D, H, iwl, nq, sw.shape[0], 12, 8, int(n_sen.max())))
        f.write(row_off.tobytes()); f.write(sw.tobytes()); f.write(qw.tobytes()); f.write(ans.tobytes())
    r = subprocess.run([str(exe), str(wdir), str(tmp_path / "batch.bin"), str(tmp_path / "pred.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = (tmp_path / "pred.bin").read_bytes()
    pred = np.frombuffer(raw[:4 * nq], np.uint32)
    match = struct.unpack("<I", raw[4 * nq:4 * nq + 4])[0]
    cost = struct.unpack("<f", raw[4 * nq + 4:])[0]
    m = oracle.make_model(cfg, wts)
    o, want_cost, checked = 0, 0.0, 0
    for q in range(nq):
        ns = int(n_sen[q])
        opred, t = oracle.forward(m, story[o:o + ns], ques[q], taps=("out_probs",))
        o += ns
        top2 = np.sort(t["out_probs"])[-2:]
        if top2[1] - top2[0] > 1e-6:
            assert int(pred[q]) == opred, q
            checked += 1
        want_cost -= float(t["out_probs"][ans[q]])
    assert checked >= nq - 2
    assert match == int((pred == ans).sum())
    assert cost == pytest.approx(want_cost, rel=1e-4)


def write_set(path, records):
    out = ["", "+NS+", str(len(records)), ""]
    for i, (sens, q, a) in enumerate(records):
        out += ["+I+", str(i), "+S+", str(len(sens))] + [x + " " for x in sens] + ["+Q+", q + " ", "+A+", a, ""]
    path.write_text("\n".join(out) + "\n")


def test_dataset_example_compiles(tmp_path):
    load_pkg()
    exe = tmp_path / "forward_dataset"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", *INC, "-I/opt/rocm/include", str(ROOT / "examples" / "forward_dataset.c"),
                        *LINK, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sharded_example_compiles_and_uses_the_dist_entry_points(tmp_path):
    """examples/forward_sharded.c (one host thread per shard, RCCL broadcast of the quantised blob) is plain C99 + pthreads;
    it runs on the GPU in tests/test_gpu_dist.py"""
    load_pkg()
    exe = tmp_path / "forward_sharded"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", *INC, "-I/opt/rocm/include", str(ROOT / "examples" / "forward_sharded.c"),
                        *LINK, "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    und = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    used = {l.split()[-1] for l in und.splitlines() if " U qmann_" in l}
    assert {"qmann_model_create_on", "qmann_model_create_from_params", "qmann_model_params", "qmann_comm_get_id", "qmann_comm_init_rank",
            "qmann_comm_broadcast_params", "qmann_shard_range", "qmann_model_forward_words", "qmann_dataset_load"} <= used
    assert not [l for l in und.splitlines() if " U nccl" in l]        # RCCL is reached through the library only


@pytest.mark.gpu
def test_dataset_example_from_record_files_to_predictions(tmp_path, oracle):
    """examples/forward_dataset.c: record files + weight files in, predictions out -- equal to the Python plumbing on the
    same inputs (which the other tests tie to the oracle)"""
    load_pkg()
    import torch
    import qmann_amd.abi as abi
    import qmann_amd.model as model
    exe = tmp_path / "forward_dataset"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", *INC, "-I/opt/rocm/include", str(ROOT / "examples" / "forward_dataset.c"),
                        *LINK, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(20)]
    recs = lambda n: [([" ".join(rng.choice(vocab, rng.integers(1, 6))) for _ in range(rng.integers(1, 9))],
                       " ".join(rng.choice(vocab, 2)), str(rng.choice(vocab))) for _ in range(n)]
    write_set(tmp_path / "train", recs(200)); write_set(tmp_path / "test", recs(90))
    ds = abi.load_dataset(tmp_path / "train", tmp_path / "test", 50)
    V, D, H, iwl = ds["dim_input"], 60, 3, 5
    cfg = model.babi_cfg(V, 2, 0, iwl=iwl)
    wts = {"w_q": rng.normal(0, 0.8, (D, V)).astype(np.float32), "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32),
           "w_a": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_c": [rng.normal(0, 0.8, (D, V)).astype(np.float32) for _ in range(H)],
           "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)]}
    wdir = tmp_path / "weights"; wdir.mkdir()
    model.save_weights(wdir, wts, cfg, fixed=False)
    r = subprocess.run([str(exe), str(tmp_path / "train"), str(tmp_path / "test"), str(wdir), str(iwl), str(tmp_path / "pred.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    nq = ds["n_query"]
    raw = (tmp_path / "pred.bin").read_bytes()
    pred = np.frombuffer(raw[:4 * nq], np.uint32)
    match = struct.unpack("<I", raw[4 * nq:4 * nq + 4])[0]
    dev = torch.device("cuda:0")
    hm = model.HostModel(cfg, wts)
    n_sen = np.diff(ds["row_off"].astype(np.int64))
    p2, _, m2 = hm.forward_words(torch.from_numpy(ds["story_words"].view(np.int16)).to(dev),
                                 torch.from_numpy(ds["question_words"].view(np.int16)).to(dev),
                                 torch.from_numpy(ds["row_off"].astype(np.int32)).to(dev), int(n_sen.max()),
                                 torch.from_numpy(ds["answer"].astype(np.int64).astype(np.int32)).to(dev))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pred, p2.cpu().numpy().astype(np.uint32))
    assert match == int(m2.item())
    hm.close()
