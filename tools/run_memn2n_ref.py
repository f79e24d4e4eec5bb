#!/usr/bin/env python3
"""Run the reference's unmodified host program (oracle/_ref/MemN2N_ref = MemN2N.c + sample.c + layer.c +
common.c compiled in place, linked against libqmann_hip.so) on bAbI task 1: train, validate, test.
usage: run_memn2n_ref.py [loops] [iwl]   -- the program's own argv is  <loops> <task_s> <task_e> <iwl>"""
import subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
loops = sys.argv[1] if len(sys.argv) > 1 else "1"
iwl = sys.argv[2] if len(sys.argv) > 2 else "5"
g = np.load(ROOT / "tests" / "golden" / "babi_qa1_en1k_sets.npz")
with tempfile.TemporaryDirectory() as td:
    d = Path(td) / "dataset" / "en_10k_parsed"           # PATH_DATA_SET of the stock define.h
    d.mkdir(parents=True)
    (d / "qa1_single-supporting-fact_train_set").write_bytes(g["train_set"].tobytes())
    (d / "qa1_single-supporting-fact_test_set").write_bytes(g["test_set"].tobytes())
    t0 = time.time()
    with open(Path(td) / "stdout.log", "w") as out:
        r = subprocess.run([str(ROOT / "oracle" / "_ref" / "MemN2N_ref"), loops, "1", "1", iwl], cwd=td, stdout=out,
                           stderr=subprocess.STDOUT)
    dt = time.time() - t0
    lines = (Path(td) / "stdout.log").read_text(errors="replace").splitlines()
    print("exit", r.returncode, "seconds", round(dt, 1), "lines", len(lines))
    keep = [l for l in lines if any(k in l.lower() for k in ("err", "epoch", "test", "match", "*e"))]
    print("\n".join(keep[:15]))
    print("...")
    print("\n".join(lines[-40:]))
    for f in ("result.csv", "result_all.csv"):
        p = Path(td) / f
        if p.exists():
            print(f"--- {f}\n" + p.read_text()[-1500:])
