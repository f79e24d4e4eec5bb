"""Runs ON THE GPU BOX: the unmodified reference host (oracle/_ref/<binary>) N times -- its training is seeded by the clock
(MemN2N/sample.c:111) -- and prints first / last training error and the test error of every run.  Sizes the thresholds of
tests/test_gpu_ref_host.py::test_unmodified_reference_program_trains_and_tests.
    python3 tools/train_spread.py MemN2N_ref_mode3 8"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
binary, n = sys.argv[1], int(sys.argv[2])
g = np.load(ROOT / "tests" / "golden" / "babi_qa1_en1k_sets.npz")
for i in range(n):
    with tempfile.TemporaryDirectory() as t:
        d = Path(t) / "dataset" / "en_10k_parsed"
        d.mkdir(parents=True)
        (d / "qa1_single-supporting-fact_train_set").write_bytes(g["train_set"].tobytes())
        (d / "qa1_single-supporting-fact_test_set").write_bytes(g["test_set"].tobytes())
        r = subprocess.run([str(ROOT / "oracle" / "_ref" / binary), "1", "1", "1", "5"], cwd=t, capture_output=True, text=True, timeout=600)
        itr = [l for l in r.stdout.splitlines() if l.startswith("< ITR")]
        err = [float(l.split("error:")[1].split(",")[0]) for l in itr]
        res = (Path(t) / "result.csv").read_text().strip().split(",")
        print(binary, "run", i, "rc", r.returncode, "train err first %.3f min %.3f last %.3f" % (err[0], min(err), err[-1]), "test err", res[10], flush=True)
