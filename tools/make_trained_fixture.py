#!/usr/bin/env python3
"""Train bAbI task 1 with the reference's UNMODIFIED host program (oracle/_ref/MemN2N_ref = MemN2N.c + sample.c + layer.c +
common.c compiled where they lie, linked against libqmann_hip.so) on the MI355X and keep what it tested with:

  <out>/w_emb_{a,c,q}_float.bin, w_float.bin, w_lin_map_float.bin   the trained matrices, written by the library from the
        test phase (QMANN_SAVE_WEIGHTS_DIR) in the reference's own weight-file layout (MemN2N.c:2853-2978)
  <out>/reference_run.json   what the reference program printed: err(test) (result.csv), the library's verify line (the test
        phase computed verb by verb AND through the batched forward), run.sh's command line

usage (on the GPU box): python tools/make_trained_fixture.py gpurun_out/trained_qa1 [binary] [iwl]
The result is committed under tests/golden/trained_qa1/ and used by bench.py --workload babi_task1_trained and by
tests/test_gpu_words.py (data: weights and numbers, no reference text)."""
import json, os, subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
out = Path(sys.argv[1]).resolve(); out.mkdir(parents=True, exist_ok=True)
binary = sys.argv[2] if len(sys.argv) > 2 else "MemN2N_ref"
iwl = sys.argv[3] if len(sys.argv) > 3 else "5"
g = np.load(ROOT / "tests" / "golden" / "babi_qa1_en1k_sets.npz")
with tempfile.TemporaryDirectory() as td:
    d = Path(td) / "dataset" / "en_10k_parsed"           # PATH_DATA_SET of the stock define.h
    d.mkdir(parents=True)
    (d / "qa1_single-supporting-fact_train_set").write_bytes(g["train_set"].tobytes())
    (d / "qa1_single-supporting-fact_test_set").write_bytes(g["test_set"].tobytes())
    t0 = time.time()
    with open(Path(td) / "stdout.log", "w") as log:
        r = subprocess.run([str(ROOT / "oracle" / "_ref" / binary), "1", "1", "1", iwl], cwd=td, stdout=log, stderr=subprocess.STDOUT,
                           env=dict(os.environ, QMANN_DEFER="verify", QMANN_DEFER_STATS="1", QMANN_SAVE_WEIGHTS_DIR=str(out)))
    text = (Path(td) / "stdout.log").read_text(errors="replace")
    assert r.returncode == 0, text[-2000:]
    ver = [l for l in text.splitlines() if l.startswith("[qmann defer verify]") and "mode 3" in l]
    res = (Path(td) / "result.csv").read_text().strip().split(",")
    itr = [l for l in text.splitlines() if l.startswith("< ITR")]
    rec = {"binary": binary, "argv": ["1", "1", "1", iwl], "seconds": round(time.time() - t0, 1), "err_test_result_csv": float(res[10]),
           "verify_line": ver[-1] if ver else None, "train_error_first_epoch": float(itr[0].split("error:")[1].split(",")[0]),
           "train_error_last_epoch": float(itr[-1].split("error:")[1].split(",")[0]), "epochs": len(itr),
           "data": "tests/golden/babi_qa1_en1k_sets.npz (bAbI en/qa1 1k train + test record files)"}
    for f in list(out.glob("*_fixed.bin")) + [out / "qmann_params.bin"]:
        if f.exists():
            f.unlink()                                   # derivable from the float files; the fixture keeps those only
    (out / "reference_run.json").write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))
