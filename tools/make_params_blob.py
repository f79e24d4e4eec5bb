#!/usr/bin/env python3
"""The QUANTISED parameter blob (include/qmann_model.h: qmann_model_params) of the trained bAbI task-1 model of
tests/golden/trained_qa1/, written next to the float matrices it is made from.  Needs the GPU (the library quantises on
the device).  usage (on the GPU box): python tools/make_params_blob.py gpurun_out/params_q.blob
The committed copy (tests/golden/trained_qa1/params_q.blob) is what the CPU-only two-rank tests broadcast and vet with
qmann_params_validate; tests/test_gpu_dist.py checks that the library still produces these very bytes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_pkg
load_pkg()
import qmann_amd.model as model


def trained_cfg():
    import json
    rec = json.loads((ROOT / "tests" / "golden" / "trained_qa1" / "reference_run.json").read_text())
    return model.babi_cfg(30, attention_mode=2, softmax_base=0, iwl=int(rec["argv"][3]), n_hop=3, D=60, en_mq=True)


if __name__ == "__main__":
    cfg = trained_cfg()
    hm = model.HostModel(cfg, model.load_weights(ROOT / "tests" / "golden" / "trained_qa1", cfg), device="cuda:0")
    raw = hm.params_bytes()
    Path(sys.argv[1]).write_bytes(raw)
    print(len(raw), "bytes")
