"""Runs ON THE GPU BOX: times qmann_model_forward_words on long memories with Hamming V0/V1 attention, tied and untied
hops, to check model_host.hip's use_planes() choice (packed planes only where a plane is scanned by more than one hop).
    QMANN_LIB_PATH=<lib> python3 tools/planes_ab.py [S] [B]
Prints one line per (mode, num_bit, tied): ms per batch.  Same predictions whichever form is taken (tests/test_gpu_words.py)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg  # noqa: E402

load_pkg()
import qmann_amd.model as model  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
V, D, H, NW = 256, 250, 3, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)


def weights(tied):
    def m():
        return rng.normal(0, 1.0, (D, V)).astype(np.float32)
    a0, c0 = m(), m()
    return {"w_q": m(), "w_a": [a0 if tied else m() for _ in range(H)], "w_c": [c0 if tied else m() for _ in range(H)],
            "w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)], "w_ans": rng.normal(0, 0.1, (V, D)).astype(np.float32)}


sw = torch.from_numpy(rng.integers(0, V, (B * S, NW)).astype(np.uint16).view(np.int16)).to(dev)
qw = torch.from_numpy(rng.integers(0, V, (B, NW)).astype(np.uint16).view(np.int16)).to(dev)
row_off = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
for mode, nb in ((10, 1), (10, 4), (11, 4)):
    for tied in (False, True):
        cfg = model.babi_cfg(V, attention_mode=mode, n_hop=H, D=D, en_mq=False)
        cfg["num_bit"] = nb
        hm = model.HostModel(cfg, weights(tied), dev)
        for _ in range(2):
            hm.forward_words(sw, qw, row_off, S)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            pred, _, _ = hm.forward_words(sw, qw, row_off, S)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / n * 1e3
        print("mode %d num_bit %d tied %d  S %d B %d : %.3f ms  (pred sum %d)" % (mode, nb, tied, S, B, ms, int(pred.sum())), flush=True)
        hm.close()
