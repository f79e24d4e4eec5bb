#!/usr/bin/env python3
"""Interleaved A/B timing of compiled tuning variants of the D=128 key scan (one process, one device;
cdna_hip_programming.md rule 24).  Development tool, not part of the product or of bench.py."""
import ctypes as C, sys, json
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
bench.load_pkg()
import qmann_amd.model as model, qmann_amd.abi as abi
abi.lib.qmann_debug_set_tune.argtypes = [C.c_int]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5, 6, 7]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
S, D, V, H = 10000, 128, 256, 3
dev = torch.device("cuda:0")
cfg = model.babi_cfg(V, 2, 0, iwl=5, n_hop=H, D=D, en_mq=False)
net = model.QNet(cfg, bench.make_params(cfg, D, V, 1), device="cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
keys = bench.gauss_i8((H, B * S, 128), 3.5, gen, dev); vals = bench.gauss_i8((H, B * S, 128), 30.0, gen, dev)
u0 = (torch.randn((B, D), device=dev, generator=gen) * 3.5).round_().clamp_(-127, 127) / 4.0
row_off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
u_out = torch.empty_like(u0)
ref = None
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        abi.lib.qmann_debug_set_tune(v)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); net.hops(keys, vals, row_off, S, u0, u_out=u_out); b.record(); torch.cuda.synchronize()
        if r == 0:
            if ref is None: ref = u_out.clone()
            assert torch.equal(ref, u_out), f"variant {v} changes results"
        else:
            res[v].append(a.elapsed_time(b))
gb = B * H * S * 128 / 1e9
for v in variants:
    t = np.array(res[v]); print(f"variant {v}: median {np.median(t):.3f} ms  min {t.min():.3f}  -> {gb/np.median(t)*1e3:.0f} GB/s (min-time {gb/t.min()*1e3:.0f})", flush=True)
