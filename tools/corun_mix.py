#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the hop launch of a batch split by story length, timed with its two kernels side by side (default) and
one after the other (QMANN_NO_CORUN), over mixes of short (6-row) and long (30-row) stories -- does any mix make the pair slower
than the sequence?   python tools/corun_mix.py [queries = 262144] [mode = 2]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    import torch
    from conftest import load_pkg
    load_pkg()
    import qmann_amd.model as model
    import test_gpu_quad as TQ
    dev = torch.device("cuda:0")
    cfg = TQ.cfg_of(mode)
    H, D = cfg["n_hop"], cfg["dim_emb"]
    rng = np.random.default_rng(3)
    wts = {"w_h": [rng.normal(0, 1.0, (D, D)).astype(np.float32) for _ in range(H)], "w_ans": rng.normal(0, 0.1, (40, D)).astype(np.float32)}
    net = model.QNet(cfg, wts, device="cuda:0")
    for frac_long in (0.0, 0.01, 0.05, 0.09, 0.15, 0.25, 0.35):
        n_slots = np.where(rng.random(B) < frac_long, 30, 6).astype(np.int64)
        row_off = np.concatenate([[0], np.cumsum(n_slots)]).astype(np.int32)
        R = int(row_off[-1])
        g = torch.Generator(device=dev); g.manual_seed(1)
        mags = torch.randint(0, 60, (H, R, 64), device=dev, generator=g, dtype=torch.int32)
        sign = torch.randint(0, 2, (H, R, 64), device=dev, generator=g, dtype=torch.int32) * 128
        keys = (mags | sign).to(torch.uint8).view(torch.int8); keys[:, :, D:] = 0
        vals = keys.flip(1).contiguous()
        u0 = (torch.randint(-80, 80, (B, D), device=dev, generator=g, dtype=torch.int32).float() / 4.0).contiguous()
        dro = torch.from_numpy(row_off).to(dev)
        out = {"side by side": [1e9, None], "in sequence": [1e9, None]}
        for rnd in range(3):                                           # alternating, the best of three each (the first timing of a process runs cold)
            for setting in ("in sequence", "side by side") if rnd % 2 else ("side by side", "in sequence"):
                if setting == "in sequence":
                    os.environ["QMANN_NO_CORUN"] = "1"
                else:
                    os.environ.pop("QMANN_NO_CORUN", None)
                model.abi.lib.qmann_tuning_reload()
                for _ in range(5):
                    u = net.hops(keys, vals, dro, 64, u0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    u = net.hops(keys, vals, dro, 64, u0)
                torch.cuda.synchronize()
                out[setting][0] = min(out[setting][0], (time.perf_counter() - t0) / 30 * 1e3)
                out[setting][1] = u
        assert torch.equal(out["side by side"][1], out["in sequence"][1])
        a, b = out["side by side"][0], out["in sequence"][0]
        print(f"long stories {frac_long:5.2f} (mean {n_slots.mean():5.1f} rows): side by side {a:.3f} ms, in sequence {b:.3f} ms, {100 * (a / b - 1):+.1f} %", flush=True)
    os.environ.pop("QMANN_NO_CORUN", None)


if __name__ == "__main__":
    main()
