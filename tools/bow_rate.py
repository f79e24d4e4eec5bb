#!/usr/bin/env python3
"""Runs ON THE GPU BOX: qmann_bow_to_words alone on synthetic bag-of-words rows (the reference's float pools), rows/s and the
bytes it reads per second.   python tools/bow_rate.py [rows = 2000000] [dim_input = 238]"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 238
    import torch
    from conftest import load_pkg
    load_pkg()
    import qmann_amd.abi as abi
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    bow = (torch.rand((R, V), device=dev, generator=g) < (6.0 / V)).float()          # ~6 words per row
    bow[:, V - 1] = 1.0                                                              # a time entry
    words = torch.zeros((R, 16), dtype=torch.int16, device=dev)
    irr = torch.zeros(R, dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        n.zero_()
        assert abi.lib.qmann_bow_to_words(p(bow), R, V, p(words), p(irr), p(n), None) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        abi.lib.qmann_bow_to_words(p(bow), R, V, p(words), p(irr), p(n), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"V = {V}: {R} rows in {dt * 1e3:.3f} ms = {R / dt / 1e9:.2f} G rows/s, {R * V * 4 / dt / 1e12:.2f} TB/s of float rows read; "
          f"irregular rows {int(n.item()) // 13}", flush=True)


if __name__ == "__main__":
    main()
