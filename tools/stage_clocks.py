#!/usr/bin/env python3
"""Runs ON THE GPU BOX with a -DQM_STAGE_CLOCKS build of the library (QMANN_LIB_PATH=q-mann_amd/lib_exp/libqmann_clk.so, made by
`tools/build_variant.sh clk '1i #define QM_STAGE_CLOCKS 1' hops_common.h`): where does a wavefront's time go in the instrumented
kernel?  Runs a bench workload's forward a few times and prints each stage's share of the wavefronts' summed shader cycles.
usage: QMANN_LIB_PATH=... python tools/stage_clocks.py <workload> [steps]"""
import ctypes as C, json, subprocess, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
wl = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
import torch
bench.torch = torch
bench.load_pkg()
import qmann_amd.abi as abi
fn = abi.lib.qmann_debug_stage_clocks
fn.restype = C.c_int; fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
args = bench.parse_args(["--workload", wl, "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-sustained"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
fn(buf, 16)                                                  # clear
out = bench.run_workload(args, wl, dev, 0, 1)
n = fn(buf, 16)
tot = sum(buf[i] for i in range(n))
print(f"{wl}: {out['value'] / 1e6:.1f} M q/s with the instrumented build; summed wavefront cycles {tot:.3e}")
for i in range(n):
    if buf[i]:
        print(f"  stage {i:2d}  {buf[i] / tot:6.3f}  {buf[i]:.3e}")
