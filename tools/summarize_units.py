#!/usr/bin/env python3
"""gpurun_out/units/<tag>/pass*.csv (tools/pmc_units.sh) -> profiles/<round>_units_<tag>.txt: per-dispatch means of the SQ / TCP
counters of the k_hops_* kernel and the ratios that say which unit is saturated.
usage: tools/summarize_units.py <round> <tag> <workload> [kernel_ms] [--kernel SUBSTR --out NAME]
(--kernel: keep only the dispatches whose kernel name contains SUBSTR -- a pass file may hold several kernels of a whole
forward; --out: the summary goes to profiles/<round>_units_<NAME>.txt)"""
import csv, sys, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
argv = sys.argv[1:]
ksub = outname = None
if "--kernel" in argv:
    i = argv.index("--kernel"); ksub = argv[i + 1]; del argv[i:i + 2]
if "--out" in argv:
    i = argv.index("--out"); outname = argv[i + 1]; del argv[i:i + 2]
rnd, tag, wl = argv[0:3]
ms = float(argv[3]) if len(argv) > 3 else None
acc, cnt, kern = collections.defaultdict(float), collections.Counter(), None
dur = {}                                            # pass file -> durations of its dispatches (ns), from the rows' timestamps
for f in sorted((ROOT / "gpurun_out" / "units" / tag).glob("pass*.csv")):
    for r in csv.DictReader(open(f)):
        if ksub and ksub not in r["Kernel_Name"]:
            continue
        kern = r["Kernel_Name"][:110]
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        dur.setdefault(f.name, {})[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ms_trace = ms
if "pass1.csv" in dur:                              # the time of the very dispatches GRBM_GUI_ACTIVE was counted on
    ms = sum(dur["pass1.csv"].values()) / len(dur["pass1.csv"]) * 1e-6
m = {k: acc[k] / cnt[k] for k in acc}
sys.path.insert(0, str(ROOT / "tests")); from conftest import load_pkg
sha = load_pkg().kernel_sources_sha16()
out = [f"rocprofv3 --pmc <8 counters per pass, 4 passes> -- python3 bench.py --workload {wl} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained (MI355X)",
       f"kernel: {kern}; kernel sources {sha}; mean per dispatch (SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves)"]
for k in sorted(m):
    out.append(f"{k:32s} {m[k]:16.0f}")
g = lambda k: m.get(k, float('nan'))
wc = g("SQ_WAVE_CYCLES")
out.append("")
out.append("per wave-cycle (each wave's residency split three ways; the three sum to ~1):")
for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
    out.append(f"  {k:24s} / SQ_WAVE_CYCLES = {g(k) / wc:.3f}")
out.append("issuing cycles by unit, per wave-cycle (a wave issuing to a unit; several waves of a SIMD can be counted in the same cycle):")
for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
    out.append(f"  {k:24s} / SQ_WAVE_CYCLES = {g(k) / wc:.3f}")
# unit occupancy per SIMD: SQ_BUSY_CYCLES counts quad-cycles per SE?  use waves: mean resident waves per SIMD = wave-cycles / (busy time x SIMDs)
if ms:
    clk = g("GRBM_GUI_ACTIVE") / 8.0 / (ms * 1e-3)
    simd_cycles = clk * ms * 1e-3 * 1024
    out.append("")
    out.append(f"dispatch time {ms:.4f} ms (timestamps of the counted dispatches; profiled passes run slower than plain ones"
               + (f": kernel trace {ms_trace:.4f} ms" if ms_trace else "") + f"); effective clock GRBM_GUI_ACTIVE / 8 / t = {clk / 1e9:.2f} GHz")
    out.append(f"SIMD-cycles available = clock x t x 1024 SIMDs = {simd_cycles:.3e}")
    out.append(f"mean resident waves per SIMD = 4 x SQ_WAVE_CYCLES / SIMD-cycles = {4 * wc / simd_cycles:.2f}")
    out.append(f"VALU issue occupancy  = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles = {4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles:.3f}   (1.0 = the SIMD's vector issue port never idle)")
    out.append(f"VALU instructions x 4 cycles / SIMD-cycles                  = {4 * g('SQ_INSTS_VALU') / simd_cycles:.3f}")
    if g("SQ_WAVES") <= 4 * 1024 * 2:           # a persistent grid: its waves live (nearly) as long as the kernel, so their mean lifetime is a second clock
        life = 4 * wc / g("SQ_WAVES")
        out.append(f"  cross-check without GRBM_GUI_ACTIVE (it reads high on sub-millisecond dispatches): a wave lives {life:.3e} cycles on average = "
                   f"{life / (clk * ms * 1e-3):.2f} of the kernel's cycles as counted above; with the wave lifetime as the kernel's length the VALU issue "
                   f"occupancy is SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD = {g('SQ_ACTIVE_INST_VALU') / wc * g('SQ_WAVES') / 1024:.3f}")
    out.append(f"LDS issue occupancy   = 4 x SQ_ACTIVE_INST_LDS / SIMD-cycles  = {4 * g('SQ_ACTIVE_INST_LDS') / simd_cycles:.3f}")
    out.append(f"VMEM issue occupancy  = 4 x SQ_ACTIVE_INST_VMEM / SIMD-cycles = {4 * g('SQ_ACTIVE_INST_VMEM') / simd_cycles:.3f}")
    out.append(f"scalar issue occupancy= 4 x SQ_ACTIVE_INST_SCA / SIMD-cycles  = {4 * g('SQ_ACTIVE_INST_SCA') / simd_cycles:.3f}")
out.append(f"LDS bank-conflict cycles / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f}")
out.append(f"VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU x 4 / 4) = {g('SQ_THREAD_CYCLES_VALU') / max(64 * g('SQ_ACTIVE_INST_VALU'), 1):.3f}")
if "TCP_TCC_READ_REQ" in m:
    out.append(f"mean L2 read latency seen by the L1 = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = {g('TCP_TCC_READ_REQ_LATENCY') / max(g('TCP_TCC_READ_REQ'), 1):.0f} cycles")
    out.append(f"TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1 = {g('TCP_PENDING_STALL_CYCLES') / max(g('TCP_GATE_EN1'), 1):.3f}")
if "FETCH_SIZE" in m:
    out.append(f"HBM traffic per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; x2 gfx950 correction on FETCH_SIZE as the guide prescribes): "
               f"read {g('FETCH_SIZE') * 1024 * 2 / 1e6:.1f} MB, written {g('WRITE_SIZE') * 1024 / 1e6:.1f} MB (uncorrected)")
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and ms:
    out.append(f"matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (dispatch time x 2.1 GHz) = {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / (ms * 1e-3 * 2.1e9):.3f}; "
               f"MFMA instructions {g('SQ_INSTS_VALU_MFMA_I8'):.0f}")
p = ROOT / "profiles" / f"{rnd}_units_{outname or tag}.txt"
p.write_text("\n".join(out) + "\n")
print(p)
