#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/<tag>_{stats,pmc_fetch,pmc_sq}) into the small summaries
kept under profiles/, and record the PMC HBM traffic of the dominant kernel in profiles/traffic.json
(read by bench.py for roofline.traffic).  usage: summarize_profiles.py <tag> <workload> <bytes_per_launch>"""
import csv, glob, json, sys, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag, workload, alg = sys.argv[1], sys.argv[2], float(sys.argv[3])
src = ROOT / "gpurun_out" / "prof"
out = ROOT / "profiles"; out.mkdir(exist_ok=True)
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_hops"
import importlib.util
_spec = importlib.util.spec_from_file_location("qmann_pkg_init", ROOT / "q-mann_amd" / "__init__.py")
_pkg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_pkg)
SRC_SHA = _pkg.kernel_sources_sha16()      # the sources the profiled library was built from (run this before editing them)

st = glob.glob(str(src / f"{tag}_stats" / "*" / "*kernel_stats.csv"))
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(out / f"{tag}_kernel_stats_{workload}.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload {workload} --steps 30 --no-cpu-baseline --no-secondary --no-sustained   (MI355X)\n")
        f.write("# kernel names cut to 100 chars; torch kernels are the synthetic-data generation\n")
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows:
        if kernel in r["Name"]:
            print("stats:", r["Name"][:60], "calls", r["Calls"], "avg_ms", float(r["AverageNs"]) / 1e6)

traffic = None
pf = glob.glob(str(src / f"{tag}_pmc_fetch" / "*" / "*counter_collection.csv"))
if pf:
    recs = [r for r in csv.DictReader(open(pf[0])) if kernel in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    v = [float(r["Counter_Value"]) for r in recs]
    import re
    kname = re.search(r"k_hops_\w+(<[^>]*>)?", recs[0]["Kernel_Name"]).group(0)
    m = sum(v) / len(v)
    traffic = 2 * m * 1024
    with open(out / f"{tag}_pmc_fetch_{workload}.txt", "w") as f:
        f.write(f"rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --workload {workload} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained (MI355X)\n")
        f.write(f"dominant kernel: {kname}; {len(v)} dispatches\n")
        f.write(f"FETCH_SIZE per dispatch (KiB, raw): {[round(x, 1) for x in v]}\n")
        f.write(f"mean raw = {m:.1f} KiB = {m * 1024 / 1e9:.3f} GB\n")
        f.write("gfx950 correction (MI355X_MICROARCH.md, HBM section): a 16 B/lane stream's 128-B requests are tallied at 64 B -> x2\n")
        f.write(f"corrected HBM read traffic per launch = {traffic / 1e9:.3f} GB\n")
        f.write(f"algorithmic bytes per launch = {alg / 1e9:.3f} GB ; traffic / algorithmic = {traffic / alg:.4f}\n")
    print(open(out / f"{tag}_pmc_fetch_{workload}.txt").read())
    tj = out / "traffic.json"
    d = json.loads(tj.read_text()) if tj.exists() else {}
    d[workload] = {"traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "source": f"profiles/{tag}_pmc_fetch_{workload}.txt",
                   "method": "rocprofv3 --pmc FETCH_SIZE (own pass), KiB x 1024 x 2 (gfx950 wide-stream correction)",
                   "kernel_sources_sha16": SRC_SHA}
    tj.write_text(json.dumps(d, indent=1) + "\n")

sq = glob.glob(str(src / f"{tag}_pmc_sq" / "*" / "*counter_collection.csv"))
if sq:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sq[0])):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out / f"{tag}_pmc_sq_{workload}.txt", "w") as f:
        f.write(f"rocprofv3 --pmc <SQ counters> --output-format csv -- python3 bench.py --workload {workload} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained (MI355X)\n")
        f.write("the k_hops_* kernel, mean per dispatch (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles)\n")
        for k, v in sorted(agg.items()):
            f.write(f"{k:24s} {sum(v) / len(v):16.0f}\n")
        if "SQ_INSTS_VALU" in agg:
            iv = sum(agg["SQ_INSTS_VALU"]) / len(agg["SQ_INSTS_VALU"])
            f.write(f"VALU lane-ops per algorithmic byte = {iv * 64 / alg:.2f}\n")
        if "SQ_WAVE_CYCLES" in agg:
            wc = sum(agg["SQ_WAVE_CYCLES"]) / len(agg["SQ_WAVE_CYCLES"])
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in agg: f.write(f"{k} / SQ_WAVE_CYCLES = {sum(agg[k]) / len(agg[k]) / wc:.3f}\n")
    print(open(out / f"{tag}_pmc_sq_{workload}.txt").read())

mf = glob.glob(str(src / f"{tag}_pmc_mfma" / "*" / "*counter_collection.csv"))
if mf:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(mf[0])):
        if "k_answer_i8_part" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    st_rows = [r for r in rows if "k_answer_i8_part" in r["Name"]] if st else []
    if agg and st_rows:
        avg_ns = float(st_rows[0]["AverageNs"])
        busy = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"])
        insts = sum(agg["SQ_INSTS_VALU_MFMA_I8"]) / len(agg["SQ_INSTS_VALU_MFMA_I8"])
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the chip's 1 024 SIMDs; the kernel's cycles at the clock SQ_BUSY_CYCLES implies
        sq_busy = sum(agg["SQ_BUSY_CYCLES"]) / len(agg["SQ_BUSY_CYCLES"])
        frac = busy / 1024.0 / (avg_ns * 1e-9 * 2.1e9)
        with open(out / f"{tag}_pmc_mfma_{workload}.txt", "w") as f:
            f.write(f"rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -- python3 bench.py --workload {workload} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained (MI355X)\n")
            f.write("k_answer_i8_part (int8 MFMA projection + softmax statistics + arg-max), mean per dispatch\n")
            for k, v in sorted(agg.items()):
                f.write(f"{k:28s} {sum(v) / len(v):16.0f}\n")
            f.write(f"v_mfma_i32_16x16x64_i8 issued = {insts:.0f} = {insts * 2 * 16 * 16 * 64 / 1e9:.2f} GOP\n")
            f.write(f"kernel average (kernel trace of the same command) = {avg_ns / 1e3:.1f} us\n")
            f.write(f"matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (kernel time x 2.1 GHz) = {frac:.4f}\n")
            f.write(f"SQ_BUSY_CYCLES (per XCD-SE aggregate, for the clock estimate) = {sq_busy:.0f}\n")
        print(open(out / f"{tag}_pmc_mfma_{workload}.txt").read())
        mj = out / "mfma.json"
        d = json.loads(mj.read_text()) if mj.exists() else {}
        d[workload] = {"mfma_busy_frac": frac, "mfma_insts": insts, "kernel_us": avg_ns / 1e3, "source": f"profiles/{tag}_pmc_mfma_{workload}.txt",
                       "kernel_sources_sha16": SRC_SHA}
        mj.write_text(json.dumps(d, indent=1) + "\n")
