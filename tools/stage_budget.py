#!/usr/bin/env python3
"""Per-stage instruction budget of a hop kernel, from the compiler's assembly (no GPU).

    tools/stage_budget.py <file.hip> <kernel-name-regex> [-D...]

Compiles csrc/<file.hip> for gfx950 with -DQM_STAGE_MARKS (hops_common.h: every stage starts with an assembly comment and a
scheduling barrier) and counts, per stage, the vector-ALU, LDS, vector-memory and scalar instructions in the text of the
matching kernel.  STATIC counts of the code between two marks: a branch-guarded part (the scan passes a short story skips, the
fetch rounds beyond the first) is counted once whether it runs or not; wavefront-uniform alternatives (e^x from the table or from
v_exp_f32) are both in the text.  The dynamic totals are the SQ counters' (profiles/*_units_*.txt)."""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
src, rx = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
with tempfile.TemporaryDirectory() as td:
    out = Path(td) / "k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off", "-DQM_STAGE_MARKS",
                    *extra, "-I", str(ROOT / "include"), "--cuda-device-only", "-S", str(ROOT / "q-mann_amd" / "csrc" / src), "-o", str(out)], check=True)
    text = out.read_text()
names = [m.group(1) for m in re.finditer(r"^(_Z\S+):\s*(?:;.*)?$", text, re.M)]
for name in names:
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not re.search(rx, dem):
        continue
    i = text.index(name + ":")
    body = text[i:text.index(".amdhsa_kernel", i)].split("\n")
    stages, cur = collections.OrderedDict(), "(before the first mark / outside the hop)"
    for ln in body:
        t = ln.strip()
        m = re.match(r"; QM_MARK (.*)", t)
        if m:
            cur = m.group(1).strip()
            continue
        op = t.split()[0] if t and not t.startswith((";", ".")) and not t.endswith(":") else None
        if not op:
            continue
        c = stages.setdefault(cur, collections.Counter())
        kind = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_", "flat_"))
                else "salu" if op.startswith("s_") else "other")
        c[kind] += 1
        if op.startswith("v_") and ("readlane" in op or "writelane" in op):
            c["lane<->sgpr"] += 1
        if op in ("s_waitcnt", "s_nop", "s_barrier"):
            c["waits"] += 1
    print(dem[:140])
    tot = collections.Counter()
    print(f"  {'stage':48s} {'VALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'SALU':>6s}  (readlane/writelane, waits/nops among them)")
    for st, c in stages.items():
        print(f"  {st:48s} {c['valu']:6d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:6d}  ({c['lane<->sgpr']}, {c['waits']})")
        tot.update(c)
    print(f"  {'whole kernel text':48s} {tot['valu']:6d} {tot['lds']:5d} {tot['vmem']:5d} {tot['salu']:6d}  ({tot['lane<->sgpr']}, {tot['waits']})")
