#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 passes for one bench workload, CSV output under gpurun_out/prof/.
#   tools/collect_profiles.sh <tag> <workload> [fetch] [sq] [mfma]
# kernel trace + stats always; "fetch" adds a FETCH_SIZE counter pass, "sq" an SQ instruction-mix pass
# (counters are collected in their own runs, never together with tracing).
set -e
tag=$1; wl=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
export QMANN_BENCH_NO_SMALL_BATCH=1     # (the 64-story serving leg of babi_task1_idx would mix 1 200 tiny launches into the kernel averages)
# only the small summaries travel back (gpurun merges at most 64 MiB): per-dispatch traces are dropped and
# counter files are cut down to our kernels' rows
prune() { find "$1" -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete; 
          for c in $(find "$1" -name '*counter_collection.csv'); do { head -1 "$c"; grep -E 'k_hops|k_answer|k_embed|k_logits|k_fwd' "$c" || true; } > "$c.tmp"; mv "$c.tmp" "$c"; done; }
# 30 timed steps (+ 3 warm-up launches, which the statistics include): the first ~5 launches of a VALU-heavy kernel run up to 15 %
# slower than the rest (tools/trace_seq.sh shows them one by one); 13 launches would average mostly that transient
rocprofv3 --kernel-trace --stats --output-format csv -d $P/${tag}_stats -- python3 $R/bench.py --workload $wl --steps 30 --no-cpu-baseline --no-secondary --no-sustained > $P/${tag}_stats.log 2>&1
prune $P/${tag}_stats
for pass in "$@"; do
  case $pass in
    fetch) rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/${tag}_pmc_fetch -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > $P/${tag}_pmc_fetch.log 2>&1; prune $P/${tag}_pmc_fetch ;;
    mfma) rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $P/${tag}_pmc_mfma -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > $P/${tag}_pmc_mfma.log 2>&1; prune $P/${tag}_pmc_mfma ;;
    sq) rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $P/${tag}_pmc_sq -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > $P/${tag}_pmc_sq.log 2>&1; prune $P/${tag}_pmc_sq ;;
  esac
done
echo "$tag $wl done"
