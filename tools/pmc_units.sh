#!/bin/bash
# Runs ON THE GPU BOX: which unit does a hop kernel saturate?  Five --pmc passes (8 SQ slots each, the TCP counters three at a time: six in one pass exceed the hardware; counters never together
# with tracing) of one bench workload; only the k_hops_* rows travel back.  tools/summarize_units.py turns them into the
# table under profiles/.
#   tools/pmc_units.sh <tag> <workload> [kernel-regex (default k_hops)] [extra]
# `extra`: three more passes: FETCH_SIZE, WRITE_SIZE (they do not fit one pass) and the MFMA counters (the embedding kernels are write-heavy)
tag=$1; wl=$2; rx=${3:-k_hops}; extra=$4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/units/$tag; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
export QMANN_BENCH_NO_SMALL_BATCH=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
           "TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ" "TCP_TCC_READ_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_ACCESSES" \
           ${extra:+"FETCH_SIZE"} ${extra:+"WRITE_SIZE"} ${extra:+"SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"}; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $P/p$i -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > $P/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $P/p$i.log; }
  for c in $(find $P/p$i -name '*counter_collection.csv'); do { head -1 "$c"; grep -E "$rx" "$c" || true; } > $P/pass$i.csv; done
  rm -rf $P/p$i
done
echo "$tag $wl units done"
