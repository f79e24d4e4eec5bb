#!/bin/bash
# Runs ON THE GPU BOX: why is the one-wavefront kernel's occupancy low?  Few counters per pass (the SPI
# block holds only a couple at a time), each pass under its own timeout.
#   tools/occ_probe.sh <workload>
set -e
wl=${1:-babi_task1_idx}
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN" "SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN" \
           "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "MeanOccupancyPerCU"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $P/occ$i -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $P/occ$i.log 2>&1
  for c in $(find $P/occ$i -name '*counter_collection.csv'); do { head -1 "$c"; grep -E 'k_hops' "$c" || true; } > "$c.tmp"; mv "$c.tmp" "$c"; done
  find $P/occ$i -type f ! -name '*counter_collection.csv' -delete
  echo "pass $i ($set) done"
done
