#!/bin/bash
# Runs ON THE GPU BOX: instruction-mix counters of one workload's kernels (ad-hoc; two passes)
#   tools/pmc_embed.sh <workload> <kernel-name regex>
wl=$1; pat=$2
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/pmc_x; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $P/p$i -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > $P/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $P/p$i.log; }
  for c in $(find $P/p$i -name '*counter_collection.csv'); do
    python3 - "$c" "$pat" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r.get('Kernel_Name','')
    if not re.search(sys.argv[2],k): continue
    k=k[:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    print(k)
    for c,x in v.items(): print('   ',c, x/cnt[(k,c)])
PY
  done
done
