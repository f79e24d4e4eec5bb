#!/bin/bash
# Runs ON THE GPU BOX: the launch-by-launch durations of a workload's hop kernel, in launch order (rocprofv3 kernel trace)
#   tools/trace_seq.sh <workload> [steps]
wl=$1; steps=${2:-30}
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/trace_seq; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $P/t -- python3 $R/bench.py --workload $wl --steps $steps --no-cpu-baseline --no-secondary --no-sustained > $P/run.log 2>&1
python3 - $P <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/t/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_hops' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
gap=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e6 for i in range(len(rows)-1)]
print('durations ms:',' '.join('%.3f'%x for x in d))
print('gaps ms     :',' '.join('%.3f'%x for x in gap))
PY
rm -rf $P/t
