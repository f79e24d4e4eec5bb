#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of the quad kernel (hops_quad.h) against the lean kernel alone (QMANN_NO_QUAD), plus a
# kernel trace of each bAbI forward.   tools/ab_quad.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_quad; mkdir -p $O
B="python bench.py --steps 30 --no-cpu-baseline --no-sustained --workload"
for r in 1 2; do
  for w in babi_task1_idx babi_joint20_appx_mq babi_joint20_v1; do
    $B $w > $O/${w}_quad_$r.json 2>/dev/null
    QMANN_NO_QUAD=1 $B $w > $O/${w}_lean_$r.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_quad/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], '%.1f M q/s' % (d['value']/1e6), '%.4f ms' % d['ms_per_step'])
    except Exception as e: print(f, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp QMANN_BENCH_NO_SMALL_BATCH=1
for w in babi_task1_idx babi_joint20_appx_mq; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 30 --no-cpu-baseline --no-secondary --no-sustained > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$w -name '*kernel_stats.csv' | head -1); cp $f $GRAFT_REPO_ROOT/$O/stats_$w.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof_$w
  head -8 $GRAFT_REPO_ROOT/$O/stats_$w.csv | cut -c1-200
done
