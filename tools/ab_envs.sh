#!/bin/bash
# Runs ON THE GPU BOX: interleaved timing of several environments of one workload (boxes differ by +-3 %: only same-box
# interleaving compares two settings).
#   tools/ab_envs.sh <workload> <rounds> <steps> "<env assignments A>" "<env assignments B>" ...   (BENCH_ARGS: extra bench.py flags)
wl=$1; rounds=$2; steps=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python3 $R/bench.py --workload $wl --steps $steps --no-cpu-baseline --no-secondary --no-sustained $BENCH_ARGS 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$BENCH_ARGS', '[$e]', 'ms %.4f value %.0f' % (d['ms_per_step'], d['value']))"
  done
done
