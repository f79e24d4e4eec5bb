#!/bin/bash
# Runs ON THE GPU BOX: interleaved timing of one workload under two settings of an environment variable (same library).
#   tools/ab_env.sh <workload> <VAR> <value A> <value B> [rounds] [steps]
wl=$1; var=$2; A=$3; B=$4; rounds=${5:-2}; steps=${6:-30}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $rounds); do
  for v in "$A" "$B"; do
    env $var=$v python3 $R/bench.py --workload $wl --steps $steps --no-cpu-baseline --no-secondary --no-sustained 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$var=$v', 'kernel_ms %.4f frac %.4f value %.0f' % (d['roofline'].get('kernel_ms', 0), d['roofline']['frac'], d['value']))"
  done
done
