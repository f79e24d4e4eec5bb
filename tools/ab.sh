#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B timing of two builds of the library inside one session (boxes differ by +-3 %, and a
# sustained stream settles below the first runs: anything but same-box interleaving measures the box).
#   tools/ab.sh <workload> <lib A> <lib B> [rounds] [steps]
wl=$1; A=$2; B=$3; rounds=${4:-3}; steps=${5:-30}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $rounds); do
  for L in "$A" "$B"; do
    QMANN_LIB_PATH=$L python3 $R/bench.py --workload $wl --steps $steps --no-cpu-baseline --no-secondary --no-sustained 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$(basename $L)', 'kernel_ms %.4f frac %.4f value %.0f' % (d['roofline'].get('kernel_ms', 0), d['roofline']['frac'], d['value']))"
  done
done
