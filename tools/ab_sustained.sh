#!/bin/bash
# Runs ON THE GPU BOX: like tools/ab.sh, but compares the >= 2 s window (a VALU-heavy kernel's first launches are slow and a
# 30-step mean moves by +-4 % with them)
#   tools/ab_sustained.sh <workload> <lib A> <lib B> [rounds]
wl=$1; A=$2; B=$3; rounds=${4:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $rounds); do
  for L in "$A" "$B"; do
    QMANN_LIB_PATH=$L python3 $R/bench.py --workload $wl --steps 20 --no-cpu-baseline --no-secondary 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sustained']; print('$wl', '$(basename $L)', 'sustained kernel_ms %.4f (first tenth %.4f, last tenth %.4f) frac %.4f over %d steps' % (s['kernel_ms'], s['kernel_ms_first_tenth'], s['kernel_ms_last_tenth'], s['frac'], s['steps']))"
  done
done
