#!/bin/bash
# Is the DEVICE code of two commits the same?  (profiles are stamped with a hash of csrc/*.h*: a host-side or comment edit changes
# the stamp but not what was measured.)  Compiles every csrc/*.hip of both commits with --cuda-device-only -S and compares.
#   tools/device_asm_equal.sh <commit A> [<commit B> = working tree]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
A=$1; B=${2:-WORKTREE}
T=$(mktemp -d)
for side in A B; do
  # (both sides are compiled at the SAME path, one after the other: the path goes into the anonymous namespaces' unique names)
  c=${!side}; rm -rf $T/work; mkdir -p $T/work/csrc $T/work/include $T/$side
  if [ "$c" = WORKTREE ]; then cp $R/q-mann_amd/csrc/*.h* $T/work/csrc/; cp $R/include/*.h $T/work/include/
  else for f in $(git -C $R ls-tree --name-only $c q-mann_amd/csrc/ include/); do git -C $R show $c:$f > $T/work/$( [[ $f == include/* ]] && echo include || echo csrc )/$(basename $f); done; fi
  sed -i "s#\.\./\.\./include/#$T/work/include/#" $T/work/csrc/*
  for s in $T/work/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -ffp-contract=off -I $T/work/include --cuda-device-only -S $s -o $T/$side/$(basename $s .hip).s 2>/dev/null &
  done; wait
done
rc=0
norm() { grep -v '^\s*\.file\|^\s*\.ident\|\.loc\b' $1 | sed 's/__hip_cuid_[0-9a-f]*/__hip_cuid_X/g'; }    # (the compilation-unit id hashes the whole source text)
for s in $T/A/*.s; do b=$(basename $s); if cmp -s <(norm $s) <(norm $T/B/$b); then echo "same      $b"; else echo "DIFFERENT $b"; rc=1; fi; done
rm -rf $T; exit $rc
