#!/bin/bash
# Build a VARIANT of the library for a same-box A/B (tools/ab.sh): the kernel sources are copied to a scratch directory, a sed
# script is applied to them, and the result is linked into q-mann_amd/lib_exp/libqmann_<name>.so (git-ignored; it travels to
# the GPU box with the snapshot).   tools/build_variant.sh <name> '<sed script>' [file ...]     (default file: hops_lean.h)
set -e
name=$1; script=$2; shift 2
files=${@:-hops_lean.h}
R=$(cd "$(dirname "$0")/.." && pwd)
X=/tmp/qmann_variant_$name; rm -rf $X; mkdir -p $X $R/q-mann_amd/lib_exp
cp $R/q-mann_amd/csrc/*.h $R/q-mann_amd/csrc/*.hip $X/
sed -i "s#\.\./\.\./include/#$R/include/#" $X/*.h $X/*.hip
for f in $files; do sed -i "$script" $X/$f; done
objs=""
for s in $X/*.hip; do
  o=$X/$(basename $s .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -Wall -Wno-unused-function -I $R/include -c $s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/q-mann_amd/lib_exp/libqmann_$name.so $objs
ls -la $R/q-mann_amd/lib_exp/libqmann_$name.so
