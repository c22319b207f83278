#!/bin/bash
# Build a variant of libplp_front.so into build_exp/<name>.so for a same-box A/B (tools/ab_libs.sh).
#   bash tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DPLP_CORUN_PRIO=2>"
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2
mkdir -p build_exp/obj_$name
cd structure-plp-slam_amd/csrc
pids=()
for f in *.hip; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mno-tgsplit -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result $extra -c -o ../../build_exp/obj_$name/${f%.hip}.o $f ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build_variant: a compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_exp/$name.so ../../build_exp/obj_$name/*.o
rm -rf ../../build_exp/obj_$name
ls -la ../../build_exp/$name.so
