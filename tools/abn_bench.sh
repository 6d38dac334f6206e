#!/bin/bash
# A/B/.. of several builds on one box, alternating, N rounds: tools/abn_bench.sh N lib1.so lib2.so ...  (paths under orb_slam3_amd/)
N=$1; shift
for i in $(seq $N); do
  for l in "$@"; do
    echo -n "$l: "; ORBGPU_LIBRARY=$PWD/orb_slam3_amd/$l bash tools/quick_bench.sh --steps 30 || exit 1
  done
done
