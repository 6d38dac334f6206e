#!/bin/bash
# A/B of two builds on one box: orb_slam3_amd/liborbgpu_base.so (a copy of an earlier build) against the in-tree library,
# alternating, N rounds.  Usage: tools/ab_bench.sh [N] [extra env assignments applied to both, e.g. ORBGPU_TUNE=early_l0=0]
N=${1:-3}; shift
for i in $(seq $N); do
  echo -n "base: "; env "$@" ORBGPU_LIBRARY=$PWD/orb_slam3_amd/liborbgpu_base.so bash tools/quick_bench.sh --steps 30 || exit 1
  echo -n "new:  "; env "$@" bash tools/quick_bench.sh --steps 30 || exit 1
done
