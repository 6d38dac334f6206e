#!/bin/bash
# Per-phase times of k_fast_cells ("Round 3: k_fast_cells").  Build step (CPU, here):
#   tools/fast_profile.sh build     -> orb_slam3_amd/liborbgpu_prof.so (k_fast.hip with -DORBGPU_FAST_PROFILE, the other objects as built)
# Run step (GPU box): tools/fast_profile.sh [W H FEATURES FRAMES]   (default 640 480 1000 256)
set -e -o pipefail
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  cd orb_slam3_amd/csrc && make
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DORBGPU_FAST_PROFILE -c k_fast.hip -o build/k_fast_prof.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbgpu_prof.so build/api.o build/k_image.o build/k_fast_prof.o build/k_describe.o build/k_match.o build/k_database.o build/k_quadtree.o
  exit 0
fi
ORBGPU_LIBRARY=$PWD/orb_slam3_amd/liborbgpu_prof.so timeout -k 10 120 python tools/qt_profile.py ${1:-640} ${2:-480} ${3:-1000} ${4:-256} 2>&1 | grep "FAST profile" | tail -2
