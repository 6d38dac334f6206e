#!/bin/bash
# One counter pass over the short bench for the base build (orb_slam3_amd/liborbgpu_base.so) and the in-tree one.
# Usage: tools/ab_pmc.sh "CTR1 CTR2 ..." [kernel regex]
echo "== base"; ORBGPU_LIBRARY=$PWD/orb_slam3_amd/liborbgpu_base.so bash tools/quick_pmc.sh "$1" "$2" || exit 1
echo "== new"; bash tools/quick_pmc.sh "$1" "$2"
