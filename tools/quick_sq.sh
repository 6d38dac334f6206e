#!/bin/bash
# SQ instruction counters per kernel for the short bench.  Usage: tools/quick_sq.sh [extra bench args]
export TMPDIR=/tmp
rm -rf gpurun_out/pmcsq
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES -d gpurun_out/pmcsq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bruteforce --no-allpairs --no-database --no-large --no-projection --no-sequence --no-host-path --one-context "$@" > gpurun_out/pmcsq.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcsq | awk '/^k_/{k=$1} /SQ_/{printf "%s %s %s %s\n", k, $1, $2, $3}' | grep -E "^k_(fast_cells|blur|orient|resize_level|quadtree|link_fused)"
