#!/bin/bash
# One counter pass (<= 8 SQ counters) over the short bench, per-kernel means.  Usage: tools/quick_pmc.sh "CTR1 CTR2 ..." [kernel regex]
export TMPDIR=/tmp
rm -rf gpurun_out/qpmc
timeout -s KILL 120 rocprofv3 --pmc $1 -d gpurun_out/qpmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bruteforce --no-allpairs --no-database --no-large --no-projection --no-sequence --no-host-path --one-context > gpurun_out/qpmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/qpmc | awk '/^k_/{k=$1} /SQ_|TCP_|TCC_|TA_/{printf "%s %s %s %s\n", k, $1, $2, $3}' | grep -E "^${2:-k_(fast_cells|blur|orient|describe|resize_level|quadtree)}"
