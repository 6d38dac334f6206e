#!/bin/bash
# short headline bench (no side sections); prints value, stage times and the link time.  Usage: tools/quick_bench.sh [extra bench args]
set -o pipefail
timeout -k 10 120 python bench.py --no-cpu-baseline --no-bruteforce --no-database --no-allpairs --no-large --no-projection --no-sequence --no-host-path --one-context --steps 10 --warmup 3 "$@" > gpurun_out/qb.log 2>&1 || { tail gpurun_out/qb.log; exit 1; }
tail -1 gpurun_out/qb.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline']['stage_ms'], d['pipeline']['link_match_ms_per_batch'], d['pipeline']['frac_of_8TBps'])"
