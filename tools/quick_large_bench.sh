#!/bin/bash
# bench of the headline + large-image sections only.  Usage: tools/quick_large_bench.sh [env assignments...]
set -o pipefail
timeout -k 10 200 env "$@" python bench.py --no-cpu-baseline --no-bruteforce --no-database --no-allpairs --no-projection --no-sequence --no-host-path --one-context --steps 10 --warmup 3 > gpurun_out/ql_b.json 2> gpurun_out/ql_b.err || { tail gpurun_out/ql_b.err; exit 1; }
python3 - <<'P'
import json
d = json.load(open("gpurun_out/ql_b.json"))
print("VGA", d["value"], d["ms_per_step"], d["pipeline"]["stage_ms"])
for k, v in d["large_images"].items():
    print(k, v["frames_per_s"], v["ms_per_batch"], v["extract_ms_per_batch"], v["frac_of_8TBps"], v["stage_ms"])
P
