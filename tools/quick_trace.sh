#!/bin/bash
# kernel trace of the short headline bench; prints the per-kernel averages.  Usage: tools/quick_trace.sh [extra bench args]
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/qt && mkdir -p gpurun_out/qt
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/qt --output-format csv -- python3 bench.py --no-cpu-baseline --no-bruteforce --no-database --no-allpairs --no-large --no-projection --steps 10 --warmup 3 "$@" > gpurun_out/qt/bench.log 2>&1 || { tail gpurun_out/qt/bench.log; exit 1; }
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/qt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = n.split("(")[0][-60:]
    if "k_resize" in n:
        key += " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        v2 = sorted(v)[: max(1, len(v) * 3 // 4)]
        print("%-75s n=%4d avg=%8.1f us  (fastest 3/4: %8.1f)" % (k, len(v), sum(v) / len(v), sum(v2) / len(v2)))
P
