#!/bin/bash
# wall times of the online call shape, then the kernel timeline of the last single-frame extraction and window search
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/online_probe.py 60 || exit 1
rm -rf gpurun_out/ot && mkdir -p gpurun_out/ot
timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/ot --output-format csv -- python3 tools/online_probe.py 12 > gpurun_out/ot/probe.log 2>&1 || { tail gpurun_out/ot/probe.log; exit 1; }
python3 - <<'P'
import csv, glob, re
f = sorted(glob.glob("gpurun_out/ot/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [i for i, r in enumerate(rows) if "k_describe" in r["Kernel_Name"]]
sel = rows[d[-2] + 1:d[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
print("-- last single-frame extraction")
for r in sel:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Kernel_Name"])
    print("%-30s %8.1f %8.1f" % (m.group(1) if m else r["Kernel_Name"][:30], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
w = [i for i, r in enumerate(rows) if "k_snn_small_resolve" in r["Kernel_Name"]]
if len(w) >= 2:
    sel = rows[w[-2] + 1:w[-1] + 1]
    t0 = int(sel[0]["Start_Timestamp"])
    print("-- last window search")
    for r in sel:
        m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Kernel_Name"])
        print("%-30s %8.1f %8.1f" % (m.group(1) if m else r["Kernel_Name"][:30], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
P
