"""Timeline of the last full extraction in a rocprofv3 kernel trace (gpurun_out/qt): kernel, start, end in us."""
import csv, glob, re, sys
f = sorted(glob.glob("gpurun_out/qt/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [r for r in rows if "k_describe" in r["Kernel_Name"]]
t1, tprev = int(d[-2]["End_Timestamp"]), int(d[-3]["End_Timestamp"])
sel = [r for r in rows if tprev < int(r["Start_Timestamp"]) <= t1]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Kernel_Name"])
    print("%-30s %8.1f %8.1f" % (m.group(1) if m else r["Kernel_Name"][:30], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
