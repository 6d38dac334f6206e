#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into per-launch HBM-side byte counts.

Following the MI355X guide's HBM section: the two counters come from separate --pmc passes; on gfx950 FETCH_SIZE
tallies 128-byte requests as 64 bytes, so it is corrected by the factor measured with tools/hbm_calib.bin on the
access width these kernels use (one dword per lane; known 1 GiB read and 1 GiB written, far beyond the Infinity
Cache) -- the factor comes out as 2.0 for FETCH_SIZE and 1.0 for WRITE_SIZE.  Counters are in KiB.
Usage: pmc_traffic.py gpurun_out/prof_TAG profiles/TAG_hbm_traffic.json"""
import collections, csv, glob, json, re, sys

root, out = sys.argv[1], sys.argv[2]


def kname(full):
    """short kernel name; the two launches of k_fast_cells per extraction (levels >= 1 / level 0) stay apart"""
    m = re.search(r"\bk_\w+", full)
    if not m:
        return None
    k = m.group(0)
    t = re.search(r"k_fast_cells<([^>]*)>", full)
    if k == "k_fast_cells" and t:
        k += "<" + t.group(1).replace(" ", "") + ">"
    return k



def mean_per_kernel(sub, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = kname(row["Kernel_Name"])
            if k:
                acc[k] += float(row["Counter_Value"])
                n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


KNOWN = float(1 << 30)
cf = mean_per_kernel("calib_fetch", "FETCH_SIZE")["k_calib_read"] * 1024
cw = mean_per_kernel("calib_write", "WRITE_SIZE")["k_calib_write"] * 1024
f_fetch, f_write = KNOWN / cf, KNOWN / cw
fetch = mean_per_kernel("pmc_fetch", "FETCH_SIZE")
write = mean_per_kernel("pmc_write", "WRITE_SIZE")
res = {"source": f"{root}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-bruteforce`; mean per launch; KiB counters x 1024 x calibration factor",
       "calibration": {"fetch_factor": round(f_fetch, 4), "write_factor": round(f_write, 4),
                       "method": "tools/hbm_calib.bin: 1 GiB dword-per-lane streaming read / write under the same two passes"},
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fb = fetch.get(k, 0.0) * 1024 * f_fetch
    wb = write.get(k, 0.0) * 1024 * f_write
    res["kernels"][k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb)}
# instruction counters of the pmc_sq pass (mean per launch), for the VALU-issue fractions bench.py prints
sq = collections.defaultdict(dict)
for f in glob.glob(f"{root}/pmc_sq/**/*counter_collection.csv", recursive=True):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = kname(row["Kernel_Name"])
        if k:
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            n[(k, row["Counter_Name"])] += 1
    for (k, c), v in acc.items():
        sq[k][c] = round(v / n[(k, c)])
res["sq_counters_per_launch"] = sq
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["calibration"]), {k: v["hbm_bytes"] for k, v in res["kernels"].items()})
