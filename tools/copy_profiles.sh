#!/bin/bash
# copies the summaries tools/profile_round.sh left under gpurun_out/prof_TAG into profiles/TAG_*.  Usage: tools/copy_profiles.sh TAG
set -e
# (gpurun merges new files into gpurun_out/: remove gpurun_out/prof_TAG before a new profile round, or stale traces stay beside the new ones)
TAG=${1:-r3}
O=gpurun_out/prof_$TAG
cp "$(ls -t $O/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_kernel_stats.csv
cp "$(ls -t $O/trace_no_large/*/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_no_large_kernel_stats.csv
tail -1 $O/bench_under_trace.json > profiles/${TAG}_bench_line_under_rocprof.json
tail -1 $O/bench_no_large_under_trace.json > profiles/${TAG}_bench_no_large_line_under_rocprof.json
tail -1 $O/bench_line.json > profiles/${TAG}_bench_line.json
for f in pmc_fetch pmc_write pmc_sq calib_fetch calib_write valu_rate; do cp $O/$f.txt profiles/${TAG}_$f.txt; done
cp $O/hbm_traffic.json profiles/${TAG}_hbm_traffic.json
