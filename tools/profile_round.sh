#!/bin/bash
# Collects the round's profiles on the GPU box: kernel-trace stats of the default bench command, then counter passes
# (each in a run of its own, never together with a trace), then the FETCH/WRITE calibration.  Usage: tools/profile_round.sh TAG
set -e -o pipefail
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py > $OUT/bench_under_trace.json 2> $OUT/trace.log
echo trace done
# the same without the large-image, sequence and host-path sections (their launches of the same kernels run on other
# batch sizes): there every k_fast_cells launch is a headline launch, so the kernel's average
# in this file is the one the bench line's roofline.avg_launch_ms has to agree with
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_no_large --output-format csv -- python3 bench.py --no-large --no-sequence --no-host-path --one-context > $OUT/bench_no_large_under_trace.json 2> $OUT/trace_no_large.log
echo trace without large images done
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bruteforce --no-database --no-large --no-projection --no-allpairs --no-sequence --no-host-path --one-context"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- $BENCH > $OUT/pmc_write.log 2>&1
echo write done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/pmc_sq --output-format csv -- $BENCH > $OUT/pmc_sq.log 2>&1
echo sq done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch --output-format csv -- ./tools/hbm_calib.bin > $OUT/calib_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_write --output-format csv -- ./tools/hbm_calib.bin > $OUT/calib_write.log 2>&1
echo calib done
timeout -k 10 200 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.log
for d in pmc_fetch pmc_write pmc_sq calib_fetch calib_write; do python3 tools/pmc_summary.py $OUT/$d > $OUT/$d.txt; done
python3 tools/pmc_traffic.py $OUT $OUT/hbm_traffic.json
./tools/valu_rate.bin > $OUT/valu_rate.txt
echo all done
