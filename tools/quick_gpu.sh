#!/bin/bash
# extraction parity tests + short bench + FETCH_SIZE pass; prints one summary
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_extract.py tests/test_golden.py tests/test_gpu_batch.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
timeout -k 10 120 python bench.py --no-cpu-baseline --no-bruteforce > gpurun_out/b.log 2>&1 || { tail gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['stage_ms'], d['pipeline']['link_match_ms_per_batch'])"
if [ "$1" = "pmc" ]; then
  export TMPDIR=/tmp
  rm -rf gpurun_out/pmcq
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bruteforce > gpurun_out/pmcq.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmcq | grep -A1 "^k_" | grep -v "^--" | paste - - | awk '{print $1, $5}'
fi
