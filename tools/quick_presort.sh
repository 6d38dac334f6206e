#!/bin/bash
# presort of the quadtree (k_qt_presort): parity of the extraction tests with every level presorted, then the
# large-image timings with and without it
set -o pipefail
ORBGPU_TUNE=qt_presort_minpx=0 timeout -k 10 400 python -u -m pytest tests/test_gpu_extract.py tests/test_gpu_batch.py tests/test_gpu_headline.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ps_t1.log 2>&1 || { tail -20 gpurun_out/ps_t1.log; exit 1; }
tail -1 gpurun_out/ps_t1.log
bash tools/quick_large.sh || exit 1
ORBGPU_TUNE=qt_presort_minpx=-1 bash tools/quick_large.sh || exit 1
