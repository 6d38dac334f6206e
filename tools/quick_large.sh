set -o pipefail
timeout -k 10 200 python bench.py --no-cpu-baseline --no-bruteforce --no-database --no-allpairs --no-projection --steps 10 --warmup 3 > gpurun_out/ql.log 2>&1 || { tail gpurun_out/ql.log; exit 1; }
tail -1 gpurun_out/ql.log | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k,v in d['large_images'].items():
    if isinstance(v, dict): print(k, v.get('frames_per_s'), v.get('ms_per_batch'), {a: round(b,3) for a,b in v.get('stage_ms').items()})
"
