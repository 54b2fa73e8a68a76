#!/bin/bash
# Mrays/s, launch time and walk statistics of the default kernel: bash scripts/quick_bench.sh [workload ...]
out=gpurun_out/quick_bench.txt
: > $out
for wl in "${@:-c3}"; do
  timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-counters --no-reference-walk --no-others \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl', d['value'], 'launch', r['avg_launch_ms'], 'build', d['lbvh_build_ms'], d['trace_ms_by_bounce'], json.dumps({k:v for k,v in (r.get('own_layout') or {}).items() if k!='note'}))" >> $out || exit 1
done
cat $out
