#!/bin/bash
# steps / leaves per ray and Mrays/s of the default kernel for several triangle-round thresholds (C3)
set -e
out=gpurun_out/threshold_sweep.txt
: > $out
for t in 1 8 16 24 32 40; do
  echo "== GPURT_LEAF_THRESHOLD=$t" >> $out
  GPURT_LEAF_THRESHOLD=$t timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-counters --no-reference-walk --no-others \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_ms'], d['trace_ms_by_bounce'], json.dumps(r.get('own_layout')))" >> $out
done
cat $out
