#!/bin/bash
# Mrays/s of the default kernel over its tuning knobs (C3): bash scripts/tuning_sweep.sh
out=gpurun_out/tuning_sweep.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-counters --no-reference-walk --no-others \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; o=r.get('own_layout') or {}; print(d['value'], r['avg_launch_ms'], d['trace_ms_by_bounce'], o.get('lanes_per_walking_turn'), o.get('lanes_per_triangle_round'), o.get('steps_per_ray'))" >> $out || exit 1
}
for r in 4 8 12; do for t in 24 32 40; do run GPURT_REFILL_THRESHOLD=$r GPURT_LEAF_THRESHOLD=$t; done; done
for s in 1 2 3 4 6 8 64; do run GPURT_STUCK_THRESHOLD=$s; done
for d in 7 8 10; do run GPURT_DEAL_SHIFT=$d; done
run GPURT_BLOCKS_PER_CU=3
cat $out
