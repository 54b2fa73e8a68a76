#!/bin/bash
# Mrays/s of the default kernel over its tuning knobs (C3): bash scripts/tuning_sweep.sh
out=gpurun_out/tuning_sweep.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-counters --no-reference-walk --no-others \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_ms'], d['trace_ms_by_bounce'])" >> $out || exit 1
}
for t in 16 24 32 40 48; do run GPURT_LEAF_THRESHOLD=$t; done
for t in 4 12 16 24; do run GPURT_REFILL_THRESHOLD=$t; done
for t in 3 5; do run GPURT_BLOCKS_PER_CU=$t; done
run GPURT_KEY_PASSES=1
run GPURT_KEY_PASSES=3
run GPURT_SORT_RAYS=0
cat $out
