#!/bin/bash
# Usage: scripts/sweep.sh "<ENV=VAL ...>" ...   -- one short bench run per argument, prints Mrays/s, trace ms, build ms
for cfg in "$@"; do
  env $cfg timeout -k 10 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-walk > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$cfg FAILED"; tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/sweep.json'));print('$cfg', d['value'], d['roofline']['avg_launch_ms'], d['lbvh_build_ms'], d['roofline']['box_tests_per_ray'])"
done
