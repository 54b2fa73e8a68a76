#!/bin/bash
# Sweep of one GPURT_* variable on one workload (GPU box): bash scripts/tuning_sweep.sh <workload> <VARIABLE> <values...>
set -u
WL=$1; VAR=$2; shift 2
for v in "$@"; do
  env "$VAR=$v" timeout -k 10 300 python bench.py --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline --no-counters --no-reference-walk > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('$WL $VAR=$v: %.1f Mrays/s, launch %.4f ms, trace call %.4f ms' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['avg_trace_call_ms']))"
done
