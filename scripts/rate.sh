#!/bin/bash
# Usage: scripts/rate.sh <triangles>... -- box-test and triangle-test rates of the trace kernel at several scene sizes
for n in "$@"; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --triangles $n > gpurun_out/rate.json 2> gpurun_out/rate.err || { echo "$n FAILED"; tail -3 gpurun_out/rate.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/rate.json'));r=d['roofline']
rays=r['rays_per_launch'];ms=r['avg_launch_ms']
print($n,'Mrays/s',d['value'],'launch ms',ms,'box/ray',r['box_tests_per_ray'],'tri/ray',r['triangle_tests_per_ray'],'Gbox/s',round(rays*r['box_tests_per_ray']/ms/1e6,1),'by bounce',d['trace_ms_by_bounce'])"
done
