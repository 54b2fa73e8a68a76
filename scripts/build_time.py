#!/usr/bin/env python3
"""Times the fused LBVH build alone (no tracing): python scripts/build_time.py <triangles> [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpuraytracer_amd import host_py                      # noqa: E402
from gpuraytracer_amd.runtime import Accel, Device        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = host_py.synthetic_scene(n)
dev = Device(0)
idx, vtx = dev.upload(sc["indices"]), dev.upload(sc["vertices"])
bvh = dev.alloc(32 * dev.lib.rt_bvh_node_count(n))
acc = Accel(dev, n)
acc.pack(n, idx, vtx)
for _ in range(3):
    acc.build(sc["scene_min"], sc["scene_max"], bvh)
times = []
for _ in range(reps):
    a, b = dev.event(), dev.event()
    a.record()
    acc.build(sc["scene_min"], sc["scene_max"], bvh)
    b.record()
    times.append(a.elapsed_ms(b))
times.sort()
print("%d triangles: build median %.4f ms, min %.4f ms, %.2f TB/s of the 252 B/triangle" % (n, times[len(times) // 2], times[0],
                                                                                          252.0 * n / (times[len(times) // 2] * 1e-3) / 1e12))
