#!/usr/bin/env python3
"""Times the fused LBVH build alone (no tracing), and the refit of the same hierarchy:
python scripts/build_time.py <triangles> [repeats] [second hierarchy 0|1] [time the refit too 0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpuraytracer_amd import host_py                      # noqa: E402
from gpuraytracer_amd.runtime import Accel, Device        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
second = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with_refit = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sc = host_py.synthetic_scene(n)
dev = Device(0)
idx, vtx = dev.upload(sc["indices"]), dev.upload(sc["vertices"])
bvh = dev.alloc(32 * dev.lib.rt_bvh_node_count(n))
acc = Accel(dev, n)
dev.lib.rt_accel_set_second_hierarchy(acc.handle, second, 21)
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
med = times[len(times) // 2]
rtimes = []
for _ in range(reps if with_refit else 0):
    a, b = dev.event(), dev.event()
    a.record()
    acc.refit(bvh)
    b.record()
    rtimes.append(a.elapsed_ms(b))
rtimes.sort()
print("%d triangles, %s: build median %.4f ms, min %.4f ms, %.2f TB/s of the 252 B/triangle = %.3f of the HBM peak"
      % (n, "both hierarchies" if second else "the reference's hierarchy alone", med, times[0], 252.0 * n / (med * 1e-3) / 1e12,
         252.0 * n / (med * 1e-3) / 8e12))
if rtimes:
    print("    refit (leaves and levels over the kept order) median %.4f ms, min %.4f ms" % (rtimes[len(rtimes) // 2], rtimes[0]))
