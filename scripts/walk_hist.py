"""Steps of the default traversal kernel by depth of the entered node (two levels per entry), per bounce:
    python scripts/walk_hist.py [triangles] [width] [height]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from gpuraytracer_amd import host_py, scenes
from gpuraytracer_amd.pipeline import Pipeline
from gpuraytracer_amd.runtime import Device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
depth = 4
scene = host_py.synthetic_scene(n)
cam = scene["camera"]
iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
rs = iter(scenes.mt19937_raw(1, 30).reshape(-1, 3))
ts = iter(scenes.mt19937_raw(2, 120).reshape(-1, 3))
dev = Device(0)
pipe = Pipeline(dev, scene, w, h, iv, ip, rpp=1, depth=depth, use_bvh=True, use_accel=True, raygen_seeds=lambda: next(rs),
                trace_seeds=lambda: next(ts), rebuild_every_cycle=True)
pipe.render(depth)
dev.sync()
buf = dev.alloc(160)
dev.lib.rt_device_set_walk_stats(dev.handle, buf.handle)
rays = w * h
prev = np.zeros(20, dtype=np.uint64)
pipe.build()
pipe.camera_rays()
for b in range(depth):
    pipe.trace()
    dev.sync()
    cur = buf.download(np.uint64, 20)
    d = (cur - prev).astype(np.float64)
    prev = cur
    print("bounce %d: steps/ray %.2f empty %.2f leaves %.2f | by depth: %s" % (
        b, d[0] / rays, d[1] / rays, d[2] / rays, " ".join("%.2f" % (x / rays) for x in d[4:16])))
dev.lib.rt_device_set_walk_stats(dev.handle, None)
