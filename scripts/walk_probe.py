"""Lane occupancy of the default traversal kernel on the C3 frame: steps per turn of the walking loop, leaves per
triangle round (counting variant). Usage (GPU box): python scripts/walk_probe.py [triangles] [width] [height]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpuraytracer_amd import host_py, scenes
from gpuraytracer_amd.pipeline import Pipeline
from gpuraytracer_amd.runtime import Device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
sc = host_py.synthetic_scene(n)
cam = sc["camera"]
iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
dev = Device(0)
rs = iter(scenes.mt19937_raw(1, 30).reshape(-1, 3)); ts = iter(scenes.mt19937_raw(2, 60).reshape(-1, 3))
p = Pipeline(dev, sc, w, h, iv, ip, depth=4, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
buf = dev.alloc(32)
dev.lib.rt_device_set_walk_stats(dev.handle, buf.handle)
for bounce in range(4):
    buf.clear()
    p.render(1)
    dev.sync()
    steps, empty, leaves, packed = [int(x) for x in buf.download(np.uint64, 4)]
    turns, rounds = packed & ((1 << 32) - 1), packed >> 32
    rays = w * h
    print("bounce %d: %.1f steps/ray (%.1f empty), %.2f leaves/ray; %.1f lanes per walking turn, %.1f lanes per triangle round; per 64 rays %.0f turns, %.1f rounds"
          % (bounce, steps / rays, empty / rays, leaves / rays, steps / max(turns, 1), leaves / max(rounds, 1), turns / (rays / 64), rounds / (rays / 64)), flush=True)
