import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpuraytracer_amd import scenes
from gpuraytracer_amd.pipeline import Pipeline
from gpuraytracer_amd.runtime import Device
from oracle import oracle_py as oracle
n, w, h = 1000, 96, 54
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = scenes.synthetic_scene(n)
cam = sc["camera"]
iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
rs = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)); ts = iter(oracle.mt19937(2, 0, 30).reshape(-1, 3))
dev = Device(0)
gp = Pipeline(dev, sc, w, h, iv, ip, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts), use_accel=(len(sys.argv) < 3))
op = oracle.Pipeline(sc, w, h, seed_raygen=1, seed_trace=2)
gp.render(calls); op.render(calls)
R = w*h
g_rays = gp.rays.download(np.float32, 8*R).reshape(-1, 8); o_rays = op.rays().view(np.float32).reshape(-1, 8)
g_em = gp.emitted.download(np.float32, 4*R).reshape(-1,4); o_em = op.emitted()
g_sc = gp.scattered.download(np.float32, 4*R).reshape(-1,4); o_sc = op.scattered()
for name, g, o in (("rays", g_rays, o_rays), ("emitted", g_em, o_em), ("scattered", g_sc, o_sc)):
    neq = (g.view(np.uint32) != o.view(np.uint32))
    rows = np.where(neq.any(axis=1))[0]
    print(name, "rows differing", len(rows), "of", len(g), "cols", neq.sum(axis=0))
    for r in rows[:6]:
        print("  row", r, "gpu", g[r], "\n        cpu", o[r])
