"""One-off check at the 50 M-triangle size: a small image traced on the library's own hierarchy and by the walk in
the reference's order must give the same buffers bit for bit. Usage (GPU box): python scripts/check_c5.py [triangles]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpuraytracer_amd import capi, host_py, scenes          # noqa: E402
from gpuraytracer_amd.pipeline import Pipeline              # noqa: E402
from gpuraytracer_amd.runtime import Device                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
w, h, calls, depth = 480, 270, 4, 4
sc = host_py.synthetic_scene(n)
cam = sc["camera"]
iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
dev = Device(0)
got = []
for kernel in (3, 0):                                       # RT_TRACE_KERNEL_SPATIAL, RT_TRACE_KERNEL_QUAD
    capi.check(dev.lib.rt_device_set_trace_kernel(dev.handle, kernel), "rt_device_set_trace_kernel")
    rs = iter(scenes.mt19937_raw(1, 6).reshape(-1, 3))
    ts = iter(scenes.mt19937_raw(2, 3 * calls + 3).reshape(-1, 3))
    p = Pipeline(dev, sc, w, h, iv, ip, rpp=1, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
    listed = []
    for _ in range(calls):
        p.render(1)
        c = C.c_uint32(0)
        capi.check(dev.lib.rt_device_listed_rays(dev.handle, C.byref(c)), "rt_device_listed_rays")
        listed.append(c.value)
    got.append((p.result_image().copy(), p.emitted.download(np.float32, 4 * w * h).copy(), p.rays.download(np.float32, 8 * w * h).copy()))
    print("kernel %d: listed rays per call %s, image mean %.6f" % (kernel, listed, float(got[-1][0][..., :3].mean())), flush=True)
    p.close()
same = all(a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes() for a, b in zip(got[0], got[1]))
print("%d triangles, %dx%d, %d bounces: buffers %s" % (n, w, h, depth, "identical" if same else "DIFFER"))
sys.exit(0 if same else 1)
