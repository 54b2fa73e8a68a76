#!/usr/bin/env python3
"""How many rays the default kernel hands to the walk in the reference's order, per trace call, on the reference's three
assets (1920x1080, the cameras of the reference's Settings.h) and on the benchmark's scene - and how many of them were
listed while they were walked rather than at the end: python scripts/listed_rays.py [cycles]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                               # noqa: E402
from gpuraytracer_amd import capi, host_py                 # noqa: E402
from gpuraytracer_amd.pipeline import Pipeline             # noqa: E402
from gpuraytracer_amd.runtime import Device                # noqa: E402
from oracle import oracle_py as oracle                     # noqa: E402   (camera matrices and seeds only)

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = Device(0)
for name, w, h, depth in (("c2_testscene2", 1920, 1080, 3), ("c2_testobject4_2", 1920, 1080, 3), ("c2_testobject5_1", 1920, 1080, 3),
                          ("c3", 3840, 2160, 4)):
    sc = bench.make_scene(name, 1000000)
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    rs = iter(oracle.mt19937(1, 0, 3 * cycles).reshape(-1, 3))
    ts = iter(oracle.mt19937(2, 0, 3 * cycles * depth).reshape(-1, 3))
    gp = Pipeline(dev, sc, w, h, iv, ip, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
    rows = []
    for call in range(cycles * depth):
        gp.render(1, previews=False)
        total, during, whole = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        capi.check(dev.lib.rt_device_listed_rays_detail(dev.handle, C.byref(total), C.byref(during), C.byref(whole)), "rt_device_listed_rays_detail")
        rows.append((total.value, during.value))
    print("%-18s %d rays per call; listed (of them during the walk) per call: %s" % (name, w * h, " ".join("%d(%d)" % r for r in rows)), flush=True)
    gp.close()
dev.close()
