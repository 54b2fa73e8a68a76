"""Debug helper: re-traces one fuzz case with both kernels and lists, for every ray whose result differs, the hits a
brute-force float32 Moeller-Trumbore finds along the traced ray. Usage: python scripts/debug_case.py <case> <seed>"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fuzz_walks", os.path.join(ROOT, "scripts", "fuzz_walks.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from gpuraytracer_amd import capi, host_py, scenes          # noqa: E402
from gpuraytracer_amd.pipeline import Pipeline              # noqa: E402
from gpuraytracer_amd.runtime import Device                 # noqa: E402

case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
for c in range(case + 1):
    sc, kind, n = fz.random_scene(rng)
w, h, calls, depth = 256, 144, 4, 4
cam = sc["camera"]
iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
dev = Device(0)
got = []
for kernel in (3, 0):
    capi.check(dev.lib.rt_device_set_trace_kernel(dev.handle, kernel), "k")
    rs = iter(scenes.mt19937_raw(1, 6).reshape(-1, 3))
    ts = iter(scenes.mt19937_raw(2, 3 * calls + 3).reshape(-1, 3))
    p = Pipeline(dev, sc, w, h, iv, ip, rpp=1, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
    per_call = []
    for _ in range(calls):
        p.render(1)
        per_call.append((p.rays.download(np.float32, 8 * w * h).reshape(-1, 8).copy(), p.old_rays.download(np.float32, 8 * w * h).reshape(-1, 8).copy()))
    got.append(per_call)
    p.close()
pos = sc["vertices"]["position"].reshape(-1, 3, 3).astype(np.float32)
for call in range(1):
    a, b = got[0][call][0], got[1][call][0]
    rows = np.nonzero(np.any(a.view(np.uint32) != b.view(np.uint32), axis=1))[0]
    for r in rows[:4]:
        old_a, old_b = got[0][call][1][r], got[1][call][1][r]
        print("call %d ray %d: traced ray default %s | other %s" % (call, r, old_a, old_b))
        ray = a[r] if not old_a.any() else old_b          # a ray that hit nothing stays what it was
        o, d = ray[0:3], ray[3:6]
        e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
        tv = o[None, :] - pos[:, 0]
        pv = np.cross(d[None, :], e2).astype(np.float32)
        qv = np.cross(tv, e1).astype(np.float32)
        det = np.einsum("ij,ij->i", e1, pv).astype(np.float32)
        inv = np.where(np.abs(det) < 1e-6, 0, 1 / det).astype(np.float32)
        t = (np.einsum("ij,ij->i", e2, qv) * inv).astype(np.float32)
        u = (np.einsum("ij,ij->i", tv, pv) * inv).astype(np.float32)
        v = (np.einsum("ij,j->i", qv, d) * inv).astype(np.float32)
        ok = (t >= 1e-6) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= ray[6]) & (t < ray[7])
        idx = np.nonzero(ok)[0]
        order = idx[np.argsort(t[idx])][:8]
        for i in order:
            lo, hi = pos[i].min(axis=0), pos[i].max(axis=0)
            with np.errstate(all="ignore"):
                t1, t2 = (lo - o) / d, (hi - o) / d
            tn = np.max(np.minimum(t1, t2))
            tf = np.min(np.maximum(t1, t2))
            print("   box lo %s hi %s  t1 %s t2 %s  exit %.9g  corners %s" % (lo, hi, t1, t2, tf, pos[i].tolist()))
            print("   tri %d: t %.9g u %.4g v %.4g det %.4g own-box entry %.9g" % (i, t[i], u[i], v[i], det[i], tn))
        for name, new in (("default", a[r]), ("other", b[r])):
            print("   %s hit point -> distance along ray %.9g" % (name, np.dot(new[0:3] - o, d) / np.dot(d, d)))
