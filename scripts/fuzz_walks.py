"""Differential run of the two traversals over random scenes and cameras: the default kernel (the library's own
hierarchy) against the walk in the reference's order, which returns the reference's closest hits by construction.
Every buffer must agree bit for bit. With a third argument "oracle" the default kernel is compared with the CPU oracle
instead (test infrastructure; slower). Usage (GPU box): python scripts/fuzz_walks.py [cases] [seed] [oracle]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpuraytracer_amd import capi, host_py, scenes          # noqa: E402
from gpuraytracer_amd.capi import vertex_dtype              # noqa: E402
from gpuraytracer_amd.pipeline import Pipeline              # noqa: E402
from gpuraytracer_amd.runtime import Device                 # noqa: E402


def mesh_scene(rng):
    """One of the reference's own meshes (as the golden fixtures hold them: shared edges and vertices, large walls) seen
    from a random place."""
    gold = os.path.join(ROOT, "tests", "golden")
    name = str(rng.choice(["testobject5_1", "testscene2"]))
    g = np.load(os.path.join(gold, "oracle_%s.npz" % name))
    v = g["vertices"]
    lo, hi = v["position"].min(axis=0), v["position"].max(axis=0)
    mid, half = (lo + hi) / 2, (hi - lo) / 2
    eye = mid + rng.uniform(-1.2, 1.2, 3) * half
    focus = mid + rng.uniform(-0.5, 0.5, 3) * half
    cam = dict(position=tuple(float(x) for x in eye), focus=tuple(float(x) for x in focus), up=(0.0, 1.0, 0.0),
               fov=float(rng.uniform(0.5, 1.6)), near_z=0.01, far_z=10000.0)
    sc = dict(vertices=v, indices=np.arange(len(v), dtype=np.uint32), materials=g["materials"], texture_ids=g["texture_ids"],
              atlas=g["atlas"], scene_min=g["scene_min"], scene_max=g["scene_max"], camera=cam, name=name)
    return sc, "mesh", len(v) // 3


def random_scene(rng):
    """A triangle soup of a random kind: sizes from specks to slabs across the scene, axis-aligned quads (flat boxes,
    shared edges), slivers (rays nearly in their plane), duplicates (equal distances)."""
    n = int(10 ** rng.uniform(2.0, float(os.environ.get("FUZZ_MAX_EXP", "5.3"))))
    kinds = ["soup", "big", "quads", "slivers", "duplicates", "clusters", "inplane", "huge", "tiny", "mesh", "sheets", "room"]
    if os.environ.get("FUZZ_KINDS"):
        kinds = os.environ["FUZZ_KINDS"].split(",")
    kind = rng.choice(kinds)
    if kind == "mesh":
        return mesh_scene(rng)
    c = rng.uniform(-10, 10, (n, 1, 3))
    if kind == "clusters":
        c = rng.choice([-6.0, 0.0, 7.0], (n, 1, 3)) + rng.normal(0, 0.4, (n, 1, 3))
    e = 20.0 / np.cbrt(n) * (8.0 if kind == "big" else 1.0)
    off = rng.uniform(-e, e, (n, 3, 3))
    if kind in ("quads", "inplane"):                      # triangles in axis-aligned planes of a 0.25 grid
        axis = rng.randint(0, 3, n)
        off[np.arange(n), :, axis] = 0.0
        c = np.round(c * 4) / 4
    if kind == "slivers":
        off[:, 2] = off[:, 1] * rng.uniform(0.98, 1.02, (n, 1)) + rng.normal(0, e * 1e-3, (n, 3))
    pos = (c + off).astype(np.float32)
    if kind == "duplicates":
        pos[n // 2:] = pos[:n - n // 2]
        pos = pos[rng.permutation(n)]
    if kind == "sheets":                                  # 1 - 150 triangles that span the scene (outliers: tested from a list,
        k = min(int(rng.randint(1, 151)), n - 1)          # beyond 64 of them the rest stays in the hierarchy) among the soup
        pos[:k] = (rng.uniform(-12, 12, (k, 1, 3)) + rng.uniform(-25, 25, (k, 3, 3))).astype(np.float32)
        pos = pos[rng.permutation(n)]
    if kind == "room" and n >= 40:                        # a closed box around the soup and large axis-aligned panels inside:
        box = scenes._box_triangles(float(rng.uniform(11, 30)))      # flat outliers whose hits tie with their boxes' entries
        k = int(rng.randint(0, 13))
        panels = np.zeros((2 * k, 3, 3), dtype=np.float32)
        for j in range(k):
            axis, at = int(rng.randint(0, 3)), float(np.round(rng.uniform(-10, 10) * 4) / 4)
            u, v2 = [a for a in range(3) if a != axis]
            lo, hi = np.sort(rng.uniform(-11, 11, (2, 2)), axis=0)
            q = np.zeros((4, 3), dtype=np.float32)
            q[:, axis] = at
            q[:, u] = (lo[0], hi[0], hi[0], lo[0])
            q[:, v2] = (lo[1], lo[1], hi[1], hi[1])
            panels[2 * j], panels[2 * j + 1] = q[[0, 1, 2]], q[[0, 2, 3]]
        extra = np.concatenate([box, panels])
        pos[:len(extra)] = extra
        pos = pos[rng.permutation(n)]
    v = np.zeros(3 * n, dtype=vertex_dtype)
    v["position"] = pos.reshape(-1, 3)
    v["normal"] = (0, 0, -1)
    v["uv"] = rng.uniform(0, 1, (3 * n, 2)).astype(np.float32)
    v["material_id"] = np.repeat(np.arange(n) % 8, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    atlas, ids = scenes.default_atlas()
    smin, smax = scenes.centroid_bounds(v, idx)
    scale = 1.0
    if kind == "huge":                                    # coordinates binary16 cannot hold: the 16-bit boxes become infinite
        scale = 5000.0
    if kind == "tiny":                                    # determinants near the reference's 1e-6 cut-off
        scale = 0.02
    v["position"] *= np.float32(scale)
    smin, smax = scenes.centroid_bounds(v, idx)
    eye = rng.uniform(-12, 12, 3) * scale
    if kind == "inplane":                                 # the camera in one of the planes: rays (nearly) inside flat boxes
        eye = np.round(eye * 4) / 4
    cam = dict(position=tuple(float(x) for x in eye), focus=tuple(float(x) for x in rng.uniform(-5, 5, 3) * scale), up=(0.0, 1.0, 0.0),
               fov=float(rng.uniform(0.5, 1.6)), near_z=0.01 * scale, far_z=10000.0 * scale)
    return dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                scene_max=smax, camera=cam, name="%s-%d" % (kind, n)), kind, n


def run(dev, cases, seed, w=256, h=144, calls=4, depth=4, verbose=True, against_oracle=False, only=None):
    """Traces `cases` random scenes with both kernels on device `dev` (or with the default kernel and the oracle);
    returns (cases that differ, rays listed)."""
    oracle = None
    if against_oracle:
        from oracle import oracle_py as oracle
    rng = np.random.RandomState(seed)
    bad = 0
    total_listed = 0
    run.in_parts = run.whole_after_parts = 0      # rays of lists short enough to be walked in parts; of them, walked whole after all
    t0 = time.time()
    for case in range(cases):
        sc, kind, n = random_scene(rng)
        if only is not None and case not in only:
            continue
        cam = sc["camera"]
        iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
        got, listed = [], 0
        for kernel in ((3,) if oracle else (3, 0)):         # RT_TRACE_KERNEL_SPATIAL, RT_TRACE_KERNEL_QUAD
            capi.check(dev.lib.rt_device_set_trace_kernel(dev.handle, kernel), "rt_device_set_trace_kernel")
            rs = iter(scenes.mt19937_raw(1, 6).reshape(-1, 3))
            ts = iter(scenes.mt19937_raw(2, 3 * calls + 3).reshape(-1, 3))
            p = Pipeline(dev, sc, w, h, iv, ip, rpp=1, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
            for _ in range(calls):
                p.render(1)
                if kernel == 3:
                    c, during, whole = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
                    capi.check(dev.lib.rt_device_listed_rays_detail(dev.handle, C.byref(c), C.byref(during), C.byref(whole)),
                               "rt_device_listed_rays_detail")
                    listed += c.value
                    run.in_parts += c.value if c.value <= 512 else 0
                    run.whole_after_parts += whole.value
            got.append((p.result_image().copy(), p.emitted.download(np.float32, 4 * w * h).copy(), p.rays.download(np.float32, 8 * w * h).copy()))
            p.close()
        if oracle:
            op = oracle.Pipeline(sc, w, h, rpp=1, depth=depth, seed_raygen=1, seed_trace=2)
            op.render(calls)
            got.append((op.result().copy(), op.emitted().copy(), op.rays().view(np.float32).reshape(-1, 8).copy()))
            op.close()
        diff = sum(int(np.count_nonzero(np.any(a.reshape(w * h, -1).view(np.uint32) != b.reshape(w * h, -1).view(np.uint32), axis=1)))
                   for a, b in zip(got[0], got[1]))
        total_listed += listed
        bad += diff != 0
        if diff and only is not None:
            for name, a, b in zip(("result", "emitted", "rays"), got[0], got[1]):
                a2, b2 = a.reshape(w * h, -1), b.reshape(w * h, -1)
                rows = np.nonzero(np.any(a2.view(np.uint32) != b2.view(np.uint32), axis=1))[0]
                for r in rows[:8]:
                    print("  %s[%d] (x %d, y %d): default %s | other %s" % (name, r, r % w, r // w, a2[r], b2[r]))
        if verbose or diff:
            print("case %3d %-10s %7d triangles: %s, %d rays listed%s" % (case, kind, n, "same" if not diff else "%d ENTRIES DIFFER" % diff,
                  listed, "" if case % 10 else "   (%.0f s)" % (time.time() - t0)), flush=True)
    capi.check(dev.lib.rt_device_set_trace_kernel(dev.handle, 3), "rt_device_set_trace_kernel")
    return bad, total_listed


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    w, h, calls = 256, 144, 4
    only = {int(x) for x in os.environ["FUZZ_ONLY"].split(",")} if os.environ.get("FUZZ_ONLY") else None
    bad, listed = run(Device(0), cases, seed, w, h, calls, against_oracle=len(sys.argv) > 3 and sys.argv[3] == "oracle", only=only)
    print("%d cases, %d with differences, %d rays listed in all (of %d traced); %d of them on lists of up to 512 rays (walked in parts "
          "where GPURT_LISTED_SPLIT or a recent call says so), %d walked whole after a part ran out of room"
          % (cases, bad, listed, cases * calls * w * h, run.in_parts, run.whole_after_parts))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
