"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bars: bit-exact for Morton codes, every sort pass, the sorted permutation and every hierarchy
node; bit-exact as well for rays and shaded images wherever both sides evaluate the same binary32
expression sequence (they are written to), with the stated tolerance as the formal bar:
max abs error <= 1e-5 on the tone-mapped image (values in [0, 1)).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

IMAGE_TOL = 1e-5


def _scene(n):
    from gpuraytracer_amd import scenes
    return scenes.synthetic_scene(n)


def _small_scene(n_tris, seed=7):
    """n_tris random triangles with no enclosing box (covers N = 3, 4, 5, odd counts...)."""
    from gpuraytracer_amd import scenes
    from gpuraytracer_amd.capi import vertex_dtype
    rng = np.random.RandomState(seed)
    v = np.zeros(3 * n_tris, dtype=vertex_dtype)
    c = rng.uniform(-1, 1, (n_tris, 1, 3)).astype(np.float32)
    v["position"] = (c + rng.uniform(-0.3, 0.3, (n_tris, 3, 3)).astype(np.float32)).reshape(-1, 3)
    v["normal"] = (0, 0, -1)
    v["uv"] = rng.uniform(0, 1, (3 * n_tris, 2)).astype(np.float32)
    v["material_id"] = np.repeat(np.arange(n_tris) % 8, 3)
    idx = np.arange(3 * n_tris, dtype=np.uint32)
    atlas, ids = scenes.default_atlas()
    smin, smax = scenes.centroid_bounds(v, idx)
    cam = dict(position=(0.0, 0.0, -3.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.2, near_z=0.01, far_z=10000.0)
    return dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                scene_max=smax, camera=cam, name="small-%d" % n_tris)


def _u4(buf, n):
    from oracle.oracle_py import uint4_dtype
    return buf.download(uint4_dtype, n)


@pytest.mark.parametrize("n", [3, 4, 5, 6, 7, 255, 256, 257, 1000, 2048, 2049, 4097, 70001])
def test_per_shader_build_matches_oracle_bit_exact(device, oracle, n):
    """Each launcher against the oracle function of the same shader, chained on device outputs."""
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.capi import aabb_dtype
    lib, d = device.lib, device.handle
    sc = _small_scene(n) if n < 1000 else _scene(n)
    olib = oracle.load()
    idx, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"])
    morton, temp, freq = device.alloc(16 * n), device.alloc(16 * n), device.alloc(16 * 256)
    freq.upload(np.full(1024, 0xDEADBEEF, dtype=np.uint32))      # the Morton shader must zero .x itself

    mi = capi.MortonCodeInfo()
    omi = oracle.MortonInfo()
    for k in range(3):
        mi.scene_min[k] = omi.scene_min[k] = float(sc["scene_min"][k])
        mi.scene_max[k] = omi.scene_max[k] = float(sc["scene_max"][k])
    mi.num_primitives = omi.num_primitives = n
    capi.check(lib.rt_generate_morton_codes(d, C.byref(mi), idx.handle, vtx.handle, morton.handle, freq.handle), "morton")
    o_m = np.zeros(n, dtype=oracle.uint4_dtype)
    o_t = np.zeros(n, dtype=oracle.uint4_dtype)
    o_f = np.zeros(256, dtype=oracle.uint4_dtype)
    o_f["x"] = 0xDEADBEEF
    o_f["y"] = 0xDEADBEEF
    olib.orc_generate_morton_codes(C.byref(omi), oracle.ptr(sc["indices"]), oracle.ptr(sc["vertices"]), oracle.ptr(o_m),
                                   oracle.ptr(o_f))
    g_m = _u4(morton, n)
    assert np.array_equal(g_m, o_m), "Morton records differ"
    assert np.array_equal(_u4(freq, 256)["x"], o_f["x"])

    for p in range(4):
        si, osi = capi.SortInfo(n, p), oracle.SortInfo(n, p)
        capi.check(lib.rt_sort_count(d, C.byref(si), morton.handle, temp.handle, freq.handle), "count")
        olib.orc_sort_count(C.byref(osi), oracle.ptr(o_m), oracle.ptr(o_t), oracle.ptr(o_f))
        assert np.array_equal(_u4(freq, 256)["x"], o_f["x"]), "digit counts differ in pass %d" % p
        g_t = _u4(temp, n)
        assert np.array_equal(g_t["x"], o_t["x"]) and np.array_equal(g_t["y"], o_t["y"])
        capi.check(lib.rt_sort_prefix_sum(d, C.byref(si), freq.handle), "prefix")
        olib.orc_sort_prefix_sum(C.byref(osi), oracle.ptr(o_f))
        g_f = _u4(freq, 256)
        assert np.array_equal(g_f["y"], o_f["y"]), "prefix sums differ in pass %d" % p
        capi.check(lib.rt_sort_sort(d, C.byref(si), morton.handle, temp.handle, freq.handle), "scatter")
        olib.orc_sort_sort(C.byref(osi), oracle.ptr(o_m), oracle.ptr(o_t), oracle.ptr(o_f))
        g_m = _u4(morton, n)
        assert np.array_equal(g_m["x"], o_m["x"]) and np.array_equal(g_m["y"], o_m["y"]), "scatter differs in pass %d" % p
        assert np.array_equal(_u4(freq, 256)["x"], o_f["x"])
    # structural invariant: the result is the stable sort of (code, 3*tri) by code
    codes = g_m["x"].astype(np.int64)
    assert np.all(np.diff(codes) >= 0)
    ties = np.diff(codes) == 0
    assert np.all(np.diff(g_m["y"].astype(np.int64))[ties] > 0)

    nodes = oracle.load().orc_bvh_node_count(n)
    assert nodes == lib.rt_bvh_node_count(n)
    bvh = device.alloc(32 * max(nodes, 1))
    o_bvh = np.zeros(nodes, dtype=aabb_dtype)
    bi, obi = capi.BVHInfo(n, 1, (n + 1) // 2 + 1, 0), oracle.BvhInfo(n, 1, (n + 1) // 2 + 1, 0)
    capi.check(lib.rt_bvh_build_leaves(d, C.byref(bi), idx.handle, vtx.handle, morton.handle, bvh.handle), "leaves")
    olib.orc_bvh_build_leaves(C.byref(obi), oracle.ptr(sc["indices"]), oracle.ptr(sc["vertices"]), oracle.ptr(o_m),
                              oracle.ptr(o_bvh))
    nc, prev, cur = (n + 1) // 2, 1, (n + 1) // 2 + 1
    while nc != 2:
        bi, obi = capi.BVHInfo(nc, prev, cur, 0), oracle.BvhInfo(nc, prev, cur, 0)
        capi.check(lib.rt_bvh_build(d, C.byref(bi), bvh.handle), "level")
        olib.orc_bvh_build(C.byref(obi), oracle.ptr(o_bvh))
        nc = (nc + 1) // 2
        prev, cur = cur, cur + nc
    bi, obi = capi.BVHInfo(2, prev, 0, 0), oracle.BvhInfo(2, prev, 0, 0)
    capi.check(lib.rt_bvh_build(d, C.byref(bi), bvh.handle), "root")
    olib.orc_bvh_build(C.byref(obi), oracle.ptr(o_bvh))
    assert cur == nodes
    g_bvh = bvh.download(aabb_dtype, nodes)
    assert g_bvh.tobytes() == o_bvh.tobytes(), "hierarchy nodes differ"
    for b in (idx, vtx, morton, temp, freq, bvh):
        b.free()


@pytest.mark.parametrize("n", [3, 4, 5, 511, 513, 1000, 4099, 131073, 1000000])
def test_fused_build_matches_oracle_bit_exact(device, oracle, n):
    from gpuraytracer_amd.capi import aabb_dtype
    from gpuraytracer_amd.runtime import Accel
    sc = _small_scene(n) if n < 1000 else _scene(n)
    o_m = oracle.sort_primitives(sc)
    o_bvh, count = oracle.build_bvh(sc, o_m)
    idx, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"])
    acc = Accel(device, n)
    acc.pack(n, idx, vtx)
    bvh, morton = device.alloc(32 * count), device.alloc(16 * n)
    for _ in range(2):                                   # a rebuild must give the same result
        acc.build(sc["scene_min"], sc["scene_max"], bvh, morton)
        g_m = _u4(morton, n)
        assert np.array_equal(g_m, o_m), "sorted Morton records differ"
        assert bvh.download(aabb_dtype, count).tobytes() == o_bvh[:count].tobytes(), "hierarchy nodes differ"
    acc.destroy()
    for b in (idx, vtx, bvh, morton):
        b.free()


def test_too_few_primitives_is_an_error(device):
    from gpuraytracer_amd import capi
    assert device.lib.rt_bvh_node_count(2) == 0
    h = C.c_void_p()
    assert device.lib.rt_accel_create(device.handle, 2, C.byref(h)) == capi.RT_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("w,h,rpp,dof", [(64, 36, 1, 0.0), (130, 50, 2, 1.0), (17, 9, 3, 0.5)])
def test_camera_rays_match_oracle(device, oracle, w, h, rpp, dof):
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.capi import ray_dtype
    iv, ip = oracle.camera_matrices((-4.0259, 2.1676, -2.9459), (0, 0, 0), (0, 1, 0), 1.2, w / h, 0.01, 10000.0)
    info, oinfo = capi.CameraRayGenInfo(), oracle.RaygenInfo()
    for k in range(16):
        info.inverse_view[k] = oinfo.inverse_view[k] = float(iv[k])
        info.inverse_projection[k] = oinfo.inverse_projection[k] = float(ip[k])
    info.screen_size[0] = oinfo.screen[0] = w
    info.screen_size[1] = oinfo.screen[1] = h
    info.aa_sample_spread = oinfo.aa_spread = 1.5
    info.dof_sample_spread = oinfo.dof_spread = dof
    info.max_rays_per_pixel = oinfo.max_rays_per_pixel = rpp
    for k, s in enumerate((123456789, 362436069, 521288629)):
        info.rng_seed[k] = oinfo.rng_seed[k] = s
    n = w * h * rpp
    rays, old, pix = device.alloc(32 * n), device.alloc(32 * n), device.alloc(4 * n)
    capi.check(device.lib.rt_camera_ray_generation(device.handle, C.byref(info), rays.handle, old.handle, pix.handle, None), "raygen")
    o_r, o_o, o_p = np.zeros(n, dtype=ray_dtype), np.zeros(n, dtype=ray_dtype), np.zeros(n, dtype=np.uint32)
    oracle.load().orc_camera_ray_generation(C.byref(oinfo), oracle.ptr(o_r), oracle.ptr(o_o), oracle.ptr(o_p))
    assert np.array_equal(pix.download(np.uint32, n), o_p)
    assert old.download(ray_dtype, n).tobytes() == o_o.tobytes()
    g_r = rays.download(ray_dtype, n)
    assert g_r.tobytes() == o_r.tobytes(), "camera rays differ"
    # unit test of the camera convention: with depth of field off every ray starts near the camera position
    if dof == 0.0:
        assert np.abs(g_r["origin"] - np.array([-4.0259, 2.1676, -2.9459], dtype=np.float32)).max() < 0.05
    # a band of rows reproduces the same rays
    r0, nrows = h // 3, h // 2
    nb = w * nrows * rpp
    rb, ob, pb = device.alloc(32 * nb), device.alloc(32 * nb), device.alloc(4 * nb)
    band = capi.ImageRows(r0, nrows, 1)
    capi.check(device.lib.rt_camera_ray_generation(device.handle, C.byref(info), rb.handle, ob.handle, pb.handle, C.byref(band)), "band")
    assert rb.download(ray_dtype, nb).tobytes() == o_r[r0 * w * rpp:(r0 + nrows) * w * rpp].tobytes()
    assert np.array_equal(pb.download(np.uint32, nb), o_p[r0 * w * rpp:(r0 + nrows) * w * rpp])
    # every third row from row 1 on, stored one after the other
    nrows3 = (h - 1 + 2) // 3
    every3 = capi.ImageRows(1, nrows3, 3)
    capi.check(device.lib.rt_camera_ray_generation(device.handle, C.byref(info), rb.handle, ob.handle, pb.handle, C.byref(every3)), "rows")
    want_r = o_r.reshape(h, w * rpp)[1::3].reshape(-1)
    want_p = o_p.reshape(h, w * rpp)[1::3].reshape(-1)
    assert nrows3 * w * rpp <= nb and len(want_r) == nrows3 * w * rpp
    assert rb.download(ray_dtype, len(want_r)).tobytes() == want_r.tobytes()
    assert np.array_equal(pb.download(np.uint32, len(want_p)), want_p)
    bad = capi.ImageRows(1, nrows3 + 1, 3)
    assert device.lib.rt_camera_ray_generation(device.handle, C.byref(info), rb.handle, ob.handle, pb.handle, C.byref(bad)) == capi.RT_ERR_INVALID_ARGUMENT
    for b in (rays, old, pix, rb, ob, pb):
        b.free()


KERNEL_QUAD, KERNEL_WAVE, KERNEL_SIMPLE, KERNEL_SPATIAL = 0, 1, 2, 3      # RT_TRACE_KERNEL_*


def _render_both(device, oracle, sc, w, h, calls, rpp=1, depth=3, use_bvh=True, use_accel=True, dof=0.0, kernel=KERNEL_SPATIAL):
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.pipeline import Pipeline
    capi.check(device.lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    ncycles = (calls + depth - 1) // depth
    rs = oracle.mt19937(1, 0, 3 * ncycles + 3).reshape(-1, 3)
    ts = oracle.mt19937(2, 0, 3 * calls + 3).reshape(-1, 3)
    ri, ti = iter(rs), iter(ts)
    gp = Pipeline(device, sc, w, h, iv, ip, rpp=rpp, depth=depth, dof=dof, use_bvh=use_bvh, use_accel=use_accel,
                  raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
    op = oracle.Pipeline(sc, w, h, rpp=rpp, depth=depth, dof=dof, use_bvh=use_bvh, seed_raygen=1, seed_trace=2)
    try:
        gp.render(calls)
    finally:
        device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
    op.render(calls)
    out = (gp.output_image(), op.output(), gp.result_image(), op.result(),
           gp.emitted.download(np.float32, 4 * w * h * rpp).reshape(-1, 4), op.emitted(),
           gp.rays.download(np.float32, 8 * w * h * rpp).reshape(-1, 8), op.rays().view(np.float32).reshape(-1, 8),
           op.stats()[1])
    gp.close()
    op.close()
    return out


def _report(name, g, o):
    diff = np.abs(g.astype(np.float64) - o.astype(np.float64))
    diff = np.nan_to_num(diff, nan=0.0)
    exact = np.mean(g.view(np.uint32) == o.view(np.uint32))
    mse = float((diff ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
    print("%s: max abs %.3g, bit-identical %.5f%%, PSNR %.1f dB" % (name, diff.max(), 100 * exact, psnr))
    return diff.max(), exact


@pytest.mark.parametrize("n,w,h,calls,use_accel", [(1000, 96, 54, 6, True), (1000, 96, 54, 6, False), (20000, 160, 90, 9, True),
                                                   (20000, 64, 36, 3, False)])
def test_traced_images_match_oracle(device, oracle, n, w, h, calls, use_accel):
    sc = _scene(n)
    g_out, o_out, g_res, o_res, g_em, o_em, g_rays, o_rays = _render_both(device, oracle, sc, w, h, calls, use_accel=use_accel)[:8]
    assert np.isfinite(o_out).all()
    m1, _ = _report("tone-mapped", g_out, o_out)
    _report("result", g_res, o_res)
    _report("emitted", g_em, o_em)
    _report("rays", g_rays, o_rays)
    assert m1 <= IMAGE_TOL
    assert o_out[..., :3].max() > 0.0, "the test image must not be black"


@pytest.mark.parametrize("n", [3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 1023, 1025, 2049])
@pytest.mark.parametrize("use_accel,kernel", [(True, KERNEL_SPATIAL), (True, KERNEL_QUAD), (False, KERNEL_QUAD), (True, KERNEL_WAVE)])
def test_small_and_odd_triangle_counts_trace_exactly(device, oracle, n, use_accel, kernel):
    """Every shape of the balanced hierarchy near its corner cases: two leaves, only children on several levels,
    counts just below, at and above powers of two. Rays that miss everything leave their pixel untouched."""
    sc = _small_scene(n, seed=100 + n)
    g = _render_both(device, oracle, sc, 48, 32, 4, depth=2, use_accel=use_accel, kernel=kernel)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs at %d triangles" % (name, n)
    assert g[8]["rays_hit"] > 0, "some ray must hit"


@pytest.mark.parametrize("kernel", [KERNEL_SPATIAL, KERNEL_QUAD, KERNEL_WAVE, KERNEL_SIMPLE])
@pytest.mark.parametrize("n", [20000, 40001])
def test_every_trace_kernel_gives_the_oracle_image(device, oracle, kernel, n):
    """The four traversal kernels (the library's own hierarchy; the reference's hierarchy two levels per step, one
    level per step, one thread per ray) produce the same buffers, bit for bit; 20000 and 40001 triangles have an
    even and an odd number of levels below the trunk."""
    g = _render_both(device, oracle, _scene(n), 128, 72, 6, kernel=kernel)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs with kernel %d" % (name, kernel)


def _wall_scene(k, copies, seed=3):
    """copies coincident walls of k x k axis-aligned squares (two triangles each) in the plane z = 2: every leaf box
    is flat, every ray that hits has `copies` hits at the same distance, and rays through the squares' diagonals
    and edges hit several triangles of one wall. The triangles are shuffled so that neither order sees a wall whole."""
    from gpuraytracer_amd import scenes
    from gpuraytracer_amd.capi import vertex_dtype
    rng = np.random.RandomState(seed)
    xs = np.linspace(-1.0, 1.0, k + 1).astype(np.float32)
    tris = []
    for c in range(copies):
        for i in range(k):
            for j in range(k):
                a, b, cc, d = (xs[i], xs[j]), (xs[i + 1], xs[j]), (xs[i + 1], xs[j + 1]), (xs[i], xs[j + 1])
                tris.append((a, b, cc))
                tris.append((a, cc, d))
    tris = np.array(tris, dtype=np.float32)                      # [T][3][2]
    tris = tris[rng.permutation(len(tris))]
    n = len(tris)
    v = np.zeros(3 * n, dtype=vertex_dtype)
    v["position"][:, :2] = tris.reshape(-1, 2)
    v["position"][:, 2] = 2.0
    v["normal"] = (0, 0, -1)
    v["uv"] = rng.uniform(0, 1, (3 * n, 2)).astype(np.float32)
    v["material_id"] = np.repeat(np.arange(n) % 8, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    atlas, ids = scenes.default_atlas()
    smin, smax = scenes.centroid_bounds(v, idx)
    cam = dict(position=(0.0, 0.0, -1.0), focus=(0.0, 0.0, 2.0), up=(0.0, 1.0, 0.0), fov=1.0, near_z=0.01, far_z=10000.0)
    return dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                scene_max=smax, camera=cam, name="walls-%d-%d" % (k, copies))


@pytest.mark.parametrize("k,copies", [(8, 1), (8, 3), (20, 2)])
def test_equal_distances_and_flat_boxes_follow_the_reference_order(device, oracle, k, copies):
    """Hits at exactly the same distance (coincident walls, shared edges) go to the triangle the reference's walk
    meets first, and boxes without thickness enter at distances that round differently from their triangles' hits:
    the walk on the library's own hierarchy must sort both out or hand the ray to the reference's walk."""
    sc = _wall_scene(k, copies)
    g = _render_both(device, oracle, sc, 128, 128, 4, depth=2, kernel=KERNEL_SPATIAL)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs" % name
    assert g[8]["rays_hit"] > 0


def _clumped_scene(kind, n, seed=11):
    """Scenes whose centroids fill their bounds unevenly, for the second hierarchy's key coordinates and cell layout:
    'clusters' = two tight clusters far apart plus a few triangles far out, 'diagonal' = centroids along a line,
    'stacked' = every triangle around the same point, 'slab' = a flat sheet (no extent on one axis of the centroids)."""
    from gpuraytracer_amd import scenes
    from gpuraytracer_amd.capi import vertex_dtype
    rng = np.random.RandomState(seed)
    if kind == "clusters":
        c = np.where(rng.uniform(size=(n, 1)) < 0.5, -0.8, 0.8) + rng.normal(0, 0.05, (n, 3))
        c[:5] = rng.uniform(-30, 30, (5, 3))
        size = 0.05
    elif kind == "diagonal":
        t = rng.uniform(-1, 1, (n, 1))
        c = np.repeat(t, 3, axis=1) * np.array([1.0, 0.7, 0.4])
        size = 0.08
    elif kind == "stacked":
        c = np.zeros((n, 3)) + np.array([0.1, -0.2, 0.3])
        size = 0.6
    else:
        c = rng.uniform(-1, 1, (n, 3))
        c[:, 2] = 0.5
        size = 0.1
    c = c.astype(np.float32).reshape(n, 1, 3)
    off = rng.uniform(-size, size, (n, 3, 3)).astype(np.float32)
    if kind in ("stacked", "slab"):
        off -= off.mean(axis=1, keepdims=True)          # every centroid (nearly) at the same point / in the same plane
        if kind == "slab":
            off[:, :, 2] = 0.0
    v = np.zeros(3 * n, dtype=vertex_dtype)
    v["position"] = (c + off).reshape(-1, 3)
    v["normal"] = (0, 0, -1)
    v["uv"] = rng.uniform(0, 1, (3 * n, 2)).astype(np.float32)
    v["material_id"] = np.repeat(np.arange(n) % 8, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    atlas, ids = scenes.default_atlas()
    smin, smax = scenes.centroid_bounds(v, idx)
    cam = dict(position=(0.0, 0.0, -3.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.2, near_z=0.01, far_z=10000.0)
    return dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                scene_max=smax, camera=cam, name="%s-%d" % (kind, n))


@pytest.mark.parametrize("kind,n", [("clusters", 6000), ("diagonal", 6000), ("stacked", 700), ("slab", 5000), ("clusters", 300000),
                                    ("diagonal", 300000)])
def test_unevenly_filled_scenes_trace_exactly(device, oracle, kind, n):
    """The second hierarchy lays its leaves out by Morton cell where the cells' populations allow it and by rank where
    they do not, on coordinates derived from the centroids' distribution: clusters, lines, a single point and a flat
    sheet must give the oracle's buffers like any other scene. (300 000 centroids along a line occupy a few of many
    cells: the long runs of cells without triangles in between are what k_cell_fill is for.)"""
    sc = _clumped_scene(kind, n)
    w, h = (96, 54) if n <= 10000 else (64, 36)
    g = _render_both(device, oracle, sc, w, h, 4, depth=2, kernel=KERNEL_SPATIAL)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs for %s" % (name, sc["name"])
    assert g[8]["rays_hit"] > 0, "some ray must hit"


@pytest.mark.parametrize("n,w,h", [(250000, 192, 108), (1000000, 160, 90)])
def test_large_scenes_against_the_oracle(device, oracle, n, w, h):
    """The default traversal at sizes where the second hierarchy's leaf slots go by fine Morton cells (2^12 to 2^15;
    1 M triangles is the benchmark's scene), on images small enough for the oracle: every buffer bit for bit."""
    g = _render_both(device, oracle, _scene(n), w, h, 4, depth=4, kernel=KERNEL_SPATIAL)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs" % name
    assert g[8]["rays_hit"] > 0


def test_random_scenes_agree_between_the_two_walks(device):
    """scripts/fuzz_walks.py: random soups, slabs, axis-aligned quads on a grid (flat boxes, overlapping coplanar
    triangles), slivers, duplicates, clusters, cameras inside the quads' planes, coordinates too large and too small
    for 16-bit boxes - traced by the default kernel and by the walk in the reference's order. Seed 1 is the run in
    which quads on a grid first showed hits lost behind boxes that were farther by a rounding."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_walks", os.path.join(ROOT, "scripts", "fuzz_walks.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, listed = fuzz.run(device, 40, 1, verbose=False)
    print("rays listed:", listed)
    assert bad == 0


def _sliver_scenes(device, seeds=range(12)):
    """The ray and the 1300:1 sliver of the one fuzzed case in which the two walks differed, placed in random soups that
    keep clear of the ray; returns per scene the ray buffers after one trace call by (SPATIAL, QUAD) and the ray."""
    from gpuraytracer_amd import capi, scenes
    from gpuraytracer_amd.capi import vertex_dtype
    from gpuraytracer_amd.pipeline import Pipeline
    sliver = np.array([[1.472400426864624, -0.08368708193302155, -1.4298924207687378],
                       [1.6871155500411987, -0.3055894076824188, -2.033355474472046],
                       [1.687436819076538, -0.30577898025512695, -2.033759355545044]], dtype=np.float32)
    ray = np.array([-11.012279, 4.101115, -6.340874, 0.8997188, -0.31224802, 0.30497092, 0.0, 11814.092], dtype=np.float32)
    outs = []
    for seed in seeds:
        rng = np.random.RandomState(seed)
        n = 3000
        pos = (rng.uniform(-10, 10, (n, 1, 3)) + rng.uniform(-0.2, 0.2, (n, 3, 3))).astype(np.float32)
        # nothing but the sliver near the ray
        o, d = ray[0:3].astype(np.float64), ray[3:6].astype(np.float64)
        c = pos.mean(axis=1).astype(np.float64) - o
        off = np.linalg.norm(c - np.outer(c @ d / (d @ d), d), axis=1)
        pos = pos[off > 1.5]
        pos[len(pos) // 2] = sliver
        n = len(pos)
        v = np.zeros(3 * n, dtype=vertex_dtype)
        v["position"] = pos.reshape(-1, 3)
        v["normal"] = (0, 0, -1)
        v["material_id"] = np.repeat(np.arange(n) % 8, 3)
        idx = np.arange(3 * n, dtype=np.uint32)
        atlas, ids = scenes.default_atlas()
        smin, smax = scenes.centroid_bounds(v, idx)
        cam = dict(position=(0.0, 0.0, -30.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.0, near_z=0.01, far_z=10000.0)
        sc = dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                  scene_max=smax, camera=cam, name="sliver")
        out = []
        for kernel in (KERNEL_SPATIAL, KERNEL_QUAD):
            capi.check(device.lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
            eye = np.eye(4, dtype=np.float32).reshape(-1)
            p = Pipeline(device, sc, 8, 8, eye, eye, rpp=1, depth=1, raygen_seeds=lambda: (1, 2, 3), trace_seeds=lambda: (4, 5, 6))
            try:
                p.build()
                p.camera_rays()
                p.rays.upload(np.tile(ray, 64))
                p.trace()
                device.sync()
                out.append(p.rays.download(np.float32, 8 * 64).reshape(64, 8).copy())
            finally:
                device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
                p.close()
        outs.append((out[0], out[1]))
    return outs, ray


def test_false_hit_beside_a_sliver_is_found_by_both_walks(device):
    """The ray and the 1300:1 sliver of the one fuzzed case in which the two walks differed: Moeller-Trumbore accepts
    the ray (determinant 2e-5) although it passes 1 % of the box size beside the triangle's bounding box. The walk in
    the reference's order tests the triangle whenever the box of its leaf - the sliver and whatever triangle is next
    to it in the reference's order - lets the ray in; the default kernel must then find the same hit."""
    outs, ray = _sliver_scenes(device)
    found = 0
    for seed, (spatial, quad) in enumerate(outs):
        assert spatial.view(np.uint32).tobytes() == quad.view(np.uint32).tobytes(), "the walks differ (scene seed %d)" % seed
        found += int(not np.array_equal(quad[0], ray))
    print("scenes in which the reference's walk takes the false hit:", found, "of 12")
    assert found > 0, "no scene in which the reference's order lets the ray into the sliver's leaf: the test is empty"


def test_walk_stats_count_the_walk_on_the_second_hierarchy(device, oracle):
    """rt_device_set_walk_stats: steps, steps without a hit and leaves tested of the default traversal kernel; the
    image is the same with and without counting, and the walk needs far fewer box tests than the reference's."""
    from gpuraytracer_amd import capi
    sc = _scene(20000)
    buf = device.alloc(32)
    capi.check(device.lib.rt_device_set_walk_stats(device.handle, buf.handle), "rt_device_set_walk_stats")
    try:
        g = _render_both(device, oracle, sc, 128, 72, 3, kernel=KERNEL_SPATIAL)
    finally:
        device.lib.rt_device_set_walk_stats(device.handle, None)
    steps, empty, leaves, packed = [int(x) for x in buf.download(np.uint64, 4)]
    buf.free()
    turns, rounds = packed & ((1 << 32) - 1), packed >> 32
    assert 0 < steps <= 64 * turns and 0 < leaves <= 64 * rounds, "at most 64 lanes work in a turn or a round"
    assert g[2].view(np.uint32).tobytes() == g[3].view(np.uint32).tobytes()
    rays = 128 * 72 * 3
    print("per ray: %.1f steps (%.1f without a hit), %.1f leaves; the reference's walk: %.1f box tests" %
          (steps / rays, empty / rays, leaves / rays, g[8]["nodes_visited"] / g[8]["rays"]))
    assert 0 < leaves < steps and empty < steps
    assert 4 * steps < g[8]["nodes_visited"], "four boxes per step: fewer box tests than the reference's walk"
    small = device.alloc(16)
    assert device.lib.rt_device_set_walk_stats(device.handle, small.handle) == capi.RT_ERR_INVALID_ARGUMENT
    small.free()


def test_invalid_trace_kernel_is_rejected(device):
    from gpuraytracer_amd import capi
    assert device.lib.rt_device_set_trace_kernel(device.handle, 4) == capi.RT_ERR_INVALID_ARGUMENT
    assert device.lib.rt_device_set_trace_kernel(device.handle, -1) == capi.RT_ERR_INVALID_ARGUMENT


def test_traverse_timing_reports_every_launch(device, oracle):
    """rt_device_set_traverse_timing brackets the traversal kernel of every trace call with events."""
    from gpuraytracer_amd.pipeline import Pipeline
    sc = _scene(2000)
    w, h = 64, 36
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    gp = Pipeline(device, sc, w, h, iv, ip)
    assert device.lib.rt_device_set_traverse_timing(device.handle, 1) == 0
    gp.render(5)
    total, launches = C.c_double(-1.0), C.c_uint32(0)
    assert device.lib.rt_device_traverse_timing(device.handle, C.byref(total), C.byref(launches)) == 0
    assert launches.value == 5 and total.value > 0.0
    assert device.lib.rt_device_traverse_timing(device.handle, C.byref(total), C.byref(launches)) == 0
    assert launches.value == 0 and total.value == 0.0            # read once, then forgotten
    assert device.lib.rt_device_set_traverse_timing(device.handle, 0) == 0
    gp.render(1)
    assert device.lib.rt_device_traverse_timing(device.handle, C.byref(total), C.byref(launches)) == 0
    assert launches.value == 0
    gp.close()


def test_traced_image_without_bvh_matches_oracle(device, oracle):
    sc = _scene(300)
    g_out, o_out = _render_both(device, oracle, sc, 48, 27, 6, use_bvh=False)[:2]
    m1, _ = _report("no-BVH tone-mapped", g_out, o_out)
    assert m1 <= IMAGE_TOL


def test_multiple_rays_per_pixel_and_dof(device, oracle):
    sc = _scene(5000)
    g_out, o_out = _render_both(device, oracle, sc, 40, 30, 8, rpp=3, depth=4, dof=1.0)[:2]
    m1, _ = _report("rpp=3 dof=1 tone-mapped", g_out, o_out)
    assert m1 <= IMAGE_TOL


def test_image_bands_reproduce_the_full_image(device, oracle):
    """Two half-image pipelines (global pixel/ray indices in the seeds) assemble to the full image."""
    from gpuraytracer_amd.pipeline import Pipeline
    sc = _scene(4000)
    w, h, calls = 64, 40, 6
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    rs = oracle.mt19937(1, 0, 30).reshape(-1, 3)
    ts = oracle.mt19937(2, 0, 30).reshape(-1, 3)

    def run(first, rows, stride=1, rpp=1):
        ri, ti = iter(rs), iter(ts)
        p = Pipeline(device, sc, w, h, iv, ip, rpp=rpp, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti),
                     first_row=first, num_rows=rows, row_stride=stride)
        p.render(calls)
        img = p.output_image()
        p.close()
        return img

    full = run(0, h)
    top, bottom = run(0, 17), run(17, h - 17)
    assert np.array_equal(np.concatenate([top, bottom], axis=0).view(np.uint32), full.view(np.uint32))
    # rows dealt out in turn to three devices, two rays per pixel
    full2 = run(0, h, rpp=2)
    dealt = np.zeros_like(full2)
    for r in range(3):
        dealt[r::3] = run(r, None, stride=3, rpp=2)
    assert np.array_equal(dealt.view(np.uint32), full2.view(np.uint32))


def test_traversal_counters_match_oracle(device, oracle):
    """The instrumented kernel variant counts exactly the box and triangle tests the oracle performs."""
    from gpuraytracer_amd.pipeline import Pipeline
    sc = _scene(30000)
    w, h, calls = 96, 54, 6
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    rs, ts = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)), iter(oracle.mt19937(2, 0, 30).reshape(-1, 3))
    for use_accel in (True, False):
        ri, ti = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)), iter(oracle.mt19937(2, 0, 30).reshape(-1, 3))
        gp = Pipeline(device, sc, w, h, iv, ip, use_accel=use_accel, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
        stats = device.alloc(32)
        device.lib.rt_device_set_trace_stats(device.handle, stats.handle)
        gp.render(calls)
        device.sync()
        device.lib.rt_device_set_trace_stats(device.handle, None)
        got = [int(x) for x in stats.download(np.uint64, 4)]
        gp.close()
        stats.free()
        op = oracle.Pipeline(sc, w, h, seed_raygen=1, seed_trace=2)
        op.render(calls)
        _, tot = op.stats()
        op.close()
        assert got == [tot["nodes_visited"], tot["tris_tested"], tot["rays_hit"], tot["rays"]]


def _golden_scene(name):
    from tests.test_oracle_kat import _load_golden_scene
    return _load_golden_scene(name)


@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2"])
@pytest.mark.parametrize("use_accel", [True, False])
def test_reference_assets_match_golden(device, oracle, name, use_accel):
    """The reference's own scenes (testobject4_2 carries the only texture): build and images against the
    committed oracle outputs."""
    from gpuraytracer_amd.capi import aabb_dtype
    from gpuraytracer_amd.pipeline import Pipeline
    sc, g = _golden_scene(name)
    w, h, calls = 64, 36, 9
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    ri, ti = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)), iter(oracle.mt19937(2, 0, 30).reshape(-1, 3))
    gp = Pipeline(device, sc, w, h, iv, ip, use_accel=use_accel, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
    gp.render(calls)
    img = gp.output_image()
    nodes = gp.bvh.download(aabb_dtype, len(g["bvh"]))
    assert nodes.tobytes() == g["bvh"].tobytes(), "hierarchy differs from the golden one"
    if not use_accel:
        m = gp.morton.download(oracle.uint4_dtype, gp.num_prims)
        assert np.array_equal(m["x"], g["sorted_codes"]) and np.array_equal(m["y"], g["sorted_values"])
    gp.close()
    worst, exact = _report(name, img, g["image"])
    assert worst <= IMAGE_TOL
    assert img[..., :3].max() > 0


def test_bvh_and_brute_force_images_agree(device, oracle):
    """RT_USE_BVH on and off differ only in tie-breaking on equal hit distances."""
    from gpuraytracer_amd.pipeline import Pipeline
    sc, _ = _golden_scene("testobject5_1")
    w, h = 64, 36
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    imgs = []
    for use_bvh in (True, False):
        ri, ti = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)), iter(oracle.mt19937(2, 0, 30).reshape(-1, 3))
        gp = Pipeline(device, sc, w, h, iv, ip, use_bvh=use_bvh, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
        gp.render(6)
        imgs.append(gp.output_image())
        gp.close()
    assert np.mean(np.any(np.abs(imgs[0] - imgs[1]) > 1e-5, axis=2)) < 0.01


@pytest.mark.parametrize("fused,reference_walk", [(True, False), (False, False), (True, True)])
def test_cpp_host_pipeline_matches_oracle(device, oracle, fused, reference_walk):
    """RaytracerPipeline (C++ host: own scene generator, camera maths, mt19937 seeds, stage sequencing), with the default
    traversal and with Settings::ReferenceOrderWalk."""
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.capi import aabb_dtype
    n, w, h, calls = 6000, 80, 45, 7
    sc = _scene(n)
    hp = host_py.HostPipeline(device, n, w, h, sc["camera"], fused_build=fused, seed_raygen=1, seed_trace=2,
                              reference_order_walk=reference_walk)
    try:
        hp.render(calls)
    finally:
        device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)     # the setting is the device's
    img, res = hp.output(), hp.result()
    count = oracle.load().orc_bvh_node_count(n)
    nodes = hp.read_buffer("bvh", aabb_dtype, count)
    morton = hp.read_buffer("morton", oracle.uint4_dtype, n)
    hp.close()
    op = oracle.Pipeline(sc, w, h, seed_raygen=1, seed_trace=2)
    op.render(calls)
    assert nodes.tobytes() == op.bvh().tobytes()
    assert np.array_equal(morton["x"], op.morton()["x"]) and np.array_equal(morton["y"], op.morton()["y"])
    worst, _ = _report("C++ host tone-mapped", img, op.output())
    _report("C++ host result", res, op.result())
    op.close()
    assert worst <= IMAGE_TOL


def test_rebuild_every_cycle_keeps_the_image(device, oracle):
    """Per-frame rebuild (new in this build) must not change anything for a static scene."""
    from gpuraytracer_amd import host_py
    n, w, h, calls = 3000, 48, 27, 9
    sc = _scene(n)
    imgs = []
    for rebuild in (False, True):
        hp = host_py.HostPipeline(device, n, w, h, sc["camera"], rebuild_every_cycle=rebuild)
        hp.render(calls)
        imgs.append(hp.output())
        hp.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))


def _bounds_of(device, sc):
    from gpuraytracer_amd import capi
    idx, vtx, out = device.upload(sc["indices"]), device.upload(sc["vertices"]), device.alloc(32)
    out.upload(np.full(8, 7.0, dtype=np.float32))
    capi.check(device.lib.rt_scene_bounds(device.handle, len(sc["indices"]) // 3, idx.handle, vtx.handle, out.handle), "rt_scene_bounds")
    b = out.download(np.float32, 8)
    for x in (idx, vtx, out):
        x.free()
    return b


@pytest.mark.parametrize("n", [0, 1, 3, 255, 257, 4097, 300001])
def test_scene_bounds_match_the_host_loop(device, oracle, n):
    """The device reduction gives the centroid bounds of the loader's host loop, bit for bit."""
    if n == 0:
        sc = dict(_small_scene(3))
        sc["indices"] = sc["indices"][:0]
    else:
        sc = _small_scene(n, seed=3 + n) if n < 1000 else _scene(n)
    b = _bounds_of(device, sc)
    mn, mx = oracle.scene_aabb(sc)
    assert b[:3].tobytes() == mn.tobytes() and b[4:7].tobytes() == mx.tobytes()
    assert b[3] == 0.0 and b[7] == 0.0
    if n == 0:
        assert b[0] == np.float32(3.402823466e+38) and b[4] == np.float32(-3.402823466e+38)


@pytest.mark.parametrize("name", ["testobject4_2", "testscene2"])
def test_scene_bounds_of_reference_assets(device, oracle, name):
    sc = _golden_scene(name)[0]
    b = _bounds_of(device, sc)
    assert b[:3].tobytes() == np.asarray(sc["scene_min"], dtype=np.float32).tobytes()
    assert b[4:7].tobytes() == np.asarray(sc["scene_max"], dtype=np.float32).tobytes()


@pytest.mark.parametrize("frame", [0, 1, 7, 31, 32, 100000])
def test_synthetic_animation_matches_oracle(device, oracle, frame):
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.capi import vertex_dtype
    sc = _scene(5003)
    n = len(sc["indices"]) // 3
    base, out = device.upload(sc["vertices"]), device.alloc(48 * 3 * n)
    capi.check(device.lib.rt_synthetic_animate(device.handle, n, frame, base.handle, out.handle), "rt_synthetic_animate")
    want = oracle.synthetic_animate(sc, frame)["vertices"]
    got = out.download(vertex_dtype, 3 * n)
    assert got.tobytes() == want.tobytes()
    assert np.abs(got["position"] - sc["vertices"]["position"]).max() <= 0.25
    base.free()
    out.free()


@pytest.mark.parametrize("use_accel", [True, False])
def test_dynamic_scene_rebuild_matches_oracle(device, oracle, use_accel):
    """The moving synthetic scene: every sample cycle animates the triangles on the device, reduces the scene
    bounds there, rebuilds the hierarchy and traces; the oracle gets the same vertices from its own animation."""
    from gpuraytracer_amd.pipeline import Pipeline
    sc = _scene(6000)
    w, h, depth, cycles = 80, 45, 3, 3
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    ri, ti = iter(oracle.mt19937(1, 0, 30).reshape(-1, 3)), iter(oracle.mt19937(2, 0, 60).reshape(-1, 3))
    gp = Pipeline(device, sc, w, h, iv, ip, depth=depth, use_accel=use_accel, raygen_seeds=lambda: next(ri),
                  trace_seeds=lambda: next(ti), dynamic=True)
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, rebuild_every_cycle=True)
    for frame in range(cycles):
        moved = oracle.synthetic_animate(sc, frame)
        op.set_vertices(moved["vertices"])
        gp.render(depth)
        op.render(depth)
        from gpuraytracer_amd.capi import aabb_dtype
        nodes = device.lib.rt_bvh_node_count(gp.num_prims)
        assert gp.bvh.download(aabb_dtype, nodes).tobytes() == op.bvh().tobytes(), "hierarchy differs in frame %d" % frame
        m, _ = _report("dynamic frame %d" % frame, gp.output_image(), op.output())
        assert m <= IMAGE_TOL
    assert gp.result_image().view(np.uint32).tobytes() == op.result().view(np.uint32).tobytes()
    gp.close()
    op.close()


@pytest.mark.parametrize("n,w,h", [(1_000_000, 3840, 2160), (10_000_000, 1920, 1080)])
def test_full_size_frames_agree_between_the_two_hierarchies(device, oracle, n, w, h):
    """The benchmark's frame (1M triangles, 3840x2160, 4 bounces) and a 10M-triangle frame, traced once on the
    library's own hierarchy and once by the walk in the reference's order, which returns the reference's closest
    hits by construction: every buffer must be the same bit for bit. Tolerance stated for this comparison: none."""
    from gpuraytracer_amd import capi, host_py
    from gpuraytracer_amd.pipeline import Pipeline
    sc = host_py.synthetic_scene(n)
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    calls, depth = 4, 4
    got = []
    listed = []
    for kernel in (KERNEL_SPATIAL, KERNEL_QUAD):
        capi.check(device.lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
        rs = iter(oracle.mt19937(1, 0, 6).reshape(-1, 3))
        ts = iter(oracle.mt19937(2, 0, 3 * calls + 3).reshape(-1, 3))
        gp = Pipeline(device, sc, w, h, iv, ip, rpp=1, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
        try:
            for _ in range(calls):
                gp.render(1)
                c = C.c_uint32(0)
                capi.check(device.lib.rt_device_listed_rays(device.handle, C.byref(c)), "rt_device_listed_rays")
                listed.append(c.value)
        finally:
            device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
        got.append((gp.result_image().copy(), gp.emitted.download(np.float32, 4 * w * h).copy(),
                    gp.rays.download(np.float32, 8 * w * h).copy()))
        gp.close()
    print("rays handed to the walk in the reference's order, per trace call:", listed[:calls], "of", w * h)
    assert listed[calls:] == [0] * calls
    assert max(listed[:calls]) < w * h // 100, "the second walk is meant for the odd ray"
    for a, b, name in zip(got[0], got[1], ("result", "emitted", "rays")):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs between the hierarchies" % name
    assert got[0][0][..., :3].max() > 0.0


def test_ten_million_triangle_build_properties(device, oracle):
    """The 10M-triangle configuration, checked through properties that do not need the oracle at that size: the
    sorted records are a permutation ordered by (code, triangle) - i.e. the stable sort - the codes are those
    of the Morton kernel (recomputed on a sample by the oracle), every node contains its children, leaves hold
    the sorted triangles pairwise, and the node count is the closed form."""
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.capi import aabb_dtype
    from gpuraytracer_amd.runtime import Accel
    n = 10_000_000
    sc = host_py.synthetic_scene(n)
    idx, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"])
    nodes = device.lib.rt_bvh_node_count(n)
    assert nodes == 10000009
    bvh, morton = device.alloc(32 * nodes), device.alloc(16 * n)
    acc = Accel(device, n)
    acc.pack(n, idx, vtx)
    acc.build(sc["scene_min"], sc["scene_max"], bvh, morton)
    m = _u4(morton, n)
    code, ci = m["x"], m["y"]
    assert np.all(code[1:] >= code[:-1]), "codes are not ascending"
    ties = code[1:] == code[:-1]
    assert np.all(ci[1:][ties] > ci[:-1][ties]), "equal codes must keep ascending triangle order"
    seen = np.zeros(n, dtype=np.uint8)
    assert np.all(ci % 3 == 0)
    seen[ci // 3] = 1
    assert seen.all(), "not a permutation of the triangles"
    # codes of a sample of triangles from the oracle's Morton function
    rng = np.random.RandomState(5)
    sample = np.sort(rng.choice(n, 200000, replace=False)).astype(np.uint32)
    sub = dict(sc)
    sub["indices"] = (3 * sample[:, None] + np.arange(3, dtype=np.uint32)[None, :]).reshape(-1).astype(np.uint32)
    omi = oracle.MortonInfo()
    for k in range(3):
        omi.scene_min[k] = float(sc["scene_min"][k])
        omi.scene_max[k] = float(sc["scene_max"][k])
    omi.num_primitives = len(sample)
    o_m = np.zeros(len(sample), dtype=oracle.uint4_dtype)
    o_f = np.zeros(256, dtype=oracle.uint4_dtype)
    oracle.load().orc_generate_morton_codes(C.byref(omi), oracle.ptr(sub["indices"]), oracle.ptr(sc["vertices"]), oracle.ptr(o_m),
                                            oracle.ptr(o_f))
    code_of_tri = np.empty(n, dtype=np.uint32)
    code_of_tri[ci // 3] = code
    assert np.array_equal(code_of_tri[sample], o_m["x"])
    # hierarchy: leaves = bounds of the sorted triangle pairs, inner nodes = union of their children
    g = bvh.download(aabb_dtype, nodes)
    pos = sc["vertices"]["position"]
    leaves = (n + 1) // 2
    tri_lo = pos[sc["indices"].reshape(-1, 3)].min(axis=1)       # [n, 3]
    tri_hi = pos[sc["indices"].reshape(-1, 3)].max(axis=1)
    order = ci // 3
    lo = np.minimum(tri_lo[order[0::2]], tri_lo[order[1::2]])
    hi = np.maximum(tri_hi[order[0::2]], tri_hi[order[1::2]])
    assert np.array_equal(g["min"][1:1 + leaves], lo) and np.array_equal(g["max"][1:1 + leaves], hi)
    assert np.array_equal(g["link"][1:1 + leaves, 0], ci[0::2] | 0x80000000)
    prev, count, cur = 1, leaves, 1 + leaves
    while count > 1:
        parents = (count + 1) // 2
        base = 0 if parents == 1 else cur
        c0 = prev + 2 * np.arange(parents)
        c1 = np.minimum(c0 + 1, prev + count - 1)
        assert np.array_equal(g["min"][base:base + parents], np.minimum(g["min"][c0], g["min"][c1]))
        assert np.array_equal(g["max"][base:base + parents], np.maximum(g["max"][c0], g["max"][c1]))
        assert np.array_equal(g["link"][base:base + parents, 0], c0.astype(np.uint32))
        prev, count, cur = cur, parents, cur + parents
    acc.destroy()
    for b in (idx, vtx, bvh, morton):
        b.free()


@pytest.mark.parametrize("fused", [True, False])
def test_cpp_host_dynamic_scene_matches_oracle(device, oracle, fused):
    """Settings::DynamicScene of the C++ host: the mesh moves on the device before every sample cycle."""
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.capi import aabb_dtype
    n, w, h, depth, cycles = 5000, 64, 36, 3, 3
    sc = _scene(n)
    hp = host_py.HostPipeline(device, n, w, h, sc["camera"], fused_build=fused, seed_raygen=1, seed_trace=2, dynamic=True)
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, rebuild_every_cycle=True)
    count = oracle.load().orc_bvh_node_count(n)
    for frame in range(cycles):
        op.set_vertices(oracle.synthetic_animate(sc, frame)["vertices"])
        hp.render(depth)
        op.render(depth)
        assert hp.read_buffer("bvh", aabb_dtype, count).tobytes() == op.bvh().tobytes(), "hierarchy differs in frame %d" % frame
    assert hp.result().view(np.uint32).tobytes() == op.result().view(np.uint32).tobytes()
    assert hp.num_samples() == cycles + 1
    hp.close()
    op.close()


def test_checkpoint_and_resume(device, oracle, tmp_path):
    """SaveState / LoadState of the C++ host: the running mean and the sample counter survive; a state is only taken
    between two sample cycles, and a state of another image size or a file that is no state is refused. (That the
    resumed run continues the random sequences: tests/test_gpu_round2.py::test_resume_continues_the_random_sequences.)"""
    from gpuraytracer_amd import host_py
    n, w, h, depth = 4000, 48, 27, 3
    sc = _scene(n)
    path = tmp_path / "state.bin"
    a = host_py.HostPipeline(device, n, w, h, sc["camera"], seed_raygen=1, seed_trace=2)
    a.render(depth)
    with pytest.raises(RuntimeError):
        a.render(1)
        a.save_state(path)                    # in the middle of a sample cycle: refused
    a.render(depth - 1)
    a.save_state(path)
    saved, samples = a.result(), a.num_samples()
    assert samples == 3
    a.close()

    b = host_py.HostPipeline(device, n, w, h, sc["camera"], seed_raygen=1, seed_trace=2)
    b.load_state(path)
    assert b.num_samples() == samples
    assert b.result().view(np.uint32).tobytes() == saved.view(np.uint32).tobytes()
    b.render(depth)
    assert b.num_samples() == samples + 1
    # a state of another image size, and a file that is no state, are refused
    d = host_py.HostPipeline(device, n, w + 1, h, sc["camera"], seed_raygen=1, seed_trace=2)
    with pytest.raises(RuntimeError):
        d.load_state(path)
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"P6\n1 1\n255\n")
    with pytest.raises(RuntimeError):
        b.load_state(junk)
    old = tmp_path / "old.bin"
    old.write_bytes(b"GPURTSTATE1 %d %d 3\n" % (w, h) + saved.tobytes())      # the format without generator states
    with pytest.raises(RuntimeError):
        b.load_state(old)
    for p in (b, d):
        p.close()


@pytest.mark.parametrize("scale,offset", [(0.05, 0.0), (3000.0, 7000.0), (1e5, -2e5), (1.0, 65000.0)])
def test_widened_boxes_stay_exact_at_extreme_coordinates(device, oracle, scale, offset):
    """The two-level kernel walks on binary16 boxes that contain the true ones. Coordinates that binary16 holds badly
    (denormal, coarse, or beyond its range, where the boxes become infinite) may cost time, never a different image."""
    sc = dict(_small_scene(300, seed=11))
    v = sc["vertices"].copy()
    v["position"] = (v["position"].astype(np.float64) * scale + offset).astype(np.float32)
    sc["vertices"] = v
    from gpuraytracer_amd import scenes
    sc["scene_min"], sc["scene_max"] = scenes.centroid_bounds(v, sc["indices"])
    cam = dict(sc["camera"])
    cam["position"] = tuple(np.float32(c * scale + offset) for c in (0.0, 0.0, -3.0))
    cam["focus"] = tuple(np.float32(offset) for _ in range(3))
    cam["near_z"], cam["far_z"] = 0.01 * scale, 10000.0 * scale
    sc["camera"] = cam
    g = _render_both(device, oracle, sc, 64, 48, 4, depth=2)
    for name, a, b in (("tone-mapped", g[0], g[1]), ("result", g[2], g[3]), ("emitted", g[4], g[5]), ("rays", g[6], g[7])):
        assert a.view(np.uint32).tobytes() == b.view(np.uint32).tobytes(), "%s differs at scale %g" % (name, scale)
    assert g[8]["rays_hit"] > 0, "some ray must hit"
