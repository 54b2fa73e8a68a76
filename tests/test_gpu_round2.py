"""GPU tests of the full-size configurations, the image tolerance against the reference's arithmetic as written,
the margins of the default traversal kernel, and the boundary's settings (tuning, slab division, second hierarchy).

Tolerances stated here:
 * every comparison between two HIP kernels, and between a HIP kernel and the oracle using the same slab-test
   arithmetic: none (bit-identical buffers);
 * the default path (one reciprocal per ray and axis) against the reference's slab test AS WRITTEN (a division per box,
   shader/CS_TraceRays.hlsl:83-84) at 16 accumulated sample cycles: PSNR >= PSNR_MIN dB on the tone-mapped image and at
   most DIFF_MAX of the pixels differing by more than 1e-5, max-abs printed.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNEL_QUAD, KERNEL_WAVE, KERNEL_SIMPLE, KERNEL_SPATIAL = 0, 1, 2, 3
PSNR_MIN = 40.0          # dB, tone-mapped image (values in [0, 1)), default path vs the slab test as written
DIFF_MAX = 0.01          # fraction of pixels that may differ by more than 1e-5 between the two


def _golden_scene(name):
    from tests.test_oracle_kat import _load_golden_scene
    return _load_golden_scene(name)[0]


def _metrics(a, b):
    a, b = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    d = np.abs(a - b)
    mse = float((d ** 2).mean())
    psnr = 99.0 if mse == 0.0 else 10.0 * np.log10(1.0 / mse)
    frac = float(np.mean(np.any(d > 1e-5, axis=-1)))
    return psnr, frac, float(d.max())


def _seeds(oracle, cycles, depth):
    rs = oracle.mt19937(1, 0, 3 * cycles + 3).reshape(-1, 3)
    ts = oracle.mt19937(2, 0, 3 * cycles * depth + 3).reshape(-1, 3)
    return iter(rs), iter(ts)


def _gpu_render(device, oracle, sc, w, h, cycles, depth=3, rpp=1, kernel=KERNEL_SPATIAL, divide=False, use_bvh=True, use_accel=True,
                tuning=None, keep=("output",), **kw):
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.pipeline import Pipeline
    lib = device.lib
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    ri, ti = _seeds(oracle, cycles, depth)
    old = capi.TraceTuning()
    capi.check(lib.rt_device_get_trace_tuning(device.handle, C.byref(old)), "rt_device_get_trace_tuning")
    capi.check(lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
    capi.check(lib.rt_device_set_slab_division(device.handle, 1 if divide else 0), "rt_device_set_slab_division")
    if tuning is not None:
        capi.check(lib.rt_device_set_trace_tuning(device.handle, C.byref(tuning)), "rt_device_set_trace_tuning")
    gp = Pipeline(device, sc, w, h, iv, ip, rpp=rpp, depth=depth, use_bvh=use_bvh, use_accel=use_accel,
                  raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti), **kw)
    try:
        gp.render(cycles * depth)
        out = {}
        if "output" in keep:
            out["output"] = gp.output_image()
        if "result" in keep:
            out["result"] = gp.result_image()
        if "emitted" in keep:
            out["emitted"] = gp.emitted.download(np.float32, 4 * gp.local_rays).reshape(-1, 4)
        if "rays" in keep:
            out["rays"] = gp.rays.download(np.float32, 8 * gp.local_rays).reshape(-1, 8)
    finally:
        lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
        lib.rt_device_set_slab_division(device.handle, 0)
        lib.rt_device_set_trace_tuning(device.handle, C.byref(old))
        gp.close()
    return out


ALL = ("output", "result", "emitted", "rays")


def _same(a, b, what):
    for k in a:
        assert a[k].view(np.uint32).tobytes() == b[k].view(np.uint32).tobytes(), "%s: %s differs" % (what, k)


# ------------------------------------------------------------------ the slab test as the reference writes it ----

@pytest.mark.parametrize("kernel", [KERNEL_WAVE, KERNEL_SIMPLE])
@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2"])
def test_slab_division_gives_the_as_written_oracle(device, oracle, name, kernel):
    """rt_device_set_slab_division: the HIP walk that divides by the direction for every box equals, bit for bit, the
    oracle run with true_division (the reference's IntersectAABB exactly as written)."""
    sc = _golden_scene(name)
    w, h, cycles, depth = 96, 54, 3, 3
    g = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=kernel, divide=True, keep=ALL)
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, true_division=True)
    op.render(cycles * depth)
    o = dict(output=op.output(), result=op.result(), emitted=op.emitted(), rays=op.rays().view(np.float32).reshape(-1, 8))
    op.close()
    _same(g, o, "%s, as-written slab test, kernel %d" % (name, kernel))
    assert g["output"][..., :3].max() > 0.0


@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2"])
def test_default_path_against_the_as_written_oracle(device, oracle, name):
    """THE image tolerance: the default path (SPATIAL kernel, reciprocal slab test) against the CPU oracle with the
    reference's slab test as written, 16 accumulated sample cycles of the reference's assets."""
    sc = _golden_scene(name)
    w, h, cycles, depth = 240, 135, 16, 3
    g = _gpu_render(device, oracle, sc, w, h, cycles, depth)["output"]
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, true_division=True)
    op.render(cycles * depth)
    o = op.output()
    op.close()
    psnr, frac, worst = _metrics(g, o)
    print("%s %dx%d x %d cycles, default HIP path vs as-written oracle: PSNR %.1f dB, %.4f%% of pixels differ, max abs %.3g"
          % (name, w, h, cycles, psnr, 100 * frac, worst))
    assert psnr >= PSNR_MIN and frac <= DIFF_MAX


# ------------------------------------------------------------------ C2 at full size ----

@pytest.mark.parametrize("name", ["testscene2", "testobject4_2", "testobject5_1"])
def test_c2_full_size_kernels_agree_and_tolerance_holds(device, oracle, name):
    """BASELINE config 2: the reference's assets at 1920x1080, 16 accumulated sample cycles of 3 bounces.
    SPATIAL and QUAD (the walk in the reference's order): identical buffers. Default path vs the as-written slab test
    (HIP walk with the division, itself bit-exact against the oracle at small sizes): the stated tolerance.
    BVH on vs off (ties between equal hit distances go to different triangles): reported, loosely bounded."""
    sc = _golden_scene(name)
    w, h, cycles, depth = 1920, 1080, 16, 3
    a = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=KERNEL_SPATIAL, keep=ALL)
    b = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=KERNEL_QUAD, keep=ALL)
    _same(a, b, "%s 1080p x16: SPATIAL vs QUAD" % name)
    d = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=KERNEL_WAVE, divide=True)
    psnr, frac, worst = _metrics(a["output"], d["output"])
    print("%s 1080p x16, default vs as-written slab test: PSNR %.1f dB, %.4f%% of pixels differ, max abs %.3g" % (name, psnr, 100 * frac, worst))
    assert psnr >= PSNR_MIN and frac <= DIFF_MAX
    n = _gpu_render(device, oracle, sc, w, h, cycles, depth, use_bvh=False)
    psnr, frac, worst = _metrics(a["output"], n["output"])
    print("%s 1080p x16, BVH vs NoBVH: PSNR %.1f dB, %.4f%% of pixels differ, max abs %.3g" % (name, psnr, 100 * frac, worst))
    assert psnr >= 30.0 and frac <= 0.05
    assert a["output"][..., :3].max() > 0.0


def test_testscene2_has_the_structure_of_the_reference_screenshot(device):
    """The only output the reference holds, results/NoBVH.png (testscene2 without a hierarchy, 1080p, camera of
    src/Settings.h:22-25), as 16 x 16 block means of linear RGB (tests/golden/ref_nobvh_lowfreq.npz, made by
    scripts/make_nobvh_fixture.py). Seeds and sample count of the screenshot are unknown, so this is qualitative: the
    luminance of our render, encoded to 8-bit sRGB as the reference presents it, correlates with the screenshot's block
    by block, and the channel means agree once the running mean's k / (k + 1) factor is taken out (the screenshot's
    noise level and brightness fit k = 3 accumulated samples). First check that does not come from reading the HLSL."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("compare_nobvh_png", os.path.join(ROOT, "scripts", "compare_nobvh_png.py"))
    cmp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmp)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_nobvh_lowfreq.npz"))["rgb"]
    for use_bvh, cycles in ((False, 3), (True, 3), (True, 12)):
        corr, log_corr, ratio = cmp.compare(cmp.low_frequency(cmp.render(device, cycles, use_bvh)), ref)
        unscaled = ratio / ((cycles / (cycles + 1.0)) / 0.75)
        print("testscene2 %s, %d cycles vs results/NoBVH.png: luminance correlation %.4f (log %.4f), channel mean ratios %s, "
              "without the k/(k+1) factors %s" % ("BVH" if use_bvh else "NoBVH", cycles, corr, log_corr, np.round(ratio, 3), np.round(unscaled, 3)))
        assert corr >= 0.97 and log_corr >= 0.95
        assert np.all(np.abs(unscaled - 1.0) <= 0.1)


# ------------------------------------------------------------------ C4: rows dealt to 8 pseudo-ranks ----

def test_c4_rows_dealt_to_eight_ranks_assemble_to_the_unsharded_image(device, oracle):
    """BASELINE config 4 on one GPU: the 10M-triangle scene at 3840x2160, 4 rays per pixel, 3 accumulated sample cycles
    of 3 bounces (towards the configuration's 64 samples per pixel: rays per pixel x cycles), traced once whole and once
    as eight pipelines that each hold the rows r, r + 8, ... (replicated scene, own rebuild): the rows, put back in
    place, are the unsharded image bit for bit."""
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.distributed import rows_for_rank
    sc = host_py.synthetic_scene(10_000_000)
    w, h, depth, rpp, world, cycles = 3840, 2160, 3, 4, 8, 3
    full = _gpu_render(device, oracle, sc, w, h, cycles, depth, rpp=rpp, keep=("output", "result"))
    parts = {k: np.zeros_like(v) for k, v in full.items()}
    for r in range(world):
        first, rows, stride = rows_for_rank(h, world, r)
        got = _gpu_render(device, oracle, sc, w, h, cycles, depth, rpp=rpp, keep=("output", "result"), first_row=first, num_rows=rows,
                          row_stride=stride)
        for k in parts:
            parts[k][r::world] = got[k]
    _same(full, parts, "C4 dealt to 8 ranks")
    assert full["output"][..., :3].max() > 0.0


# ------------------------------------------------------------------ C5: 50M moving triangles ----

def test_c5_fifty_million_moving_triangles(device, oracle):
    """BASELINE config 5: the 50M-triangle scene, moved on the device and rebuilt for two consecutive frames. The build
    is checked through properties (the oracle would take minutes): 50 000 007 nodes; the sorted records are a
    permutation in (code, triangle) order, i.e. the stable sort; codes of a sample recomputed by the oracle from the
    moved vertices and the device's bounds; leaves hold the bounds of the sorted pairs; every node is the union of its
    children. A reduced image is then traced by SPATIAL and QUAD to identical buffers."""
    from gpuraytracer_amd import capi, host_py
    from gpuraytracer_amd.capi import aabb_dtype, vertex_dtype
    from gpuraytracer_amd.runtime import Accel
    lib = device.lib
    n = 50_000_000
    sc = host_py.synthetic_scene(n)
    nodes = lib.rt_bvh_node_count(n)
    assert nodes == 50000007
    idx, base, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"]), device.alloc(48 * 3 * n)
    bounds, bvh, morton = device.alloc(32), device.alloc(32 * nodes), device.alloc(16 * n)
    acc = Accel(device, n)
    rng = np.random.RandomState(9)
    leaves = (n + 1) // 2
    for frame in (0, 1):
        capi.check(lib.rt_synthetic_animate(device.handle, n, frame, base.handle, vtx.handle), "rt_synthetic_animate")
        acc.pack_bounds(n, idx, vtx, bounds)
        acc.build_device_bounds(bounds, bvh, morton)
        device.sync()
        b = bounds.download(np.float32, 8)
        m = morton.download(oracle.uint4_dtype, n)
        code, ci = m["x"].copy(), m["y"].copy()
        del m
        assert np.all(code[1:] >= code[:-1]), "codes are not ascending (frame %d)" % frame
        ties = code[1:] == code[:-1]
        assert np.all(ci[1:][ties] > ci[:-1][ties]), "equal codes must keep ascending triangle order"
        del ties
        assert np.all(ci % 3 == 0)
        seen = np.zeros(n, dtype=np.uint8)
        seen[ci // 3] = 1
        assert seen.all(), "not a permutation of the triangles"
        del seen
        # a sample of triangles: moved vertices and bounds from the device, codes from the oracle's Morton function
        allv = vtx.download(vertex_dtype, 3 * n)
        sample = np.sort(rng.choice(n, 100000, replace=False)).astype(np.int64)
        moved = np.ascontiguousarray(allv[(3 * sample[:, None] + np.arange(3)[None, :]).reshape(-1)])
        assert np.abs(moved["position"] - sc["vertices"]["position"][(3 * sample[:, None] + np.arange(3)[None, :]).reshape(-1)]).max() <= 0.2501
        omi = oracle.MortonInfo()
        for k in range(3):
            omi.scene_min[k], omi.scene_max[k] = float(b[k]), float(b[4 + k])
        omi.num_primitives = len(sample)
        o_m = np.zeros(len(sample), dtype=oracle.uint4_dtype)
        o_f = np.zeros(256, dtype=oracle.uint4_dtype)
        sidx = np.arange(3 * len(sample), dtype=np.uint32)
        oracle.load().orc_generate_morton_codes(C.byref(omi), oracle.ptr(sidx), oracle.ptr(moved), oracle.ptr(o_m), oracle.ptr(o_f))
        code_of_tri = np.empty(n, dtype=np.uint32)
        code_of_tri[ci // 3] = code
        assert np.array_equal(code_of_tri[sample], o_m["x"]), "Morton codes differ from the oracle's on the sample"
        del code_of_tri
        # the centroid bounds the codes are scaled by: min / max over ((p1 + p2) + p3) * (1/3), as the loader's host loop
        pos = allv["position"].reshape(n, 3, 3)
        del allv
        cen = ((pos[:, 0] + pos[:, 1]) + pos[:, 2]) * np.float32(1.0 / 3.0)
        assert np.array_equal(b[0:3], cen.min(axis=0)) and np.array_equal(b[4:7], cen.max(axis=0)), "scene bounds differ"
        del cen
        g = bvh.download(aabb_dtype, nodes)
        # leaves of a sample of leaf indices: bounds of the two sorted triangles' (moved) corners
        ls = np.sort(rng.choice(leaves, 50000, replace=False)).astype(np.int64)
        t0, t1 = ci[2 * ls] // 3, ci[np.minimum(2 * ls + 1, n - 1)] // 3
        lo3 = np.minimum(pos[t0].min(axis=1), pos[t1].min(axis=1))
        hi3 = np.maximum(pos[t0].max(axis=1), pos[t1].max(axis=1))
        del pos
        assert np.array_equal(g["min"][1 + ls], lo3) and np.array_equal(g["max"][1 + ls], hi3), "leaf boxes differ"
        assert np.array_equal(g["link"][1:1 + leaves, 0], ci[0::2] | 0x80000000)
        assert np.array_equal(g["link"][1:1 + leaves - 1, 1], ci[1::2][:leaves - 1] | 0x80000000)
        prev, count, cur = 1, leaves, 1 + leaves
        while count > 1:
            parents = (count + 1) // 2
            at = 0 if parents == 1 else cur
            c0 = prev + 2 * np.arange(parents)
            c1 = np.minimum(c0 + 1, prev + count - 1)
            assert np.array_equal(g["min"][at:at + parents], np.minimum(g["min"][c0], g["min"][c1]))
            assert np.array_equal(g["max"][at:at + parents], np.maximum(g["max"][c0], g["max"][c1]))
            assert np.array_equal(g["link"][at:at + parents, 0], c0.astype(np.uint32))
            prev, count, cur = cur, parents, cur + parents
        assert cur == nodes + 1          # the walk above counted the trunk, which is stored at node 0
        del g, code, ci
    acc.destroy()
    for x in (idx, base, vtx, bounds, bvh, morton):
        x.free()
    # a reduced image of the moving scene, two frames, both walks
    w, h, depth = 480, 270, 4
    a = _gpu_render(device, oracle, sc, w, h, 2, depth, kernel=KERNEL_SPATIAL, keep=ALL, dynamic=True)
    q = _gpu_render(device, oracle, sc, w, h, 2, depth, kernel=KERNEL_QUAD, keep=ALL, dynamic=True)
    _same(a, q, "C5 480x270: SPATIAL vs QUAD")
    assert a["output"][..., :3].max() > 0.0
    del a, q
    # The configuration's own frame, 7680x4320 and 4 bounces, by the default kernel - and every 64th row of it alone
    # (rt_image_rows) by the walk in the reference's order, which would take minutes for the whole frame. Seeds and pixel
    # coordinates are those of the whole image, so the half million rays of those rows must equal the same rows of the
    # full frame bit for bit.
    w, h, depth, first, stride = 7680, 4320, 4, 17, 64
    keep = ("output", "emitted", "rays")
    rows = len(range(first, h, stride))
    part = _gpu_render(device, oracle, sc, w, h, 1, depth, kernel=KERNEL_QUAD, keep=keep, dynamic=True, first_row=first, num_rows=rows,
                       row_stride=stride)
    full = _gpu_render(device, oracle, sc, w, h, 1, depth, kernel=KERNEL_SPATIAL, keep=keep, dynamic=True)
    assert full["output"].shape[0] == h and part["output"].shape[0] == rows
    assert full["output"][first::stride].tobytes() == part["output"].tobytes(), "C5 at 8K: tone-mapped rows differ"
    assert full["rays"].reshape(h, w, 8)[first::stride].tobytes() == part["rays"].tobytes(), "C5 at 8K: rays differ"
    assert full["emitted"].reshape(h, w, 4)[first::stride].tobytes() == part["emitted"].tobytes(), "C5 at 8K: emitted light differs"
    assert part["output"][..., :3].max() > 0.0


# ------------------------------------------------------------------ what the default kernel's two margins are for ----

def test_sliver_room_decides_the_false_hit(device):
    """The room around slivers in the library's own leaf boxes (default: 1/32 of the longest edge): with it every scene
    of the sliver case agrees with the walk in the reference's order; without it (pad 0) or with too little (1/1024)
    the default kernel loses the hit the reference finds in at least one scene."""
    from gpuraytracer_amd import capi
    from tests.test_gpu_parity import _sliver_scenes
    lib = device.lib
    differing = {}
    try:
        for pad in (0.0, 1.0 / 1024.0, 1.0 / 32.0, 1.0 / 8.0):
            capi.check(lib.rt_device_set_walk_margins(device.handle, 1.00390625, pad), "rt_device_set_walk_margins")
            outs, _ = _sliver_scenes(device)
            differing[pad] = sum(int(a.view(np.uint32).tobytes() != b.view(np.uint32).tobytes()) for a, b in outs)
    finally:
        lib.rt_device_set_walk_margins(device.handle, 1.00390625, 1.0 / 32.0)
    print("scenes (of 12) in which the two walks differ, by sliver pad:", differing)
    assert differing[1.0 / 32.0] == 0 and differing[1.0 / 8.0] == 0
    assert differing[0.0] > 0, "without the room the false hit beside the sliver must be lost"
    assert lib.rt_device_set_walk_margins(device.handle, 0.5, 0.1) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_device_set_walk_margins(device.handle, 1.0, 2.0) == capi.RT_ERR_INVALID_ARGUMENT


def test_entry_margin_decides_hits_behind_rounded_boxes(device):
    """Boxes are entered up to reach * (1 + 2^-8): a hit distance and the entry distance of a flat box around the
    triangle are two roundings of the same number. On axis-aligned quads of a grid (overlapping coplanar triangles, flat
    boxes) a margin of 1 loses hits the reference finds; the default margin loses none."""
    import importlib.util
    from gpuraytracer_amd import capi
    spec = importlib.util.spec_from_file_location("fuzz_walks", os.path.join(ROOT, "scripts", "fuzz_walks.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    lib = device.lib
    os.environ["FUZZ_KINDS"], os.environ["FUZZ_MAX_EXP"] = "quads,inplane", "4.3"
    bad = {}
    try:
        for margin in (1.0, 1.00390625):
            capi.check(lib.rt_device_set_walk_margins(device.handle, margin, 1.0 / 32.0), "rt_device_set_walk_margins")
            bad[margin], _ = fuzz.run(device, 24, 1, verbose=False)
    finally:
        os.environ.pop("FUZZ_KINDS"), os.environ.pop("FUZZ_MAX_EXP")
        lib.rt_device_set_walk_margins(device.handle, 1.00390625, 1.0 / 32.0)
    print("grid-quad scenes (of 24) in which the two walks differ, by entry margin:", bad)
    assert bad[1.00390625] == 0
    assert bad[1.0] > 0, "without the margin hits behind boxes that are farther by a rounding must be lost"


# ------------------------------------------------------------------ settings of the boundary ----

def test_every_tuning_knob_keeps_the_oracle_image(device, oracle):
    """rt_device_set_trace_tuning (and the GPURT_* variables that initialise it): every knob at a non-default value,
    one at a time and all together, for the default kernel and the walk in the reference's order - the buffers stay the
    oracle's."""
    from gpuraytracer_amd import capi, scenes
    sc = scenes.synthetic_scene(20000)
    w, h, cycles, depth = 128, 72, 2, 3
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2)
    op.render(cycles * depth)
    want = dict(output=op.output(), result=op.result(), emitted=op.emitted(), rays=op.rays().view(np.float32).reshape(-1, 8))
    op.close()
    base = capi.TraceTuning()
    capi.check(device.lib.rt_device_get_trace_tuning(device.handle, C.byref(base)), "rt_device_get_trace_tuning")
    knobs = dict(leaf_threshold=5, refill_threshold=40, blocks_per_cu=2, sort_rays=0, deal_shift=9, cost_bits=9, cost_depth=7,
                 cost_stride=3, cost_chord=0, key_passes=1, stuck_threshold=1, retire_us=7)
    cases = [{k: v} for k, v in knobs.items()] + [dict(knobs), dict(cost_chord=1, key_passes=3), dict(leaf_threshold=64, refill_threshold=1),
                                                            dict(sort_rays=1), dict(sort_rays=2), dict(stuck_threshold=64), dict(stuck_threshold=9, leaf_threshold=50),
                                                            dict(retire_us=0), dict(retire_us=1000), dict(retire_us=200, blocks_per_cu=3),
                                                            # (the probe on the library's own hierarchy reads its sibling-pair records, for which all
                                                            # of its node levels are made on demand: builds store only the levels the walk needs)
                                                            dict(sort_rays=1, cost_depth=7, cost_chord=0), dict(sort_rays=1, cost_depth=3, cost_stride=1, cost_chord=0)]
    for case in cases:
        t = capi.TraceTuning()
        C.memmove(C.byref(t), C.byref(base), C.sizeof(t))
        for k, v in case.items():
            setattr(t, k, v)
        for kernel in (KERNEL_SPATIAL, KERNEL_QUAD, KERNEL_WAVE):
            g = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=kernel, tuning=t, keep=ALL)
            _same(g, want, "tuning %s, kernel %d" % (case, kernel))
    # out-of-range values are clamped, not refused
    t = capi.TraceTuning(*([5000] * 12))
    capi.check(device.lib.rt_device_set_trace_tuning(device.handle, C.byref(t)), "rt_device_set_trace_tuning")
    got = capi.TraceTuning()
    device.lib.rt_device_get_trace_tuning(device.handle, C.byref(got))
    assert (got.leaf_threshold, got.blocks_per_cu, got.deal_shift, got.cost_bits, got.key_passes, got.sort_rays, got.stuck_threshold,
            got.retire_us) == (64, 8, 20, 12, 3, 1, 64, 1000)
    device.lib.rt_device_set_trace_tuning(device.handle, C.byref(base))


def test_second_hierarchy_is_an_accel_property(device, oracle):
    """rt_accel_set_second_hierarchy: an accel built without the library's own hierarchy is traced by the walk in the
    reference's order (no ray is ever listed), one built with coarser Morton cells by the default kernel - same buffers."""
    from gpuraytracer_amd import capi, scenes
    from gpuraytracer_amd.pipeline import Pipeline
    sc = scenes.synthetic_scene(30000)
    w, h, depth = 96, 54, 3
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    outs = []
    for enable, bits in ((1, 99), (0, 21), (1, 0), (1, 6)):
        ri, ti = _seeds(oracle, 2, depth)
        gp = Pipeline(device, sc, w, h, iv, ip, depth=depth, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
        capi.check(device.lib.rt_accel_set_second_hierarchy(gp.accel.handle, enable, bits), "rt_accel_set_second_hierarchy")
        gp.render(2 * depth)
        outs.append((gp.result_image(), gp.rays.download(np.float32, 8 * w * h)))
        gp.close()
    for o in outs[1:]:
        assert o[0].view(np.uint32).tobytes() == outs[0][0].view(np.uint32).tobytes()
        assert o[1].view(np.uint32).tobytes() == outs[0][1].view(np.uint32).tobytes()
    assert device.lib.rt_accel_set_second_hierarchy(None, 1, 21) == capi.RT_ERR_INVALID_ARGUMENT


def test_trace_rejects_rows_that_do_not_fit_the_image(device, oracle):
    """rt_trace_rays checks the rows it is given against the image, as rt_camera_ray_generation does, and the ray count
    against the image when no rows are given: inconsistent arguments are an error, not an out-of-bounds write."""
    from gpuraytracer_amd import capi, scenes
    from gpuraytracer_amd.pipeline import Pipeline
    sc = scenes.synthetic_scene(2000)
    w, h = 64, 36
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    gp = Pipeline(device, sc, w, h, iv, ip)
    gp.render(1)
    ok_rows = gp.rows
    try:
        for bad in (capi.ImageRows(30, 36, 1), capi.ImageRows(0, 36, 2), capi.ImageRows(0, 36, 0), capi.ImageRows(37, 0, 1)):
            gp.rows = bad
            with pytest.raises(capi.GpurtError) as e:
                gp.trace()
            assert e.value.code == capi.RT_ERR_INVALID_ARGUMENT
        gp.rows = ok_rows
        gp.tr.num_rays = w * h - 1                     # whole image, wrong ray count
        args = (gp.indices.handle, gp.vertices.handle, gp.materials.handle, gp.texture_ids.handle, gp.atlas.handle, gp.atlas_elements,
                gp.rays.handle, gp.old_rays.handle, gp.ray_pixels.handle, gp.scattered.handle, gp.emitted.handle, gp.bvh.handle, None)
        assert device.lib.rt_trace_rays_accel(device.handle, C.byref(gp.tr), gp.accel.handle, *args) == capi.RT_ERR_INVALID_ARGUMENT
        gp.tr.num_rays = w * h
        assert device.lib.rt_trace_rays_accel(device.handle, C.byref(gp.tr), gp.accel.handle, *args) == capi.RT_OK
    finally:
        gp.rows = ok_rows
        gp.close()


def test_listed_rays_are_zero_after_other_kernels(device, oracle):
    """rt_device_listed_rays reports the SPATIAL kernel's list of the last trace call, and 0 after a call traced by any
    other kernel (also the brute-force and one-thread-per-ray paths, which never touch the list)."""
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.pipeline import Pipeline
    from tests.test_gpu_parity import _wall_scene
    sc = _wall_scene(20, 2)                       # flat boxes and overlapping coplanar triangles: some rays need the reference's order
    w = h = 128
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    lib = device.lib

    def listed():
        c = C.c_uint32(99)
        capi.check(lib.rt_device_listed_rays(device.handle, C.byref(c)), "rt_device_listed_rays")
        return c.value

    try:
        for use_bvh in (True, False):
            ri, ti = _seeds(oracle, 4, 2)
            gp = Pipeline(device, sc, w, h, iv, ip, depth=2, use_bvh=use_bvh, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
            counts = []
            for kernel in (KERNEL_SPATIAL, KERNEL_SPATIAL, KERNEL_SIMPLE, KERNEL_QUAD, KERNEL_WAVE, KERNEL_SPATIAL):
                capi.check(lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
                gp.render(1)
                counts.append(listed())
            gp.close()
            print("rays listed per call (bvh %s):" % use_bvh, counts)
            if use_bvh:
                assert counts[0] + counts[1] + counts[5] > 0, "coincident walls list rays"
                assert counts[2:5] == [0, 0, 0]
            else:
                assert counts == [0] * 6
    finally:
        lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)


def test_resume_continues_the_random_sequences(device, oracle, tmp_path):
    """SaveState / LoadState carry both std::mt19937 states and the frame of a moving scene: k sample cycles, a
    checkpoint, a fresh pipeline that resumes and k more cycles give the image of 2k uninterrupted cycles bit for bit."""
    from gpuraytracer_amd import host_py, scenes
    n, w, h, depth, k = 4000, 48, 27, 3, 3
    cam = scenes.synthetic_scene(n)["camera"]
    for dynamic in (False, True):
        path = tmp_path / ("state%d.bin" % dynamic)
        whole = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=1, seed_trace=2, dynamic=dynamic)
        whole.render(2 * k * depth)
        want, want_samples = whole.result(), whole.num_samples()
        whole.close()
        a = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=1, seed_trace=2, dynamic=dynamic)
        a.render(k * depth)
        a.save_state(path)
        a.close()
        b = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=77, seed_trace=78, dynamic=dynamic)     # seeds come from the file
        b.load_state(path)
        b.render(k * depth)
        assert b.num_samples() == want_samples == 2 * k + 1
        assert b.result().view(np.uint32).tobytes() == want.view(np.uint32).tobytes(), "resumed run differs (dynamic=%s)" % dynamic
        b.close()


def test_previews_between_bounces_change_no_image(device, oracle):
    """Settings::PreviewEveryBounce = false (Pipeline.render(previews=False)): the final-image kernel only after the last
    bounce of a sample cycle. Accumulated image, displayed image and sample count after whole cycles are those of the
    reference's sequence, which runs the kernel after every bounce."""
    from gpuraytracer_amd import host_py, scenes
    from gpuraytracer_amd.pipeline import Pipeline
    n, w, h, depth, cycles = 4000, 64, 36, 3, 3
    cam = scenes.synthetic_scene(n)["camera"]
    got = []
    for previews in (True, False):
        hp = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=1, seed_trace=2, previews=previews)
        hp.render(cycles * depth)
        got.append((hp.result().tobytes(), hp.output().tobytes(), hp.num_samples()))
        hp.close()
    assert got[0] == got[1] and got[0][2] == cycles + 1
    sc = scenes.synthetic_scene(n)
    iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    images = []
    for previews in (True, False):
        rs, ts = _seeds(oracle, cycles, depth)
        p = Pipeline(device, sc, w, h, iv, ip, depth=depth, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
        p.render(cycles * depth, previews=previews)
        images.append((p.result_image().tobytes(), p.output_image().tobytes()))
        p.close()
    assert images[0] == images[1]


# ------------------------------------------------------------------ multi-GPU entry points of the C-ABI ----

@pytest.mark.parametrize("world,w,h", [(1, 33, 9), (2, 64, 36), (3, 50, 31), (8, 130, 67), (16, 7, 5)])
def test_rows_interleave_puts_gathered_rows_in_image_order(device, world, w, h):
    """rt_rows_interleave: the ranks' rows one rank after the other (what the gather delivers) -> rows in image order."""
    from gpuraytracer_amd import capi
    rng = np.random.RandomState(world)
    image = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    gathered = np.concatenate([image[r::world] for r in range(world)], axis=0)
    src, dst = device.upload(gathered), device.alloc(16 * w * h)
    capi.check(device.lib.rt_rows_interleave(device.handle, src.handle, w, h, world, dst.handle), "rt_rows_interleave")
    device.sync()
    assert dst.download(np.float32, 4 * w * h).tobytes() == image.tobytes()
    assert device.lib.rt_rows_interleave(device.handle, src.handle, w, h, 0, dst.handle) == capi.RT_ERR_INVALID_ARGUMENT
    assert device.lib.rt_rows_interleave(device.handle, src.handle, w, h, world, src.handle) == capi.RT_ERR_INVALID_ARGUMENT
    small = device.alloc(16)
    assert device.lib.rt_rows_interleave(device.handle, src.handle, w, h, world, small.handle) == capi.RT_ERR_BUFFER_TOO_SMALL
    for b in (src, dst, small):
        b.free()


def test_gather_rows_over_rccl_with_one_rank(device, oracle):
    """rt_comm_unique_id / rt_comm_init / rt_gather_rows through RCCL itself with the one GPU a test box has: the rank
    sends its rows to itself (an RCCL send/receive pair) and the root puts them in image order. Also through
    RaytracerPipeline::GatherOutput of the C++ host."""
    from gpuraytracer_amd import capi, host_py, scenes
    # (PyTorch brings a ROCm stack of its own along, RCCL included; the library must bind the RCCL that belongs to the HIP
    # runtime it runs on whether PyTorch was imported before the library was loaded, after it, or not at all)
    import gpuraytracer_amd.distributed  # noqa: F401 - imports torch
    lib = device.lib
    uid = (C.c_ubyte * capi.RT_COMM_ID_BYTES)()
    capi.check(lib.rt_comm_unique_id(uid), "rt_comm_unique_id")
    comm = C.c_void_p()
    capi.check(lib.rt_comm_init(device.handle, 0, 1, uid, C.byref(comm)), "rt_comm_init")
    assert lib.rt_comm_rank(comm) == 0 and lib.rt_comm_world(comm) == 1
    w, h = 96, 54
    rows = np.random.RandomState(3).uniform(0, 1, (h, w, 4)).astype(np.float32)
    src, dst = device.upload(rows), device.alloc(16 * w * h)
    capi.check(lib.rt_gather_rows(comm, src.handle, w, h, 0, dst.handle), "rt_gather_rows")
    device.sync()
    assert dst.download(np.float32, 4 * w * h).tobytes() == rows.tobytes()
    assert lib.rt_gather_rows(comm, src.handle, w, h, 1, dst.handle) == capi.RT_ERR_INVALID_ARGUMENT      # no such root
    assert lib.rt_gather_rows(comm, src.handle, w, h, 0, None) == capi.RT_ERR_INVALID_ARGUMENT            # the root needs an image
    n = 3000
    hp = host_py.HostPipeline(device, n, w, h, scenes.synthetic_scene(n)["camera"], seed_raygen=1, seed_trace=2, device_rank=0, device_count=1)
    hp.render(3)
    hp.gather_output(comm, dst.handle)
    device.sync()
    assert dst.download(np.float32, 4 * w * h).tobytes() == hp.output().tobytes()
    hp.close()
    capi.check(lib.rt_comm_destroy(comm), "rt_comm_destroy")
    bad = C.c_void_p()
    assert lib.rt_comm_init(device.handle, 2, 2, uid, C.byref(bad)) == capi.RT_ERR_INVALID_ARGUMENT
    for b in (src, dst):
        b.free()


def test_host_pipelines_of_three_ranks_assemble_to_the_whole_image(device, oracle):
    """Settings::DeviceRank / DeviceCount of the C++ host: three RaytracerPipelines that hold the rows r, r + 3, ... of
    the image (same scene, same seeds), their outputs one after the other and rt_rows_interleave = the single-device image."""
    from gpuraytracer_amd import capi, host_py, scenes
    n, w, h, depth, world = 5000, 80, 47, 3, 3
    cam = scenes.synthetic_scene(n)["camera"]
    whole = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=1, seed_trace=2)
    whole.render(2 * depth)
    want = whole.output()
    whole.close()
    parts = []
    for r in range(world):
        hp = host_py.HostPipeline(device, n, w, h, cam, seed_raygen=1, seed_trace=2, device_rank=r, device_count=world)
        hp.render(2 * depth)
        parts.append(hp.output())
        assert parts[-1].shape[0] == len(range(r, h, world))
        hp.close()
    src, dst = device.upload(np.concatenate(parts, axis=0)), device.alloc(16 * w * h)
    capi.check(device.lib.rt_rows_interleave(device.handle, src.handle, w, h, world, dst.handle), "rt_rows_interleave")
    device.sync()
    assert dst.download(np.float32, 4 * w * h).tobytes() == want.tobytes()
    for b in (src, dst):
        b.free()


def test_headless_cli_with_ranks_writes_the_same_image(device, tmp_path):
    """rt_headless --ranks 1 (device group over RCCL, rows gathered to the first device) writes the image of the plain
    single-device run, as PNG and PFM."""
    import subprocess
    exe = os.path.join(ROOT, "gpuraytracer_amd", "bin", "rt_headless")
    common = [exe, "--synthetic", "4000", "--width", "96", "--height", "54", "--cycles", "2"]
    a_png, a_pfm, b_png, b_pfm = (str(tmp_path / x) for x in ("a.png", "a.pfm", "b.png", "b.pfm"))
    ra = subprocess.run(common + ["--out", a_png, "--pfm", a_pfm], capture_output=True, text=True, timeout=120)
    rb = subprocess.run(common + ["--out", b_png, "--pfm", b_pfm, "--ranks", "1"], capture_output=True, text=True, timeout=120)
    assert ra.returncode == 0, ra.stdout + ra.stderr
    assert rb.returncode == 0, rb.stdout + rb.stderr
    assert "Mrays/s" in rb.stdout
    assert open(a_pfm, "rb").read() == open(b_pfm, "rb").read()
    assert open(a_png, "rb").read() == open(b_png, "rb").read()


# ------------------------------------------------------------------ dropping missed rays, merged vertices ----

def _open_scene(n, seed=5):
    """Random triangles without an enclosing box: most camera rays miss, and scattered rays leave the scene."""
    from tests.test_gpu_parity import _small_scene
    return _small_scene(n, seed=seed)


@pytest.mark.parametrize("kernel,use_bvh,sort_rays", [(KERNEL_SPATIAL, True, 1), (KERNEL_QUAD, True, 1), (KERNEL_WAVE, True, 1),
                                                        (KERNEL_SIMPLE, True, 1), (KERNEL_SPATIAL, False, 1), (KERNEL_SPATIAL, True, 0),
                                                        (KERNEL_SPATIAL, True, 2), (KERNEL_QUAD, True, 0)])
def test_dropping_missed_rays_changes_no_buffer(device, oracle, kernel, use_bvh, sort_rays):
    """rt_device_set_drop_missed_rays: rays that missed are left out of the later bounces of their sample cycle instead
    of being traced again (the reference's behaviour, shader/CS_TraceRays.hlsl:248,290-291). A ray that missed would
    miss again, so every buffer equals the oracle's, which re-traces them - with fewer rays walked."""
    from gpuraytracer_amd import capi
    sc = _open_scene(700)
    w, h, cycles, depth = 96, 54, 3, 4
    op = oracle.Pipeline(sc, w, h, depth=depth, use_bvh=use_bvh, seed_raygen=1, seed_trace=2)
    op.render(cycles * depth)
    want = dict(output=op.output(), result=op.result(), emitted=op.emitted(), rays=op.rays().view(np.float32).reshape(-1, 8))
    total = op.stats()[1]
    op.close()
    assert 0 < total["rays_hit"] < total["rays"] // 2, "the scene must let most rays miss"
    tuning = capi.TraceTuning()
    device.lib.rt_device_get_trace_tuning(device.handle, C.byref(tuning))
    tuning.sort_rays = sort_rays
    counted = {}
    for drop in (0, 1):
        stats = device.alloc(32)
        capi.check(device.lib.rt_device_set_drop_missed_rays(device.handle, drop), "rt_device_set_drop_missed_rays")
        if kernel in (KERNEL_WAVE, KERNEL_SIMPLE) or not use_bvh:
            device.lib.rt_device_set_trace_stats(device.handle, stats.handle)
        try:
            g = _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=kernel, use_bvh=use_bvh, tuning=tuning, keep=ALL)
        finally:
            device.lib.rt_device_set_drop_missed_rays(device.handle, 0)
            device.lib.rt_device_set_trace_stats(device.handle, None)
        _same(g, want, "drop_missed_rays=%d, kernel %d" % (drop, kernel))
        counted[drop] = int(stats.download(np.uint64, 4)[3])
        stats.free()
    if (kernel in (KERNEL_WAVE, KERNEL_SIMPLE) or not use_bvh) and (sort_rays or kernel == KERNEL_SIMPLE or not use_bvh):
        print("rays walked: %d re-tracing misses, %d dropping them" % (counted[0], counted[1]))
        assert counted[0] == total["rays"] and 0 < counted[1] < counted[0]


def test_dropped_rays_come_back_with_new_camera_rays_and_new_scenes(device, oracle):
    """The marks of missed rays end with the sample cycle (new camera rays) and with a rebuilt hierarchy: the moving
    scene, rebuilt every cycle, in drop mode equals the oracle over several cycles."""
    from gpuraytracer_amd import capi
    sc = _open_scene(900, seed=8)
    w, h, depth, cycles = 80, 45, 3, 3
    capi.check(device.lib.rt_device_set_drop_missed_rays(device.handle, 1), "rt_device_set_drop_missed_rays")
    try:
        g = _gpu_render(device, oracle, sc, w, h, cycles, depth, keep=ALL, dynamic=True)
    finally:
        device.lib.rt_device_set_drop_missed_rays(device.handle, 0)
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, rebuild_every_cycle=True)
    for frame in range(cycles):
        op.set_vertices(oracle.synthetic_animate(sc, frame)["vertices"])
        op.render(depth)
    want = dict(output=op.output(), result=op.result(), emitted=op.emitted(), rays=op.rays().view(np.float32).reshape(-1, 8))
    op.close()
    _same(g, want, "moving open scene in drop mode")


def test_merged_vertices_render_the_same_image(device, oracle, tmp_path):
    """A mesh loaded with vertex de-duplication (shared vertices behind a real index buffer) renders the image of the
    triangle soup the default loader makes: positions, normals, texture coordinates and materials per corner are equal."""
    from gpuraytracer_amd import host_py
    from tests.test_host_cpu import _write_shared_vertex_obj
    obj = tmp_path / "box.obj"
    _write_shared_vertex_obj(obj)
    cam = dict(position=(0.2, 0.1, 0.3), focus=(1.0, 0.4, 0.0), up=(0.0, 1.0, 0.0), fov=1.2, near_z=0.01, far_z=100.0)
    outs = []
    for dedup in (False, True):
        sc = host_py.load_scene(str(obj), camera=cam, dedup=dedup)
        for use_accel in (True, False):
            outs.append(_gpu_render(device, oracle, sc, 96, 54, 2, 3, use_accel=use_accel, keep=ALL))
    assert len(sc["vertices"]) < len(sc["indices"])
    for o in outs[1:]:
        _same(o, outs[0], "merged vertices")
    assert outs[0]["output"][..., :3].max() > 0.0


# ------------------------------------------------------------------ refit ----

@pytest.mark.parametrize("kernel", [KERNEL_SPATIAL, KERNEL_QUAD])
def test_refit_is_leaves_and_levels_over_the_kept_order(device, oracle, kernel):
    """rt_accel_refit: the moving scene built once, then refitted at every sample cycle. The oracle does the same with the
    reference's own build steps (CS_BVHBuildLeaves + CS_BVHBuild over the Morton records of the first build and the moved
    vertices): hierarchy and all buffers stay identical, frame after frame, for the default kernel (whose own hierarchy
    is refitted too) and the walk in the reference's order."""
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.capi import aabb_dtype
    from gpuraytracer_amd.pipeline import Pipeline
    from gpuraytracer_amd import scenes
    sc = scenes.synthetic_scene(6000)
    w, h, depth, cycles = 80, 45, 3, 4
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    ri, ti = _seeds(oracle, cycles, depth)
    capi.check(device.lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
    gp = Pipeline(device, sc, w, h, iv, ip, depth=depth, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti), dynamic=True, refit=True)
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, rebuild_every_cycle=2)
    try:
        first = None
        for frame in range(cycles):
            op.set_vertices(oracle.synthetic_animate(sc, frame)["vertices"])
            gp.render(depth)
            op.render(depth)
            nodes = gp.bvh.download(aabb_dtype, device.lib.rt_bvh_node_count(gp.num_prims))
            assert nodes.tobytes() == op.bvh().tobytes(), "hierarchy differs in frame %d" % frame
            if first is None:
                first = nodes.copy()
            else:
                assert np.array_equal(nodes["link"], first["link"]), "a refit keeps every link"
                assert nodes.tobytes() != first.tobytes(), "the boxes follow the triangles"
            assert gp.output_image().view(np.uint32).tobytes() == op.output().view(np.uint32).tobytes(), "image differs in frame %d" % frame
        assert gp.result_image().view(np.uint32).tobytes() == op.result().view(np.uint32).tobytes()
        assert gp.rays.download(np.float32, 8 * w * h).tobytes() == op.rays().view(np.float32).tobytes()
        if kernel == KERNEL_SPATIAL:
            # Settings::RefitBVH of the C++ host: the same frames
            from gpuraytracer_amd import host_py
            hp = host_py.HostPipeline(device, 6000, w, h, cam, depth=depth, seed_raygen=1, seed_trace=2, dynamic=True, refit=True)
            hp.render(cycles * depth)
            assert hp.result().view(np.uint32).tobytes() == op.result().view(np.uint32).tobytes()
            hp.close()
    finally:
        device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
        gp.close()
        op.close()


def test_refit_needs_a_build_of_the_same_size(device):
    from gpuraytracer_amd import capi, scenes
    from gpuraytracer_amd.runtime import Accel
    sc = scenes.synthetic_scene(300)
    n = 300
    idx, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"])
    bvh = device.alloc(32 * device.lib.rt_bvh_node_count(n))
    acc = Accel(device, n)
    acc.pack(n, idx, vtx)
    assert device.lib.rt_accel_refit(acc.handle, bvh.handle) == capi.RT_ERR_INVALID_ARGUMENT          # nothing built yet
    acc.build(sc["scene_min"], sc["scene_max"], bvh)
    small = device.alloc(64)
    assert device.lib.rt_accel_refit(acc.handle, small.handle) == capi.RT_ERR_BUFFER_TOO_SMALL
    built = bvh.download(np.uint8, 32 * device.lib.rt_bvh_node_count(n)).copy()
    acc.refit(bvh)                                                                                    # same vertices: same hierarchy
    device.sync()
    assert bvh.download(np.uint8, 32 * device.lib.rt_bvh_node_count(n)).tobytes() == built.tobytes()
    acc.pack(200, idx, vtx)
    assert device.lib.rt_accel_refit(acc.handle, bvh.handle) == capi.RT_ERR_INVALID_ARGUMENT          # another triangle count
    acc.destroy()
    for b in (idx, vtx, bvh, small):
        b.free()
