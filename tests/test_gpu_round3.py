"""GPU tests of what the default traversal kernel leaves out of its own hierarchy (outliers: triangles whose box is
a large part of the scene's), of pixel data that belongs to another device's rows, and of the full-size
configurations as far as they were open: every comparison is bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KERNEL_QUAD, KERNEL_WAVE, KERNEL_SIMPLE, KERNEL_SPATIAL = 0, 1, 2, 3


def _scene_of(pos, name, camera=None):
    """A scene from triangle corners pos[n, 3, 3]: eight materials in turn, one of them a light."""
    from gpuraytracer_amd import scenes
    from gpuraytracer_amd.capi import vertex_dtype
    pos = np.asarray(pos, dtype=np.float32)
    n = pos.shape[0]
    v = np.zeros(3 * n, dtype=vertex_dtype)
    v["position"] = pos.reshape(-1, 3)
    d1, d2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    nrm = np.cross(d1, d2).astype(np.float32)
    ln = np.sqrt((nrm * nrm).sum(axis=1, dtype=np.float32)).astype(np.float32)
    nrm = (nrm / np.where(ln == 0, np.float32(1), ln)[:, None]).astype(np.float32)
    v["normal"] = np.repeat(nrm, 3, axis=0)
    v["uv"] = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (n, 1))
    v["material_id"] = np.repeat(np.arange(n, dtype=np.uint32) % np.uint32(8), 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    atlas, ids = scenes.default_atlas()
    smin, smax = scenes.centroid_bounds(v, idx)
    cam = camera or dict(position=(0.0, 0.0, -22.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.2, near_z=0.01, far_z=10000.0)
    return dict(vertices=v, indices=idx, materials=scenes.synthetic_materials(), texture_ids=ids, atlas=atlas, scene_min=smin,
                scene_max=smax, camera=cam, name=name)


def _soup(n, rng, half=10.0, edge=None):
    e = edge if edge is not None else 20.0 / np.cbrt(n)
    c = rng.uniform(-half, half, (n, 1, 3))
    p = c + np.concatenate([np.zeros((n, 1, 3)), rng.uniform(-e, e, (n, 2, 3))], axis=1)
    return p.astype(np.float32)


def _sheets(k, rng, half=25.0):
    """k large triangles: random planes through the scene, each several times the size of the soup."""
    c = rng.uniform(-0.5 * half, 0.5 * half, (k, 1, 3))
    return (c + rng.uniform(-half, half, (k, 3, 3))).astype(np.float32)


def _render(device, oracle, sc, w, h, cycles, depth, kernel, **kw):
    from tests.test_gpu_round2 import ALL, _gpu_render
    return _gpu_render(device, oracle, sc, w, h, cycles, depth, kernel=kernel, keep=ALL, **kw)


def _oracle(oracle, sc, w, h, cycles, depth, **kw):
    op = oracle.Pipeline(sc, w, h, depth=depth, seed_raygen=1, seed_trace=2, **kw)
    op.render(cycles * depth)
    o = dict(output=op.output(), result=op.result(), emitted=op.emitted(), rays=op.rays().view(np.float32).reshape(-1, 8))
    op.close()
    return o


def _outliers(device, sc):
    """(found, listed) of a fused build of the scene."""
    from gpuraytracer_amd.runtime import Accel
    n = len(sc["indices"]) // 3
    idx, vtx = device.upload(sc["indices"]), device.upload(sc["vertices"])
    bvh = device.alloc(32 * device.lib.rt_bvh_node_count(n))
    acc = Accel(device, n)
    acc.pack(n, idx, vtx)
    acc.build(sc["scene_min"], sc["scene_max"], bvh)
    found, listed = C.c_uint32(0), C.c_uint32(0)
    assert device.lib.rt_accel_outliers(acc.handle, C.byref(found), C.byref(listed)) == 0
    acc.destroy()
    for b in (idx, vtx, bvh):
        b.free()
    return found.value, listed.value


@pytest.mark.parametrize("big,small", [(12, 3000), (64, 3000), (65, 3000), (200, 2000), (20, 0), (7, 1), (3, 0), (40, 5)])
def test_outliers_are_tested_from_the_list_and_change_nothing(device, oracle, big, small):
    """Scenes with `big` triangles that span the scene among `small` little ones: up to 64 of the big ones leave the
    library's own hierarchy for the list (the rest stay in it), whole scenes may consist of them, and every buffer is
    the oracle's."""
    from tests.test_gpu_round2 import _same
    rng = np.random.RandomState(100 + big + small)
    parts = [_sheets(big, rng)]
    if small:
        parts.append(_soup(small, rng))
    pos = np.concatenate(parts)
    pos = pos[rng.permutation(len(pos))]
    sc = _scene_of(pos, "sheets-%d-%d" % (big, small))
    found, listed = _outliers(device, sc)
    assert listed == min(found, 64) and found >= min(big, 3), (found, listed)
    if small >= 1000:
        assert found <= big + 4            # the little triangles are none
    w, h, cycles, depth = 96, 54, 2, 3
    want = _oracle(oracle, sc, w, h, cycles, depth)
    for kernel in (KERNEL_SPATIAL, KERNEL_QUAD):
        _same(_render(device, oracle, sc, w, h, cycles, depth, kernel), want, "%d sheets + %d, kernel %d" % (big, small, kernel))
    if big + small >= 8:                   # (fewer triangles than materials: there may be no light among them)
        assert want["output"][..., :3].max() > 0.0


def test_synthetic_scene_lists_its_twelve_wall_triangles(device):
    from gpuraytracer_amd import scenes
    assert _outliers(device, scenes.synthetic_scene(20000)) == (12, 12)
    assert _outliers(device, scenes.synthetic_scene(15)) == (12, 12)


def test_flat_scene_has_no_outliers_by_area_zero(device, oracle):
    """All centroids in one plane and on one line: the box of the centroids has no surface, so nothing is an outlier."""
    from tests.test_gpu_round2 import _same
    rng = np.random.RandomState(5)
    n = 500
    c = np.zeros((n, 1, 3), dtype=np.float32)
    c[:, 0, 0] = rng.uniform(-5, 5, n)
    tri = np.array([[0, -0.2, -0.2], [0, 0.2, -0.2], [0, 0.0, 0.2]], dtype=np.float32)      # same shape: centroids on the x axis
    sc = _scene_of(c + tri[None], "line", camera=dict(position=(0.0, 0.0, -12.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.2,
                                                   near_z=0.01, far_z=10000.0))
    assert _outliers(device, sc) == (0, 0)
    w, h, cycles, depth = 64, 36, 1, 3
    _same(_render(device, oracle, sc, w, h, cycles, depth, KERNEL_SPATIAL), _oracle(oracle, sc, w, h, cycles, depth), "line of triangles")


@pytest.mark.parametrize("kernel", [KERNEL_SPATIAL, KERNEL_SIMPLE])
def test_pixel_data_of_other_rows_still_scatters_the_ray(device, oracle, kernel):
    """A ray whose pixel data points outside the rows a device holds: the reference's out-of-range buffer access reads zeros
    and drops the light writes, but the ray is scattered like any other (shader/CS_TraceRays.hlsl:245-291)."""
    from gpuraytracer_amd import capi, scenes
    from gpuraytracer_amd.pipeline import Pipeline
    sc = scenes.synthetic_scene(3000)
    w, h = 64, 40
    cam = sc["camera"]
    iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    capi.check(device.lib.rt_device_set_trace_kernel(device.handle, kernel), "rt_device_set_trace_kernel")
    try:
        def run(tamper):
            seeds = iter([(11, 12, 13), (21, 22, 23)])
            gp = Pipeline(device, sc, w, h, iv, ip, depth=3, raygen_seeds=lambda: next(seeds), trace_seeds=lambda: next(seeds),
                          first_row=0, num_rows=h // 2)
            gp.build()
            gp.camera_rays()
            n = gp.local_rays
            px = gp.ray_pixels.download(np.uint32, n)
            if tamper is not None:
                px[tamper] = (px[tamper] & np.uint32(0xffff0000)) | np.uint32(h - 1)      # a row of the other half
                gp.ray_pixels.upload(px)
            gp.trace()
            out = dict(rays=gp.rays.download(np.float32, 8 * n).reshape(-1, 8), old=gp.old_rays.download(np.float32, 8 * n).reshape(-1, 8),
                       s=gp.scattered.download(np.float32, 4 * n).reshape(-1, 4), e=gp.emitted.download(np.float32, 4 * n).reshape(-1, 4))
            gp.close()
            return out
        plain = run(None)
        victims = np.arange(5, w * (h // 2), 97)
        odd = run(victims)
    finally:
        device.lib.rt_device_set_trace_kernel(device.handle, KERNEL_SPATIAL)
    assert (plain["rays"][victims, 3:6] != 0).any()                  # they were hit and scattered in the plain run
    for k in ("rays", "old"):
        assert odd[k].tobytes() == plain[k].tobytes(), k + ": the ray itself does not depend on where its light goes"
    keep = np.ones(len(plain["s"]), dtype=bool)
    keep[victims] = False
    assert odd["s"][keep].tobytes() == plain["s"][keep].tobytes() and odd["e"][keep].tobytes() == plain["e"][keep].tobytes()
    assert not odd["s"][victims].any() and not odd["e"][victims].any()      # dropped writes: still the zeros of a new buffer


def test_c1_settings_defaults_on_the_substitute_scene(device, oracle):
    """BASELINE config 1 is the reference's default build: assets/testscene1.obj at the defaults of src/Settings.h:4-20
    (1280x720, 1 ray per pixel, depth 3, AA spread 1.5, no depth of field, hierarchy on), one sample cycle. That asset is
    not part of the reference's checkout (a missing large file); assets/testscene2.obj with its own camera
    (src/Settings.h:22-25) stands in for it. The C++ host with default-constructed Settings and the Python pipeline both
    give the oracle's image, buffer for buffer."""
    from gpuraytracer_amd import host_py
    from tests.test_gpu_round2 import _golden_scene, _same
    sc = _golden_scene("testscene2")
    w, h, depth = 1280, 720, 3                     # RT_WINDOW_WIDTH, RT_WINDOW_HEIGHT, RT_MAX_RAY_DEPTH
    want = _oracle(oracle, sc, w, h, 1, depth)
    _same(_render(device, oracle, sc, w, h, 1, depth, KERNEL_SPATIAL), want, "C1 defaults, Python pipeline")
    hp = host_py.HostPipeline(device, sc, w, h, sc["camera"])      # rpp 1, depth 3, aa 1.5, dof 0, hierarchy on: the defaults
    try:
        hp.render(depth)
        assert hp.output().tobytes() == want["output"].tobytes(), "C1 defaults, C++ host: tone-mapped image differs"
        assert hp.result().tobytes() == want["result"].tobytes(), "C1 defaults, C++ host: accumulated image differs"
    finally:
        hp.close()
    assert want["output"][..., :3].max() > 0.0


def test_a_short_list_of_rays_is_walked_in_parts_with_the_same_result(device, oracle):
    """The rays the default kernel lists for the walk in the reference's order are few on a real scene (a few dozen per
    call on the reference's room, assets/testscene2.obj) and their walk is long; a short list is dealt out in parts, one
    subtree of the reference's hierarchy per lane, and put together by k_listed_merge. Always in parts, never in parts and
    by expectation (the default: in parts once a call has listed rays) - every buffer is the oracle's; and a scene of
    coincident copies, whose parts run out of room for their candidates, takes the way back to the whole walk."""
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.pipeline import Pipeline
    from tests.test_gpu_round2 import _golden_scene, _same, _seeds
    lib = device.lib
    sc = _golden_scene("testscene2")
    w, h, cycles, depth = 640, 360, 2, 3
    want = _oracle(oracle, sc, w, h, cycles, depth)
    try:
        for mode in (1, 2, 0):
            capi.check(lib.rt_device_set_listed_split(device.handle, mode), "rt_device_set_listed_split")
            _same(_render(device, oracle, sc, w, h, cycles, depth, KERNEL_SPATIAL), want, "testscene2, listed rays in parts: mode %d" % mode)
        # the list really is short and not empty there (else the three modes would have nothing to tell apart)
        capi.check(lib.rt_device_set_listed_split(device.handle, 1), "rt_device_set_listed_split")
        cam = sc["camera"]
        iv, ip = oracle.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
        ri, ti = _seeds(oracle, cycles, depth)
        gp = Pipeline(device, sc, w, h, iv, ip, depth=depth, raygen_seeds=lambda: next(ri), trace_seeds=lambda: next(ti))
        listed = 0
        for _ in range(cycles * depth):
            gp.render(1)
            total, during, whole = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            capi.check(lib.rt_device_listed_rays_detail(device.handle, C.byref(total), C.byref(during), C.byref(whole)), "rt_device_listed_rays_detail")
            assert total.value <= 512 and whole.value <= total.value
            listed += total.value
        gp.close()
        assert listed > 0, "testscene2 lists rays at this size"
        # twelve coincident copies of a soup in front of a wall: equal distances, flat boxes - parts with more candidates than
        # they have room for
        rng = np.random.RandomState(5)
        soup = _soup(40, rng, half=4.0, edge=3.0)
        wall = np.array([[[-30, -30, 9], [30, -30, 9], [30, 30, 9]], [[-30, -30, 9], [30, 30, 9], [-30, 30, 9]]], dtype=np.float32)
        crowd = _scene_of(np.concatenate([np.tile(soup, (12, 1, 1)), wall]), "coincident-copies")
        want2 = _oracle(oracle, crowd, 160, 90, 2, 3)
        overflowed = 0
        for mode in (1, 2):
            capi.check(lib.rt_device_set_listed_split(device.handle, mode), "rt_device_set_listed_split")
            _same(_render(device, oracle, crowd, 160, 90, 2, 3, KERNEL_SPATIAL), want2, "coincident copies, mode %d" % mode)
            if mode == 1:
                whole = C.c_uint32(0)
                lib.rt_device_listed_rays_detail(device.handle, C.byref(C.c_uint32(0)), C.byref(C.c_uint32(0)), C.byref(whole))
                overflowed = whole.value
        print("coincident copies: rays of the last call walked whole after their parts ran out of room:", overflowed)
    finally:
        lib.rt_device_set_listed_split(device.handle, 0)


def test_an_accel_too_large_for_the_second_hierarchy_says_so_when_asked(device):
    """Beyond 2^26 leaf slots (134 M triangles) an accel is created without the library's own hierarchy and traced by the
    walk in the reference's order: rt_accel_set_second_hierarchy(enable) on it returns RT_ERR_LIMIT_EXCEEDED (switching it off
    is fine), on an accel that has room RT_OK. Creation only: nothing is built."""
    from gpuraytracer_amd import capi
    lib = device.lib
    big, small = C.c_void_p(), C.c_void_p()
    capi.check(lib.rt_accel_create(device.handle, 140_000_000, C.byref(big)), "rt_accel_create")
    try:
        assert lib.rt_accel_set_second_hierarchy(big, 1, 21) == capi.RT_ERR_LIMIT_EXCEEDED
        assert lib.rt_accel_set_second_hierarchy(big, 0, 21) == capi.RT_OK
    finally:
        lib.rt_accel_destroy(big)
    capi.check(lib.rt_accel_create(device.handle, 1000, C.byref(small)), "rt_accel_create")
    try:
        assert lib.rt_accel_set_second_hierarchy(small, 1, 21) == capi.RT_OK
    finally:
        lib.rt_accel_destroy(small)


def test_headless_cli_refuses_more_ranks_than_the_machine_has_gpus(device):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuraytracer_amd", "bin", "rt_headless")
    n = C.c_int(0)
    assert device.lib.rt_device_count(C.byref(n)) == 0 and n.value >= 1
    r = subprocess.run([exe, "--synthetic", "100", "--width", "32", "--height", "18", "--ranks", str(n.value + 1)], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 2 and "this machine shows %d" % n.value in r.stdout, r.stdout + r.stderr


def test_gather_rows_between_two_gpus(device):
    """rt_comm_init from two host threads of one process (one per GPU) and rt_gather_rows of an image whose height is odd,
    so that the ranks hold different numbers of rows: the root's image is the rows in image order. Needs two GPUs (the
    test boxes have one: skipped there; every wait is bounded so that a failure cannot hang the run)."""
    import threading
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.runtime import Device
    lib = device.lib
    n = C.c_int(0)
    assert lib.rt_device_count(C.byref(n)) == 0
    if n.value < 2:
        pytest.skip("one GPU on this machine")
    world, w, h = 2, 64, 45
    image = np.random.RandomState(4).uniform(0, 1, (h, w, 4)).astype(np.float32)
    uid = (C.c_ubyte * capi.RT_COMM_ID_BYTES)()
    capi.check(lib.rt_comm_unique_id(uid), "rt_comm_unique_id")
    got, errors = {}, []

    def rank(r):
        try:
            dev = Device(r)
            comm = C.c_void_p()
            capi.check(lib.rt_comm_init(dev.handle, r, world, uid, C.byref(comm)), "rt_comm_init")
            rows = capi.ImageRows()
            capi.check(lib.rt_rows_for_rank(h, world, r, C.byref(rows)), "rt_rows_for_rank")
            assert (rows.first_row, rows.row_stride, rows.num_rows) == (r, world, len(range(r, h, world)))
            src = dev.upload(np.ascontiguousarray(image[r::world]))
            dst = dev.alloc(16 * w * h) if r == 0 else None
            capi.check(lib.rt_gather_rows(comm, src.handle, w, h, 0, dst.handle if dst else None), "rt_gather_rows")
            dev.sync()
            if r == 0:
                got["image"] = dst.download(np.float32, 4 * w * h).reshape(h, w, 4)
            lib.rt_comm_destroy(comm)
        except Exception as e:          # noqa: BLE001 - reported by the main thread
            errors.append("rank %d: %r" % (r, e))

    threads = [threading.Thread(target=rank, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive(), "a rank did not come back from the collective"
    assert not errors, errors
    assert got["image"].tobytes() == image.tobytes()


def test_reference_screenshot_pins_the_picture_to_the_pixel_and_the_sample_count(device):
    """results/NoBVH.png of the reference (testscene2 without a hierarchy, 1920 x 1080 window in a 1922 x 1081 screenshot), as
    the profiles and noise energies of tests/golden/ref_nobvh_detail.npz (scripts/make_nobvh_fixture.py; data, not source).
    Seeds are unknown, so pixels cannot be compared, but two things can:
     * where the picture lies. Column means of eight horizontal bands line up with ours at a shift of exactly 0 pixels
       (correlation of their first differences ~0.5 there, below 0.05 one pixel to either side); row means of 64 vertical
       bands at -1: the screenshot's last two rows are window border and its first row is the picture's second (the capture
       is one row off). That pins the flipped, un-centred pixel-to-NDC map (shader/CS_CameraRayGeneration.hlsl:73-77), the
       column-major reading of the matrices and the camera of src/Settings.h:22-25 to a pixel - no transposed or mirrored
       variant of them survives a shift of one.
     * how many samples it holds. The pixel noise (Laplacian energy over squared brightness, median of the 16 x 16
       blocks) falls with the samples per pixel, the brightness follows the running mean's k / (k + 1) (k sample cycles,
       shader/CS_GenerateFinalImage.hlsl:28-60). One ray per pixel explains only one of the two: at k = 3, where the brightness
       fits, the noise is three times the screenshot's; at k = 12, where the noise fits, the picture is a quarter too bright.
       Three sample cycles of three to four rays per pixel give both: the screenshot's noise lies between those two renders',
       and their channel means are within 8 % (red and green within 5 %) of its."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("compare_nobvh_png", os.path.join(root, "scripts", "compare_nobvh_png.py"))
    cmp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmp)
    detail = np.load(os.path.join(root, "tests", "golden", "ref_nobvh_detail.npz"))
    low = np.load(os.path.join(root, "tests", "golden", "ref_nobvh_lowfreq.npz"))["rgb"].astype(np.float64)
    shot_noise = cmp.relative_noise(detail["laplacian_energy"].astype(np.float64), cmp.luminance(low))
    shot_mean = low.mean(axis=(0, 1))

    def look(cycles, rpp):
        img = cmp.render(device, cycles, True, rpp=rpp)
        rgb = cmp.linear_rgb(img)
        lum = cmp.luminance(rgb)
        noise = cmp.relative_noise(cmp.laplacian_energy(lum), cmp.luminance(cmp.low_frequency(img)))
        return lum, noise, rgb.mean(axis=(0, 1)) / shot_mean

    seen = {}
    for cycles, rpp in ((3, 1), (12, 1), (3, 3), (3, 4)):
        lum, noise, ratio = look(cycles, rpp)
        cols, rows = cmp.band_profiles(lum)
        cx = cmp.shift_correlation(detail["column_profiles"], cols)
        cy = cmp.shift_correlation(detail["row_profiles"], rows, smooth=3)
        print("%2d cycles x %d rays/pixel: x shift %+d (%.3f, next %.3f), y shift %+d (%.3f); noise %.3f (screenshot %.3f); channel means / "
              "screenshot's %s" % (cycles, rpp, cx.argmax() - 4, cx.max(), np.sort(cx)[-2], cy.argmax() - 4, cy.max(), noise, shot_noise, np.round(ratio, 3)))
        assert cx.argmax() - 4 == 0 and cx.max() >= 0.3 and np.sort(cx)[-2] <= 0.1, "columns do not line up at zero shift"
        assert cy.argmax() - 4 == -1 and cy.max() > cy[4] and cy.max() > cy[2], "rows do not line up at the capture's offset of one"
        seen[(cycles, rpp)] = (noise, ratio)
    assert seen[(3, 1)][0] > 2.5 * shot_noise and abs(seen[(3, 1)][1][1] - 1.0) < 0.1            # brightness fits, noise does not
    assert seen[(12, 1)][0] < 1.1 * shot_noise and seen[(12, 1)][1][1] > 1.2                    # noise about right, a quarter too bright
    assert seen[(3, 4)][0] < shot_noise < seen[(3, 3)][0]
    for key in ((3, 3), (3, 4)):
        ratio = seen[key][1]
        assert np.all(np.abs(ratio - 1.0) <= 0.08) and np.all(np.abs(ratio[:2] - 1.0) <= 0.05), (key, ratio)
