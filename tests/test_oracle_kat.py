"""Pins of the CPU oracle: known-answer values derived from the reference's shader sources, structural
invariants, and the committed golden outputs (tests/golden/oracle_*.npz). CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_xorshift_sequence_from_seed_1(oracle):
    lib = oracle.load()
    s, out = 1, []
    for _ in range(4):
        s = lib.orc_xorshift(s)
        out.append(s)
    assert out == [270369, 67634689, 2647435461, 307599695]       # shader/Random.hlsli:3-9


def test_per_thread_seed(oracle):
    lib = oracle.load()
    got = [lib.orc_thread_seed(t) for t in (0, 1, 2, 1279, 921599)]
    assert got == [0, 4866675, 10273956, 28087347, 431529254]      # shader/CS_TraceRays.hlsl:255-257


def test_left_shift3_keeps_the_reference_quirk(oracle):
    lib = oracle.load()
    got = [lib.orc_left_shift3(v) for v in (1, 2, 3, 7, 512, 1023)]
    assert got == [0x1, 0xA, 0xB, 0x4F, 0x0A00A200, 0x0B2CF3FF]    # shader/CS_GenerateMortonCodes.hlsl:30-38


def test_morton_interleave(oracle):
    lib = oracle.load()
    assert lib.orc_morton(1, 0, 0) == 1 and lib.orc_morton(0, 1, 0) == 2 and lib.orc_morton(0, 0, 1) == 4
    assert lib.orc_morton(5, 9, 1) == 1367 and lib.orc_morton(1023, 1023, 1023) == 0x3FFFFFFF


def test_random_scale_is_two_to_minus_32(oracle):
    lib = oracle.load()
    assert np.float32(2.3283064e-10) == np.float32(2.0 ** -32)
    s = C.c_uint32(1)
    assert lib.orc_random_float(C.byref(s)) == np.float32(270369) * np.float32(2.0 ** -32)


@pytest.mark.parametrize("n,nodes", [(992, 992), (1330, 1335), (10528, 10533), (10 ** 6, 1000007), (10 ** 7, 10000009),
                                     (5 * 10 ** 7, 50000007), (3, 3), (4, 3), (5, 6), (2, 0), (0, 0)])
def test_node_count_closed_form(oracle, n, nodes):
    assert oracle.load().orc_bvh_node_count(n) == nodes           # src/RaytracerPipeline.cpp:465-467, 497-559


def test_sincos_polynomial_accuracy(oracle):
    lib = oracle.load()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(0.0, 2 * np.pi, 4001):
        lib.orc_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(x)), abs(c.value - np.cos(x)))
    assert worst < 1e-6


def test_mt19937_matches_the_standard_sequence(oracle):
    assert oracle.mt19937(5489, 0, 1)[0] == 3499211612            # first output of the default-seeded std::mt19937
    assert oracle.mt19937(5489, 9999, 1)[0] == 4123659995         # its 10000th output (C++ standard, [rand.predef])


def test_sort_as_written_equals_stable_sort(oracle):
    from gpuraytracer_amd import scenes
    sc = scenes.synthetic_scene(3000)
    a = oracle.sort_primitives(sc, as_written=True)
    b = oracle.sort_primitives(sc, as_written=False)
    assert np.array_equal(a, b)
    order = np.argsort(a["x"], kind="stable")
    assert np.array_equal(order, np.arange(len(a)))
    # ties keep ascending triangle index
    same = np.diff(a["x"].astype(np.int64)) == 0
    assert np.all(np.diff(a["y"].astype(np.int64))[same] > 0)
    assert sorted(a["y"].tolist()) == list(range(0, 9000, 3))


def test_prefix_sum_as_written_is_an_exclusive_scan(oracle):
    lib = oracle.load()
    rng = np.random.RandomState(3)
    freq = np.zeros(256, dtype=oracle.uint4_dtype)
    freq["x"] = rng.randint(0, 1000, 256)
    info = oracle.SortInfo(0, 0)
    lib.orc_sort_prefix_sum(C.byref(info), oracle.ptr(freq))
    assert np.array_equal(freq["y"], np.concatenate([[0], np.cumsum(freq["x"])[:-1]]).astype(np.uint32))


def _load_golden_scene(name):
    from gpuraytracer_amd import scenes
    g = np.load(os.path.join(GOLD, "oracle_%s.npz" % name))
    if name.startswith("synthetic"):
        return scenes.synthetic_scene(int(name[len("synthetic"):])), g
    cam = g["camera"]
    sc = dict(vertices=g["vertices"], indices=np.arange(len(g["vertices"]), dtype=np.uint32), materials=g["materials"],
              texture_ids=g["texture_ids"], scene_min=g["scene_min"], scene_max=g["scene_max"],
              camera=dict(position=tuple(cam[0:3]), focus=tuple(cam[3:6]), up=tuple(cam[6:9]), fov=float(cam[9]),
                          near_z=float(cam[10]), far_z=float(cam[11])), name=name)
    if "atlas" in g:
        sc["atlas"] = g["atlas"]
    else:
        # textured scene: rebuild the atlas bytes from the decoded-texture fixture (tight packing behind 8 bytes of 0xff)
        t = np.load(os.path.join(GOLD, "ref_texture_Testure1.npz"))
        rgb = t["rgb8"].astype(np.uint16) * 257
        px = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 65535, dtype=np.uint16)], axis=2)
        size = 16 + ((px.shape[0] * px.shape[1] * 8 + 15) & ~15)
        atlas = np.zeros(size, dtype=np.uint8)
        atlas[:8] = 0xFF
        atlas[8:8 + px.size * 2] = px.reshape(-1).view(np.uint8)
        sc["atlas"] = atlas.view(np.uint32).reshape(-1, 4)
    return sc, g


@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2", "synthetic4096"])
def test_oracle_reproduces_its_golden_outputs(oracle, name):
    sc, g = _load_golden_scene(name)
    m = oracle.sort_primitives(sc)
    assert np.array_equal(m["x"], g["sorted_codes"]) and np.array_equal(m["y"], g["sorted_values"])
    bvh, count = oracle.build_bvh(sc, m)
    assert count == len(g["bvh"]) and bvh[:count].tobytes() == g["bvh"].tobytes()
    p = oracle.Pipeline(sc, 64, 36, seed_raygen=1, seed_trace=2)
    p.render(9)
    img = p.output()
    _, total = p.stats()
    p.close()
    assert total["nodes_visited"] == int(g["nodes_visited"]) and total["tris_tested"] == int(g["tris_tested"])
    assert np.allclose(img, g["image"], rtol=0, atol=1e-6)
    assert img[..., :3].max() > 0


def test_bvh_and_brute_force_agree_within_tolerance(oracle):
    """BVH-on and BVH-off differ only in tie-breaking (sorted position vs original index)."""
    sc, _ = _load_golden_scene("testobject5_1")
    a = oracle.Pipeline(sc, 48, 27, use_bvh=True)
    b = oracle.Pipeline(sc, 48, 27, use_bvh=False)
    a.render(3)
    b.render(3)
    ia, ib = a.output(), b.output()
    a.close()
    b.close()
    assert np.mean(np.abs(ia - ib) > 1e-5) < 0.01


def test_true_division_slab_test_changes_few_pixels(oracle):
    """The slab test by one reciprocal per ray (what the GPU kernel does) against per-box IEEE division."""
    sc, _ = _load_golden_scene("testscene2")
    a = oracle.Pipeline(sc, 64, 36, true_division=False)
    b = oracle.Pipeline(sc, 64, 36, true_division=True)
    a.render(3)
    b.render(3)
    ia, ib = a.output(), b.output()
    a.close()
    b.close()
    assert np.mean(np.any(np.abs(ia - ib) > 1e-5, axis=2)) < 0.005


def test_camera_origin_is_the_camera_position(oracle):
    """With depth of field off, every generated ray starts at the near plane around the camera position."""
    from gpuraytracer_amd.capi import ray_dtype
    w, h = 32, 18
    pos = (-4.0259, 2.1676, -2.9459)
    iv, ip = oracle.camera_matrices(pos, (0, 0, 0), (0, 1, 0), 1.2, w / h, 0.01, 10000.0)
    info = oracle.RaygenInfo()
    for k in range(16):
        info.inverse_view[k], info.inverse_projection[k] = float(iv[k]), float(ip[k])
    info.screen[0], info.screen[1], info.aa_spread, info.dof_spread, info.max_rays_per_pixel = w, h, 1.5, 0.0, 1
    rays = np.zeros(w * h, dtype=ray_dtype)
    old = np.zeros(w * h, dtype=ray_dtype)
    pix = np.zeros(w * h, dtype=np.uint32)
    oracle.load().orc_camera_ray_generation(C.byref(info), oracle.ptr(rays), oracle.ptr(old), oracle.ptr(pix))
    assert np.abs(rays["origin"] - np.array(pos, dtype=np.float32)).max() < 0.03
    assert np.allclose(np.linalg.norm(rays["direction"], axis=1), 1.0, atol=1e-5)
    # both image axes are flipped: pixel (0, 0) looks up and to the left of the view direction... and x shares seeds per column
    assert pix[5] == (5 << 16) and pix[w + 5] == ((5 << 16) | 1)
    centre = rays["direction"][(h // 2) * w + w // 2]
    to_focus = -np.array(pos) / np.linalg.norm(pos)
    assert np.dot(centre, to_focus) > 0.99
