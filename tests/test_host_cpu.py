"""CPU tests of the host side: library symbols, the from-scratch OBJ/MTL/PNG loaders against what the
reference's vendored parsers return (tests/golden/ref_*.npz), camera matrices, atlas packing, the
synthetic scene generator, band sharding. No GPU compute is called."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_ASSETS = "/root/reference/assets"
needs_reference_assets = pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="the reference's asset files are not on this machine")


def test_library_exports_every_declared_symbol():
    from gpuraytracer_amd import capi
    lib = capi.load()
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert [os.path.basename(h) for h in headers] == ["gpurt.h", "gpurt_debug.h"]
    declared = set()
    for h in headers:
        declared |= set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", open(h).read()))
    assert len(declared) >= 30
    assert declared == set(capi.SIGNATURES), "include/*.h and the ctypes table disagree"
    # what a maintainer binds carries no test-only switch or counter
    main = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", open(headers[0]).read()))
    assert not main & {"rt_device_set_walk_margins", "rt_device_set_trace_stats", "rt_device_set_walk_stats", "rt_device_listed_rays",
                       "rt_device_set_traverse_timing", "rt_device_traverse_timing", "rt_device_listed_rays_detail",
                       "rt_device_set_listed_split"}
    for name in declared:
        assert hasattr(lib, name), "libgpurt.so does not export %s" % name


def test_error_strings_and_struct_sizes():
    from gpuraytracer_amd import capi
    assert capi.error_string(0) == "ok"
    assert capi.error_string(capi.RT_ERR_TOO_FEW_PRIMITIVES) == "fewer than 3 primitives"
    lib = capi.load()
    assert lib.rt_bvh_node_count(992) == 992 and lib.rt_bvh_node_count(2) == 0 and lib.rt_bvh_node_count(50000000) == 50000007


def test_missing_library_fails_loudly(monkeypatch):
    from gpuraytracer_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libgpurt.so")
    with pytest.raises(capi.GpurtMissing):
        capi.load()


def test_parse_real_follows_the_digit_accumulation_rule():
    from gpuraytracer_amd import host_py
    cases = ["3.793523", "-1.000000", "0.375000", "867.454773", "1e-3", "-.5", "+2.5E+2", "0.0000001234", "12", "7.", "abc"]
    for text in cases:
        got = host_py.parse_real(text)
        try:
            want = np.float32(float(text))
        except ValueError:
            want = np.float32(0.0)
        assert abs(got - want) <= abs(want) * 2e-7, text
    assert host_py.parse_real("0.1") == np.float32(0.1)


@needs_reference_assets
@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2"])
def test_obj_loader_matches_the_reference_parser(name, oracle):
    from gpuraytracer_amd import host_py
    g = np.load(os.path.join(GOLD, "ref_mesh_%s.npz" % name))
    cwd = os.getcwd()
    os.chdir("/root/reference")           # texture paths in the MTL are relative to the reference's working directory
    try:
        sc = host_py.load_scene("assets/%s.obj" % name)
    finally:
        os.chdir(cwd)
    v = sc["vertices"]
    assert len(v) == len(g["corners"]) and np.array_equal(sc["indices"], np.arange(len(v)))
    assert np.array_equal(v["position"].view(np.uint32), g["corners"][:, 0:3].view(np.uint32)), "positions differ in some bit"
    assert np.array_equal(v["uv"].view(np.uint32), g["corners"][:, 3:5].view(np.uint32))
    assert np.array_equal(v["normal"].view(np.uint32), g["corners"][:, 5:8].view(np.uint32))
    assert np.array_equal(v["material_id"], g["material_ids"].astype(np.uint32))
    m, mtl = sc["materials"], g["mtl"]
    assert len(m) == len(mtl)
    assert np.array_equal(m["albedo"], mtl[:, 0:3]) and np.array_equal(m["f0_color"], mtl[:, 3:6])
    assert np.array_equal(m["emissive"], mtl[:, 6:9]) and np.array_equal(m["metallic"], mtl[:, 11])
    lib = oracle.load()
    want_rough = np.array([lib.orc_roughness_from_mtl(float(r[9]), float(r[10])) for r in mtl], dtype=np.float32)
    assert np.array_equal(m["roughness"], want_rough)
    # texture table order: "" first, then first appearance
    names = [""]
    for row in g["texture_names"]:
        for t in row:
            if t not in names:
                names.append(str(t))
    assert sc["texture_names"] == names
    # post-processing against the oracle's restatement of the same host code
    o = np.load(os.path.join(GOLD, "oracle_%s.npz" % name))
    assert np.array_equal(sc["scene_min"], o["scene_min"]) and np.array_equal(sc["scene_max"], o["scene_max"])
    assert np.array_equal(v["tangent"].view(np.uint32), o["vertices"]["tangent"].view(np.uint32))
    assert sc["materials"].tobytes() == o["materials"].tobytes()
    assert sc["texture_ids"].tobytes() == o["texture_ids"].tobytes()
    if "atlas" in o:
        assert np.array_equal(sc["atlas"], o["atlas"])
    else:
        t = np.load(os.path.join(GOLD, "ref_texture_Testure1.npz"))
        a = sc["atlas"].reshape(-1).view(np.uint8)
        assert len(a) == 16 + 1024 * 1024 * 8 and np.all(a[:8] == 0xFF) and np.all(a[8 + 1024 * 1024 * 8:] == 0)
        px = a[8:8 + 1024 * 1024 * 8].view(np.uint16).reshape(1024, 1024, 4)
        assert np.array_equal(px[..., :3], t["rgb8"].astype(np.uint16) * 257) and np.all(px[..., 3] == 65535)
        assert tuple(sc["texture_ids"][1]) == (16, 1024, 1024, 1024)      # byte offset, as the reference stores it


@needs_reference_assets
def test_png_decoder_matches_the_reference_decoder():
    from gpuraytracer_amd import host_py
    t = np.load(os.path.join(GOLD, "ref_texture_Testure1.npz"))
    px = host_py.load_texture(os.path.join(REF_ASSETS, "Testure1.png"))
    assert px.shape == (1024, 1024, 4)
    assert np.array_equal(px[0, 0], t["first_texel"]) and list(px[0, 0]) == [52171, 43176, 8738, 65535]
    assert np.array_equal(px[..., :3], t["rgb8"].astype(np.uint16) * 257) and np.all(px[..., 3] == 65535)


@pytest.mark.parametrize("mode,bits", [("RGB", 8), ("RGBA", 8), ("L", 8), ("LA", 8), ("P", 8), ("I;16", 16)])
def test_png_decoder_on_generated_files(tmp_path, mode, bits):
    from PIL import Image
    from gpuraytracer_amd import host_py
    rng = np.random.RandomState(11)
    w, h = 37, 23
    if mode == "I;16":
        data = rng.randint(0, 65536, (h, w)).astype(np.uint16)
        im = Image.fromarray(data, mode="I;16")
        want = np.stack([data, data, data, np.full_like(data, 65535)], axis=2)
    elif mode == "P":
        idx = rng.randint(0, 16, (h, w)).astype(np.uint8)
        pal = rng.randint(0, 256, (16, 3)).astype(np.uint8)
        im = Image.fromarray(idx, mode="P")
        im.putpalette(pal.reshape(-1).tolist())
        rgb = pal[idx].astype(np.uint16) * 257
        want = np.concatenate([rgb, np.full((h, w, 1), 65535, dtype=np.uint16)], axis=2)
    else:
        ch = len(mode)
        data = rng.randint(0, 256, (h, w, ch)).astype(np.uint8)
        im = Image.fromarray(data[..., 0] if ch == 1 else data, mode=mode)
        d16 = data.astype(np.uint16) * 257
        if mode == "L":
            want = np.concatenate([d16, d16, d16, np.full((h, w, 1), 65535, dtype=np.uint16)], axis=2)
        elif mode == "LA":
            want = np.concatenate([d16[..., :1]] * 3 + [d16[..., 1:2]], axis=2)
        elif mode == "RGB":
            want = np.concatenate([d16, np.full((h, w, 1), 65535, dtype=np.uint16)], axis=2)
        else:
            want = d16
    path = str(tmp_path / "t.png")
    im.save(path)
    got = host_py.load_texture(path)
    assert np.array_equal(got, want)


def test_loader_rejects_bad_input(tmp_path):
    from gpuraytracer_amd import host_py
    with pytest.raises(RuntimeError):
        host_py.load_scene(str(tmp_path / "missing.obj"))
    quad = tmp_path / "quad.obj"
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\n")
    with pytest.raises(RuntimeError):
        host_py.load_scene(str(quad))
    with pytest.raises(RuntimeError):
        host_py.load_texture(str(quad))


def test_loader_without_materials_or_normals(tmp_path):
    """No MTL: one glowing default material; missing normals are generated; negative indices resolve."""
    from gpuraytracer_amd import host_py
    p = tmp_path / "tri.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\nf -4/1 -2/3 -1/2\n")
    sc = host_py.load_scene(str(p))
    assert len(sc["vertices"]) == 6 and len(sc["materials"]) == 1
    assert tuple(sc["materials"][0]["emissive"]) == (0.5, 0.5, 0.5)
    assert np.all(sc["vertices"]["material_id"] == 0xFFFFFFFF)        # tinyobj's "no material" id, kept as the reference keeps it
    n = sc["vertices"]["normal"]
    assert np.allclose(np.abs(n[0]), (0, 0, 1)) and np.allclose(np.linalg.norm(n, axis=1), 1.0)
    assert np.array_equal(sc["vertices"]["position"][3], (0, 0, 0)) and np.array_equal(sc["vertices"]["position"][5], (0, 0, 1))
    assert sc["texture_names"] == [""] and len(sc["texture_ids"]) == 1 and sc["atlas"].shape == (1, 4)


def test_camera_matrices_match_the_oracle(oracle):
    from gpuraytracer_amd import host_py
    for pos, focus, aspect in [((-4.0259, 2.1676, -2.9459), (0, 0, 0), 1280 / 720), ((0, 0, 0), (1, 0, 0), 16 / 9),
                               ((0, 0, -22), (0, 0, 0), 3840 / 2160), ((-2.5, 5.0, -2.0), (0, 0, 0), 1.0)]:
        a = host_py.camera_matrices(pos, focus, (0, 1, 0), 1.2, aspect, 0.01, 10000.0)
        b = oracle.camera_matrices(pos, focus, (0, 1, 0), 1.2, aspect, 0.01, 10000.0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # inverse(transpose(V)) read column-major is V^-1: its last row (row-vector convention) is the camera position
        vinv = a[0].reshape(4, 4).T
        assert np.allclose(vinv[3, :3], pos, atol=1e-5) and np.allclose(vinv[:, 3], (0, 0, 0, 1), atol=1e-6)


def test_synthetic_scene_generators_agree():
    from gpuraytracer_amd import host_py, scenes
    for n in (15, 1000, 20011):
        a, b = scenes.synthetic_scene(n), host_py.synthetic_scene(n)
        assert a["vertices"].tobytes() == b["vertices"].tobytes()
        assert np.array_equal(a["indices"], b["indices"]) and a["materials"].tobytes() == b["materials"].tobytes()
        assert np.array_equal(a["scene_min"], b["scene_min"]) and np.array_equal(a["scene_max"], b["scene_max"])
        assert np.array_equal(a["atlas"], b["atlas"]) and a["texture_ids"].tobytes() == b["texture_ids"].tobytes()
    with pytest.raises(ValueError):
        host_py.synthetic_scene(14)


def test_band_partition_covers_the_image():
    from gpuraytracer_amd.distributed import band_for_rank
    for h in (1, 7, 720, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            rows = [band_for_rank(h, world, r) for r in range(world)]
            assert rows[0][0] == 0 and sum(n for _, n in rows) == h
            for (f0, n0), (f1, _) in zip(rows, rows[1:]):
                assert f0 + n0 == f1
            assert max(n for _, n in rows) - min(n for _, n in rows) <= 1


def test_image_writers_round_trip(tmp_path):
    """WriteImagePPM / WriteImagePNG / WriteImagePFM: the 8-bit files hold the sRGB encoding of the R8G8B8A8_UNORM_SRGB
    back buffer the reference presents to (src/RenderTarget.cpp:66 of the reference); the PFM holds the floats."""
    import struct
    import zlib
    from gpuraytracer_amd import host_py
    lib = host_py.load()
    rng = np.random.RandomState(4)
    w, h = 37, 21
    img = rng.uniform(0.0, 1.0, (h, w, 4)).astype(np.float32)
    img[0, 0, :3] = (0.0, 1.0, 0.0031308)
    img[0, 1, :3] = (-1.0, 2.0, np.nan)                     # clamped; NaN presents as black
    img[0, 2, :3] = (0.5, 0.21404114, 0.001)

    def srgb(x):                                            # the sRGB transfer function, rounded to 8 bits
        x = np.where(np.isnan(x), 0.0, np.clip(x.astype(np.float64), 0.0, 1.0))
        e = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
        return np.floor(e * 255.0 + 0.5).astype(np.int64)

    want = srgb(img[..., :3])
    assert [lib.rth_encode_srgb8(v) for v in (0.0, 1.0, 0.5, 0.21404114, -3.0, 7.0)] == [0, 255, 188, 128, 0, 255]

    ppm, png, pfm = tmp_path / "a.ppm", tmp_path / "a.png", tmp_path / "a.pfm"
    for p in (ppm, png, pfm):
        host_py.write_image(p, img)
    raw = ppm.read_bytes()
    head = b"P6\n%d %d\n255\n" % (w, h)
    assert raw.startswith(head) and len(raw) == len(head) + 3 * w * h
    got = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(h, w, 3).astype(np.int64)
    assert np.abs(got - want).max() <= 1 and np.mean(got != want) < 0.01        # float pow vs double pow at a rounding boundary

    data = png.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        size, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + size]
        assert struct.unpack(">I", data[pos + 8 + size:pos + 12 + size])[0] == (zlib.crc32(kind + body) & 0xffffffff)
        chunks.append((kind, body))
        pos += 12 + size
    assert [k for k, _ in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[2][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()                             # filter type 0 on every row
    assert np.array_equal(rows[:, 1:].reshape(h, w, 3).astype(np.int64), got), "PNG and PPM hold the same samples"
    # ... and the library's own PNG decoder reads it back (8-bit samples widened to 16 bits)
    back = host_py.load_texture(str(png))
    assert back.shape == (h, w, 4) and np.array_equal(back[..., :3] >> 8, got.astype(np.uint16))

    raw = pfm.read_bytes()
    head = b"PF\n%d %d\n-1.0\n" % (w, h)
    assert raw.startswith(head)
    f = np.frombuffer(raw[len(head):], dtype="<f4").reshape(h, w, 3)[::-1]      # PFM rows run bottom to top
    assert f.tobytes() == np.ascontiguousarray(img[..., :3]).tobytes()


def test_rows_dealt_to_ranks_cover_the_image_once():
    """rt_rows_for_rank (host arithmetic only): rank r of N holds rows r, r + N, ...; together the ranks hold every row
    once, ranks beyond the image hold none, and the Python twin used with torch.distributed agrees."""
    import ctypes as C
    from gpuraytracer_amd import capi
    from gpuraytracer_amd.distributed import rows_for_rank
    lib = capi.load()
    for height in (0, 1, 2, 7, 8, 9, 1080, 2160, 4320):
        for world in (1, 2, 3, 4, 8, 16):
            seen = np.zeros(height, dtype=np.int32)
            for rank in range(world):
                r = capi.ImageRows()
                assert lib.rt_rows_for_rank(height, world, rank, C.byref(r)) == capi.RT_OK
                assert r.row_stride == world
                assert (min(r.first_row, height), r.num_rows, r.row_stride) == tuple(min(x, height) if i == 0 else x for i, x in enumerate(rows_for_rank(height, world, rank)))
                rows = r.first_row + r.row_stride * np.arange(r.num_rows)
                assert np.all(rows < height)
                seen[rows] += 1
            assert np.all(seen == 1)
    r = capi.ImageRows()
    assert lib.rt_rows_for_rank(10, 0, 0, C.byref(r)) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_rows_for_rank(10, 4, 4, C.byref(r)) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_rows_for_rank(10, 4, 1, None) == capi.RT_ERR_INVALID_ARGUMENT


def _write_shared_vertex_obj(path):
    """A closed box of 12 triangles over 8 corners with per-face normals, one face listed twice, and a corner written once
    as -0.0 and once as 0.0 (equal for the reference's float comparison)."""
    corners = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    faces = [((0, 1, 2, 3), (0, 0, -1)), ((4, 5, 6, 7), (0, 0, 1)), ((0, 1, 5, 4), (0, -1, 0)), ((2, 3, 7, 6), (0, 1, 0)),
             ((1, 2, 6, 5), (1, 0, 0)), ((0, 3, 7, 4), (-1, 0, 0))]
    (path.parent / "box.mtl").write_text("newmtl wall\nKd 0.8 0.7 0.6\nKe 0.4 0.5 0.6\nPr 0.5\n")
    lines = ["mtllib box.mtl", "usemtl wall"] + ["v %g %g %g" % c for c in corners] + ["v -0.0 0 0", "v 0.0 0 0", "v 0 1 0"]
    lines += ["vn %g %g %g" % n for _, n in faces] + ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1"]
    for k, (q, _) in enumerate(faces):
        a, b, c, d = (i + 1 for i in q)
        lines.append("f %d/1/%d %d/2/%d %d/3/%d" % (a, k + 1, b, k + 1, c, k + 1))
        lines.append("f %d/1/%d %d/3/%d %d/4/%d" % (a, k + 1, c, k + 1, d, k + 1))
    lines.append("f 1/1/1 2/2/1 3/3/1")                    # the first triangle once more
    lines.append("f 9/1/1 11/2/1 10/1/1")                  # corners at -0 and +0 with equal attributes
    path.write_text("\n".join(lines) + "\n")


def test_vertex_deduplication_follows_the_switched_off_reference_loop(tmp_path):
    """LoadMeshFromFile(bDeduplicate): every vertex gets the index of the first earlier unique vertex that the reference's
    VerticesAreEqual accepts (position, UV, normal, material; src/RaytracerMesh.cpp:157-169, 249-275 of the reference) -
    compared with that quadratic loop restated in numpy. The default stays Indices[i] == i."""
    from gpuraytracer_amd import host_py
    obj = tmp_path / "box.obj"
    _write_shared_vertex_obj(obj)
    plain = host_py.load_scene(str(obj))
    dedup = host_py.load_scene(str(obj), dedup=True)
    v = plain["vertices"]
    assert np.array_equal(plain["indices"], np.arange(len(v), dtype=np.uint32))
    want_idx, uniq = [], []
    for i in range(len(v)):
        for j, u in enumerate(uniq):
            if (np.all(v["position"][i] == v["position"][u]) and np.all(v["uv"][i] == v["uv"][u]) and
                    np.all(v["normal"][i] == v["normal"][u]) and v["material_id"][i] == v["material_id"][u]):
                want_idx.append(j)
                break
        else:
            want_idx.append(len(uniq))
            uniq.append(i)
    assert np.array_equal(dedup["indices"], np.array(want_idx, dtype=np.uint32))
    assert len(dedup["vertices"]) == len(uniq) < len(v)
    for field in ("position", "uv", "normal", "material_id"):
        assert np.array_equal(dedup["vertices"][field], v[field][uniq])
        assert np.array_equal(dedup["vertices"][field][dedup["indices"]], v[field])       # same triangles
    assert dedup["indices"][-3] == dedup["indices"][-1] != dedup["indices"][-2]     # -0.0 and 0.0 are one vertex
    assert np.array_equal(dedup["scene_min"], plain["scene_min"]) and np.array_equal(dedup["scene_max"], plain["scene_max"])


def test_no_kernel_uses_scratch():
    """The compiler's resource remarks saved by the build (gpuraytracer_amd/lib/*.resources.txt): no kernel of the library
    keeps data in scratch memory, and the default traversal kernel still fits four waves per SIMD. A kernel argument
    that is written to, or a dynamically indexed local array, silently moves to scratch and costs the walk 15 %."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "gpuraytracer_amd", "lib", "*.resources.txt")))
    if not files:
        pytest.skip("the library was built without resource remarks")
    kernels = {}
    for path in files:
        name = None
        for line in open(path):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                kernels[name] = {}
                continue
            m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
            if m and name:
                kernels[name][m.group(1)] = int(m.group(2))
    assert len(kernels) >= 30
    for name, r in kernels.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0 and r.get("VGPRs Spill", 0) == 0, "%s uses scratch: %s" % (name, r)
    walk = [r for n, r in kernels.items() if "k_traverse_spatialILb0" in n]
    assert len(walk) == 1 and walk[0]["Occupancy [waves/SIMD]"] >= 4


def test_counter_file_of_the_round_travels_with_the_tree():
    """bench.py takes its roofline fractions from the committed counter passes under profiles/: the file of this round
    exists, names the default traversal kernel, and nothing keeps profiles/ from being sent to the GPU box."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    counters, source, provenance = bench.load_pmc("c3", "k_traverse_spatial")
    assert counters is not None and source.startswith("profiles/"), provenance
    assert provenance["traffic_source_sha"] == provenance["source_sha"] == bench.source_sha()
    assert "FETCH_SIZE" in counters and "WRITE_SIZE" in counters
    fr = bench.counter_fractions(counters, 2.4e-3)
    assert 0.0 < fr["memory"]["frac"] <= 1.0 and 0.0 < fr["vector"]["frac"] <= 1.0
    ignore = os.path.join(ROOT, ".gpurunignore")
    if os.path.exists(ignore):
        lines = [l.strip() for l in open(ignore) if l.strip()]
        assert not any(l.strip("./").startswith("profiles") for l in lines)


# ------------------------------------------------------------------ the OBJ file parsed by several threads ----

def _write_mixed_obj(path, faces=400, seed=3, crlf=False, final_newline=True):
    """An OBJ file that uses what the parser knows, in an order that makes "the counts so far" matter: v / vt / vn lines
    between the faces, 1-based and relative indices, all four corner forms, comments, blank lines, trailing blanks,
    materials switched (known, unknown, before their library is named), a second library half way."""
    rng = np.random.RandomState(seed)
    d = os.path.dirname(path)
    with open(os.path.join(d, "first.mtl"), "w") as f:
        f.write("newmtl red\nKd 0.8 0.1 0.1\nKe 0.2 0.2 0.2\nnewmtl blue\nKd 0.1 0.1 0.9\nNs 50\n")
    with open(os.path.join(d, "second.mtl"), "w") as f:
        f.write("newmtl green\nKd 0.1 0.7 0.1\nPr 0.4\nPm 0.6\n")
    lines = ["# mixed test file", "usemtl red", "mtllib missing.mtl first.mtl", ""]
    nv = nvt = nvn = 0
    for k in range(faces):
        for _ in range(3):
            lines.append("v %.6f %.6f %.6f" % tuple(rng.uniform(-5, 5, 3)) + (" " if k % 7 == 0 else ""))
            nv += 1
        if k % 3 != 2:
            for _ in range(3):
                lines.append("vt %.5f %.5f" % tuple(rng.uniform(0, 1, 2)))
                nvt += 1
        if k % 4 != 3:
            for _ in range(2):
                n = rng.normal(size=3)
                lines.append("vn %.6f %.6f %.6f" % tuple(n / np.linalg.norm(n)))
                nvn += 1
        if k % 11 == 0:
            lines.append("usemtl %s" % ["red", "blue", "nothing", "green"][(k // 11) % 4])
        if k == faces // 2:
            lines.append("mtllib second.mtl")
        if k % 13 == 5:
            lines += ["", "   # a comment", "g group%d" % k, "s off"]
        form = k % 4
        corners = []
        for c in range(3):
            v = -(3 - c) if k % 2 else nv - 2 + c                               # relative / 1-based
            vt = (-(c + 1) if k % 5 == 0 else max(nvt - c, 1)) if nvt else 1
            vn = (-1 if k % 6 == 0 else max(nvn - (c % 2), 1)) if nvn else 1
            if k % 17 == 3:
                vt = nvt + 50                                                    # beyond the counts so far: left at zero
            corners.append(["%d" % v, "%d/%d" % (v, vt), "%d//%d" % (v, vn), "%d/%d/%d" % (v, vt, vn)][form])
        lines.append("f " + "  ".join(corners) + ("\t " if k % 9 == 0 else ""))
    text = ("\r\n" if crlf else "\n").join(lines) + (("\r\n" if crlf else "\n") if final_newline else "")
    with open(path, "w", newline="") as f:
        f.write(text)


def _same_scene(a, b):
    for k in ("vertices", "indices", "materials", "texture_ids", "scene_min", "scene_max"):
        assert a[k].tobytes() == b[k].tobytes(), k
    assert a["texture_names"] == b["texture_names"]


@pytest.mark.parametrize("crlf,final_newline", [(False, True), (True, True), (False, False)])
def test_obj_parsed_by_several_threads_gives_the_same_mesh(tmp_path, crlf, final_newline):
    from gpuraytracer_amd import host_py
    path = str(tmp_path / "mixed.obj")
    _write_mixed_obj(path, crlf=crlf, final_newline=final_newline)
    for dedup in (False, True):
        want = host_py.load_scene(path, dedup=dedup, threads=1)
        assert len(want["indices"]) == 1200 and want["vertices"]["material_id"].max() == 0xFFFFFFFF
        assert len(set(want["vertices"]["material_id"].tolist())) == 4          # red, blue, green and "none"
        for threads in (2, 3, 5, 16, 61):
            _same_scene(host_py.load_scene(path, dedup=dedup, threads=threads), want)


@needs_reference_assets
@pytest.mark.parametrize("name", ["testobject4_2", "testobject5_1", "testscene2"])
def test_reference_assets_parsed_by_several_threads(name):
    from gpuraytracer_amd import host_py
    cwd = os.getcwd()
    os.chdir("/root/reference")
    try:
        want = host_py.load_scene("assets/%s.obj" % name, threads=1)
        for threads in (2, 7):
            _same_scene(host_py.load_scene("assets/%s.obj" % name, threads=threads), want)
    finally:
        os.chdir(cwd)


def test_parser_threads_report_the_first_error_as_the_serial_parser_does(tmp_path, capfd):
    """Bad files: the message (kind of error and line) of the threaded parser is the line-by-line parser's."""
    from gpuraytracer_amd import host_py
    good = ["v 0 0 0", "v 1 0 0", "v 0 1 0", "f 1 2 3"] * 40
    cases = {
        "forward": good[:80] + ["f 1 2 500"] + good[80:],                                  # an index beyond the counts so far
        "zero": good[:120] + ["f 0 1 2"] + good[120:],
        "quad": good[:60] + ["f 1 2 3 4"] + good[60:] + ["f 1 2 9999"],                     # two errors: the first counts
        "quad_bad_fourth": good[:100] + ["f 1 2 3 0"] + good[100:],                         # fourth corner: its index is looked at first
        "two_corners": good + ["f 1 2"],
        "empty_slot": good[:30] + ["f 1/ 2/ 3/"] + good[30:],
    }
    for name, lines in cases.items():
        path = str(tmp_path / (name + ".obj"))
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")
        messages = []
        for threads in (1, 2, 4, 9):
            capfd.readouterr()
            with pytest.raises(RuntimeError):
                host_py.load_scene(path, threads=threads)
            messages.append(capfd.readouterr().out.strip())
        assert messages[0].startswith("Error loading scene:") and "line" in messages[0], (name, messages[0])
        assert all(m == messages[0] for m in messages), (name, messages)


# ------------------------------------------------------------------ the gate of a device group (rt_headless --ranks) ----

def _gate_run(world, plan, timeout=20.0):
    """Runs one thread per rank; plan[r] is a list of steps: ("arrive", well) -> records what the gate said, ("leave",).
    Returns the records per rank; fails if a thread does not come back (a rank left waiting for ever)."""
    import threading
    from gpuraytracer_amd import host_py
    lib = host_py.load()
    gate = lib.rth_gate_create(world)
    said = [[] for _ in range(world)]

    def rank(r):
        for step in plan[r]:
            if step[0] == "arrive":
                said[r].append(lib.rth_gate_arrive(gate, 1 if step[1] else 0))
            elif step[0] == "leave":
                lib.rth_gate_leave(gate)
                return
            elif step[0] == "sleep":
                import time
                time.sleep(step[1])

    threads = [threading.Thread(target=rank, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
        assert not t.is_alive(), "a rank is still waiting at the gate"
    failed = lib.rth_gate_failed(gate)
    lib.rth_gate_destroy(gate)
    return said, failed


def test_rank_gate_opens_for_a_healthy_group():
    said, failed = _gate_run(4, [[("arrive", True), ("arrive", True), ("arrive", True)]] * 4)
    assert said == [[1, 1, 1]] * 4 and failed == 0


def test_rank_gate_tells_everyone_about_a_rank_that_arrives_unwell():
    plan = [[("arrive", True), ("arrive", True)] for _ in range(3)]
    plan[1] = [("arrive", True), ("arrive", False)]
    said, failed = _gate_run(3, plan)
    assert said == [[1, 0]] * 3 and failed == 1


def test_rank_gate_does_not_wait_for_a_rank_that_has_left():
    """A rank that fails before the first collective (its device could not be created, say) leaves; the others are told at
    the gate instead of waiting inside the collective for ever, and every later gate opens at once, closed for business."""
    plan = [[("arrive", True), ("arrive", True)] for _ in range(4)]
    plan[2] = [("leave",)]
    said, failed = _gate_run(4, plan)
    assert failed == 1
    assert said[2] == [] and all(s == [0, 0] for r, s in enumerate(said) if r != 2)
    # ... also when the others are already waiting when it leaves
    plan = [[("arrive", True)] for _ in range(3)]
    plan[0] = [("sleep", 0.3), ("leave",)]
    said, failed = _gate_run(3, plan)
    assert said[1] == [0] and said[2] == [0] and failed == 1
    # ... and when it leaves between two gates
    plan = [[("arrive", True), ("arrive", True)] for _ in range(3)]
    plan[1] = [("arrive", True), ("leave",)]
    said, failed = _gate_run(3, plan)
    assert said[0] == [1, 0] and said[2] == [1, 0] and said[1] == [1]


def test_headless_cli_refuses_more_ranks_than_devices():
    """rt_headless --ranks N beyond the GPUs of the machine ends at once with a non-zero code (on a machine without a GPU:
    because the devices cannot even be counted), instead of starting threads that wait for each other."""
    import subprocess
    exe = os.path.join(ROOT, "gpuraytracer_amd", "bin", "rt_headless")
    if not os.path.exists(exe):
        pytest.skip("rt_headless is not built")
    r = subprocess.run([exe, "--synthetic", "100", "--width", "32", "--height", "18", "--ranks", "99"], capture_output=True, text=True, timeout=60)
    assert r.returncode in (1, 2), r.stdout + r.stderr
    assert "ranks" in r.stdout or "devices" in r.stdout


# ------------------------------------------------------------------ committed hardware counters and the kernels they belong to ----

def test_committed_counter_files_belong_to_the_kernels_in_the_tree():
    """bench.py takes roofline.traffic and the vector fractions from profiles/<round>_<workload>_pmc.json. A counter file
    measured before the last change to the kernel sources says nothing about the kernels that run now: it has to be
    taken again (scripts/profile_round.sh + scripts/collect_round.py), and until then bench.py withholds the fractions."""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_*_pmc.json" % bench.ROUND)))
    assert files, "no counter file of this round is committed"
    sha = bench.source_sha()
    for f in files:
        table = json.load(open(f))
        assert table.get("source_sha") == sha, "%s was measured on kernel sources %s, the tree holds %s" % (os.path.basename(f), table.get("source_sha"), sha)
        assert table.get("head"), "%s does not name the commit it was collected at" % os.path.basename(f)
