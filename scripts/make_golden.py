#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ (run in the build container, where /root/reference exists).

Two kinds of fixture, both plain data:
  ref_*.npz     what the reference's own vendored parsers (tinyobjloader 2.0.0, stb_image 2.27), compiled
                in place as oracle/_ref/ref_loader, return for the reference's assets with the options
                the reference's loader uses. They pin the host's from-scratch OBJ/MTL/PNG code.
  oracle_*.npz  scenes assembled from those dumps plus the CPU oracle's outputs on them (sorted Morton
                records, hierarchy nodes, small images at fixed seeds). They pin the oracle against
                regressions and give the GPU tests real-asset inputs on a box without the reference.

Usage: python scripts/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpuraytracer_amd import scenes  # noqa: E402
from gpuraytracer_amd.capi import material_dtype, texture_id_dtype, vertex_dtype  # noqa: E402
from oracle import oracle_py as oracle  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
LOADER = os.path.join(ROOT, "oracle", "_ref", "ref_loader")

CAMERAS = {
    "testscene2": dict(position=(0.0, 0.0, 0.0), focus=(1.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)),        # src/Settings.h:22-25
    "testobject4_2": dict(position=(-2.5, 5.0, -2.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)),    # src/Settings.h:32-35
    "testobject5_1": dict(position=(-2.5, 5.0, -2.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)),
}
for c in CAMERAS.values():
    c.update(fov=1.2, near_z=0.01, far_z=10000.0)


def ref_mesh(name, tmp):
    prefix = os.path.join(tmp, name)
    subprocess.run([LOADER, "obj", os.path.join(REF, "assets", name + ".obj"), prefix], check=True, cwd=REF)
    rec = np.fromfile(prefix + ".corners.bin", dtype=np.dtype([("f", "<f4", 8), ("m", "<i4")]))
    lines = open(prefix + ".materials.txt").read().split("\n")
    mtl, names = [], []
    for i in range(0, len(lines) - 1, 2):
        mtl.append([float.fromhex(t) for t in lines[i].split()])
        names.append(lines[i + 1].split("|")[1:6])
    return rec["f"].copy(), rec["m"].copy(), np.array(mtl, dtype=np.float32).reshape(-1, 12), names


def ref_texture(name, tmp):
    out = os.path.join(tmp, name + ".bin")
    subprocess.run([LOADER, "png", os.path.join(REF, "assets", name + ".png"), out], check=True)
    dims = np.fromfile(out, dtype=np.uint32, count=2)
    px = np.fromfile(out, dtype=np.uint16, offset=8).reshape(dims[1], dims[0], 4)
    return px


def assemble_scene(name, corners, mat_ids, mtl, texnames, textures):
    """LoadMeshFromFile's post-processing (src/RaytracerMesh.cpp:277-383) on the parser's output, via the oracle."""
    lib = oracle.load()
    n = len(corners)
    v = np.zeros(n, dtype=vertex_dtype)
    v["position"], v["uv"], v["normal"] = corners[:, 0:3], corners[:, 3:5], corners[:, 5:8]
    v["material_id"] = mat_ids.astype(np.uint32)
    idx = np.arange(n, dtype=np.uint32)
    table = {"": 0}
    m = np.zeros(max(len(mtl), 1), dtype=material_dtype)
    for i, (row, tn) in enumerate(zip(mtl, texnames)):
        m[i]["albedo"], m[i]["f0_color"], m[i]["emissive"] = row[0:3], row[3:6], row[6:9]
        m[i]["roughness"] = lib.orc_roughness_from_mtl(float(row[9]), float(row[10]))
        m[i]["metallic"] = row[11]
        for t in tn:
            table.setdefault(t, len(table))
        (m[i]["albedo_texture"], m[i]["roughness_texture"], m[i]["f0_texture"], m[i]["metallic_texture"],
         m[i]["emissive_texture"]) = [table[t] for t in tn]
    if len(mtl) == 0:
        m[0]["emissive"] = (0.5, 0.5, 0.5)
    import ctypes as C
    mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
    lib.orc_scene_aabb(n, oracle.ptr(idx), oracle.ptr(v), mn, mx)
    lib.orc_calculate_tangents(n, oracle.ptr(idx), oracle.ptr(v))
    # texture atlas (src/TextureAtlas.cpp:36-205): byte offsets in the table, tightly packed pixels
    ordered = sorted(table, key=table.get)
    ids = np.zeros(len(ordered), dtype=texture_id_dtype)
    ids[0] = (0, 1, 1, 1)
    size, chunks = 16, [np.full(8, 0xFF, dtype=np.uint8)]
    for k, tname in enumerate(ordered[1:], start=1):
        px = textures[os.path.basename(tname)]
        h, w = px.shape[:2]
        ids[k] = (size, w, h, w)
        size += (w * h * 8 + 15) & ~15
        chunks.append(np.ascontiguousarray(px).view(np.uint8).reshape(-1))
    packed = np.concatenate(chunks)
    atlas = np.zeros(size, dtype=np.uint8)
    atlas[:len(packed)] = packed
    return dict(vertices=v, indices=idx, materials=m, texture_ids=ids, atlas=atlas.view(np.uint32).reshape(-1, 4),
                scene_min=np.array(mn, dtype=np.float32), scene_max=np.array(mx, dtype=np.float32), camera=CAMERAS[name],
                texture_names=ordered, name=name)


def oracle_outputs(sc, w=64, h=36, calls=9):
    morton = oracle.sort_primitives(sc)
    assert np.array_equal(morton, oracle.sort_primitives(sc, as_written=True))
    bvh, count = oracle.build_bvh(sc, morton)
    out = dict(sorted_codes=morton["x"].copy(), sorted_values=morton["y"].copy(), bvh=bvh[:count].copy())
    p = oracle.Pipeline(sc, w, h, seed_raygen=1, seed_trace=2)
    p.render(calls)
    out["image"], out["result"] = p.output(), p.result()
    last, total = p.stats()
    out["nodes_visited"], out["tris_tested"], out["rays_hit"] = total["nodes_visited"], total["tris_tested"], total["rays_hit"]
    p.close()
    return out


def save_scene(path, sc, extra):
    cam = sc["camera"]
    np.savez_compressed(path, vertices=sc["vertices"], materials=sc["materials"], texture_ids=sc["texture_ids"],
                        scene_min=sc["scene_min"], scene_max=sc["scene_max"],
                        camera=np.array(list(cam["position"]) + list(cam["focus"]) + list(cam["up"]) +
                                        [cam["fov"], cam["near_z"], cam["far_z"]], dtype=np.float64),
                        texture_names=np.array(sc.get("texture_names", [""])), **extra)


def main():
    os.makedirs(GOLD, exist_ok=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref", "-s"], check=True)
    with tempfile.TemporaryDirectory() as tmp:
        textures = {}
        for t in ("Testure1",):
            px = ref_texture(t, tmp)
            rgb8 = (px[..., :3] >> 8).astype(np.uint8)
            # the decoder's 16-bit output for this 8-bit RGB file is exactly v * 257 with opaque alpha
            assert np.array_equal(px[..., :3], rgb8.astype(np.uint16) * 257) and np.all(px[..., 3] == 65535)
            np.savez_compressed(os.path.join(GOLD, "ref_texture_%s.npz" % t), rgb8=rgb8,
                                first_texel=px[0, 0].copy(), widened="v*257, alpha 65535")
            textures[t + ".png"] = px
        for name in ("testobject4_2", "testobject5_1", "testscene2"):
            corners, mat_ids, mtl, texnames = ref_mesh(name, tmp)
            np.savez_compressed(os.path.join(GOLD, "ref_mesh_%s.npz" % name), corners=corners, material_ids=mat_ids, mtl=mtl,
                                texture_names=np.array(texnames))
            sc = assemble_scene(name, corners, mat_ids, mtl, texnames, textures)
            extra = oracle_outputs(sc)
            if name != "testobject4_2":
                extra["atlas"] = sc["atlas"]          # the textured atlas is rebuilt from ref_texture_*.npz by the tests
            save_scene(os.path.join(GOLD, "oracle_%s.npz" % name), sc, extra)
            print(name, len(corners) // 3, "triangles,", len(mtl), "materials, hit rays", extra["rays_hit"])
    sc = scenes.synthetic_scene(4096)
    extra = oracle_outputs(sc)
    np.savez_compressed(os.path.join(GOLD, "oracle_synthetic4096.npz"), **{k: extra[k] for k in extra})
    print("synthetic4096 done")


if __name__ == "__main__":
    main()
