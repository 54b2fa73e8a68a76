"""Deterministic synthetic scenes for the large benchmark configurations.

The reference ships no large scene, so the multi-million-triangle configurations use a
generated one: N - 12 small random triangles inside the cube [-10, 10]^3, eight materials
assigned round-robin (one of them a light), and a 12-triangle box [-25, 25]^3 around both the
triangles and the camera so that every path keeps bouncing until it reaches a light or its depth
limit. All values derive from the 32-bit Mersenne Twister seeded with 12345 (the same sequence
as std::mt19937), mapped to [0, 1) by (u >> 8) * 2^-24, in binary32 arithmetic.
"""
import numpy as np

from .capi import material_dtype, texture_id_dtype, vertex_dtype

SYNTHETIC_CAMERA = dict(position=(0.0, 0.0, -22.0), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                        fov=1.2, near_z=0.01, far_z=10000.0)
BOX_HALF = 25.0


def mt19937_raw(seed, count):
    """First `count` 32-bit outputs of the Mersenne Twister initialised with init_genrand(seed)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed))
    return bg.random_raw(int(count)).astype(np.uint32)


def _unit01(raw):
    return (raw >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def _normalize(v):
    n = np.sqrt((v * v).sum(axis=1, dtype=np.float32)).astype(np.float32)
    n = np.where(n == 0, np.float32(1), n)
    return (v / n[:, None]).astype(np.float32)


def _box_triangles(h):
    c = np.array([[-h, -h, -h], [h, -h, -h], [h, h, -h], [-h, h, -h], [-h, -h, h], [h, -h, h], [h, h, h], [-h, h, h]],
                 dtype=np.float32)
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    tris = []
    for a, b, cc, d in quads:
        tris.append((c[a], c[b], c[cc]))
        tris.append((c[a], c[cc], c[d]))
    return np.array(tris, dtype=np.float32)  # [12, 3, 3]


def synthetic_materials():
    m = np.zeros(8, dtype=material_dtype)
    albedo = [(0.8, 0.8, 0.8), (0.8, 0.2, 0.2), (0.2, 0.8, 0.2), (0.2, 0.2, 0.8), (0.8, 0.8, 0.2), (0.2, 0.8, 0.8),
              (0.8, 0.2, 0.8), (0.8, 0.8, 0.8)]
    for i in range(8):
        m[i]["albedo"] = albedo[i]
        m[i]["roughness"] = 0.5
        m[i]["f0_color"] = (0.04, 0.04, 0.04)
        m[i]["metallic"] = 0.0
    m[7]["emissive"] = (10.0, 10.0, 10.0)
    return m


def default_atlas():
    """Atlas with only the 1x1 white default texture (8 bytes of 0xff, padded to one 16-byte element)."""
    atlas = np.zeros((1, 4), dtype=np.uint32)
    atlas[0, 0] = 0xFFFFFFFF
    atlas[0, 1] = 0xFFFFFFFF
    ids = np.zeros(1, dtype=texture_id_dtype)
    ids[0] = (0, 1, 1, 1)
    return atlas, ids


def centroid_bounds(vertices, indices):
    """Scene bounds over triangle centroids, (p1 + p2 + p3) * (1/3) in binary32."""
    p = vertices["position"][indices.astype(np.int64)].reshape(-1, 3, 3)
    c = ((p[:, 0] + p[:, 1]).astype(np.float32) + p[:, 2]).astype(np.float32) * np.float32(1.0 / 3.0)
    return c.min(axis=0).astype(np.float32), c.max(axis=0).astype(np.float32)


def synthetic_scene(num_triangles, seed=12345):
    """Returns a dict: vertices, indices, materials, texture_ids, atlas, scene_min, scene_max, camera."""
    n = int(num_triangles)
    if n < 15:
        raise ValueError("need at least 15 triangles (12 belong to the enclosing box)")
    nr = n - 12
    r = _unit01(mt19937_raw(seed, 9 * nr)).reshape(nr, 9)
    e = np.float32(20.0) / np.float32(np.cbrt(np.float64(n)))
    centre = (np.float32(-10.0) + np.float32(20.0) * r[:, 0:3]).astype(np.float32)
    e1 = (e * (np.float32(2.0) * r[:, 3:6] - np.float32(1.0))).astype(np.float32)
    e2 = (e * (np.float32(2.0) * r[:, 6:9] - np.float32(1.0))).astype(np.float32)
    pos = np.empty((n, 3, 3), dtype=np.float32)
    pos[:nr, 0] = centre
    pos[:nr, 1] = centre + e1
    pos[:nr, 2] = centre + e2
    pos[nr:] = _box_triangles(BOX_HALF)
    d1 = pos[:, 1] - pos[:, 0]
    d2 = pos[:, 2] - pos[:, 0]
    normal = _normalize(np.cross(d1, d2).astype(np.float32))
    tangent = _normalize(d1.astype(np.float32))
    v = np.zeros(3 * n, dtype=vertex_dtype)
    v["position"] = pos.reshape(-1, 3)
    uv = np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32)
    v["uv"] = np.tile(uv, (n, 1))
    v["normal"] = np.repeat(normal, 3, axis=0)
    v["tangent"] = np.repeat(tangent, 3, axis=0)
    mat = (np.arange(n, dtype=np.uint32) % np.uint32(8))
    mat[nr:] = 0
    v["material_id"] = np.repeat(mat, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    atlas, ids = default_atlas()
    smin, smax = centroid_bounds(v, idx)
    return dict(vertices=v, indices=idx, materials=synthetic_materials(), texture_ids=ids, atlas=atlas,
                scene_min=smin, scene_max=smax, camera=dict(SYNTHETIC_CAMERA), name="synthetic-%d" % n)
