"""ctypes binding of libgpurt_host.so: the C++ host classes (RaytracerPipeline, RaytracerMesh,
TextureAtlas, scene loading) wrapped for scripts. Camera matrices, scene loading and atlas packing
used by the benchmark come from here, i.e. from the product's own host code."""
import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgpurt_host.so")


class HostSettings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_rays_per_pixel", C.c_uint32), ("max_ray_depth", C.c_uint32),
                ("aa_spread", C.c_float), ("dof_spread", C.c_float), ("use_bvh", C.c_int32), ("fov", C.c_float),
                ("near_z", C.c_float), ("far_z", C.c_float), ("cam_position", C.c_float * 3), ("cam_focus", C.c_float * 3),
                ("cam_up", C.c_float * 3), ("rebuild_every_cycle", C.c_int32), ("fused_build", C.c_int32),
                ("seed_raygen", C.c_uint32), ("seed_trace", C.c_uint32), ("dynamic_scene", C.c_int32),
                ("reference_order_walk", C.c_int32), ("drop_missed_rays", C.c_int32), ("no_previews", C.c_int32), ("refit", C.c_int32), ("device_rank", C.c_uint32),
                ("device_count", C.c_uint32)]


_P = C.c_void_p
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()      # libgpurt.so first (RTLD_GLOBAL not needed: the host library links it by rpath)
    if not os.path.exists(LIB_PATH):
        raise capi.GpurtMissing("%s is missing: build it with `make -C gpuraytracer_amd/csrc host`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    f3 = C.POINTER(C.c_float)
    sig = {
        "rth_camera_matrices": (None, [f3, f3, f3, C.c_float, C.c_float, C.c_float, C.c_float, f3, f3]),
        "rth_parse_real": (C.c_float, [C.c_char_p]),
        "rth_load_mesh": (_P, [C.c_char_p]),
        "rth_load_mesh_dedup": (_P, [C.c_char_p]),
        "rth_load_mesh_threads": (_P, [C.c_char_p, C.c_int, C.c_uint32]),
        "rth_synthetic_scene": (_P, [C.c_uint32, C.c_uint32]),
        "rth_mesh_from_arrays": (_P, [_P, C.c_uint64, _P, C.c_uint64]),
        "rth_gate_create": (_P, [C.c_int]), "rth_gate_arrive": (C.c_int, [_P, C.c_int]), "rth_gate_leave": (None, [_P]),
        "rth_gate_failed": (C.c_int, [_P]), "rth_gate_destroy": (None, [_P]),
        "rth_mesh_index_count": (C.c_uint64, [_P]), "rth_mesh_vertex_count": (C.c_uint64, [_P]),
        "rth_mesh_material_count": (C.c_uint64, [_P]), "rth_mesh_texture_count": (C.c_uint64, [_P]),
        "rth_mesh_texture_name": (C.c_char_p, [_P, C.c_uint64]),
        "rth_mesh_copy": (None, [_P, _P, _P, _P, f3, f3]),
        "rth_mesh_free": (None, [_P]),
        "rth_load_texture": (_P, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "rth_free": (None, [_P]),
        "rth_build_atlas": (C.c_uint64, [_P, C.c_char_p, _P, C.c_uint64, _P, C.c_uint64]),
        "rth_pipeline_create": (_P, [_P, _P, C.POINTER(HostSettings)]),
        "rth_pipeline_render": (C.c_int, [_P, C.c_uint32]),
        "rth_pipeline_read_output": (C.c_int, [_P, _P]), "rth_pipeline_read_result": (C.c_int, [_P, _P]),
        "rth_pipeline_bvh": (_P, [_P]), "rth_pipeline_morton": (_P, [_P]), "rth_pipeline_slot": (_P, [_P, C.c_uint32]),
        "rth_pipeline_save_state": (C.c_int, [_P, C.c_char_p]), "rth_pipeline_load_state": (C.c_int, [_P, C.c_char_p]),
        "rth_pipeline_num_samples": (C.c_uint32, [_P]),
        "rth_pipeline_local_rows": (C.c_uint32, [_P]),
        "rth_pipeline_gather_output": (C.c_int, [_P, _P, _P]),
        "rth_pipeline_destroy": (None, [_P]),
        "rth_write_ppm": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
        "rth_write_pfm": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
        "rth_write_png": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
        "rth_encode_srgb8": (C.c_uint32, [C.c_float]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_matrices(position, focus, up, fov, aspect, near_z, far_z):
    iv, ip = (C.c_float * 16)(), (C.c_float * 16)()
    load().rth_camera_matrices(_f3(position), _f3(focus), _f3(up), fov, aspect, near_z, far_z, iv, ip)
    return np.array(iv, dtype=np.float32), np.array(ip, dtype=np.float32)


def parse_real(text):
    return load().rth_parse_real(text.encode())


def _mesh_to_scene(lib, h, texture_prefix="", camera=None, name=""):
    ni, nv, nm = lib.rth_mesh_index_count(h), lib.rth_mesh_vertex_count(h), lib.rth_mesh_material_count(h)
    idx = np.zeros(ni, dtype=np.uint32)
    vtx = np.zeros(nv, dtype=capi.vertex_dtype)
    mat = np.zeros(nm, dtype=capi.material_dtype)
    mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
    lib.rth_mesh_copy(h, idx.ctypes.data_as(_P), vtx.ctypes.data_as(_P), mat.ctypes.data_as(_P), mn, mx)
    names = [lib.rth_mesh_texture_name(h, i).decode() for i in range(lib.rth_mesh_texture_count(h))]
    packed = lib.rth_build_atlas(h, texture_prefix.encode(), None, 0, None, 0)
    if packed == 0:
        raise RuntimeError("texture atlas could not be built (missing texture file?)")
    nbytes, nids = packed & ((1 << 40) - 1), packed >> 40
    atlas = np.zeros(nbytes, dtype=np.uint8)
    ids = np.zeros(nids, dtype=capi.texture_id_dtype)
    lib.rth_build_atlas(h, texture_prefix.encode(), atlas.ctypes.data_as(_P), nbytes, ids.ctypes.data_as(_P), nids)
    return dict(vertices=vtx, indices=idx, materials=mat, texture_ids=ids, atlas=atlas.view(np.uint32).reshape(-1, 4),
                scene_min=np.array(mn, dtype=np.float32), scene_max=np.array(mx, dtype=np.float32), texture_names=names,
                camera=camera, name=name)


def load_scene(obj_path, texture_prefix="", camera=None, dedup=False, threads=0):
    """LoadMeshFromFile + TextureAtlas packing through the C++ host; returns a scene dict. dedup: equal vertices are
    merged (LoadMeshFromFile's bDeduplicate); threads: 1 = the file is parsed line by line, more = by that many threads,
    0 = chosen by the file's size."""
    lib = load()
    h = lib.rth_load_mesh_threads(obj_path.encode(), int(dedup), int(threads))
    if not h:
        raise RuntimeError("LoadMeshFromFile failed for %s" % obj_path)
    try:
        return _mesh_to_scene(lib, h, texture_prefix, camera, os.path.basename(obj_path))
    finally:
        lib.rth_mesh_free(h)


def synthetic_scene(n, seed=12345):
    from .scenes import SYNTHETIC_CAMERA
    lib = load()
    h = lib.rth_synthetic_scene(n, seed)
    if not h:
        raise ValueError("synthetic scene needs at least 15 triangles")
    try:
        return _mesh_to_scene(lib, h, "", dict(SYNTHETIC_CAMERA), "synthetic-%d" % n)
    finally:
        lib.rth_mesh_free(h)


def load_texture(path):
    lib = load()
    w, h = C.c_uint32(), C.c_uint32()
    p = lib.rth_load_texture(path.encode(), C.byref(w), C.byref(h))
    if not p:
        raise RuntimeError("LoadTextureFromFile failed for %s" % path)
    n = w.value * h.value * 4
    out = np.frombuffer((C.c_uint16 * n).from_address(p), dtype=np.uint16).copy().reshape(h.value, w.value, 4)
    lib.rth_free(p)
    return out


def write_image(path, pixels, kind=None):
    """WriteImagePPM / WriteImagePNG / WriteImagePFM of the C++ host: pixels is a (height, width, 4) float32 image."""
    lib = load()
    px = np.ascontiguousarray(pixels, dtype=np.float32)
    h, w = px.shape[:2]
    kind = kind or str(path).rsplit(".", 1)[-1].lower()
    fn = {"ppm": lib.rth_write_ppm, "png": lib.rth_write_png, "pfm": lib.rth_write_pfm}[kind]
    if not fn(str(path).encode(), px.ctypes.data_as(_P), w, h):
        raise RuntimeError("writing %s failed" % path)


class HostPipeline:
    """RaytracerPipeline (C++) driven from Python: Initialize once, Render() per bounce."""

    def __init__(self, dev, mesh_source, width, height, camera, rpp=1, depth=3, aa=1.5, dof=0.0, use_bvh=True, fused_build=True,
                 rebuild_every_cycle=False, seed_raygen=1, seed_trace=2, dynamic=False, reference_order_walk=False,
                 drop_missed_rays=False, device_rank=0, device_count=1, previews=True, refit=False):
        self.lib = load()
        self.dev = dev
        self.width, self.height = width, height
        self.local_rows = max(0, (height - device_rank + device_count - 1) // device_count)
        if isinstance(mesh_source, int):
            mesh = self.lib.rth_synthetic_scene(mesh_source, 12345)
        elif isinstance(mesh_source, dict):          # a scene held in arrays (a triangle soup without textures)
            v = np.ascontiguousarray(mesh_source["vertices"])
            m = np.ascontiguousarray(mesh_source["materials"])
            mesh = self.lib.rth_mesh_from_arrays(v.ctypes.data_as(C.c_void_p), len(v), m.ctypes.data_as(C.c_void_p), len(m))
        else:
            mesh = self.lib.rth_load_mesh(str(mesh_source).encode())
        if not mesh:
            raise RuntimeError("scene could not be loaded")
        s = HostSettings()
        s.width, s.height, s.max_rays_per_pixel, s.max_ray_depth = width, height, rpp, depth
        s.aa_spread, s.dof_spread, s.use_bvh = aa, dof, 1 if use_bvh else 0
        s.fov, s.near_z, s.far_z = camera["fov"], camera["near_z"], camera["far_z"]
        for k in range(3):
            s.cam_position[k] = camera["position"][k]
            s.cam_focus[k] = camera["focus"][k]
            s.cam_up[k] = camera["up"][k]
        s.rebuild_every_cycle, s.fused_build = int(rebuild_every_cycle), int(fused_build)
        s.seed_raygen, s.seed_trace = seed_raygen, seed_trace
        s.dynamic_scene = int(dynamic)
        s.reference_order_walk = int(reference_order_walk)
        s.drop_missed_rays = int(drop_missed_rays)
        s.no_previews = int(not previews)
        s.refit = int(refit)
        s.device_rank, s.device_count = int(device_rank), int(device_count)
        self.handle = self.lib.rth_pipeline_create(dev.handle, mesh, C.byref(s))
        if not self.handle:
            raise RuntimeError("RaytracerPipeline::Initialize failed")

    def render(self, calls=1):
        if not self.lib.rth_pipeline_render(self.handle, calls):
            raise RuntimeError("RaytracerPipeline::Render failed")

    def gather_output(self, comm, image):
        """RaytracerPipeline::GatherOutput: collective; comm is an rt_comm handle, image an rt_buffer handle (rank 0) or None."""
        if not self.lib.rth_pipeline_gather_output(self.handle, comm, image):
            raise RuntimeError("GatherOutput failed")

    def output(self):
        out = np.zeros((self.local_rows, self.width, 4), dtype=np.float32)
        if not self.lib.rth_pipeline_read_output(self.handle, out.ctypes.data_as(_P)):
            raise RuntimeError("ReadOutput failed")
        return out

    def result(self):
        out = np.zeros((self.local_rows, self.width, 4), dtype=np.float32)
        if not self.lib.rth_pipeline_read_result(self.handle, out.ctypes.data_as(_P)):
            raise RuntimeError("ReadResult failed")
        return out

    def save_state(self, path):
        """Checkpoint of the accumulation (running-mean image + sample counter), between two sample cycles."""
        if not self.lib.rth_pipeline_save_state(self.handle, str(path).encode()):
            raise RuntimeError("SaveState failed")

    def load_state(self, path):
        if not self.lib.rth_pipeline_load_state(self.handle, str(path).encode()):
            raise RuntimeError("LoadState failed")

    def num_samples(self):
        return int(self.lib.rth_pipeline_num_samples(self.handle))

    def read_buffer(self, which, dtype, count):
        h = self.lib.rth_pipeline_bvh(self.handle) if which == "bvh" else self.lib.rth_pipeline_morton(self.handle)
        out = np.zeros(count, dtype=dtype)
        capi.check(self.dev.lib.rt_buffer_readback(h, out.ctypes.data_as(_P), out.nbytes, 0), "rt_buffer_readback")
        return out

    def close(self):
        if self.handle:
            self.lib.rth_pipeline_destroy(self.handle)
            self.handle = None
