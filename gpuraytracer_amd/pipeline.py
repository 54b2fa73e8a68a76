"""Python driver of the hot path over the gpurt C-ABI.

Mirrors the call order of the reference's RaytracerPipeline::Render
(src/RaytracerPipeline.cpp:939-991): [build] -> camera rays every MAX_RAY_DEPTH calls -> trace ->
final image. It exists for the benchmark and the parity tests; the C++ classes under
gpuraytracer_amd/host/ are the drop-in host surface. (first_row, num_rows, row_stride) restricts a
pipeline to some rows of the image for multi-GPU sharding; random seeds use global pixel and ray
indices, so any split of the rows reproduces the single-device image exactly.
"""
import ctypes as C

import numpy as np

from . import capi
from .runtime import Accel, Device


class Pipeline:
    def __init__(self, dev, scene, width, height, inverse_view, inverse_projection, rpp=1, depth=3, aa=1.5, dof=0.0,
                 use_bvh=True, use_accel=True, raygen_seeds=None, trace_seeds=None, first_row=0, num_rows=None,
                 row_stride=1, rebuild_every_cycle=False, output_ptr=None, dynamic=False, refit=False):
        self.dev = dev
        self.lib = dev.lib
        self.width, self.height, self.rpp, self.depth = int(width), int(height), int(rpp), int(depth)
        self.first_row = int(first_row)
        self.row_stride = int(row_stride)
        left = self.height - self.first_row
        self.num_rows = (left + self.row_stride - 1) // self.row_stride if num_rows is None else int(num_rows)
        self.rows = capi.ImageRows(self.first_row, self.num_rows, self.row_stride)
        self.use_bvh, self.use_accel = bool(use_bvh), bool(use_accel and use_bvh)
        self.rebuild_every_cycle = bool(rebuild_every_cycle) or bool(dynamic)
        # dynamic: before every rebuild the synthetic animation moves the triangles on the device, and the
        # centroid bounds the Morton codes are scaled by are reduced on the device, too
        self.dynamic = bool(dynamic)
        # refit: after the first build of a moving scene the sorted order is kept and only leaves and levels are made
        # again (rt_accel_refit); needs the fused build
        self.refit = bool(refit)
        self.builds = 0
        self.frame = 0
        self.raygen_seeds = raygen_seeds    # callable returning 3 uint32 per camera-ray generation
        self.trace_seeds = trace_seeds      # callable returning 3 uint32 per trace dispatch
        self.num_indices = len(scene["indices"])
        self.num_prims = self.num_indices // 3
        self.scene_min = np.asarray(scene["scene_min"], dtype=np.float32)
        self.scene_max = np.asarray(scene["scene_max"], dtype=np.float32)

        self.indices = dev.upload(scene["indices"])
        self.vertices = dev.upload(scene["vertices"])
        self.base_vertices = dev.upload(scene["vertices"]) if self.dynamic else None
        self.bounds = dev.alloc(32) if self.dynamic else None
        self.materials = dev.upload(scene["materials"])
        self.texture_ids = dev.upload(scene["texture_ids"])
        atlas = np.ascontiguousarray(scene["atlas"], dtype=np.uint32).reshape(-1, 4)
        self.atlas_elements = atlas.shape[0]
        self.atlas = dev.upload(atlas)

        self.local_rays = self.num_rows * self.width * self.rpp
        nr = max(self.local_rays, 1)
        self.rays = dev.alloc(32 * nr)
        self.old_rays = dev.alloc(32 * nr)
        self.ray_pixels = dev.alloc(4 * nr)
        self.scattered = dev.alloc(16 * nr)
        self.emitted = dev.alloc(16 * nr)
        npx = max(self.num_rows * self.width, 1)
        self.result = dev.alloc(16 * npx)
        # the tone-mapped band may live in caller memory (e.g. a torch tensor that a collective will read)
        self.output = dev.wrap(output_ptr, 16 * npx) if output_ptr else dev.alloc(16 * npx)

        self.bvh = None
        self.accel = None
        self.morton = self.temp = self.freq = None
        if self.use_bvh:
            nodes = self.lib.rt_bvh_node_count(self.num_prims)
            if nodes == 0:
                raise capi.GpurtError(capi.RT_ERR_TOO_FEW_PRIMITIVES, "rt_bvh_node_count")
            self.bvh = dev.alloc(32 * nodes)
            if self.use_accel:
                self.accel = Accel(dev, self.num_prims)
                self.accel.pack(self.num_prims, self.indices, self.vertices)
            else:
                self.morton = dev.alloc(16 * self.num_prims)
                self.temp = dev.alloc(16 * self.num_prims)
                self.freq = dev.alloc(16 * 256)

        self.rg = capi.CameraRayGenInfo()
        for k in range(16):
            self.rg.inverse_view[k] = float(inverse_view[k])
            self.rg.inverse_projection[k] = float(inverse_projection[k])
        self.rg.screen_size[0], self.rg.screen_size[1] = self.width, self.height
        self.rg.aa_sample_spread, self.rg.dof_sample_spread = aa, dof
        self.rg.max_rays_per_pixel = self.rpp
        self.tr = capi.TraceRaysInfo()
        self.tr.screen_dimensions[0], self.tr.screen_dimensions[1] = self.width, self.height
        self.tr.num_indices = self.num_indices
        self.tr.num_rays = self.local_rays
        self.tr.max_rays_per_pixel = self.rpp
        self.im = capi.GenerateFinalImageInfo()
        self.im.screen_dimensions[0], self.im.screen_dimensions[1] = self.width, self.num_rows
        self.im.max_rays_per_pixel = self.rpp
        self.im.num_samples = 1
        self.built = False
        self.iteration = 0

    # ---- LBVH build ----
    def build_fused(self):
        if self.dynamic and self.refit and self.builds > 0:
            self.accel.pack(self.num_prims, self.indices, self.vertices)
            self.accel.refit(self.bvh)
        elif self.dynamic:
            self.accel.pack_bounds(self.num_prims, self.indices, self.vertices, self.bounds)
            self.accel.build_device_bounds(self.bounds, self.bvh)
        else:
            self.accel.build(self.scene_min, self.scene_max, self.bvh)
        self.builds += 1

    def animate(self):
        """Moves the triangles to their places of frame self.frame (synthetic dynamic workload)."""
        capi.check(self.lib.rt_synthetic_animate(self.dev.handle, self.num_prims, self.frame, self.base_vertices.handle,
                                                 self.vertices.handle), "rt_synthetic_animate")
        self.frame += 1

    def build_per_shader(self):
        """SortPrimitives::Sort + BuildBVH::Build with one launch per reference dispatch."""
        lib, d = self.lib, self.dev.handle
        n = self.num_prims
        mi = capi.MortonCodeInfo()
        for k in range(3):
            mi.scene_min[k] = float(self.scene_min[k])
            mi.scene_max[k] = float(self.scene_max[k])
        mi.num_primitives = n
        capi.check(lib.rt_generate_morton_codes(d, C.byref(mi), self.indices.handle, self.vertices.handle, self.morton.handle,
                                                self.freq.handle), "rt_generate_morton_codes")
        for p in range(4):
            si = capi.SortInfo(n, p)
            capi.check(lib.rt_sort_count(d, C.byref(si), self.morton.handle, self.temp.handle, self.freq.handle), "rt_sort_count")
            capi.check(lib.rt_sort_prefix_sum(d, C.byref(si), self.freq.handle), "rt_sort_prefix_sum")
            capi.check(lib.rt_sort_sort(d, C.byref(si), self.morton.handle, self.temp.handle, self.freq.handle), "rt_sort_sort")
        bi = capi.BVHInfo(n, 1, (n + 1) // 2 + 1, 0)
        capi.check(lib.rt_bvh_build_leaves(d, C.byref(bi), self.indices.handle, self.vertices.handle, self.morton.handle,
                                           self.bvh.handle), "rt_bvh_build_leaves")
        bi.num_children = (bi.num_children + 1) // 2
        while bi.num_children != 2:
            snap = capi.BVHInfo(bi.num_children, bi.previous_index, bi.current_index, 0)
            bi.num_children = (bi.num_children + 1) // 2
            capi.check(lib.rt_bvh_build(d, C.byref(snap), self.bvh.handle), "rt_bvh_build")
            bi.previous_index = bi.current_index
            bi.current_index += bi.num_children
        root = capi.BVHInfo(2, bi.previous_index, 0, 0)
        capi.check(lib.rt_bvh_build(d, C.byref(root), self.bvh.handle), "rt_bvh_build")

    def build(self):
        if self.dynamic:
            self.animate()
            if not self.use_accel:        # the per-shader launchers take the bounds from the host
                capi.check(self.lib.rt_scene_bounds(self.dev.handle, self.num_prims, self.indices.handle, self.vertices.handle,
                                                    self.bounds.handle), "rt_scene_bounds")
                b = self.bounds.download(np.float32, 8)
                self.scene_min, self.scene_max = b[0:3].copy(), b[4:7].copy()
        if self.use_accel:
            self.build_fused()
        else:
            self.build_per_shader()
        self.built = True

    # ---- stages ----
    def camera_rays(self):
        s = self.raygen_seeds() if self.raygen_seeds else (0, 0, 0)
        for k in range(3):
            self.rg.rng_seed[k] = int(s[k])
        capi.check(self.lib.rt_camera_ray_generation(self.dev.handle, C.byref(self.rg), self.rays.handle, self.old_rays.handle,
                                                     self.ray_pixels.handle, C.byref(self.rows)),
                   "rt_camera_ray_generation")

    def trace(self):
        s = self.trace_seeds() if self.trace_seeds else (0, 0, 0)
        for k in range(3):
            self.tr.rng_seed[k] = int(s[k])
        args = (self.indices.handle, self.vertices.handle, self.materials.handle, self.texture_ids.handle, self.atlas.handle,
                self.atlas_elements, self.rays.handle, self.old_rays.handle, self.ray_pixels.handle, self.scattered.handle,
                self.emitted.handle, self.bvh.handle if self.bvh else None, C.byref(self.rows))
        if self.use_accel:
            capi.check(self.lib.rt_trace_rays_accel(self.dev.handle, C.byref(self.tr), self.accel.handle, *args),
                       "rt_trace_rays_accel")
        else:
            capi.check(self.lib.rt_trace_rays(self.dev.handle, C.byref(self.tr), 1 if self.use_bvh else 0, *args), "rt_trace_rays")

    def final_image(self, apply):
        if apply:
            self.im.num_samples = ((self.im.num_samples | 0x80000000) + 1) & 0xFFFFFFFF
        else:
            self.im.num_samples &= 0x7FFFFFFF
        capi.check(self.lib.rt_generate_final_image(self.dev.handle, C.byref(self.im), self.emitted.handle, self.result.handle,
                                                    self.output.handle, None), "rt_generate_final_image")

    def render(self, calls=1, previews=True):
        """One call = one bounce, as RaytracerPipeline::Render. previews=False leaves out the final-image kernel
        between the bounces of a sample cycle, where it only refreshes the displayed image (the accumulated image
        and the display after the cycle's last bounce are the same either way)."""
        for _ in range(calls):
            if self.use_bvh and (not self.built or (self.rebuild_every_cycle and self.iteration == 0)):
                self.build()
            if self.iteration == 0:
                self.camera_rays()
            self.iteration += 1
            if self.iteration == self.depth:
                self.iteration = 0
            self.trace()
            if previews or self.iteration == 0:
                self.final_image(self.iteration == 0)

    # ---- readback ----
    def output_image(self):
        self.dev.sync()
        return self.output.download(np.float32, 4 * self.num_rows * self.width).reshape(self.num_rows, self.width, 4)

    def result_image(self):
        self.dev.sync()
        return self.result.download(np.float32, 4 * self.num_rows * self.width).reshape(self.num_rows, self.width, 4)

    def close(self):
        self.dev.sync()
        if self.accel:
            self.accel.destroy()
        for b in (self.base_vertices, self.bounds, self.indices, self.vertices, self.materials, self.texture_ids, self.atlas, self.rays, self.old_rays,
                  self.ray_pixels, self.scattered, self.emitted, self.result, self.output, self.bvh, self.morton, self.temp,
                  self.freq):
            if b is not None:
                b.free()
