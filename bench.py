#!/usr/bin/env python3
"""Benchmark of the LBVH-build + ray-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one frame of the workload: rebuild the LBVH (Morton codes, radix sort, leaves, levels),
generate camera rays, trace `depth` bounces (each followed by the final-image pass), and - with more
than one GPU - gather the finished image rows to rank 0. Inputs are resident in HBM before the timed
region. Default workload ("c3"): the 1M-triangle synthetic scene at 3840x2160, 1 ray per pixel,
4 bounces, LBVH rebuilt every frame. With N GPUs the image rows are dealt out in turn (rank r owns rows
r, r + N, ...), the scene and hierarchy are replicated and rebuilt on every GPU, and the frame carries
N rays per pixel (weak scaling: every GPU traces width x height rays per bounce, as the single GPU
does; --scaling strong keeps 1 ray per pixel instead).

Rank 0 prints ONE JSON line: Mrays/s over the whole job (a ray = one ray segment handled by one trace
thread in one trace launch), the LBVH build time, a roofline object for the dominant kernel (the
traversal launch: bytes across the memory side from the committed hardware-counter passes over the launch duration
measured live, the vector units' utilisation from the same passes) and the CPU baseline (the scalar
OpenMP oracle timed on a bounded sample of the same workload on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak
# Algorithmic bytes of one LBVH build per triangle in the reference's data layout: Morton codes 56 (12 B of indices,
# 36 B of positions, 8 B key + value out), radix sort 80 (4 passes x (4 B key read for the count + 8 B read + 8 B
# written)), leaves 68, upper levels 48.
BUILD_BYTES_PER_TRIANGLE = 252
ROUND = "r3"
NUM_SIMDS = 1024                # 256 compute units x 4
NUM_XCDS = 8


def source_sha():
    """Hash of everything the kernels are compiled from (csrc/*.hip, csrc/*.h, include/*.h): what ties a committed
    counter file to the kernels it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "gpuraytracer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gpuraytracer_amd", "csrc", "*.h")) +
                       glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def tuning_from_environment():
    return sorted(k for k in os.environ if k.startswith("GPURT_"))


def load_pmc(workload, kernel):
    """Per-dispatch hardware counters of `kernel` from the committed counter passes of this workload
    (profiles/<round>_<workload>_pmc.json, written by scripts/pmc_summary.py --json from scripts/pmc_passes.sh runs:
    one rocprofv3 --pmc run per counter group). Returns (counters, file, provenance of the file); counters are None when
    there is no such file or kernel, or when the file was measured on other kernel sources than the ones in this tree."""
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (ROUND, workload))
    if not os.path.exists(path):
        return None, None, {"reason": "no counter file profiles/%s_%s_pmc.json" % (ROUND, workload)}
    with open(path) as f:
        table = json.load(f)
    prov = {"file": os.path.relpath(path, ROOT), "traffic_source_sha": table.get("source_sha"), "traffic_source_head": table.get("head"),
            "source_sha": source_sha()}
    if table.get("source_sha") != prov["source_sha"]:
        prov["reason"] = "the counter file was measured on other kernel sources (source_sha differs): fractions withheld"
        return None, prov["file"], prov
    if tuning_from_environment():
        prov["reason"] = "GPURT_* tuning variables are set: the committed counters are those of the defaults"
        return None, prov["file"], prov
    for name in (kernel + "<false>", kernel):
        if name in table["kernels"]:
            return table["kernels"][name], prov["file"], prov
    prov["reason"] = "no kernel %s in the counter file" % kernel
    return None, prov["file"], prov


def counter_fractions(c, avg_launch_s):
    """The two roofline fractions of a traversal launch from its counters:
    memory side = bytes that crossed the L2's memory side (FETCH_SIZE + WRITE_SIZE, KiB; 64-byte gathers, for which
    FETCH_SIZE counts exactly) / launch duration / HBM peak;
    vector side = share of cycles the SIMDs' vector units were issuing (SQ_ACTIVE_INST_VALU, quad-cycles summed over
    the 1024 SIMDs, against GRBM_GUI_ACTIVE summed over the 8 XCDs) x share of the 64 lanes active in those
    instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))."""
    out = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["traffic"] = int(traffic)
        out["memory"] = {"fetch_bytes": int(c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024),
                         "achieved": round(traffic / avg_launch_s / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBPS, 4)}
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            out["memory"]["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 4)
    if all(k in c for k in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")):
        busy = (c["SQ_ACTIVE_INST_VALU"] * 4.0 / NUM_SIMDS) / (c["GRBM_GUI_ACTIVE"] / NUM_XCDS)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        out["vector"] = {"valu_busy": round(busy, 4), "active_lane_fraction": round(lanes, 4), "frac": round(busy * lanes, 4),
                         "unit": "share of the vector units' lane slots doing work", "valu_instructions": int(c.get("SQ_INSTS_VALU", 0))}
    return out


WORKLOADS = {
    # name: (triangles, width, height, rays/pixel, bounces, description)
    "c3": (1000000, 3840, 2160, 1, 4, "C3: synthetic 1M-triangle scene, 3840x2160, 1 ray/pixel, 4 bounces, LBVH rebuilt every frame"),
    "c4": (10000000, 3840, 2160, 1, 4, "C4: synthetic 10M-triangle scene, 3840x2160, 1 ray/pixel, 4 bounces, LBVH rebuilt every frame"),
    "c5": (50000000, 7680, 4320, 1, 4, "C5: synthetic 50M-triangle dynamic scene (triangles move every frame, scene bounds reduced on "
           "the device), 7680x4320, 1 ray/pixel, 4 bounces, LBVH rebuilt every frame"),
    "small": (100000, 1280, 720, 1, 3, "small: synthetic 100k-triangle scene, 1280x720, 1 ray/pixel, 3 bounces, LBVH rebuilt every frame"),
    # the C3 scene without the twelve triangles of its enclosing box: most rays leave the scene after a bounce or two
    # (what rt_device_set_drop_missed_rays is for)
    "open": (1000000, 3840, 2160, 1, 4, "open: the C3 scene without its enclosing box, 3840x2160, 1 ray/pixel, 4 bounces, LBVH rebuilt every frame"),
    # C2: the reference's own assets with the cameras of src/Settings.h:22-35 of the reference, at 1920x1080 and the
    # reference's default depth of 3; the triangles come from the committed loader fixtures (tests/golden/oracle_*.npz)
    "c2_testscene2": (0, 1920, 1080, 1, 3, "C2: the reference's assets/testscene2.obj (10 528 triangles), 1920x1080, 1 ray/pixel, 3 bounces"),
    "c2_testobject4_2": (0, 1920, 1080, 1, 3, "C2: the reference's assets/testobject4_2.obj (992 triangles, textured), 1920x1080, 1 ray/pixel, 3 bounces"),
    "c2_testobject5_1": (0, 1920, 1080, 1, 3, "C2: the reference's assets/testobject5_1.obj (1 330 triangles), 1920x1080, 1 ray/pixel, 3 bounces"),
}
# what a plain run reports beside its headline workload: (workload, frames timed, warm-up frames)
OTHER_WORKLOADS = [("c2_testscene2", 16, 2), ("c2_testobject4_2", 16, 2), ("c2_testobject5_1", 16, 2), ("c4", 5, 2), ("c5", 3, 1), ("open", 5, 2)]


def asset_scene(name):
    """One of the reference's three assets as the loader fixtures hold it (vertices, materials, camera; the atlas of the
    textured one is rebuilt from the decoded-texture fixture: tight packing behind 8 bytes of 0xff)."""
    gold = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gold, "oracle_%s.npz" % name))
    cam = g["camera"]
    sc = dict(vertices=g["vertices"], indices=np.arange(len(g["vertices"]), dtype=np.uint32), materials=g["materials"],
              texture_ids=g["texture_ids"], scene_min=g["scene_min"], scene_max=g["scene_max"],
              camera=dict(position=tuple(cam[0:3]), focus=tuple(cam[3:6]), up=tuple(cam[6:9]), fov=float(cam[9]),
                          near_z=float(cam[10]), far_z=float(cam[11])), name=name)
    if "atlas" in g:
        sc["atlas"] = g["atlas"]
    else:
        t = np.load(os.path.join(gold, "ref_texture_Testure1.npz"))
        rgb = t["rgb8"].astype(np.uint16) * 257
        px = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 65535, dtype=np.uint16)], axis=2)
        size = 16 + ((px.shape[0] * px.shape[1] * 8 + 15) & ~15)
        atlas = np.zeros(size, dtype=np.uint8)
        atlas[:8] = 0xFF
        atlas[8:8 + px.size * 2] = px.reshape(-1).view(np.uint8)
        sc["atlas"] = atlas.view(np.uint32).reshape(-1, 4)
    return sc


def make_scene(workload, ntri):
    from gpuraytracer_amd import host_py, scenes
    if workload.startswith("c2_"):
        return asset_scene(workload[3:])
    scene = host_py.synthetic_scene(ntri)
    if workload == "open":
        scene = dict(scene)
        scene["vertices"] = scene["vertices"][:-36].copy()
        scene["indices"] = scene["indices"][:-36].copy()
        scene["scene_min"], scene["scene_max"] = scenes.centroid_bounds(scene["vertices"], scene["indices"])
        scene["name"] += "-open"
    return scene


def trace_bytes(rays, hits, nodes, tris):
    """Algorithmic bytes of trace launches in the reference's data layout: 100 B of per-ray state read,
    96 B written on a hit (ray, previous ray, two light values) or 32 B on a miss, 32 B per box test,
    12 + 36 B per triangle test, and on a hit three 48-B vertices, a 64-B material and five texture
    lookups of 16 + 8 B."""
    return rays * 100 + hits * (96 + 3 * 48 + 64 + 5 * 24) + (rays - hits) * 32 + nodes * 32 + tris * 48


def measure(args, ctx, workload, steps, warmup, full, use_bvh=True, drop_missed=False):
    """Times `steps` frames of one workload (after `warmup` untimed ones) and returns its record. full: the headline's side
    figures as well (the walk in the reference's order, the reference's test counts, the CPU baseline)."""
    import ctypes

    import torch
    import torch.distributed as dist

    from gpuraytracer_amd import host_py, scenes
    from gpuraytracer_amd.distributed import gather_rows, rows_for_rank
    from gpuraytracer_amd.pipeline import Pipeline
    from gpuraytracer_amd.runtime import Device

    rank, world, local_rank = ctx["rank"], ctx["world"], ctx["local_rank"]
    ntri, width, height, rpp, depth, desc = WORKLOADS[workload]
    custom = full and bool(args.triangles or args.width or args.height)
    if full:
        ntri = args.triangles or ntri
        width = args.width or width
        height = args.height or height

    # scene, camera and host seeds come from the product's own host code (C++ generator and camera maths)
    scene = make_scene(workload, ntri)
    ntri = len(scene["indices"]) // 3
    cam = scene["camera"]
    iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], width / height, cam["near_z"],
                                     cam["far_z"])
    total_frames = steps + warmup + 8      # + the counting frames and the frames of the walk in the reference's order
    rs = iter(scenes.mt19937_raw(1, 3 * total_frames).reshape(-1, 3))
    ts = iter(scenes.mt19937_raw(2, 3 * total_frames * depth).reshape(-1, 3))

    share_rank, share_world = rank, world
    if full and args.band and world == 1:
        share_rank, share_world = (int(x) for x in args.band.split("/"))
    if args.scaling == "weak":
        rpp *= share_world
    first_row, num_rows, row_stride = rows_for_rank(height, share_world, share_rank)
    dynamic = workload == "c5" or (full and args.dynamic)
    refit = full and args.refit
    stream = torch.cuda.Stream()
    dev = Device(local_rank)
    with torch.cuda.stream(stream):
        dev.set_stream(stream.cuda_stream)
        dev.lib.rt_device_set_drop_missed_rays(dev.handle, 1 if drop_missed else 0)
        band = torch.zeros((num_rows, width, 4), dtype=torch.float32, device="cuda")
        pipe = Pipeline(dev, scene, width, height, iv, ip, rpp=rpp, depth=depth, use_bvh=use_bvh, use_accel=use_bvh,
                        raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts), first_row=first_row, num_rows=num_rows,
                        row_stride=row_stride, dynamic=dynamic, rebuild_every_cycle=True, output_ptr=band.data_ptr(),
                        refit=refit)

        build_events, trace_events = [], []
        raw_build, raw_trace = pipe.build, pipe.trace
        timing = {"on": False}

        def timed(fn, store):
            def call():
                if not timing["on"]:
                    return fn()
                a, b = dev.event(), dev.event()
                a.record()
                fn()
                b.record()
                store.append((a, b))
            return call

        pipe.build = timed(raw_build, build_events)
        pipe.trace = timed(raw_trace, trace_events)

        # The gather of a frame's rows runs beside the next frame's kernels: the rows are copied aside on the compute
        # stream, and the collective reads that copy on a stream of its own. Every frame is gathered in full inside
        # the timed region (the final synchronisation waits for the last one).
        comm = torch.cuda.Stream() if world > 1 else None
        aside = torch.empty_like(band) if world > 1 else None

        def step():
            # one frame: build, camera rays, `depth` bounces, the final image once (the reference's Render() also
            # refreshes its display after every bounce; nothing consumes those previews here)
            pipe.render(depth, previews=False)
            if world > 1:
                stream.wait_stream(comm)             # the previous frame's rows have left the copy
                aside.copy_(band, non_blocking=True)
                comm.wait_stream(stream)
                with torch.cuda.stream(comm):
                    return gather_rows(aside, height, width)
            return band

        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timing["on"] = True
        dev.lib.rt_device_set_traverse_timing(dev.handle, 1)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing["on"] = False

        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if ctx["backend"] == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        kernel_ms, kernel_launches = ctypes.c_double(0.0), ctypes.c_uint32(0)
        dev.lib.rt_device_traverse_timing(dev.handle, ctypes.byref(kernel_ms), ctypes.byref(kernel_launches))
        dev.lib.rt_device_set_traverse_timing(dev.handle, 0)
        build_ms = [a.elapsed_ms(b) for a, b in build_events]
        trace_ms = [a.elapsed_ms(b) for a, b in trace_events]
        dev.lib.rt_device_set_drop_missed_rays(dev.handle, 0)

        outliers = None
        if use_bvh:
            found, listed = ctypes.c_uint32(0), ctypes.c_uint32(0)
            dev.lib.rt_accel_outliers(pipe.accel.handle, ctypes.byref(found), ctypes.byref(listed))
            outliers = {"found": found.value, "listed": listed.value}

        # the same frames by the walk in the reference's order (untimed for `value`; reported beside it)
        ref_walk = None
        kernel_name = (os.environ.get("GPURT_TRACE_KERNEL", "") or "spatial") if use_bvh else "simple"
        if full and kernel_name == "spatial" and world == 1 and not args.no_reference_walk:
            dev.lib.rt_device_set_trace_kernel(dev.handle, 0)        # RT_TRACE_KERNEL_QUAD
            dev.lib.rt_accel_set_second_hierarchy(pipe.accel.handle, 0, 21)      # that walk never reads it
            pipe.render(depth)
            torch.cuda.synchronize()
            dev.lib.rt_device_set_traverse_timing(dev.handle, 1)
            ref_build_events = []
            pipe.build = timed(raw_build, ref_build_events)      # builds the reference's hierarchy alone
            timing["on"] = True
            t1 = time.perf_counter()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ref_elapsed = time.perf_counter() - t1
            timing["on"] = False
            pipe.build = timed(raw_build, build_events)
            rk_ms, rk_n = ctypes.c_double(0.0), ctypes.c_uint32(0)
            dev.lib.rt_device_traverse_timing(dev.handle, ctypes.byref(rk_ms), ctypes.byref(rk_n))
            dev.lib.rt_device_set_traverse_timing(dev.handle, 0)
            dev.lib.rt_device_set_trace_kernel(dev.handle, 3)        # RT_TRACE_KERNEL_SPATIAL
            dev.lib.rt_accel_set_second_hierarchy(pipe.accel.handle, 1, 21)
            ref_walk = {"kernel": "k_traverse_quad", "value": round(width * height * rpp * depth * 2 / ref_elapsed / 1e6, 3),
                        "unit": "Mrays/s", "steps": 2, "avg_launch_ms": round(rk_ms.value / max(rk_n.value, 1), 4),
                        "lbvh_build_ms": round(float(np.mean([a.elapsed_ms(b) for a, b in ref_build_events])), 4)}

        # counters of the walk on the library's own hierarchy (one more frame, untimed): its own algorithmic bytes
        walk = None
        if kernel_name == "spatial":
            walk_buf = dev.alloc(32)
            dev.lib.rt_device_set_walk_stats(dev.handle, walk_buf.handle)
            pipe.render(depth)
            dev.sync()
            dev.lib.rt_device_set_walk_stats(dev.handle, None)
            walk = [int(x) for x in walk_buf.download(np.uint64, 4)]
            walk_buf.free()

        # exact traversal counters of one more frame (untimed, instrumented kernel variant), per bounce
        nodes = tris = hits = counted = 0
        hits_by_bounce = None
        want_counts = (full and not args.no_counters) or workload == "open"
        if want_counts and use_bvh:
            stats_buf = dev.alloc(32)
            dev.lib.rt_device_set_trace_stats(dev.handle, stats_buf.handle)
            pipe.build()
            pipe.camera_rays()
            hits_by_bounce, before = [], 0
            for _ in range(depth):
                pipe.trace()
                dev.sync()
                h_now = int(stats_buf.download(np.uint64, 4)[2])
                hits_by_bounce.append(h_now - before)
                before = h_now
            pipe.iteration = 0
            dev.lib.rt_device_set_trace_stats(dev.handle, None)
            nodes, tris, hits, counted = [int(x) for x in stats_buf.download(np.uint64, 4)]
            stats_buf.free()
        img = band.float().cpu().numpy()

    local_rays = num_rows * width * rpp
    rays_per_step = width * height * rpp * depth
    mrays = rays_per_step * steps / elapsed / 1e6
    launches_per_frame = depth
    bytes_per_launch = trace_bytes(counted, hits, nodes, tris) / launches_per_frame
    # the dominant kernel alone (events recorded by the library around k_traverse on its stream); a trace call
    # also runs the shading kernel after it (and, where rays are ordered, the ray-key sort before it)
    avg_trace_s = kernel_ms.value / max(kernel_launches.value, 1) * 1e-3 if kernel_launches.value else float(np.mean(trace_ms)) * 1e-3
    avg_trace_call_ms = float(np.mean(trace_ms))
    achieved = bytes_per_launch / avg_trace_s / 1e9
    avg_build_ms = float(np.mean(build_ms)) if build_ms else 0.0

    plain = not custom and not (full and (args.band or args.dynamic or args.refit)) and args.scaling == "weak" and world == 1 and not drop_missed
    kernel_symbol = {"spatial": "k_traverse_spatial", "quad": "k_traverse_quad", "wave": "k_traverse", "simple": "k_trace"}.get(kernel_name, kernel_name)
    counters, counter_file, prov = load_pmc(workload, kernel_symbol) if plain else (None, None, {"reason": "not the workload as defined"})
    fr = counter_fractions(counters, avg_trace_s) if counters else {}
    mem, vec = fr.get("memory"), fr.get("vector")
    # the roofline of the dominant kernel, the traversal launch: HBM is the contract's bound for this path (no dense
    # contraction anywhere); frac is what crossed the memory side over the launch's duration, against the HBM peak.
    # The unit that binds the kernel today is named in `binding`.
    roofline = {"kernel": kernel_symbol, "bound": "hbm",
                "achieved": mem["achieved"] if mem else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": mem["frac"] if mem else None,
                "traffic": fr.get("traffic"),
                "traffic_source": (counter_file + ": FETCH_SIZE + WRITE_SIZE per launch x 1024 (separate --pmc passes)") if mem else None,
                "source_sha": prov.get("source_sha") or source_sha(), "traffic_source_sha": prov.get("traffic_source_sha"),
                "traffic_source_head": prov.get("traffic_source_head"), "counters_withheld": prov.get("reason"),
                "memory": mem, "vector": vec,
                "binding": (("vector units (vector.frac %.2f against memory.frac %.2f)" % (vec["frac"], mem["frac"]))
                            if vec["frac"] >= mem["frac"] else ("memory side (memory.frac %.2f against vector.frac %.2f)" % (mem["frac"], vec["frac"])))
                if mem and vec else None,
                "avg_launch_ms": round(avg_trace_s * 1e3, 4), "launches_timed": int(kernel_launches.value),
                "avg_trace_call_ms": round(avg_trace_call_ms, 4), "rays_per_launch": local_rays}
    if full:
        roofline["note"] = ("achieved = bytes across the L2's memory side per launch (hardware counters of the committed passes of this "
                            "workload, used only while their source_sha is that of the kernels in this tree) / the launch duration measured "
                            "live in this run (HIP events on the library's stream around the traversal kernel). algorithmic_* are the "
                            "reference walk's own counts in the reference's data layout (100 B per ray + 32 B per box test + 48 B per "
                            "triangle test + hit traffic), kept for reference; the default kernel finds the same closest hits with far "
                            "fewer tests, so they are not a fraction of anything it does")
    if counted and full:
        roofline.update({"algorithmic_bytes_per_launch": int(bytes_per_launch), "algorithmic_gbps_if_performed": round(achieved, 1),
                         "algorithmic_box_tests_per_ray": round(nodes / max(counted, 1), 2),
                         "algorithmic_triangle_tests_per_ray": round(tris / max(counted, 1), 3), "hit_fraction": round(hits / max(counted, 1), 4)})
    build_bytes = BUILD_BYTES_PER_TRIANGLE * ntri
    out = {
        "metric": "Mrays/sec", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not workload.startswith("c2_") else
        "the reference's asset (committed loader fixture), synthetic host seeds",
        "config": {"workload": (desc + (" - weak scaling: %d rays per pixel over %d GPUs" % (rpp, share_world)
                                        if args.scaling == "weak" and share_world > 1 else ""))
                   if not custom else
                   "custom: synthetic %d-triangle scene, %dx%d, %d ray/pixel, %d bounces, LBVH rebuilt every frame" % (ntri, width, height, rpp, depth),
                   "triangles": ntri, "width": width, "height": height, "rays_per_pixel": rpp, "bounces": depth,
                   "rays_per_step": rays_per_step,
                   "hierarchy": "LBVH + the library's own second hierarchy" if use_bvh else "none (RT_USE_BVH 0: every ray tests every triangle)",
                   "drop_missed_rays": bool(drop_missed),
                   "final_image": "once per frame (after the last bounce)",
                   "build": "refit after the first frame (not the workload as defined)" if refit else "rebuilt every frame",
                   "parallelism": "image rows dealt to %d GPU(s) in turn, replicated scene + per-GPU rebuild, gather to rank 0" % world},
        "lbvh_build_ms": round(avg_build_ms, 4),
        "roofline": roofline,
        # 252 B per triangle (reference layout) over the time a frame pays for its build - both hierarchies, and for a
        # moving scene also the animation and the bounds - and, below, over the reference's hierarchy alone
        "build_roofline": {"kernels": "per frame: k_morton_hist, radix passes, k_leaves, k_upsweep for the reference's hierarchy; "
                                      "keys, radix passes, cell layout, k_leaves, k_upsweep for the library's own",
                           "bound": "hbm", "achieved": round(build_bytes / (avg_build_ms * 1e-3) / 1e9, 1) if avg_build_ms else None,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(build_bytes / (avg_build_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if avg_build_ms else None,
                           "algorithmic_bytes_per_triangle": BUILD_BYTES_PER_TRIANGLE, "build_ms_per_frame": round(avg_build_ms, 4),
                           "note": "252 B per triangle count the reference's hierarchy once; build_ms_per_frame covers both hierarchies "
                                   "(for c5 also moving the mesh and its bounds), reference_hierarchy_alone the first one only"},
        "trace_ms_by_bounce": [round(float(x), 3) for x in np.asarray(trace_ms).reshape(-1, depth).mean(axis=0)],
        "image_mean": round(float(img[..., :3].mean()), 6),
    }
    if outliers:
        out["outliers"] = outliers
    if hits_by_bounce:
        out["rays_hit_by_bounce"] = hits_by_bounce
        out["live_rays_by_bounce"] = [local_rays] + hits_by_bounce[:-1]      # a ray that has missed stays as it is and misses again
    if ref_walk:
        out["reference_order_walk"] = ref_walk
        # lbvh_build_ms is what a frame pays: the reference's hierarchy and the library's own; this is the former alone
        out["reference_lbvh_build_ms"] = ref_walk["lbvh_build_ms"]
        rb = ref_walk["lbvh_build_ms"] * 1e-3
        out["build_roofline"]["reference_hierarchy_alone"] = {"build_ms": ref_walk["lbvh_build_ms"], "achieved": round(build_bytes / rb / 1e9, 1),
                                                              "frac": round(build_bytes / rb / 1e9 / HBM_PEAK_GBPS, 4)}
    if walk:
        # what k_traverse_spatial itself reads and writes per launch in its own layout: 32 B ray in, 20 B hit record out,
        # 48 B of a 64-byte record per step, 88 B of a leaf record per leaf tested
        launches = depth
        own_bytes = (local_rays * launches * 52 + walk[0] * 48 + walk[2] * 88) / launches
        out["roofline"]["own_layout"] = {
            "steps_per_ray": round(walk[0] / (local_rays * launches), 2), "empty_steps_per_ray": round(walk[1] / (local_rays * launches), 2),
            "leaves_per_ray": round(walk[2] / (local_rays * launches), 2), "bytes_per_launch": int(own_bytes),
            "lanes_per_walking_turn": round(walk[0] / max(walk[3] & ((1 << 32) - 1), 1), 1),
            "lanes_per_triangle_round": round(walk[2] / max(walk[3] >> 32, 1), 1),
            "requested_gbps": round(own_bytes / avg_trace_s / 1e9, 2),
            "note": "bytes the kernel requests in its own data layout (mostly L1/L2 hits: requested, not fetched - compare "
                    "with `traffic`); not a roofline fraction; triangles tested from the list of outliers are not counted"}

    if full and rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (scalar OpenMP restatement of the reference shaders) on a bounded sample -
        # the same scene and camera, the image reduced per axis, one frame including its build
        from oracle import oracle_py as oracle
        div = max(args.cpu_sample_div, 1)
        cw, ch = max(width // div, 16), max(height // div, 9)
        tb = time.perf_counter()
        m = oracle.sort_primitives(scene)
        oracle.build_bvh(scene, m)
        cpu_build_ms = (time.perf_counter() - tb) * 1e3
        # the sort exactly as the shaders are written (count, 256-entry scan, and CS_SortSort's scatter in which thread d
        # walks over all N records, shader/CS_SortSort.hlsl:18-35) on a bounded share of the triangles: O(256 N) per pass
        n_aw = min(ntri, 200000)
        sub = dict(scene)
        sub["indices"] = scene["indices"][:3 * n_aw]
        ta = time.perf_counter()
        oracle.sort_primitives(sub, as_written=True)
        sort_as_written_ms = (time.perf_counter() - ta) * 1e3
        ta = time.perf_counter()
        oracle.sort_primitives(sub)
        sort_stable_ms = (time.perf_counter() - ta) * 1e3
        op = oracle.Pipeline(scene, cw, ch, rpp=rpp, depth=depth, seed_raygen=1, seed_trace=2)
        tt = time.perf_counter()
        op.render(depth)
        cpu_s = time.perf_counter() - tt - cpu_build_ms * 1e-3      # render() includes one build
        _, tot = op.stats()
        op.close()
        out["cpu_baseline"] = {"value": round(cw * ch * rpp * depth / max(cpu_s, 1e-9) / 1e6, 4), "unit": "Mrays/s",
                               "cores": oracle.load().orc_num_threads(), "kind": "port",
                               "sample": "one frame of the same scene and camera at %dx%d (1/%d of the rays), %d bounces; "
                                         "LBVH build of the full %d triangles timed separately" % (cw, ch, div * div, depth, ntri),
                               "lbvh_build_ms": round(cpu_build_ms, 2),
                               "lbvh_build_variant": "Morton codes + four std::stable_sort passes by digit (what the reference's sort "
                                                     "computes) + leaves + levels, %d triangles" % ntri,
                               "sort_as_written_ms": round(sort_as_written_ms, 2), "sort_stable_sort_ms": round(sort_stable_ms, 2),
                               "sort_sample": "Morton codes + 4 passes on the first %d triangles: as written (CS_SortCount, "
                                              "CS_SortPrefixSum, CS_SortSort with one thread per digit over all records) beside the "
                                              "std::stable_sort variant" % n_aw,
                               "box_tests_per_ray": round(tot["nodes_visited"] / max(tot["rays"], 1), 2)}
    pipe.close()
    dev.close()
    del band
    torch.cuda.empty_cache()
    return out


def brief(rec):
    """What the headline line carries of another workload's record."""
    r = rec["roofline"]
    own = r.get("own_layout") or {}
    b = {"workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"], "steps": rec["steps"], "warmup": rec["warmup"],
         "ms_per_step": rec["ms_per_step"], "rays_per_step": rec["config"]["rays_per_step"], "triangles": rec["config"]["triangles"],
         "lbvh_build_ms": rec["lbvh_build_ms"], "traversal_kernel": r["kernel"], "avg_launch_ms": r["avg_launch_ms"],
         "trace_ms_by_bounce": rec["trace_ms_by_bounce"], "steps_per_ray": own.get("steps_per_ray"), "leaves_per_ray": own.get("leaves_per_ray"),
         "roofline_frac": r["frac"], "vector_frac": (r.get("vector") or {}).get("frac"),
         "l2_hit_rate": (r.get("memory") or {}).get("l2_hit_rate"), "traffic_source": r.get("traffic_source"),
         "counters_withheld": r.get("counters_withheld"), "build_roofline_frac": rec["build_roofline"]["frac"], "image_mean": rec["image_mean"]}
    for k in ("outliers", "rays_hit_by_bounce", "live_rays_by_bounce"):
        if k in rec:
            b[k] = rec[k]
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--triangles", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true",
                    help="skip the frame that counts the reference's box and triangle tests (minutes at 50 M triangles); "
                         "the roofline object then carries no algorithmic bytes")
    ap.add_argument("--no-reference-walk", action="store_true",
                    help="skip the two extra frames traced by the walk in the reference's order (reported beside `value`)")
    ap.add_argument("--no-others", action="store_true",
                    help="the headline workload only: without it a plain one-GPU run of c3 also measures the reference's assets (C2, with "
                         "and without hierarchy), C4, C5 and the open scene (with and without dropping missed rays) and reports them "
                         "under other_workloads")
    ap.add_argument("--no-bvh", action="store_true", help="RT_USE_BVH 0: every ray tests every triangle (small scenes only)")
    ap.add_argument("--drop-missed-rays", action="store_true", help="rt_device_set_drop_missed_rays for the timed frames")
    ap.add_argument("--dynamic", action="store_true", help="move the triangles on the device before every rebuild (always on for c5)")
    ap.add_argument("--refit", action="store_true",
                    help="moving scenes: build once, then refit (rt_accel_refit) instead of the per-frame rebuild the workloads are defined with")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: rays per pixel grow with the GPU count; strong: the frame is fixed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo lets several ranks rehearse on one GPU")
    ap.add_argument("--band", default="", help="R/N: single process doing the share of rank R of N (load-balance probe)")
    ap.add_argument("--cpu-sample-div", type=int, default=4, help="CPU baseline traces the image reduced by this factor per axis")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed launcher (one rank per GPU)" % args.gpus)
        args.gpus = world
    if args.backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    ctx = {"rank": rank, "world": world, "local_rank": local_rank, "backend": args.backend}

    out = measure(args, ctx, args.workload, args.steps, args.warmup, True, use_bvh=not args.no_bvh, drop_missed=args.drop_missed_rays)

    plain = (args.workload == "c3" and world == 1 and not (args.triangles or args.width or args.height or args.band or args.dynamic or
                                                            args.refit or args.no_bvh or args.drop_missed_rays or args.no_others))
    if plain:
        # the other configurations of the baseline, each a short run of its own (their hardware-counter fractions come
        # from their own committed counter files); the open scene twice, the assets also without a hierarchy
        others = {}
        for wl, steps, warmup in OTHER_WORKLOADS:
            t0 = time.perf_counter()
            rec = brief(measure(args, ctx, wl, steps, warmup, False))
            if wl.startswith("c2_"):
                nb = measure(args, ctx, wl, 2, 1, False, use_bvh=False)
                rec["without_hierarchy"] = {"value": nb["value"], "unit": "Mrays/s", "steps": nb["steps"], "ms_per_step": nb["ms_per_step"],
                                            "rays_per_step": nb["config"]["rays_per_step"], "traversal_kernel": "k_trace",
                                            "image_mean": nb["image_mean"]}
                rec["hierarchy_speedup"] = round(rec["value"] / max(nb["value"], 1e-9), 1)
            if wl == "open":
                dm = brief(measure(args, ctx, wl, steps, warmup, False, drop_missed=True))
                rec["drop_missed_rays"] = {k: dm[k] for k in ("value", "unit", "steps", "ms_per_step", "rays_per_step", "avg_launch_ms",
                                                              "trace_ms_by_bounce", "image_mean")}
                rec["drop_missed_rays"]["note"] = ("rays counted as the reference counts them (every ray, every bounce); the same image, "
                                                   "rays that have left the scene are not walked again")
                rec["drop_missed_rays_speedup"] = round(dm["value"] / max(rec["value"], 1e-9), 3)
            rec["measured_in_s"] = round(time.perf_counter() - t0, 1)
            others[wl] = rec
        out["other_workloads"] = others

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
