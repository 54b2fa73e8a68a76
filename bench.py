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
ROUND = "r2"
NUM_SIMDS = 1024                # 256 compute units x 4
NUM_XCDS = 8


def load_pmc(workload, kernel):
    """Per-dispatch hardware counters of `kernel` from the committed counter passes of this workload
    (profiles/<round>_<workload>_pmc.json, written by scripts/pmc_summary.py --json from scripts/pmc_passes.sh runs:
    one rocprofv3 --pmc run per counter group). None when there is no such file or kernel."""
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (ROUND, workload))
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        table = json.load(f)
    for name in (kernel + "<false>", kernel):
        if name in table["kernels"]:
            return table["kernels"][name], os.path.relpath(path, ROOT)
    return None, None


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
}


def trace_bytes(rays, hits, nodes, tris):
    """Algorithmic bytes of trace launches in the reference's data layout: 100 B of per-ray state read,
    96 B written on a hit (ray, previous ray, two light values) or 32 B on a miss, 32 B per box test,
    12 + 36 B per triangle test, and on a hit three 48-B vertices, a 64-B material and five texture
    lookups of 16 + 8 B."""
    return rays * 100 + hits * (96 + 3 * 48 + 64 + 5 * 24) + (rays - hits) * 32 + nodes * 32 + tris * 48


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

    from gpuraytracer_amd import host_py, scenes
    from gpuraytracer_amd.distributed import gather_rows, rows_for_rank
    from gpuraytracer_amd.pipeline import Pipeline
    from gpuraytracer_amd.runtime import Device

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

    ntri, width, height, rpp, depth, desc = WORKLOADS[args.workload]
    ntri = args.triangles or ntri
    width = args.width or width
    height = args.height or height

    # scene, camera and host seeds come from the product's own host code (C++ generator and camera maths)
    scene = host_py.synthetic_scene(ntri)
    cam = scene["camera"]
    iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], width / height, cam["near_z"],
                                     cam["far_z"])
    total_frames = args.steps + args.warmup + 7      # + the counting frame and the frames of the walk in the reference's order
    rs = iter(scenes.mt19937_raw(1, 3 * total_frames).reshape(-1, 3))
    ts = iter(scenes.mt19937_raw(2, 3 * total_frames * depth).reshape(-1, 3))

    share_rank, share_world = rank, world
    if args.band and world == 1:
        share_rank, share_world = (int(x) for x in args.band.split("/"))
    if args.scaling == "weak":
        rpp *= share_world
    first_row, num_rows, row_stride = rows_for_rank(height, share_world, share_rank)
    stream = torch.cuda.Stream()
    dev = Device(local_rank)
    with torch.cuda.stream(stream):
        dev.set_stream(stream.cuda_stream)
        band = torch.zeros((num_rows, width, 4), dtype=torch.float32, device="cuda")
        pipe = Pipeline(dev, scene, width, height, iv, ip, rpp=rpp, depth=depth, use_bvh=True, use_accel=True,
                        raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts), first_row=first_row, num_rows=num_rows,
                        row_stride=row_stride, dynamic=args.dynamic or args.workload == "c5", rebuild_every_cycle=True, output_ptr=band.data_ptr(),
                        refit=args.refit)

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

        for _ in range(args.warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timing["on"] = True
        dev.lib.rt_device_set_traverse_timing(dev.handle, 1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing["on"] = False

        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        import ctypes
        kernel_ms, kernel_launches = ctypes.c_double(0.0), ctypes.c_uint32(0)
        dev.lib.rt_device_traverse_timing(dev.handle, ctypes.byref(kernel_ms), ctypes.byref(kernel_launches))
        dev.lib.rt_device_set_traverse_timing(dev.handle, 0)
        build_ms = [a.elapsed_ms(b) for a, b in build_events]
        trace_ms = [a.elapsed_ms(b) for a, b in trace_events]

        # the same frames by the walk in the reference's order (untimed for `value`; reported beside it)
        ref_walk = None
        kernel_name = os.environ.get("GPURT_TRACE_KERNEL", "") or "spatial"
        if kernel_name == "spatial" and world == 1 and not args.no_reference_walk:
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

        # exact traversal counters of one more frame (untimed, instrumented kernel variant)
        nodes = tris = hits = counted = 0
        if not args.no_counters:
            stats_buf = dev.alloc(32)
            dev.lib.rt_device_set_trace_stats(dev.handle, stats_buf.handle)
            pipe.render(depth)
            dev.sync()
            dev.lib.rt_device_set_trace_stats(dev.handle, None)
            nodes, tris, hits, counted = [int(x) for x in stats_buf.download(np.uint64, 4)]
        img = band.float().cpu().numpy()

    local_rays = num_rows * width * rpp
    rays_per_step = width * height * rpp * depth
    mrays = rays_per_step * args.steps / elapsed / 1e6
    launches_per_frame = depth
    bytes_per_launch = trace_bytes(counted, hits, nodes, tris) / launches_per_frame
    # the dominant kernel alone (events recorded by the library around k_traverse on its stream); a trace call
    # also runs the ray-key sort before it and the shading kernel after it
    avg_trace_s = kernel_ms.value / max(kernel_launches.value, 1) * 1e-3
    avg_trace_call_ms = float(np.mean(trace_ms))
    achieved = bytes_per_launch / avg_trace_s / 1e9
    avg_build_ms = float(np.mean(build_ms))

    plain = not (args.triangles or args.width or args.height or args.band or args.dynamic or args.refit) and args.scaling == "weak" and world == 1
    kernel_symbol = {"spatial": "k_traverse_spatial", "quad": "k_traverse_quad", "wave": "k_traverse", "simple": "k_trace"}.get(kernel_name, kernel_name)
    counters, counter_file = load_pmc(args.workload, kernel_symbol) if plain else (None, None)
    fr = counter_fractions(counters, avg_trace_s) if counters else {}
    mem, vec = fr.get("memory"), fr.get("vector")
    # the roofline of the dominant kernel, the traversal launch: HBM is the contract's bound for this path (no dense
    # contraction anywhere); frac is what crossed the memory side over the launch's duration, against the HBM peak.
    # The unit that binds the kernel today is named in `binding`.
    roofline = {"kernel": kernel_symbol, "bound": "hbm",
                "achieved": mem["achieved"] if mem else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": mem["frac"] if mem else None,
                "traffic": fr.get("traffic"),
                "traffic_source": (counter_file + ": FETCH_SIZE + WRITE_SIZE per launch x 1024 (separate --pmc passes)") if mem else None,
                "memory": mem, "vector": vec,
                "binding": (("vector units (vector.frac %.2f against memory.frac %.2f)" % (vec["frac"], mem["frac"]))
                            if vec["frac"] >= mem["frac"] else ("memory side (memory.frac %.2f against vector.frac %.2f)" % (mem["frac"], vec["frac"])))
                if mem and vec else None,
                "avg_launch_ms": round(avg_trace_s * 1e3, 4), "launches_timed": int(kernel_launches.value),
                "avg_trace_call_ms": round(avg_trace_call_ms, 4), "rays_per_launch": local_rays,
                "note": "achieved = bytes across the L2's memory side per launch (hardware counters of the committed passes of this "
                        "workload) / the launch duration measured live in this run (HIP events on the library's stream around the "
                        "traversal kernel). algorithmic_* are the reference walk's own counts in the reference's data layout "
                        "(100 B per ray + 32 B per box test + 48 B per triangle test + hit traffic), kept for "
                        "reference; the default kernel finds the same closest hits with far fewer tests, so they are not a fraction "
                        "of anything it does"}
    if counted:
        roofline.update({"algorithmic_bytes_per_launch": int(bytes_per_launch), "algorithmic_gbps_if_performed": round(achieved, 1),
                         "algorithmic_box_tests_per_ray": round(nodes / max(counted, 1), 2),
                         "algorithmic_triangle_tests_per_ray": round(tris / max(counted, 1), 3), "hit_fraction": round(hits / max(counted, 1), 4)})
    build_bytes = BUILD_BYTES_PER_TRIANGLE * ntri
    out = {
        "metric": "Mrays/sec", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (desc + (" - weak scaling: %d rays per pixel over %d GPUs" % (rpp, share_world)
                                        if args.scaling == "weak" and share_world > 1 else ""))
                   if not (args.triangles or args.width or args.height) else
                   "custom: synthetic %d-triangle scene, %dx%d, %d ray/pixel, %d bounces, LBVH rebuilt every frame" % (ntri, width, height, rpp, depth),
                   "triangles": ntri, "width": width, "height": height, "rays_per_pixel": rpp, "bounces": depth,
                   "final_image": "once per frame (after the last bounce)",
                   "build": "refit after the first frame (not the workload as defined)" if args.refit else "rebuilt every frame",
                   "parallelism": "image rows dealt to %d GPU(s) in turn, replicated scene + per-GPU rebuild, gather to rank 0" % world},
        "lbvh_build_ms": round(avg_build_ms, 4),
        "roofline": roofline,
        # 252 B per triangle (reference layout) over the time a frame pays for its build - both hierarchies, and for a
        # moving scene also the animation and the bounds - and, below, over the reference's hierarchy alone
        "build_roofline": {"kernels": "per frame: k_morton_hist, radix passes, k_leaves, k_upsweep for the reference's hierarchy; "
                                      "keys, radix passes, cell layout, k_leaves, k_upsweep for the library's own",
                           "bound": "hbm", "achieved": round(build_bytes / (avg_build_ms * 1e-3) / 1e9, 1),
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(build_bytes / (avg_build_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                           "algorithmic_bytes_per_triangle": BUILD_BYTES_PER_TRIANGLE, "build_ms_per_frame": round(avg_build_ms, 4)},
        "trace_ms_by_bounce": [round(float(x), 3) for x in np.asarray(trace_ms).reshape(-1, depth).mean(axis=0)],
        "image_mean": round(float(img[..., :3].mean()), 6),
    }
    if ref_walk:
        out["reference_order_walk"] = ref_walk
        # lbvh_build_ms is what a frame pays: the reference's hierarchy and the library's own; this is the former alone
        out["reference_lbvh_build_ms"] = ref_walk["lbvh_build_ms"]
        rb = ref_walk["lbvh_build_ms"] * 1e-3
        out["build_roofline"]["reference_hierarchy_alone"] = {"build_ms": ref_walk["lbvh_build_ms"], "achieved": round(build_bytes / rb / 1e9, 1),
                                                              "frac": round(build_bytes / rb / 1e9 / HBM_PEAK_GBPS, 4)}
    if walk:
        # what k_traverse_spatial itself reads and writes per launch in its own layout: 32 B ray + 8 B order entry in,
        # 20 B hit record out, 48 B of a 64-byte record per step, 88 B of a leaf record per leaf tested
        launches = depth
        own_bytes = (local_rays * launches * 60 + walk[0] * 48 + walk[2] * 88) / launches
        out["roofline"]["own_layout"] = {
            "steps_per_ray": round(walk[0] / (local_rays * launches), 2), "empty_steps_per_ray": round(walk[1] / (local_rays * launches), 2),
            "leaves_per_ray": round(walk[2] / (local_rays * launches), 2), "bytes_per_launch": int(own_bytes),
            "lanes_per_walking_turn": round(walk[0] / max(walk[3] & ((1 << 32) - 1), 1), 1),
            "lanes_per_triangle_round": round(walk[2] / max(walk[3] >> 32, 1), 1),
            "requested_gbps": round(own_bytes / avg_trace_s / 1e9, 2),
            "note": "bytes the kernel requests in its own data layout (mostly L1/L2 hits: requested, not fetched - compare "
                    "with `traffic`); not a roofline fraction"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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

    if rank == 0:
        print(json.dumps(out), flush=True)
    pipe.close()
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
