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
trace kernel; algorithmic bytes from exact traversal counters) and the CPU baseline (the scalar
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
# Morton 56 + sort 80 + leaves 68 + upper levels 48 (reference data layout) for the reference's hierarchy, and the same
# again for the library's own hierarchy (its keys are made from the sorted pairs and the packed triangles: 56 as well)
BUILD_BYTES_PER_TRIANGLE = 2 * 252

# HBM-side bytes of ONE k_traverse_quad launch of the default workload on one GPU, from the hardware-counter passes
# committed as profiles/r1_c3_final_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in runs of their
# own, kilobytes -> bytes): the kernel's reads past L2 are gathers of 64-byte requests, for which FETCH_SIZE counts
# the bytes exactly (scripts/calib_fetch.hip; the doubling the microarchitecture guide prescribes applies to wide
# streaming reads, which this kernel does not issue). Counters cannot be read inside a timed run, so the figure is
# carried here and reported only for the configuration it was measured on.
MEASURED_TRAFFIC = {("c3", 1, "spatial"): {"fetch": 7374910.8e3, "write": 463644.2e3, "source": "profiles/r1_c3_spatial_pmc_summary.txt"},
                    ("c3", 1, "quad"): {"fetch": 147772356.0e3, "write": 489338.2e3, "source": "profiles/r1_c3_final_pmc_summary.txt"}}

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
                        row_stride=row_stride, dynamic=args.dynamic or args.workload == "c5", rebuild_every_cycle=True, output_ptr=band.data_ptr())

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
            pipe.render(depth)                       # one frame: build, camera rays, `depth` bounces
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

    tkey = (args.workload, world)
    plain = not (args.triangles or args.width or args.height or args.band or args.dynamic) and args.scaling == "weak"
    traffic = MEASURED_TRAFFIC.get(tkey + (kernel_name,)) if plain else None
    kernel_symbol = {"spatial": "k_traverse_spatial", "quad": "k_traverse_quad", "wave": "k_traverse", "simple": "k_trace"}.get(kernel_name, kernel_name)
    out = {
        "metric": "Mrays/sec", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (desc + (" - weak scaling: %d rays per pixel over %d GPUs" % (rpp, share_world)
                                        if args.scaling == "weak" and share_world > 1 else ""))
                   if not (args.triangles or args.width or args.height) else
                   "custom: synthetic %d-triangle scene, %dx%d, %d ray/pixel, %d bounces, LBVH rebuilt every frame" % (ntri, width, height, rpp, depth),
                   "triangles": ntri, "width": width, "height": height, "rays_per_pixel": rpp, "bounces": depth,
                   "parallelism": "image rows dealt to %d GPU(s) in turn, replicated scene + per-GPU rebuild, gather to rank 0" % world},
        "lbvh_build_ms": round(avg_build_ms, 4),
        "roofline": {"kernel": kernel_symbol, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": int(traffic["fetch"] + traffic["write"]) if traffic else None,
                     "traffic_source": traffic["source"] + " (FETCH_SIZE + WRITE_SIZE per launch)" if traffic else None,
                     "note": "algorithmic bytes are those of the reference's walk in the reference's data layout (its box and "
                             "triangle tests per ray, counted exactly, x 32 B and 48 B). k_traverse_quad performs those tests, "
                             "mostly from L1/L2; k_traverse_spatial returns the same closest hits from a hierarchy of its own "
                             "with far fewer tests, so achieved exceeds the HBM peak - traffic is what crossed the memory side",
                     "avg_launch_ms": round(avg_trace_s * 1e3, 4), "launches_timed": int(kernel_launches.value),
                     "avg_trace_call_ms": round(avg_trace_call_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                     "rays_per_launch": local_rays, "box_tests_per_ray": round(nodes / max(counted, 1), 2),
                     "triangle_tests_per_ray": round(tris / max(counted, 1), 3), "hit_fraction": round(hits / max(counted, 1), 4)},
        "build_roofline": {"kernels": "both hierarchies: (k_morton_packed | k_spatial_keys) + 4 x (k_sort_hist, k_sort_scan, k_sort_scatter) + k_leaves + k_upsweep, each",
                           "bound": "hbm", "achieved": round(BUILD_BYTES_PER_TRIANGLE * ntri / (avg_build_ms * 1e-3) / 1e9, 2),
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(BUILD_BYTES_PER_TRIANGLE * ntri / (avg_build_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                           "algorithmic_bytes_per_triangle": BUILD_BYTES_PER_TRIANGLE},
        "trace_ms_by_bounce": [round(float(x), 3) for x in np.asarray(trace_ms).reshape(-1, depth).mean(axis=0)],
        "image_mean": round(float(img[..., :3].mean()), 6),
    }
    if ref_walk:
        out["reference_order_walk"] = ref_walk
        # lbvh_build_ms is what a frame pays: the reference's hierarchy and the library's own; this is the former alone
        out["reference_lbvh_build_ms"] = ref_walk["lbvh_build_ms"]
    if walk:
        # what k_traverse_spatial itself reads and writes per launch in its own layout: 32 B ray + 8 B order entry in,
        # 20 B hit record out, 48 B of a 64-byte record per step, 88 B of a leaf record per leaf tested
        launches = depth
        own_bytes = (local_rays * launches * 60 + walk[0] * 48 + walk[2] * 88) / launches
        out["roofline"]["own_layout"] = {
            "steps_per_ray": round(walk[0] / (local_rays * launches), 2), "empty_steps_per_ray": round(walk[1] / (local_rays * launches), 2),
            "leaves_per_ray": round(walk[2] / (local_rays * launches), 2), "bytes_per_launch": int(own_bytes),
            "achieved": round(own_bytes / avg_trace_s / 1e9, 2), "unit": "GB/s", "frac": round(own_bytes / avg_trace_s / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "bytes the kernel requests in its own data layout (mostly L1/L2 hits); the kernel is bound by vector "
                    "instructions and divergence, see DESIGN.md section 5"}

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
                               "box_tests_per_ray": round(tot["nodes_visited"] / max(tot["rays"], 1), 2)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    pipe.close()
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
