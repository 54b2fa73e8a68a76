#!/usr/bin/env python3
"""Experiment: S image bands of one frame in flight on S streams of one GPU (tails of one band's launch
overlap with another band's work). Usage: python scripts/bands_experiment.py S [S ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpuraytracer_amd import host_py, scenes
from gpuraytracer_amd.distributed import band_for_rank
from gpuraytracer_amd.pipeline import Pipeline
from gpuraytracer_amd.runtime import Device

ntri, width, height, rpp, depth = 1000000, 3840, 2160, 1, 4
scene = host_py.synthetic_scene(ntri)
cam = scene["camera"]
iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], width / height, cam["near_z"], cam["far_z"])
for S in [int(a) for a in sys.argv[1:]]:
    frames = 4
    rs = scenes.mt19937_raw(1, 3 * frames).reshape(-1, 3)
    ts = scenes.mt19937_raw(2, 3 * frames * depth).reshape(-1, 3)
    devs, pipes = [], []
    for b in range(S):
        first_row, num_rows = band_for_rank(height, S, b)
        dev = Device(0)
        state = {"f": 0, "t": 0}
        def rseed(state=state):
            v = rs[state["f"]]; state["f"] += 1; return v
        def tseed(state=state):
            v = ts[state["t"]]; state["t"] += 1; return v
        pipes.append(Pipeline(dev, scene, width, height, iv, ip, rpp=rpp, depth=depth, use_bvh=True, use_accel=True,
                              raygen_seeds=rseed, trace_seeds=tseed, first_row=first_row, num_rows=num_rows, rebuild_every_cycle=True))
        devs.append(dev)
    def frame():
        for bounce in range(depth):
            for p in pipes:
                p.render(1)
    frame()
    for d in devs: d.sync()
    t0 = time.perf_counter()
    for _ in range(frames - 1):
        frame()
    for d in devs: d.sync()
    dt = (time.perf_counter() - t0) / (frames - 1)
    img = np.concatenate([p.output_image() for p in pipes], axis=0)
    print("bands", S, "ms/frame", round(dt * 1e3, 2), "Mrays/s", round(width * height * rpp * depth / dt / 1e6, 2), "image mean", float(img[..., :3].mean()), flush=True)
    for p in pipes: p.close()
    for d in devs: d.close()
