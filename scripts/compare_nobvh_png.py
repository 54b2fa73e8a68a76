"""Qualitative pin against the only output the reference holds: renders testscene2 at 1920 x 1080 with the camera of the
reference's screenshot (src/Settings.h:22-25), without and with the hierarchy, encodes it as the reference presents it
(tone-mapped, 8-bit sRGB) and compares its low-frequency structure with tests/golden/ref_nobvh_lowfreq.npz (made from
results/NoBVH.png by scripts/make_nobvh_fixture.py). Seeds and the number of accumulated samples of the screenshot are
unknown, so the comparison is of 16 x 16 block means: correlation of luminance, per-channel mean ratios.

    python scripts/compare_nobvh_png.py [cycles] [out_dir]      (GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BLOCK, ROWS, COLS = 16, 1072, 1920


def low_frequency(tone_mapped):
    """What the fixture holds, for one of our images: sRGB-encode to 8 bits as the writers do, decode, block means."""
    from gpuraytracer_amd import host_py
    lib = host_py.load()
    table = np.array([lib.rth_encode_srgb8(float(v)) for v in np.linspace(0.0, 1.0, 4097)], dtype=np.float64)
    x = np.clip(np.nan_to_num(tone_mapped[:ROWS, :COLS, :3].astype(np.float64)), 0.0, 1.0)
    u8 = np.floor(np.interp(x, np.linspace(0.0, 1.0, 4097), table) + 0.5)      # the 8-bit sample (interpolated table of the encoder)
    s = u8 / 255.0
    lin = np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)
    return lin.reshape(ROWS // BLOCK, BLOCK, COLS // BLOCK, BLOCK, 3).mean(axis=(1, 3))


def compare(ours_lowfreq, ref_lowfreq):
    lum = lambda a: 0.2126 * a[..., 0] + 0.7152 * a[..., 1] + 0.0722 * a[..., 2]
    a, b = lum(ours_lowfreq).reshape(-1), lum(ref_lowfreq.astype(np.float64)).reshape(-1)
    corr = float(np.corrcoef(a, b)[0, 1])
    log_corr = float(np.corrcoef(np.log(a + 1e-4), np.log(b + 1e-4))[0, 1])
    ratio = ours_lowfreq.mean(axis=(0, 1)) / ref_lowfreq.astype(np.float64).mean(axis=(0, 1))
    return corr, log_corr, ratio


def render(device, cycles, use_bvh):
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.pipeline import Pipeline
    from gpuraytracer_amd import scenes
    from tests.test_oracle_kat import _load_golden_scene
    sc = _load_golden_scene("testscene2")[0]
    w, h, depth = 1920, 1080, 3
    cam = dict(sc["camera"])
    cam.update(position=(0.0, 0.0, 0.0), focus=(1.0, 0.0, 0.0), up=(0.0, 1.0, 0.0))
    iv, ip = host_py.camera_matrices(cam["position"], cam["focus"], cam["up"], cam["fov"], w / h, cam["near_z"], cam["far_z"])
    rs = iter(scenes.mt19937_raw(1, 3 * cycles + 3).reshape(-1, 3))
    ts = iter(scenes.mt19937_raw(2, 3 * cycles * depth + 3).reshape(-1, 3))
    p = Pipeline(device, sc, w, h, iv, ip, depth=depth, use_bvh=use_bvh, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
    p.render(cycles * depth)
    img = p.output_image()
    p.close()
    return img


def main():
    from gpuraytracer_amd import host_py
    from gpuraytracer_amd.runtime import Device
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_nobvh_lowfreq.npz"))["rgb"]
    dev = Device(0)
    for use_bvh in (False, True):
        img = render(dev, cycles, use_bvh)
        name = "testscene2_%s_%dcycles.png" % ("bvh" if use_bvh else "nobvh", cycles)
        os.makedirs(out_dir, exist_ok=True)
        host_py.write_image(os.path.join(out_dir, name), img)
        corr, log_corr, ratio = compare(low_frequency(img), ref)
        print("%s: luminance correlation of 16x16 block means %.4f (log %.4f), mean ratio ours/reference R %.3f G %.3f B %.3f"
              % (name, corr, log_corr, ratio[0], ratio[1], ratio[2]), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
