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


def linear_rgb(tone_mapped):
    """One of our images as the screenshot shows the reference's: 8-bit sRGB samples, decoded to linear RGB."""
    from gpuraytracer_amd import host_py
    lib = host_py.load()
    table = np.array([lib.rth_encode_srgb8(float(v)) for v in np.linspace(0.0, 1.0, 4097)], dtype=np.float64)
    x = np.clip(np.nan_to_num(tone_mapped[:ROWS, :COLS, :3].astype(np.float64)), 0.0, 1.0)
    s = np.floor(np.interp(x, np.linspace(0.0, 1.0, 4097), table) + 0.5) / 255.0
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)


def luminance(rgb):
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


H_BANDS, V_BANDS = 8, 64


def band_profiles(lum):
    """Column means of H_BANDS horizontal bands [H_BANDS][COLS] and row means of V_BANDS vertical bands [V_BANDS][ROWS] of a
    luminance image [ROWS][COLS]: the picture's edges, with the pixel noise averaged away along them."""
    lum = lum[:ROWS, :COLS]
    return lum.reshape(H_BANDS, ROWS // H_BANDS, COLS).mean(axis=1), lum.reshape(ROWS, V_BANDS, COLS // V_BANDS).mean(axis=2).T


def shift_correlation(theirs, ours, max_shift=4, smooth=1):
    """Correlation of the first differences of two sets of profiles at shifts -max_shift .. max_shift (entry i: shift
    i - max_shift; positive = their picture lies that many pixels further along than ours); smooth: box filter first."""
    def prep(p):
        if smooth > 1:
            k = np.ones(smooth) / smooth
            p = np.array([np.convolve(q, k, mode="valid") for q in p])
        return np.diff(p, axis=1)
    a, b = prep(np.asarray(theirs, dtype=np.float64)), prep(np.asarray(ours, dtype=np.float64))
    out = []
    for s in range(-max_shift, max_shift + 1):
        x, y = (a[:, s:], b[:, :b.shape[1] - s]) if s >= 0 else (a[:, :s], b[:, -s:])
        x, y = x - x.mean(axis=1, keepdims=True), y - y.mean(axis=1, keepdims=True)
        out.append(float((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum())))
    return np.array(out)


def laplacian_energy(lum):
    """Mean squared 4-neighbour Laplacian of the luminance per BLOCK x BLOCK block: what is left of a picture when its
    structure is taken out is its pixel noise, which falls with the number of samples."""
    lum = lum[:ROWS, :COLS]
    lap = np.zeros_like(lum)
    lap[1:-1, 1:-1] = lum[1:-1, 1:-1] - 0.25 * (lum[:-2, 1:-1] + lum[2:, 1:-1] + lum[1:-1, :-2] + lum[1:-1, 2:])
    return (lap.reshape(ROWS // BLOCK, BLOCK, COLS // BLOCK, BLOCK) ** 2).mean(axis=(1, 3))


def relative_noise(energy, block_mean_luminance):
    """Median over the blocks of (Laplacian energy / squared mean luminance): one number for a picture's noise level."""
    return float(np.median(energy / (block_mean_luminance * block_mean_luminance + 1e-6)))


def compare(ours_lowfreq, ref_lowfreq):
    lum = lambda a: 0.2126 * a[..., 0] + 0.7152 * a[..., 1] + 0.0722 * a[..., 2]
    a, b = lum(ours_lowfreq).reshape(-1), lum(ref_lowfreq.astype(np.float64)).reshape(-1)
    corr = float(np.corrcoef(a, b)[0, 1])
    log_corr = float(np.corrcoef(np.log(a + 1e-4), np.log(b + 1e-4))[0, 1])
    ratio = ours_lowfreq.mean(axis=(0, 1)) / ref_lowfreq.astype(np.float64).mean(axis=(0, 1))
    return corr, log_corr, ratio


def render(device, cycles, use_bvh, rpp=1):
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
    p = Pipeline(device, sc, w, h, iv, ip, rpp=rpp, depth=depth, use_bvh=use_bvh, raygen_seeds=lambda: next(rs), trace_seeds=lambda: next(ts))
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
