"""Makes tests/golden/ref_nobvh_lowfreq.npz and ref_nobvh_detail.npz from the only output image the reference holds, results/NoBVH.png
(testscene2 without a hierarchy, camera {0,0,0} -> {1,0,0} of src/Settings.h:22-25, a 1922 x 1081 screenshot of the
1920 x 1080 window): the picture decoded by this repository's own PNG decoder, sRGB -> linear, cropped to 1072 x 1920 and
averaged over 16 x 16 blocks; and, for the pixel-level checks (ref_nobvh_detail.npz): profiles of its luminance - column means of
eight horizontal bands, row means of 64 vertical bands, from which the place of the picture is found to the pixel - and the
mean squared Laplacian of the luminance per 16 x 16 block, from which the noise level, hence the number of samples, is
estimated. Runs in the build container only (needs /root/reference); the fixtures are data, not source.

    python scripts/make_nobvh_fixture.py [/root/reference/results/NoBVH.png]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpuraytracer_amd import host_py          # noqa: E402

BLOCK, ROWS, COLS = 16, 1072, 1920


def srgb_to_linear(u8):
    x = u8.astype(np.float64) / 255.0
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def block_means(linear_rgb):
    """(ROWS, COLS, 3) -> (ROWS / 16, COLS / 16, 3) means"""
    h, w = linear_rgb.shape[:2]
    return linear_rgb.reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, 3).mean(axis=(1, 3))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/results/NoBVH.png"
    rgba16 = host_py.load_texture(path)                       # the repository's own decoder: 8-bit samples widened by * 257
    u8 = (rgba16[..., :3] >> 8).astype(np.uint8)
    lin = srgb_to_linear(u8[:ROWS, :COLS])
    means = block_means(lin).astype(np.float32)
    out = os.path.join(ROOT, "tests", "golden", "ref_nobvh_lowfreq.npz")
    np.savez_compressed(out, rgb=means, source="results/NoBVH.png of the reference, %dx%d, rows 0..%d cols 0..%d, 16x16 block means of linear RGB"
                        % (u8.shape[1], u8.shape[0], ROWS - 1, COLS - 1))
    print("wrote", out, means.shape, "mean linear RGB", means.mean(axis=(0, 1)))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import compare_nobvh_png as cmp
    lum = cmp.luminance(lin)
    cols, rows = cmp.band_profiles(lum)
    out = os.path.join(ROOT, "tests", "golden", "ref_nobvh_detail.npz")
    np.savez_compressed(out, column_profiles=cols.astype(np.float32), row_profiles=rows.astype(np.float32),
                        laplacian_energy=cmp.laplacian_energy(lum).astype(np.float32),
                        source="results/NoBVH.png of the reference, rows 0..%d cols 0..%d of the screenshot: luminance of linear RGB; column "
                               "means of 8 horizontal bands, row means of 64 vertical bands, mean squared 4-neighbour Laplacian per 16x16 block"
                               % (ROWS - 1, COLS - 1))
    print("wrote", out, cols.shape, rows.shape)


if __name__ == "__main__":
    main()
