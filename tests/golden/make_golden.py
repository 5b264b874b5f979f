#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/.  Runs ONLY in the build container
(needs /root/reference and oracle/_ref built by `make -C oracle ref`); the fixtures themselves are
plain data and travel to the GPU box.

  alley_1_gray.npz      frames 0001/0002 of the reference's images/alley_1 as 8-bit gray
                        (OpenCV BGR2GRAY fixed-point formula, what cv::imread(GRAYSCALE) feeds
                        kroeger/run_dense.cpp:208-209) + a 256x128 RGB crop of both frames
  alley_0001_flo.npz    the reference's only golden output, kroeger/flows/alley_0001.flo
  fdf_ref_gray.npz      inputs and OUTPUTS OF THE REFERENCE'S OWN FDF1.0.1 CODE (oracle/_ref) for the
  fdf_ref_rgb.npz       variational-refinement chain (kroeger/refine_variational.cpp:153-241):
                        every intermediate plane of the last inner iteration + the refined flow
  fdf_ref_l4_*.npz      the same chain on a 120 x 68 level (1080p level 4: images/road_HD.jpg and a shifted copy, padded to 1088 rows
                        and halved four times; five inner iterations) -- the size class of the engine's streaming solver
                        (65..96 rows)
  fdf_ref_depth_*.npz   the same inputs through the reference's stereo-depth chain (RefLevelDE, :243-330)
  natural_images.npz    the reference's natural test images as 8-bit gray (same formula): images/road_HD.jpg (1920x1080) and
                        images/yosemite_4k.jpg (3840x2160) -- the inputs SURVEY.md 8(d) names for C2-C4 (the second frame of
                        a pair is a shifted copy, made by the test)
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from oracle import fdf_ref as R         # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def rgb(path):
    return np.asarray(Image.open(path).convert("RGB"))


def gray_cv(a):
    a = a.astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def smooth_flow(h, w, seed, amp=1.5):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    u = amp * np.sin(xx / 7.0 + rng.uniform(0, 3)) * np.cos(yy / 5.0) + rng.uniform(-1, 1)
    v = amp * np.cos(xx / 9.0) * np.sin(yy / 6.0 + rng.uniform(0, 3)) + rng.uniform(-1, 1)
    return u.astype(np.float32), v.astype(np.float32)


def fdf_cases(noc):
    ref = R.FdfRef(noc)
    a0 = rgb(REF + "/images/alley_1/frame_0001.png")
    a1 = rgb(REF + "/images/alley_1/frame_0002.png")
    if noc == 1:
        f0, f1 = gray_cv(a0).astype(np.float32)[..., None], gray_cv(a1).astype(np.float32)[..., None]
    else:
        f0, f1 = a0[..., ::-1].astype(np.float32), a1[..., ::-1].astype(np.float32)   # BGR like cv::imread
    out = {}
    # (name, crop y0,x0,h,w at full res, levels down, lvl id used for inner-iteration count)
    cases = [("w64h28", 0, 0, 448 - 12, 1024, 4, 4), ("w30h17", 100, 200, 17 * 8, 30 * 8, 3, 2),
             ("w41h23", 50, 300, 23 * 4, 41 * 4, 2, 1)]
    for name, y0, x0, hh, ww, down, lvl in cases:
        c0 = f0[y0:y0 + hh, x0:x0 + ww]
        c1 = f1[y0:y0 + hh, x0:x0 + ww]
        if name == "w64h28":
            c0 = O.pad_frame(f0 if noc > 1 else f0[..., 0], 5)
            c1 = O.pad_frame(f1 if noc > 1 else f1[..., 0], 5)
            c0 = c0.reshape(c0.shape[0], c0.shape[1], noc)
            c1 = c1.reshape(c1.shape[0], c1.shape[1], noc)
        for _ in range(down):                       # 2x2 means (exact for 8-bit input)
            c0 = ((c0[0::2, 0::2] + c0[1::2, 0::2]) + (c0[0::2, 1::2] + c0[1::2, 1::2])) * np.float32(0.25)
            c1 = ((c1[0::2, 0::2] + c1[1::2, 0::2]) + (c1[0::2, 1::2] + c1[1::2, 1::2])) * np.float32(0.25)
        im1 = np.ascontiguousarray(c0.transpose(2, 0, 1))
        im2 = np.ascontiguousarray(c1.transpose(2, 0, 1))
        h, w = im1.shape[1:]
        wx, wy = smooth_flow(h, w, 7 + w)
        dump = {}
        ox, oy = ref.ref_level_of(im1, im2, wx, wy, lvl, dump=dump)
        out[name + "/im1"], out[name + "/im2"] = im1, im2
        out[name + "/wx"], out[name + "/wy"], out[name + "/lvl"] = wx, wy, np.int32(lvl)
        out[name + "/out_x"], out[name + "/out_y"] = ox, oy
        for k, v in dump.items():
            out[name + "/" + k] = v
    return out


def fdf_level4_case(noc):
    """1080p level 4 (120 x 68, lvl 4 -> five inner iterations) through the reference's FDF code"""
    ref = R.FdfRef(noc)
    a0 = rgb(REF + "/images/road_HD.jpg")
    a1 = np.roll(a0, (2, 5), (0, 1))
    if noc == 1:
        f0, f1 = gray_cv(a0).astype(np.float32), gray_cv(a1).astype(np.float32)
    else:
        f0, f1 = a0[..., ::-1].astype(np.float32), a1[..., ::-1].astype(np.float32)
    c0, c1 = O.pad_frame(f0, 6), O.pad_frame(f1, 6)
    c0 = c0.reshape(c0.shape[0], c0.shape[1], noc); c1 = c1.reshape(c1.shape[0], c1.shape[1], noc)
    for _ in range(4):
        c0 = ((c0[0::2, 0::2] + c0[1::2, 0::2]) + (c0[0::2, 1::2] + c0[1::2, 1::2])) * np.float32(0.25)
        c1 = ((c1[0::2, 0::2] + c1[1::2, 0::2]) + (c1[0::2, 1::2] + c1[1::2, 1::2])) * np.float32(0.25)
    im1 = np.ascontiguousarray(c0.transpose(2, 0, 1)); im2 = np.ascontiguousarray(c1.transpose(2, 0, 1))
    h, w = im1.shape[1:]
    assert (w, h) == (120, 68)
    wx, wy = smooth_flow(h, w, 7 + w, amp=0.6)
    wx += np.float32(5.0 / 16); wy += np.float32(2.0 / 16)             # around the true shift at this level
    dump = {}
    ox, oy = ref.ref_level_of(im1, im2, wx, wy, 4, dump=dump)
    name = "w120h68"
    out = {name + "/im1": im1, name + "/im2": im2, name + "/wx": wx, name + "/wy": wy, name + "/lvl": np.int32(4),
           name + "/out_x": ox, name + "/out_y": oy}
    for k, v in dump.items():
        out[name + "/" + k] = v
    return out


def depth_cases(noc):
    """stereo depth (SELECTMODE 2) refinement: same inputs as fdf_cases (read back from its fixture), outputs of the
    reference's compute_data_DE / sor_coupled_slow_but_readable_DE chain (RefLevelDE) for both camera sides.
    The start displacement is -|wx| for the left camera (camlr 0, disparity <= 0) and +|wx| for the right one."""
    ref = R.FdfRef(noc)
    z = np.load(os.path.join(OUT, "fdf_ref_%s.npz" % ("gray" if noc == 1 else "rgb")))
    out = {}
    for name in sorted({k.split("/")[0] for k in z.files}):
        im1, im2, wx, lvl = z[name + "/im1"], z[name + "/im2"], z[name + "/wx"], int(z[name + "/lvl"])
        for camlr in (0, 1):
            dump = {}
            w0 = (-np.abs(wx) if camlr == 0 else np.abs(wx)).astype(np.float32)
            ox = ref.ref_level_de(im1, im2, w0, lvl, camlr=camlr, dump=dump)
            out["%s/out_de%d" % (name, camlr)] = ox
            for k, v in dump.items():
                out["%s/%s_de%d" % (name, k, camlr)] = v
    return out


def main():
    a0 = rgb(REF + "/images/alley_1/frame_0001.png")
    a1 = rgb(REF + "/images/alley_1/frame_0002.png")
    np.savez_compressed(os.path.join(OUT, "alley_1_gray.npz"), frame_0001=gray_cv(a0), frame_0002=gray_cv(a1),
                        rgb_crop_0001=a0[120:248, 400:656].copy(), rgb_crop_0002=a1[120:248, 400:656].copy())
    gold = O.read_flo(REF + "/kroeger/flows/alley_0001.flo")
    np.savez_compressed(os.path.join(OUT, "alley_0001_flo.npz"), flow=gold)
    np.savez_compressed(os.path.join(OUT, "fdf_ref_gray.npz"), **fdf_cases(1))
    np.savez_compressed(os.path.join(OUT, "fdf_ref_rgb.npz"), **fdf_cases(3))
    np.savez_compressed(os.path.join(OUT, "fdf_ref_l4_gray.npz"), **fdf_level4_case(1))
    np.savez_compressed(os.path.join(OUT, "fdf_ref_l4_rgb.npz"), **fdf_level4_case(3))
    np.savez_compressed(os.path.join(OUT, "fdf_ref_depth_gray.npz"), **depth_cases(1))
    np.savez_compressed(os.path.join(OUT, "fdf_ref_depth_rgb.npz"), **depth_cases(3))
    np.savez_compressed(os.path.join(OUT, "natural_images.npz"), road_HD=gray_cv(rgb(REF + "/images/road_HD.jpg")),
                        yosemite_4k=gray_cv(rgb(REF + "/images/yosemite_4k.jpg")))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
