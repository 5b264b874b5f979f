import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The built libraries are git-ignored: build them before collection (make is incremental, so a stale libfotg.so is
    # rebuilt and an up-to-date one costs nothing; hipcc cross-compiles gfx950 without a GPU).  A broken build fails here,
    # loudly, not later as an unrelated error.
    import subprocess
    global BUILD_ERROR
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "flowonthego_amd", "csrc")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        # (no hipcc, a partial ROCm install ...): the oracle / shard / distributed tests do not need libfotg.so -- only the
        # tests that load it fail, each with this message (fixture `libfotg` below and flowonthego_amd._lib itself)
        BUILD_ERROR = "building libfotg.so failed:\n" + (r.stderr or "")[-2000:]
    # the checker (and, where the reference tree exists, oracle/_ref): the live-reference tests are skipped by a
    # collection-time test for oracle/_ref
    try:
        from oracle import oracle as _O
        _O.build()
    except Exception as e:               # (no gcc): the tests that use the oracle fail on their own
        print("building the oracle failed: %s" % e)


BUILD_ERROR = None


@pytest.fixture(autouse=True)
def _needs_libfotg(request):
    """a broken HIP build FAILS every test that loads the library (gpu-marked tests, tests/test_host.py) -- loudly, never a
    skip -- and leaves the oracle / shard / distributed tests running"""
    if BUILD_ERROR and ("gpu" in request.keywords or "test_host" in request.node.nodeid):
        pytest.fail(BUILD_ERROR)


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing in hipSetDevice"""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def alley():
    z = np.load(os.path.join(GOLDEN, "alley_1_gray.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def natural_images():
    """the reference's natural test images as 8-bit gray: road_HD (1080, 1920), yosemite_4k (2160, 3840)"""
    z = np.load(os.path.join(GOLDEN, "natural_images.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def alley_golden_flow():
    return np.load(os.path.join(GOLDEN, "alley_0001_flo.npz"))["flow"]


def load_fdf(noc, level4=True):
    """golden vectors of the reference's own FDF code: three small levels (+ a 120 x 68 level = 1080p level 4; the depth-mode
    fixtures cover the small levels only)"""
    cases = {}
    for stem in ("fdf_ref_%s.npz", "fdf_ref_l4_%s.npz")[:2 if level4 else 1]:
        z = np.load(os.path.join(GOLDEN, stem % ("gray" if noc == 1 else "rgb")))
        for k in z.files:
            name, key = k.split("/")
            cases.setdefault(name, {})[key] = z[k]
    return cases


def synth_pair(h, w, seed=1234, noc=1, shift=(5.0, 2.0), truth=False):
    """Seeded synthetic frame pair: band-limited 8-bit texture, frame1 = frame0 warped by a
    smooth known flow.  Same generator on the CPU (oracle) and GPU sides."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, noc), np.float64)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for o, g in enumerate((8, 16, 32, 64, 128, 256)):
        gh, gw = max(2, h * g // max(h, w) + 2), g + 2
        n = rng.uniform(-1, 1, (gh, gw, noc))
        fy, fx = yy * (gh - 1.001) / h, xx * (gw - 1.001) / w
        y0, x0 = fy.astype(int), fx.astype(int)
        ay, ax = (fy - y0)[..., None], (fx - x0)[..., None]
        v = (n[y0, x0] * (1 - ay) * (1 - ax) + n[y0, x0 + 1] * (1 - ay) * ax +
             n[y0 + 1, x0] * ay * (1 - ax) + n[y0 + 1, x0 + 1] * ay * ax)
        img += v / (o + 1)
    img = (img - img.min()) / (img.max() - img.min()) * 255.0
    f0 = np.round(img).astype(np.float32)
    u = shift[0] + 2.0 * np.sin(yy / h * 3.0) * np.cos(xx / w * 2.0)
    v = shift[1] + 2.0 * np.cos(yy / h * 2.0 + 1.0) * np.sin(xx / w * 3.0)
    sx, sy = np.clip(xx - u, 0, w - 1.001), np.clip(yy - v, 0, h - 1.001)
    x0, y0 = sx.astype(int), sy.astype(int)
    ax, ay = (sx - x0)[..., None], (sy - y0)[..., None]
    f1 = (f0[y0, x0] * (1 - ay) * (1 - ax) + f0[y0, x0 + 1] * (1 - ay) * ax +
          f0[y0 + 1, x0] * ay * (1 - ax) + f0[y0 + 1, x0 + 1] * ay * ax)
    f1 = np.round(f1).astype(np.float32)
    if noc == 1:
        f0, f1 = f0[..., 0], f1[..., 0]
    if truth:
        return f0, f1, np.stack([u, v], -1).astype(np.float32)
    return f0, f1
