"""Middlebury .flo files, the reference's output format: SaveFlowFile (kroeger/run_dense.cpp:16-57 == src/run_dense.cpp:26-67)
writes "PIEH", int32 width, int32 height, then height x width x nc float32 row-major (nc = 2 for optical flow).
Host-side only; nothing here is on the timed path."""
import numpy as np

TAG = b"PIEH"


def write_flo(path, flow):
    """flow: (h, w, 2) float32 (a torch tensor is copied to the host first)"""
    if hasattr(flow, "detach"):
        flow = flow.detach().cpu().numpy()
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError("flow must be (h, w, 2), got %s" % (flow.shape,))
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        f.write(TAG)
        f.write(np.array([w, h], dtype="<i4").tobytes())
        f.write(flow.astype("<f4", copy=False).tobytes())


def read_flo(path):
    with open(path, "rb") as f:
        if f.read(4) != TAG:
            raise ValueError("%s: not a .flo file (missing PIEH tag)" % path)
        w, h = (int(v) for v in np.frombuffer(f.read(8), "<i4"))
        data = np.frombuffer(f.read(), "<f4")
    if w <= 0 or h <= 0 or data.size != w * h * 2:
        raise ValueError("%s: header says %dx%d, payload has %d floats" % (path, w, h, data.size))
    return data.reshape(h, w, 2).astype(np.float32)


def write_pfm(path, disp):
    """Stereo depth mode output, SavePFMFile (kroeger/run_dense.cpp:60-81): header "Pf\\n<w> <h>\\n-1.000000\\n" (negative scale =
    little endian), then the rows BOTTOM-UP, each value NEGATED (the left camera's displacement is <= 0, the file holds the
    positive disparity).  disp: (h, w) or (h, w, 1) float32."""
    if hasattr(disp, "detach"):
        disp = disp.detach().cpu().numpy()
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    if disp.ndim == 3 and disp.shape[2] == 1:
        disp = disp[..., 0]
    if disp.ndim != 2:
        raise ValueError("disparity must be (h, w) or (h, w, 1), got %s" % (disp.shape,))
    h, w = disp.shape
    with open(path, "wb") as f:
        f.write(("Pf\n%d %d\n%f\n" % (w, h, -1.0)).encode("ascii"))
        f.write((-disp[::-1]).astype("<f4").tobytes())


def read_pfm(path):
    """inverse of write_pfm: returns the displacement field (h, w) as the engine produced it"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"Pf":
            raise ValueError("%s: not a single-channel PFM file" % path)
        w, h = (int(v) for v in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4")
    if data.size != w * h:
        raise ValueError("%s: header says %dx%d, payload has %d floats" % (path, w, h, data.size))
    return (-data.reshape(h, w)[::-1]).astype(np.float32)
