import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import flowonthego_amd as F
from flowonthego_amd.oflow import OFClass, VarRefClass
from conftest import load_fdf
def run(bands, inner_lvl=None, sweeps=3):
    os.environ["FOTG_VR_BANDS"] = bands
    c = load_fdf(1)["w64h28"]
    im1, im2, wx, wy, lvl = c["im1"], c["im2"], c["wx"], c["wy"], int(c["lvl"])
    if inner_lvl is not None: lvl = inner_lvl
    _, h, w = im1.shape
    op = F.operating_point(2, 1024, 1); op.coarsest_scale = op.finest_scale = lvl; op.var_ref_iter = sweeps
    ofc = OFClass(op, F.img_params(width=w << lvl, height=h << lvl, padding=8))
    ps = 8
    padlvl = lambda a: np.pad(a.transpose(1, 2, 0), ((ps, ps), (ps, ps), (0, 0)), mode="edge")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    flow = d(np.stack([wx, wy], -1))[None].contiguous()
    VarRefClass(ofc, d(padlvl(im1))[None], d(padlvl(im2))[None], ofc.iparams[0], op, flow)
    return flow[0].cpu().numpy()
for lv in (0,):
    a = run("0", lv); b = run("1", lv)
    df = np.abs(a - b).max(-1)
    print("lvl", lv, "max diff", df.max(), "n diff", (df > 0).sum(), "of", df.size)
    ys, xs = np.nonzero(df > 0)
    if len(ys): print("first diffs (row,col):", list(zip(ys[:12], xs[:12])), "rows with diffs:", sorted(set(ys))[:40])
c = load_fdf(1)["w64h28"]
b = run("1", 0); a = run("0", 0)
print("banded - input:", np.abs(b[...,0]-c["wx"]).max(), " nonband - input:", np.abs(a[...,0]-c["wx"]).max())
print("a[0,:4]", a[0,:4,0], "b[0,:4]", b[0,:4,0], "wx", c["wx"][0,:4])

for sw in (1, 2, 3):
    a = run("0", 0, sw); b = run("1", 0, sw)
    df = np.abs(a - b).max(-1)
    print("sweeps", sw, "max diff", df.max(), "n diff", (df > 0).sum())
    ys, xs = np.nonzero(df > 0)
    if len(ys): print("   first diffs:", list(zip(ys[:8].tolist(), xs[:8].tolist())))
