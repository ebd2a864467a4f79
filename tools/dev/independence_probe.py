"""dev probe: is a point's (gather, view transformer) result independent of the other points in the launch?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import case_inputs, load_weights
from uforecon_amd import ops
DEV = "cuda:0"
fr, idx, U1, U2, g = case_inputs("c2_hier_small")
W = ops.PackedWeights({k: v.to(DEV) for k, v in load_weights().items()})
f = fr.to(DEV)
fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
i = idx.reshape(-1)
ray_d = fr.batch["ray_d"][0][:, i].t().contiguous().to(DEV)
cz = fr.batch["cam_ray_d"][0][2, i]
near = (fr.batch["near_fars"][0, 0, 0] / cz).contiguous().to(DEV)
far = (fr.batch["near_fars"][0, 0, 1] / cz).contiguous().to(DEV)
ray_o = fr.batch["ray_o"][0].contiguous().to(DEV)
z = ops.sample_fixed(near, far, torch.rand(128, i.numel()).to(DEV))
RN = z.shape[0]
x, rgbm, dirs, _ = ops.project_gather(fh, W, ray_o, ray_d, z)
rad, srdf, dbg = ops.aggregate(W, x, rgbm, dirs, RN, 128, debug=True)
zs = z[:, 1::2].contiguous()
xs, rgbms, dirss, _ = ops.project_gather(fh, W, ray_o, ray_d, zs)
rads, srdfs, dbgs = ops.aggregate(W, xs, rgbms, dirss, RN, 64, debug=True)
sel = lambda t: t.reshape(RN, 128, *t.shape[1:])[:, 1::2].reshape(RN * 64, *t.shape[1:])
for name, a, b in (("x", sel(x), xs), ("rgbm", sel(rgbm), rgbms), ("dirs", sel(dirs), dirss),
                   ("view_out", sel(dbg["view_out"]), dbgs["view_out"]), ("radiance", sel(rad), rads)):
    d = (a - b).abs()
    print(name, "equal" if torch.equal(a, b) else f"DIFF max {float(d.max()):.3e} n {(d > 0).sum().item()} of {d.numel()}")
    if not torch.equal(a, b) and name == "view_out":
        bad = (d > 0).reshape(RN * 64, -1).any(1).nonzero().flatten()
        print("  first bad points", bad[:10].tolist(), "tokens", (d > 0).reshape(RN * 64, 4, 80).any(-1)[bad[:5]].tolist())

# which features / tokens differ, and does it follow position or neighbours?
a, b = sel(dbg["view_out"]), dbgs["view_out"]
d = (a != b).reshape(RN * 64, 4, 80)
print("diff count per token", d.sum((0, 2)).tolist())
print("diff count per feature tile (16)", d.reshape(RN * 64, 4, 5, 16).sum((0, 1, 3)).tolist())
print("diff count per point position mod 8", [int(d.reshape(-1, 8, 4, 80)[:, k].sum()) for k in range(8)])
# experiment 2: aggregate the SAME tokens twice, second time shifted by 4 points (one column tile) inside the launch
P = RN * 128
x2 = torch.cat([x[4:], x[:4]]).contiguous(); r2 = torch.cat([rgbm[4:], rgbm[:4]]).contiguous(); d2 = torch.cat([dirs[4:], dirs[:4]]).contiguous()
_, _, dbg2 = ops.aggregate(W, x2, r2, d2, RN, 128, debug=True)
v2 = torch.cat([dbg2["view_out"][-4:], dbg2["view_out"][:-4]])
print("shift by 4 points:", "equal" if torch.equal(v2, dbg["view_out"]) else f"DIFF n {(v2 != dbg['view_out']).sum().item()}")
x3 = torch.cat([x[1:], x[:1]]).contiguous(); r3 = torch.cat([rgbm[1:], rgbm[:1]]).contiguous(); d3 = torch.cat([dirs[1:], dirs[:1]]).contiguous()
_, _, dbg3 = ops.aggregate(W, x3, r3, d3, RN, 128, debug=True)
v3 = torch.cat([dbg3["view_out"][-1:], dbg3["view_out"][:-1]])
print("shift by 1 point:", "equal" if torch.equal(v3, dbg["view_out"]) else f"DIFF n {(v3 != dbg['view_out']).sum().item()}")
_, _, dbg4 = ops.aggregate(W, x, rgbm, dirs, RN, 128, debug=True)
print("same launch twice:", "equal" if torch.equal(dbg4["view_out"], dbg["view_out"]) else "DIFF")
