import os, sys, argparse
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
H, Wd = f.batch["source_imgs"].shape[-2:]
HW = H * Wd
gen = torch.Generator().manual_seed(5)
U1f, U2f = torch.rand(64, HW, generator=gen).to(DEV), torch.rand(64, HW, generator=gen).to(DEV)
ridx = torch.arange(HW, device=DEV)
def run(chunk, streams):
    ws = ops.RenderWorkspace(DEV, 64, 64, 3, chunk_rays=chunk, n_streams=streams)
    o = ops.render_rays(fh, W, ridx, U1f, U2f, workspace=ws)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
base = run(8192, 1)
for chunk, streams in ((800, 1), (1024, 1), (1024, 2), (4096, 2), (4096, 3), (512, 4)):
    o = run(chunk, streams)
    msg = []
    for k in ("depth", "rgb", "srdf", "z_all"):
        d = (o[k] - base[k]).abs()
        msg.append(f"{k}: {'eq' if torch.equal(o[k], base[k]) else 'DIFF max %.3e n %d first %d' % (float(d.max()), int((d > 0).sum()), int((d.reshape(HW, -1).amax(1) > 0).nonzero()[0]))}")
    print(chunk, streams, " | ".join(msg))
print("--- repeatability and coarse-only")
def run2(chunk, streams, coarse_only):
    ws = ops.RenderWorkspace(DEV, 64, 0 if coarse_only else 64, 3, chunk_rays=chunk, n_streams=streams)
    o = ops.render_rays(fh, W, ridx, U1f, None if coarse_only else U2f, workspace=ws, coarse_only=coarse_only)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
for co in (True, False):
    b = run2(8192, 1, co)
    for rep in range(4):
        o = run2(1024, 3, co)
        bad = ((o["depth"] != b["depth"]) | (o["srdf"] != b["srdf"]).any(1) | (o["z_all"] != b["z_all"]).any(1)).nonzero().flatten().tolist()
        print("coarse_only" if co else "hier", rep, "bad rays", bad[:12], "count", len(bad))
