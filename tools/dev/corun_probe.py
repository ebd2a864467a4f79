"""Development probe: does the gather kernel co-run with the ray / view transformer of another chunk?
Two streams, each repeating one kernel; wall time of the pair vs the two alone."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops
from uforecon_amd.scene import make_frame
DEV = "cuda:0"
RN, NV = 16384, 3
wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
W = ops.PackedWeights({k: torch.from_numpy(wz[k]).to(DEV) for k in wz.files})
fr = make_frame(512, 640, NV, 0).to(DEV)
fh = ops.FrameHandle(fr.batch, fr.source_imgs_feat, fr.feature_volume, fr.match_feature)
idx = torch.arange(RN, device=DEV) + 200 * 640
ray_d = fr.batch["ray_d"][0][:, idx].t().contiguous(); ray_o = fr.batch["ray_o"][0].contiguous()
cz = fr.batch["cam_ray_d"][0][2, idx]
near = (fr.batch["near_fars"][0, 0, 0] / cz).contiguous(); far = (fr.batch["near_fars"][0, 0, 1] / cz).contiguous()
z = ops.sample_fixed(near, far, torch.rand(64, RN, device=DEV))
x, rgbm, dirs, _ = ops.project_gather(fh, W, ray_o, ray_d, z)
tok = torch.randn(RN * 192, 80, device=DEV)
REP = 10
def f_gather(): ops.project_gather(fh, W, ray_o, ray_d, z)
def f_rt(): ops.ray_transform(W, tok, RN, 192)
def f_vt(): ops.view_transform(W, x, rgbm, dirs)
def run(fa, fb):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(REP):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / REP * 1e3
for f in (f_gather, f_rt, f_vt): f()
for name, fa, fb in (("gather", f_gather, None), ("rt192", f_rt, None), ("vt", f_vt, None), ("gather+rt", f_gather, f_rt), ("gather+vt", f_gather, f_vt),
                     ("2x gather+rt", lambda: (f_gather(), f_gather()), f_rt)):
    ts = sorted(run(fa, fb) for _ in range(5))
    print(f"{name:14s} {ts[2]:.3f} ms per round")
