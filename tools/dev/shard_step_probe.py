"""dev probe: per-step time of ONE rank's share (row tile 1/N of a 512x640 frame) incl. the replicated per-frame work,
without torch.distributed -- how much fixed overhead stands between N ranks and N-fold throughput."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops
from uforecon_amd.dist import RayShard
from uforecon_amd.scene import make_frame
dev = torch.device("cuda", 0)
wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
W = ops.PackedWeights({k: torch.from_numpy(wz[k]).to(dev) for k in wz.files})
frame = make_frame(512, 640, 3, seed=0).to(dev)
import itertools
for world, chunk in ((8, 0), (6, 0), (4, 0), (3, 0), (2, 0), (1, 0)):
    shard = RayShard(512, 640, world, 0)
    ray_idx = shard.ray_indices(dev)
    RN = ray_idx.numel()
    ws = ops.RenderWorkspace(dev, 64, 64, 3, chunk_rays=chunk, n_streams=3)
    out = dict(depth=torch.empty(RN, device=dev), depth_z=torch.empty(RN, device=dev), rgb=torch.empty(RN, 3, device=dev))
    def step():
        fh = ops.FrameHandle(frame.batch, frame.source_imgs_feat, frame.feature_volume, frame.match_feature)
        U1 = torch.rand(64, RN, device=dev); U2 = torch.rand(64, RN, device=dev)
        ops.render_rays(fh, W, ray_idx, U1, U2, workspace=ws, want_srdf=False, out=out)
        return fh
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        fh = ops.FrameHandle(frame.batch, frame.source_imgs_feat, frame.feature_volume, frame.match_feature)
    torch.cuda.synchronize()
    prep = (time.perf_counter() - t0) / 5
    print(f"world {world} chunk {chunk}: {RN} rays/rank, step {dt*1e3:.2f} ms (frame prep alone {prep*1e3:.2f} ms) -> projected {512*640/dt/1e6:.3f} M rays/s aggregate, "
          f"efficiency vs world=1 see first line")
