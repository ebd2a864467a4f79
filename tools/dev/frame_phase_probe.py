"""Per-phase cycles and shader clock of the view transformer INSIDE a whole-frame render (compact token layout, the
chunked launch sequence of ufr_render_rays), from the -DUFR_PHASE_TIMING build: UFR_LIB=.../libufr_phase.so.
usage: frame_phase_probe.py H W NV SN PN [frames]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops  # noqa: E402
from uforecon_amd.scene import make_frame  # noqa: E402

H, W, NV, SN, PN = (int(a) for a in sys.argv[1:6])
frames = int(sys.argv[6]) if len(sys.argv) > 6 else 2
DEV = "cuda:0"
wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
Wt = ops.PackedWeights({k: torch.from_numpy(wz[k]).to(DEV) for k in wz.files})
fr = make_frame(H, W, NV, 0).to(DEV)
fh = ops.FrameHandle(fr.batch, fr.source_imgs_feat, fr.feature_volume, fr.match_feature)
HW = H * W
idx = torch.arange(HW, device=DEV)
U1, U2 = torch.rand(SN, HW, device=DEV), torch.rand(PN, HW, device=DEV)
ws = ops.RenderWorkspace(DEV, SN, PN, NV, n_streams=1)
lib = ops._lib.load()
buf = (C.c_ulonglong * 32)()
ops.render_rays(fh, Wt, idx, U1, U2, workspace=ws, want_srdf=False)
torch.cuda.synchronize()
lib.ufr_debug_vt_phases(buf, 32, 1)
t0 = time.perf_counter()
for _ in range(frames):
    ops.render_rays(fh, Wt, idx, U1, U2, workspace=ws, want_srdf=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / frames
lib.ufr_debug_vt_phases(buf, 32, 1)
names = ["load", "qk gemm", "scores", "v gemm", "message", "merge gemm", "LN1", "MLP0", "relu+MLP2", "LN2+stores", "radiance MLP", "softmax"]
tot = sum(buf[:12])
print(f"{H}x{W} NV={NV} {SN}+{PN}: {dt * 1e3:.1f} ms/frame (single stream)")
for i, nm in enumerate(names):
    print(f"   phase {nm:14s} {buf[i] / frames / 1e6:10.2f} Mcyc/frame  {100.0 * buf[i] / tot:5.1f} %")
print(f"   total {tot / frames / 1e6:.2f} Mcyc per frame (wave 0 of every launch)")
if buf[20]:
    print(f"   wave 5: {buf[21] / buf[20] * 100:.0f} MHz")
