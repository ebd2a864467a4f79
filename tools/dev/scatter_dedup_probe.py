"""How many volume-gradient atomics does a training step need?  numpy model of gather_bwd's scatter for the configs[4]
shape (1024 random rays of a 512x640 frame, 128 merged samples, 3 views, 3 stages): raw corner updates, what the folding
of aligned lane groups leaves, what a full segmented reduction per corner kind leaves, and the distinct voxels."""
import numpy as np
import torch
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uforecon_amd.scene import make_cameras

H, W, NV, RN, SN = 512, 640, 3, 1024, 128
b = make_cameras(H, W, NV, 0.08, True)
rng = np.random.default_rng(0)
idx = rng.choice(H * W, RN, replace=False)
ray_d = b["ray_d"][0].numpy()[:, idx].T          # (RN,3)
ray_o = b["ray_o"][0].numpy()
near, far = [float(v) for v in b["near_fars"][0, 0]]
zc = np.linspace(0, 1, 64)[None] * (far - near) + near + (rng.random((RN, 64)) - 0.5) / 63 * (far - near)
surf = near + (far - near) * rng.uniform(0.3, 0.7, (RN, 1))
zf = surf + rng.normal(0, 0.02 * (far - near), (RN, 64))
z = np.sort(np.concatenate([zc, np.clip(zf, near, far)], 1), 1)
pts = ray_o[None, None] + z[..., None] * ray_d[:, None]           # (RN,SN,3)
poses = b["source_poses"][0].numpy()                                 # (NV,4,4)
tot = dict(raw=0, fold=0, kind_runs=0, kind_runs_zhand=0, distinct=0)
for v in range(NV):
    q = pts @ poses[v, :3, :3].T + poses[v, :3, 3]
    x, y = q[..., 0] / q[..., 2], q[..., 1] / q[..., 2]
    zn = (q[..., 2] - near) / (far - near) * 2 - 1
    for (D, s) in ((48, 4), (32, 2), (8, 1)):
        Hs, Ws = H // s, W // s
        ix, iy, iz = (x + 1) / 2 * (Ws - 1), (y + 1) / 2 * (Hs - 1), (zn + 1) / 2 * (D - 1)
        fx, fy, fz = np.floor(ix), np.floor(iy), np.floor(iz)
        keys = {}
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    cx, cy, cz = fx + dx, fy + dy, fz + dz
                    ok = (cx >= 0) & (cx <= Ws - 1) & (cy >= 0) & (cy <= Hs - 1) & (cz >= 0) & (cz <= D - 1)
                    k = np.where(ok, (cz * Hs + cy) * Ws + cx, -1).astype(np.int64)
                    keys[(dx, dy, dz)] = k
        allk = np.stack(list(keys.values()), 2)     # (RN,SN,8)
        tot["raw"] += int((allk >= 0).sum())
        tot["distinct"] += sum(len(np.unique(r[r >= 0])) for r in allk.reshape(RN, -1))
        for kind, k in keys.items():
            valid = k >= 0
            newrun = np.ones_like(k, bool)
            newrun[:, 1:] = (k[:, 1:] != k[:, :-1])
            tot["kind_runs"] += int((valid & newrun).sum())
            # aligned folding within 16-lane rows, groups of 2,4,8 (all lanes of the group equal)
            kk = k.reshape(RN, SN // 8, 8)
            surv = 0
            a = np.ones(kk.shape, bool) & (kk >= 0)
            for d in (1, 2, 4):
                g = kk.reshape(RN, SN // 8, 8 // (2 * d), 2, d)
                al = a.reshape(RN, SN // 8, 8 // (2 * d), 2, d)
                # lane pos 0 of each 2d-group absorbs lane pos d if both alive & equal key (only representative lanes matter)
                eq = (g[..., 0, 0] == g[..., 1, 0]) & al[..., 0, 0] & al[..., 1, 0]
                al[..., 1, 0] &= ~eq
            tot["fold"] += int(a.sum())
        # z hand-off after per-kind run reduction: a dz=1 run whose key equals the next sample's dz=0 key is merged
        for dy in (0, 1):
            for dx in (0, 1):
                k1, k0 = keys[(dx, dy, 1)], keys[(dx, dy, 0)]
                last = np.ones_like(k1, bool)
                last[:, :-1] = k1[:, 1:] != k1[:, :-1]
                handed = np.zeros_like(k1, bool)
                handed[:, :-1] = last[:, :-1] & (k1[:, :-1] >= 0) & (k1[:, :-1] == k0[:, 1:])
                tot["kind_runs_zhand"] -= int(handed.sum())
tot["kind_runs_zhand"] += tot["kind_runs"]
print({k: f"{v/1e6:.2f} M corner updates (x9 atomics)" for k, v in tot.items()})
