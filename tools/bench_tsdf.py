"""Micro-benchmark of the TSDF integrate kernel on a DTU-scale volume (voxel 1.5 mm over a ~0.6 m cube is 400^3;
here 384^3 = 56.6 M voxels, 226 MB per volume) and a 512x640 depth map: time per observation (HIP events via
ufr_profile_*), HBM roofline fraction on the algorithmic bytes (16 B per updated voxel + 4 B per visited voxel's
weight... see DESIGN.md).  (The numpy oracle is timed by tests/test_tsdf.py, the only place allowed to run it.)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uforecon_amd import ops, tsdf  # noqa: E402


def scene(n, H=512, W=640):
    K = np.array([[0.9 * W, 0, (W - 1) / 2], [0, 0.9 * W, (H - 1) / 2], [0, 0, 1]], np.float32)
    P = np.eye(4, dtype=np.float32)
    P[2, 3] = -3.0                                      # camera at z = -3 looking at +z
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (3.0 + 0.3 * np.sin(xs / 40.0) * np.cos(ys / 30.0)).astype(np.float32)   # a wavy wall through the volume
    bnds = np.array([[-1.0, 1.0], [-0.8, 0.8], [-1.0, 1.0]])
    vs = 2.0 / n
    return K, P, depth, bnds, vs


def measure(n: int = 384, reps: int = 10, verbose: bool = False):
    a = argparse.Namespace(n=n, reps=reps)
    K, P, depth, bnds, vs = scene(a.n)
    vol = tsdf.TSDFVolume(bnds.copy(), voxel_size=vs, margin=3)
    d = torch.from_numpy(depth).cuda()
    vol.integrate(None, d, K, P)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(a.reps):
        vol.integrate(None, d, K, P)
    torch.cuda.synchronize()
    p = ops.profile_read()["tsdf_integrate"]
    ops.profile_enable(False)
    ms = p["ms"] / p["launches"]
    n_vox = int(np.prod(vol._vol_dim))
    upd = int((vol._weight_vol_gpu > 0).sum())
    algo = 16 * upd
    if verbose:
        print(f"volume {tuple(vol._vol_dim)} = {n_vox / 1e6:.1f} M voxels, {upd / 1e6:.2f} M updated per observation: "
              f"{ms * 1e3:.1f} us/launch, {n_vox / ms / 1e6:.1f} G voxels/s, algorithmic {algo / 1e6:.1f} MB -> "
              f"{algo / ms / 1e6:.0f} GB/s = {algo / ms / 1e6 / 8000:.1%} of the 8 TB/s roof")
    return dict(volume=[int(v) for v in vol._vol_dim], voxels=n_vox, updated_voxels=upd, ms_per_observation=ms,
                voxels_per_s=n_vox / ms * 1e3, algorithmic_bytes=algo, achieved_gbps=algo / ms / 1e6,
                hbm_frac=algo / ms / 1e6 / 8000, depth_map="512x640")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=384)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    measure(a.n, a.reps, verbose=True)


if __name__ == "__main__":
    main()
