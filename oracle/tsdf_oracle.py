"""CPU restatement of the reference's TSDF `integrate` kernel.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/bench_tsdf.py's cpu leg).

What is restated: the arithmetic of the reference's GPU kernel -- the path production runs -- in numpy fp32, one
operation per line, unfused (tsdf_fusion.py:88-152), plus the volume set-up of `TSDFVolume.__init__` (:40-56) and the
bounds loop of `save_tsdf` (:459-472, `get_view_frustum` :367-381).
Pinned: partially.  pycuda is absent, so that kernel cannot be run here; tests/golden/tsdf_*.npz come from the
reference's own CPU mode (same algorithm, fp64 camera transform, np.round, `z > 0`), and tests/test_tsdf.py requires
this restatement to equal them on every voxel except a bounded handful that sit on a rounding boundary of the pixel
index or of the truncation test, where the reference's two paths themselves disagree.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def rigid_transform(xyz, transform):
    """tsdf_fusion.py:359-364"""
    xyz_h = np.hstack([xyz, np.ones((len(xyz), 1), dtype=np.float32)])
    return np.dot(transform, xyz_h.T).T[:, :3]


def get_view_frustum(depth_im, cam_intr, cam_pose):
    """tsdf_fusion.py:367-381: the 5 corners of the camera frustum out to the largest depth, in world coordinates."""
    im_h, im_w = depth_im.shape
    max_depth = np.max(depth_im)
    pts = np.array([
        (np.array([0, 0, 0, im_w, im_w]) - cam_intr[0, 2]) * np.array([0, max_depth, max_depth, max_depth, max_depth]) / cam_intr[0, 0],
        (np.array([0, 0, im_h, 0, im_h]) - cam_intr[1, 2]) * np.array([0, max_depth, max_depth, max_depth, max_depth]) / cam_intr[1, 1],
        np.array([0, max_depth, max_depth, max_depth, max_depth])])
    return rigid_transform(pts.T, cam_pose).T


def volume_bounds(depths, intrinsics, poses):
    """save_tsdf: tsdf_fusion.py:459-472 (the hull starts from the origin: np.zeros((3,2)))."""
    b = np.zeros((3, 2))
    for d, K, P in zip(depths, intrinsics, poses):
        pts = get_view_frustum(d, K, P)
        b[:, 0] = np.minimum(b[:, 0], np.amin(pts, axis=1))
        b[:, 1] = np.maximum(b[:, 1], np.amax(pts, axis=1))
    return b


def volume_layout(vol_bnds, voxel_size):
    """TSDFVolume.__init__: tsdf_fusion.py:40-56 -> (vol_dim int[3], vol_origin f32[3])."""
    vol_bnds = np.asarray(vol_bnds, dtype=np.float64)
    vol_dim = np.round((vol_bnds[:, 1] - vol_bnds[:, 0]) / float(voxel_size)).copy(order="C").astype(int)
    return vol_dim, vol_bnds[:, 0].copy(order="C").astype(np.float32)


def fold_color(color_im):
    """tsdf_fusion.py:235-238"""
    c = np.asarray(color_im).astype(np.float32)
    return np.floor(c[..., 2] * F(256 * 256) + c[..., 1] * F(256) + c[..., 0]).astype(np.float32)


def _roundf(v):
    """C roundf (half away from zero) of fp32 values, exactly (in float64)."""
    v64 = v.astype(np.float64)
    return (np.sign(v64) * np.floor(np.abs(v64) + 0.5)).astype(np.float32)


def integrate(tsdf, weight, color, vol_origin, voxel_size, trunc_margin, cam_intr, cam_pose, depth_im, color_folded=None,
              obs_weight=1.0, integrate_color=False):
    """One observation into (X,Y,Z) fp32 volumes, in place.  Kernel arithmetic of tsdf_fusion.py:99-152, fp32, unfused.
    integrate_color=False reproduces the reference (its colour block is unreachable: `return;` at :139)."""
    X, Y, Z = tsdf.shape
    im_h, im_w = depth_im.shape
    K = np.asarray(cam_intr, F).reshape(3, 3)
    P = np.asarray(cam_pose, F).reshape(4, 4)
    vs, tm, ow = F(voxel_size), F(trunc_margin), F(obs_weight)
    vx, vy, vz = np.meshgrid(np.arange(X, dtype=F), np.arange(Y, dtype=F), np.arange(Z, dtype=F), indexing="ij")
    with np.errstate(all="ignore"):
        pt = [F(vol_origin[i]) + v * vs for i, v in enumerate((vx, vy, vz))]                    # :104-106
        t = [pt[i] - P[i, 3] for i in range(3)]                                                 # :108-110
        cam = [(P[0, j] * t[0] + P[1, j] * t[1]) + P[2, j] * t[2] for j in range(3)]            # :111-113 (R^T)
        px = _roundf(K[0, 0] * (cam[0] / cam[2]) + K[0, 2])                                     # :115
        py = _roundf(K[1, 1] * (cam[1] / cam[2]) + K[1, 2])
        inside = (px >= 0) & (px < im_w) & (py >= 0) & (py < im_h) & ~(cam[2] < 0) & np.isfinite(px) & np.isfinite(py)
        ix = np.where(inside, px, 0).astype(np.int64)
        iy = np.where(inside, py, 0).astype(np.int64)
        depth = np.asarray(depth_im, F)[iy, ix]
        diff = depth - cam[2]                                                                   # :128
        upd = inside & (depth != 0) & ~(diff < -tm)                                             # :123-130
        dist = np.minimum(F(1.0), diff / tm)                                                    # :131
        w_old = weight.copy()
        w_new = w_old + ow
        new_tsdf = (tsdf * w_old + ow * dist) / w_new                                           # :136
        weight[upd] = w_new[upd]
        tsdf[upd] = new_tsdf[upd]
        if integrate_color and color_folded is not None:                                        # :140-151 (dead code upstream)
            c256 = F(256 * 256)
            old = color.copy()
            old_b = np.floor(old / c256)
            old_g = np.floor((old - old_b * c256) / F(256))
            old_r = (old - old_b * c256) - old_g * F(256)
            new = np.asarray(color_folded, F)[iy, ix]
            new_b = np.floor(new / c256)
            new_g = np.floor((new - new_b * c256) / F(256))
            new_r = (new - new_b * c256) - new_g * F(256)
            mix = lambda o, n: np.minimum(_roundf((o * w_old + ow * n) / w_new), F(255.0))
            nb, ng, nr = mix(old_b, new_b), mix(old_g, new_g), mix(old_r, new_r)
            color[upd] = ((nb * c256 + ng * F(256)) + nr)[upd]
    return int(upd.sum())
