"""Host side of TSDF fusion: a mirror of the reference's ``TSDFVolume`` (tsdf_fusion.py:20-357) whose ``integrate``
runs the HIP kernel of csrc/tsdf.hip instead of a pycuda-compiled CUDA string.  Same constructor arguments, same
attribute names, same ``integrate`` / ``get_volume`` signatures; the volumes live in HBM.  No CPU fallback: without a
GPU or libufr.so the constructor raises ``UfrError``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import UfrError, _stream


def rigid_transform(xyz, transform):
    """tsdf_fusion.py:359-364"""
    xyz_h = np.hstack([xyz, np.ones((len(xyz), 1), dtype=np.float32)])
    return np.dot(transform, xyz_h.T).T[:, :3]


def get_view_frustum(depth_im, cam_intr, cam_pose):
    """Corners of the camera frustum out to the largest depth, world coordinates (tsdf_fusion.py:367-381)."""
    im_h, im_w = depth_im.shape
    max_depth = np.max(depth_im)
    far = np.array([0, max_depth, max_depth, max_depth, max_depth])
    pts = np.array([(np.array([0, 0, 0, im_w, im_w]) - cam_intr[0, 2]) * far / cam_intr[0, 0],
                    (np.array([0, 0, im_h, 0, im_h]) - cam_intr[1, 2]) * far / cam_intr[1, 1], far])
    return rigid_transform(pts.T, cam_pose).T


class TSDFVolume:
    """Volumetric TSDF fusion of depth maps (tsdf_fusion.py:20).  ``use_gpu`` is accepted for signature
    compatibility; there is only the GPU path.  ``integrate_color=False`` is the reference's behaviour (its kernel
    returns before the colour block, tsdf_fusion.py:139): the colour volume stays zero."""

    def __init__(self, vol_bnds, voxel_size, use_gpu=True, margin=5, device="cuda:0", integrate_color=False):
        if not torch.cuda.is_available():
            raise UfrError("TSDFVolume needs a GPU: there is no CPU fallback")
        self._lib = _lib.load()
        vol_bnds = np.asarray(vol_bnds, dtype=np.float64)
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self._vol_bnds = vol_bnds
        self._voxel_size = float(voxel_size)
        self._trunc_margin = margin * self._voxel_size
        self._color_const = 256 * 256
        self._vol_dim = np.round((self._vol_bnds[:, 1] - self._vol_bnds[:, 0]) / self._voxel_size).copy(order="C").astype(int)
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].copy(order="C").astype(np.float32)
        self.device = torch.device(device)
        self.integrate_color = bool(integrate_color)
        dim = tuple(int(d) for d in self._vol_dim)
        self._tsdf_vol_gpu = torch.ones(dim, dtype=torch.float32, device=self.device)
        self._weight_vol_gpu = torch.zeros(dim, dtype=torch.float32, device=self.device)
        self._color_vol_gpu = torch.zeros(dim, dtype=torch.float32, device=self.device)
        self.gpu_mode = True

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.0):
        """color_im (H,W,3) or None, depth_im (H,W), cam_intr 3x3, cam_pose 4x4 camera-to-world (tsdf_fusion.py:220-265).
        Images may be numpy arrays (copied to the device) or device tensors."""
        dev = self.device
        depth = torch.as_tensor(np.asarray(depth_im, np.float32) if not torch.is_tensor(depth_im) else depth_im,
                                dtype=torch.float32).to(dev).contiguous()
        im_h, im_w = depth.shape
        folded = None
        if color_im is not None and self.integrate_color:
            c = torch.as_tensor(np.asarray(color_im, np.float32) if not torch.is_tensor(color_im) else color_im,
                                dtype=torch.float32).to(dev)
            folded = torch.floor(c[..., 2] * float(self._color_const) + c[..., 1] * 256.0 + c[..., 0]).contiguous()   # :235-238
        f32p = C.POINTER(C.c_float)
        dim = (C.c_int32 * 3)(*[int(d) for d in self._vol_dim])
        org = np.ascontiguousarray(self._vol_origin, np.float32)
        K = np.ascontiguousarray(np.asarray(cam_intr, np.float32).reshape(-1)[:9])
        P = np.ascontiguousarray(np.asarray(cam_pose, np.float32).reshape(-1)[:16])
        _lib.check(self._lib.ufr_tsdf_integrate(
            self._tsdf_vol_gpu.data_ptr(), self._weight_vol_gpu.data_ptr(), self._color_vol_gpu.data_ptr(), dim,
            org.ctypes.data_as(f32p), self._voxel_size, self._trunc_margin, K.ctypes.data_as(f32p), P.ctypes.data_as(f32p),
            depth.data_ptr(), folded.data_ptr() if folded is not None else None, im_h, im_w, float(obs_weight),
            1 if folded is not None else 0, _stream()), "ufr_tsdf_integrate")

    def get_volume(self):
        """(tsdf, color, weight) as numpy arrays (tsdf_fusion.py:312-317)."""
        return (self._tsdf_vol_gpu.cpu().numpy(), self._color_vol_gpu.cpu().numpy(), self._weight_vol_gpu.cpu().numpy())

    def get_mesh(self):
        """Marching cubes over the fused volume (tsdf_fusion.py:340-357); needs scikit-image like the reference."""
        try:
            from skimage import measure
        except ImportError as e:  # pragma: no cover
            raise UfrError("get_mesh needs scikit-image (marching cubes), as the reference does") from e
        tsdf_vol, color_vol, _ = self.get_volume()
        mc = getattr(measure, "marching_cubes_lewiner", None) or measure.marching_cubes
        verts, faces, norms, _ = mc(tsdf_vol, level=0)
        verts_ind = np.round(verts).astype(int)
        verts = verts * self._voxel_size + self._vol_origin
        rgb = color_vol[verts_ind[:, 0], verts_ind[:, 1], verts_ind[:, 2]]
        b = np.floor(rgb / self._color_const)
        g = np.floor((rgb - b * self._color_const) / 256)
        r = rgb - b * self._color_const - g * 256
        return verts, faces, norms, np.floor(np.asarray([r, g, b])).T.astype(np.uint8)


def fuse_depth_maps(depths, intrinsics, extrinsics, voxel_size=1.5, margin=3, colors=None, integrate_color=False):
    """The loop of the reference's save_tsdf (tsdf_fusion.py:459-499) over in-memory frames: bounds = hull of the view
    frusta (starting from the origin), then one `integrate` per view with cam_pose = inverse(extrinsic)."""
    poses = [np.linalg.inv(E) for E in extrinsics]
    vol_bnds = np.zeros((3, 2))
    for d, K, P in zip(depths, intrinsics, poses):
        pts = get_view_frustum(np.asarray(d), K, P)
        vol_bnds[:, 0] = np.minimum(vol_bnds[:, 0], np.amin(pts, axis=1))
        vol_bnds[:, 1] = np.maximum(vol_bnds[:, 1], np.amax(pts, axis=1))
    vol = TSDFVolume(vol_bnds, voxel_size=voxel_size, margin=margin, integrate_color=integrate_color)
    for i, (d, K, P) in enumerate(zip(depths, intrinsics, poses)):
        vol.integrate(None if colors is None else colors[i], d, K, P, obs_weight=1.0)
    return vol
