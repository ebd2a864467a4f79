"""CPU restatement of the correlation-volume construction step (first part).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/bench_correlate.py's cpu leg); the product path
(uforecon_amd.frustum -> libufr.so) never touches this module.

Pinned: tests/golden/correlate_*.npz hold outputs of the reference's own `homo_warping_trans` +
similarity + view aggregation run in the build container (tests/golden/make_golden_correlate.py);
tests/test_oracle_golden.py checks this restatement against them.

Follows
  homo_warping_trans      code1/encoder_utils/fmt/module.py:329-367
  DepthNet.forward step 2 code1/encoder_utils/fmt/TransMVSNet.py:66-97
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def fold_projection(proj_pair: torch.Tensor) -> torch.Tensor:
    """(2,4,4) [extrinsic, intrinsic] -> 4x4 with the top 3x4 = K[:3,:3] @ E[:3,:4]  (TransMVSNet.py:73-76)."""
    pp = proj_pair[None]                                     # the reference works on (B=1, ...) tensors: keep its
    out = pp[:, 0].clone()                                   # batched matmul / inverse code paths (they round differently
    out[:, :3, :4] = torch.matmul(pp[:, 1, :3, :3], pp[:, 0, :3, :4])   # from the 2-D ones)
    return out[0]


def relative_projection(src_proj: torch.Tensor, ref_proj: torch.Tensor) -> torch.Tensor:
    """src_proj @ inverse(ref_proj), 4x4  (module.py:340)."""
    return torch.matmul(src_proj[None], torch.inverse(ref_proj[None]))[0]


def warp_grid(proj: torch.Tensor, depth_values: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Normalised sampling grid (D, H*W, 2) of one source view (module.py:341-362); invalid -> -99."""
    D = depth_values.shape[0]
    rot, trans = proj[:3, :3], proj[:3, 3:4]
    y, x = torch.meshgrid([torch.arange(0, H, dtype=torch.float32), torch.arange(0, W, dtype=torch.float32)],
                          indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(H * W)))        # (3, HW)
    rot_xyz = torch.matmul(rot[None], xyz[None])[0]                              # (3, HW), batched like the reference
    rot_depth_xyz = rot_xyz.unsqueeze(1) * depth_values.reshape(1, D, -1)        # (3, D, HW)
    proj_xyz = rot_depth_xyz + trans.reshape(3, 1, 1)
    invalid = proj_xyz[2] < 1e-6
    proj_xy = proj_xyz[:2] / proj_xyz[2:3]
    xn = proj_xy[0] / ((W - 1) / 2) - 1
    yn = proj_xy[1] / ((H - 1) / 2) - 1
    xn[invalid] = -99.0
    yn[invalid] = -99.0
    return torch.stack((xn, yn), dim=-1)                                         # (D, HW, 2)


def view_similarity(src_fea: torch.Tensor, ref_fea: torch.Tensor, proj: torch.Tensor,
                    depth_values: torch.Tensor) -> torch.Tensor:
    """(C,H,W) source / reference features, relative projection, (D,H,W) hypotheses -> similarity (D,H,W):
    mean over channels of warped source x reference  (module.py:364-367; TransMVSNet.py:77-78)."""
    C, H, W = src_fea.shape
    D = depth_values.shape[0]
    grid = warp_grid(proj, depth_values, H, W)
    warped = F.grid_sample(src_fea[None], grid.reshape(1, D * H, W, 2), mode="bilinear", padding_mode="zeros",
                           align_corners=True).reshape(C, D, H, W)
    return (warped[None] * ref_fea[None].unsqueeze(2)).mean(1)[0]


def aggregate_views(similarity: torch.Tensor, view_weights: torch.Tensor) -> torch.Tensor:
    """(NS,D,H,W) per-view similarities, (NS,H,W) pixel-wise view weights -> (D,H,W)  (TransMVSNet.py:69-70, 86-97:
    the sums start from 0 and 1e-5 and take the views in order)."""
    s = torch.zeros_like(similarity[0])
    w = torch.full_like(view_weights[0], 1e-5)
    for i in range(similarity.shape[0]):
        s = s + similarity[i] * view_weights[i].unsqueeze(0)
        w = w + view_weights[i]
    return s / w.unsqueeze(0)


def correlate(ref_fea, src_feas, ref_proj_pair, src_proj_pairs, depth_values, view_weights=None):
    """Step 2 of DepthNet.forward for one frame: per-view similarities (NS,D,H,W) and, when view weights are given,
    their aggregate (D,H,W).  *_proj_pair: (2,4,4) as in batch['proj_matrices'] stage entries."""
    ref_new = fold_projection(ref_proj_pair)
    sims = []
    for src_fea, pp in zip(src_feas, src_proj_pairs):
        sims.append(view_similarity(src_fea, ref_fea, relative_projection(fold_projection(pp), ref_new), depth_values))
    sims = torch.stack(sims)
    return sims, (aggregate_views(sims, view_weights) if view_weights is not None else None)
