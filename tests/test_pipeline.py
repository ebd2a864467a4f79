"""The pieces compose: source images -> FeatureNet (deformable convolutions) -> FMT -> frustum cascade + matching features -> per-ray renderer ->
depth maps -> TSDF fusion, every stage through the product code (HIP kernels behind the C ABI + library ops), on the GPU.
No reference numbers here (each stage has its own parity tests); this checks the hand-offs: shapes, layouts, value ranges."""
import argparse

import numpy as np
import pytest
import torch

from helpers import load_weights
from uforecon_amd import cascade, featurenet, model as M, tsdf
from uforecon_amd.scene import fill_state_dict, make_cascade_case, make_frame

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_images_to_depth_maps_to_tsdf():
    H, W, NV = 32, 64, 3
    fr = make_frame(H, W, NV, seed=0).to(DEV)
    batch = fr.batch
    # 1. encoder: backbone on the NV rotations of the source images (model.py:139-160 build_pairs), FMT, cascade, matching
    cc = make_cascade_case("small3")                      # projection pairs / hypotheses of the same cameras, NV rotations
    prod = fill_state_dict(cascade.FrustumBuilder(), 7).eval().to(DEV)
    backbone = fill_state_dict(featurenet.FeatureNet(8), 9).eval().to(DEV)
    imgs = batch["source_imgs"][0]                                                  # (NV, 3, H, W)
    rot = [list(range(i, NV)) + list(range(0, i)) for i in range(NV)]
    with torch.no_grad():
        feats = [backbone(torch.stack([imgs[rot[b][v]] for b in range(NV)])) for v in range(NV)]   # TransMVSNet.py:175-178
        assert feats[0]["stage1"].shape == (NV, 32, H // 4, W // 4) and feats[0]["stage3"].shape == (NV, 8, H, W)
        feats = prod.transmvsnet.encode(feats, ref_idx=0)
        frustums, info = prod(feats, cc["proj_matrices"], cc["depth_values"].to(DEV), (H, W))
        for f in feats:
            f["stage1"] = f["stage1"][0:1]                                          # model.py:782-783
        match_feature = prod.transmvsnet.get_match_feat(feats, cur_n_src_views=NV)
        source_imgs_feat = torch.stack([f["stage1"] for f in feats], dim=1)        # (1, NV, 32, h, w)   model.py:788-790
    assert source_imgs_feat.shape == (1, NV, 32, H // 4, W // 4) and match_feature[0].shape == (1, NV, 64, H // 4, W // 4)
    assert frustums["stage2"]["feature_volume"].shape == (NV, 8, 32, H // 2, W // 2)
    batch["depth_info"] = (info["stage3"]["depth"] * batch["scale_factor"].to(DEV))[None]          # model.py:804-806
    # 2. the per-ray path, whole frame in one call
    args = argparse.Namespace(extract_geometry=True, test_sample_coarse=64, test_sample_fine=64, coarse_sample=64,
                              fine_sample=64, volume_type="correlation", volume_reso=96, mvs_depth_guide=1,
                              depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
                              test_coarse_only=False, test_ray_num=800)
    net = M.UFORecon(args).to(DEV)
    net.load_state_dict(load_weights(), strict=True)
    with torch.no_grad():
        depths, rgbs = net.render_depth_map(batch, source_imgs_feat, frustums, match_feature)
    assert depths.shape == (H, W) and rgbs.shape == (H, W, 3)
    d = depths.cpu().numpy()
    near, far = float(batch["near_fars"][0, 0, 0]), float(batch["near_fars"][0, 0, 1])
    assert np.isfinite(d).all() and (d > 0.5 * near).all() and (d < 1.5 * far).all()
    # 3. fuse (the same depth map seen from the render camera three times is enough to exercise the hand-off)
    K = batch["intrinsics"][0, 0].cpu().numpy()
    E = batch["w2cs"][0, 0].cpu().numpy()
    vol = tsdf.fuse_depth_maps([d, d], [K, K], [E, E], voxel_size=0.1, margin=3)
    t, _, w = vol.get_volume()
    assert (w > 0).sum() > 100 and np.isfinite(t).all() and t.min() >= -1.0 and t.max() <= 1.0
