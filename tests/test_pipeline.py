"""The pieces compose: source images -> FeatureNet (deformable convolutions) -> FMT -> frustum cascade + matching
features -> per-ray renderer -> depth map files -> TSDF fusion, every stage through the product code (HIP kernels behind
the C ABI + library ops).  No reference numbers here (each stage has its own parity tests); this checks the hand-offs:
parameter names, shapes, layouts, value ranges."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from helpers import load_weights
from uforecon_amd import pipeline, tsdf
from uforecon_amd.scene import fill_state_dict, make_frame

HERE = os.path.dirname(os.path.abspath(__file__))


def _args():
    return argparse.Namespace(extract_geometry=True, test_sample_coarse=64, test_sample_fine=64, coarse_sample=64,
                              fine_sample=64, volume_type="correlation", volume_reso=96, mvs_depth_guide=1,
                              depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
                              test_coarse_only=False, test_ray_num=800, test_n_view=3, out_dir=None)


def test_state_dict_is_the_references():
    """Every one of the reference model's 530 state_dict entries, same shapes: its checkpoints load with strict=True."""
    ref = json.load(open(os.path.join(HERE, "golden", "reference_state_dict_shapes.json")))
    sd = pipeline.UFOReconInference(_args()).state_dict()
    assert set(sd) == set(ref)
    assert all(list(sd[k].shape) == ref[k] for k in ref)


@pytest.mark.gpu
def test_images_to_depth_map_files_to_tsdf(tmp_path):
    dev = "cuda:0"
    H, W, NV = 32, 64, 3
    fr = make_frame(H, W, NV, seed=0).to(dev)
    batch = fr.batch
    # the batch entries the encoder half reads (dtu_test_sparse.py:382-436): projection pairs per stage, initial hypotheses
    pm = {}
    for st, s in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        p = torch.zeros(1, NV, 2, 4, 4, device=dev)
        p[0, :, 0] = batch["w2cs"][0, :NV]
        K = batch["intrinsics"][0, :NV].clone()
        K[:, :2] = K[:, :2] / s
        p[0, :, 1, :3, :3] = K
        p[0, :, 1, 3, 3] = 1.0
        pm[st] = p
    batch["proj_matrices"] = pm
    near, far = float(batch["near_fars"][0, 0, 0]), float(batch["near_fars"][0, 0, 1])
    batch["depth_values_org_scale"] = torch.linspace(near, far, 48, device=dev)[None]
    batch["meta"] = ["dtu-scan24-3-00000000"]
    batch["extrinsic_render_view"] = batch["w2cs"][:, 0]
    batch["intrinsic_render_view"] = batch["intrinsics"][:, 0]
    net = fill_state_dict(pipeline.UFOReconInference(_args()), 21).eval().to(dev)
    net.load_state_dict(load_weights(), strict=False)            # the per-ray weights of the parity fixtures
    depths, rgbs = net.extract_geometry(batch, out_dir=str(tmp_path))
    assert depths.shape == (H, W) and rgbs.shape == (H, W, 3)
    assert np.isfinite(depths).all() and (depths > 0.5 * near).all() and (depths < 1.5 * far).all()
    assert batch["depth_info"].shape == (1, NV, H, W)
    d = np.load(tmp_path / "depth" / "scan24" / "00000000.npy", allow_pickle=True).item()
    assert np.array_equal(d["depth"], depths) and d["extrinsic"].shape == (4, 4) and d["intrinsic"].shape == (3, 3)
    # fuse what was written, the way tsdf_fusion.save_tsdf reads it back
    vol = tsdf.fuse_depth_maps([d["depth"], d["depth"]], [d["intrinsic"]] * 2, [d["extrinsic"]] * 2, voxel_size=0.1, margin=3)
    t, _, w = vol.get_volume()
    assert (w > 0).sum() > 100 and np.isfinite(t).all() and t.min() >= -1.0 and t.max() <= 1.0


@pytest.mark.gpu
def test_backbone_once_per_image_equals_once_per_rotation():
    """encode_frame runs FeatureNet on the N distinct images instead of on every view of every rotation (N x N images,
    TransMVSNet.py:175-178): same feature tensors."""
    dev = "cuda:0"
    fr = make_frame(32, 64, 3, seed=1).to(dev)
    net = fill_state_dict(pipeline.UFOReconInference(_args()), 22).eval().to(dev)
    imgs = fr.batch["source_imgs"]
    dummy = {st: torch.zeros(1, 3, 2, 4, 4, device=dev) for st in ("stage1", "stage2", "stage3")}
    rot_imgs, _, _ = net.build_pairs(imgs, dummy, torch.zeros(1, 4, device=dev))
    with torch.no_grad():
        base = net.transmvsnet.feature(imgs[0])
        for v in range(3):
            per_rotation = net.transmvsnet.feature(rot_imgs[:, v])
            comb = [(r + v) % 3 for r in range(3)]
            for st in ("stage1", "stage2", "stage3"):
                assert torch.allclose(per_rotation[st], base[st][comb], rtol=0, atol=1e-6), (v, st)
