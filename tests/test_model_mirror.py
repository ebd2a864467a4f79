"""The host-side mirror of the reference API: state_dict keys (CPU) and infer() parity (GPU)."""
import argparse

import pytest
import torch

from helpers import CASES, REL_TOL, border_degenerate_rays, case_inputs, load_weights, max_rel_elem, rel_err
from uforecon_amd import model as M
from uforecon_amd import ops


def _args(**over):
    a = dict(extract_geometry=True, test_sample_coarse=64, test_sample_fine=64, coarse_sample=64, fine_sample=64,
             volume_type="correlation", volume_reso=96, mvs_depth_guide=1, depth_pos_encoding=True,
             use_dir_srdf=False, explicit_similarity=True, test_coarse_only=False, test_ray_num=800)
    a.update(over)
    return argparse.Namespace(**a)


def test_state_dict_keys_match_the_reference():
    """Every per-ray key of the reference checkpoint exists with the same shape (golden weights file =
    reference state_dict filtered to ray_transformer.* / deviation_network.*)."""
    ref = load_weights()
    m = M.UFORecon(_args())
    sd = m.state_dict()
    assert set(sd) == set(ref), set(sd) ^ set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict(ref, strict=True)
    assert set(ops.RAW_WEIGHT_KEYS) <= set(sd)


def test_unsupported_configurations_fail_loudly():
    with pytest.raises(ops.UfrError, match="use_dir_srdf"):
        M.RayTransformer(args=_args(use_dir_srdf=True))
    with pytest.raises(ops.UfrError, match="volume_type"):
        M.RayTransformer(args=_args(volume_type="featuregrid"))
    m = M.UFORecon(_args())
    with pytest.raises(ops.UfrError, match="fused"):
        m.ray_transformer(None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_hier_small", "c1_coarse_only"])
def test_infer_extract_geometry_matches_golden(name):
    c = CASES[name]
    fr, idx, U1, U2, g = case_inputs(name)
    dev = "cuda:0"
    m = M.UFORecon(_args(test_coarse_only=c.get("coarse_only", False))).to(dev)
    m.load_state_dict(load_weights(), strict=True)
    f = fr.to(dev)
    with torch.no_grad():
        srdf, pts, depth, rgb = m.infer(f.batch, idx.to(dev), f.source_imgs_feat, f.feature_volume, extract_geometry=True,
                                        match_feature=f.match_feature, is_train=False, uniforms=(U1, U2))
    assert tuple(depth.shape) == (1, c["RN"]) and tuple(rgb.shape) == (1, c["RN"], 3)
    assert tuple(srdf.shape) == g["srdf"][None].shape and tuple(pts.shape) == g["points"][None].shape
    assert max_rel_elem(depth[0], g["depth"], 1e-3) < REL_TOL
    assert rel_err(pts[0], g["points"]) < 1e-5
    # weights updated in place -> packed copy must follow
    with torch.no_grad():
        m.deviation_network.variance.add_(0.05)
        _, _, depth2, _ = m.infer(f.batch, idx.to(dev), f.source_imgs_feat, f.feature_volume, extract_geometry=True,
                                  match_feature=f.match_feature, is_train=False, uniforms=(U1, U2))
    assert float((depth2 - depth).abs().max()) > 1e-5


@pytest.mark.gpu
def test_infer_training_signature_forward_matches_golden():
    fr, idx, U1, U2, g = case_inputs("c5_train_fwd")
    dev = "cuda:0"
    m = M.UFORecon(_args(extract_geometry=False)).to(dev)
    m.load_state_dict(load_weights(), strict=True)
    f = fr.to(dev)
    with torch.no_grad():
        out = m.infer(f.batch, idx.to(dev), f.source_imgs_feat, f.feature_volume, extract_geometry=False,
                      match_feature=f.match_feature, uniforms=(U1, U2))
    names = ["rgb_gt", "rgb", "depth", "depth_gt", "srdf", "opacity", "weight", "points_in_pixel",
             "rgb_2", "depth_2", "srdf_2", "opacity_2", "weight_2", "points_in_pixel_2", "z_val", "z_val_all", "variance"]
    assert len(out) == 17
    got = dict(zip(names, out))
    for k in ("rgb_gt", "depth_gt", "z_val"):
        assert rel_err(got[k], g[k]) < 1e-6, k
    for k in ("depth", "depth_2", "opacity", "opacity_2", "variance"):
        assert rel_err(got[k], g[k]) < REL_TOL, k
    for k in ("weight", "srdf"):
        assert rel_err(got[k], g[k]) < 1e-4, k
    assert rel_err(got["z_val_all"], g["z_val_all"]) < 1e-5
    # with gradients enabled the same call is differentiable (tests/test_gpu_backward.py checks the gradients)
    out = m.infer(f.batch, idx.to(dev), f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature, uniforms=(U1, U2))
    assert out[1].requires_grad and out[9].requires_grad and not out[14].requires_grad
    assert rel_err(out[9], g["depth_2"]) < REL_TOL


def test_save_depth_outputs_wire_format(tmp_path):
    """The files tsdf fusion consumes (model.py:828-842): pickled dict .npy + 8-bit previews, truncating casts."""
    import numpy as np
    from PIL import Image

    rng = np.random.default_rng(0)
    depths = (rng.random((12, 16)) * 900 + 100).astype(np.float32)
    rgbs = rng.random((12, 16, 3)).astype(np.float32)
    E, K = np.eye(4, dtype=np.float32), np.eye(3, dtype=np.float32) * 2
    M.save_depth_outputs(str(tmp_path), "scan24", "00000000", depths, rgbs, E, K)
    d = np.load(tmp_path / "depth" / "scan24" / "00000000.npy", allow_pickle=True).item()
    assert set(d) == {"depth", "extrinsic", "intrinsic"}
    assert d["depth"].dtype == np.float32 and np.array_equal(d["depth"], depths)
    assert np.array_equal(d["extrinsic"], E) and np.array_equal(d["intrinsic"], K)
    png = np.asarray(Image.open(tmp_path / "scan24" / "depth" / "00000000.png"))
    assert png.dtype == np.uint8 and np.array_equal(png, ((depths / depths.max()).astype(np.float32) * 255).astype(np.uint8))
    jpg = Image.open(tmp_path / "rgb" / "scan24" / "00000000.jpg")
    assert jpg.size == (16, 12) and jpg.mode == "RGB"


@pytest.mark.gpu
def test_extract_geometry_frame_matches_per_chunk_infer(tmp_path):
    """One frame-level call == the reference's loop over ray chunks (model.py:815-823) + depth post-processing."""
    import numpy as np

    fr, idx, U1, U2, g = case_inputs("c2_hier_small")
    dev = "cuda:0"
    m = M.UFORecon(_args()).to(dev)
    m.load_state_dict(load_weights(), strict=True)
    f = fr.to(dev)
    H, W = f.batch["source_imgs"].shape[-2:]
    HW = H * W
    gen = torch.Generator().manual_seed(5)
    U1f, U2f = torch.rand(64, HW, generator=gen), torch.rand(64, HW, generator=gen)
    f.batch["scale_mat"] = torch.eye(4)[None] * 2.5
    f.batch["meta"] = ["dtu-scan24-3-00000000"]
    f.batch["extrinsic_render_view"] = torch.eye(4)[None]
    f.batch["intrinsic_render_view"] = torch.eye(3)[None]
    with torch.no_grad():
        depths, rgbs = m.extract_geometry(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature,
                                          out_dir=str(tmp_path), uniforms=(U1f, U2f))
        # the reference's loop: chunks of 800 rays, depth * cam_ray_d.z, concatenated, * scale_mat[0][0,0]
        parts = []
        for r0 in range(0, HW, 800):
            ridx = torch.arange(r0, min(r0 + 800, HW))[None].to(dev)
            _, _, depth, _ = m.infer(f.batch, ridx, f.source_imgs_feat, f.feature_volume, extract_geometry=True,
                                     match_feature=f.match_feature, is_train=False,
                                     uniforms=(U1f[:, r0:r0 + 800], U2f[:, r0:r0 + 800]))
            parts.append(depth[0] * f.batch["cam_ray_d"][0][2, ridx[0]])
    want = (torch.cat(parts).view(H, W) * 2.5).cpu().numpy()
    assert np.array_equal(depths, want)                      # rays are independent: chunking is invisible
    d = np.load(tmp_path / "depth" / "scan24" / "00000000.npy", allow_pickle=True).item()
    assert np.array_equal(d["depth"], depths) and rgbs.shape == (H, W, 3)
