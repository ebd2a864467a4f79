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
    with pytest.raises(ops.UfrError, match="fea_volume"):       # featuregrid-style call without the frustum lookup
        m.ray_transformer(torch.zeros(1, 2, 16, 3), {}, torch.zeros(1, 3, 32, 4, 4))


def test_batch_elements_are_split_frame_major():
    """B > 1 (main.py:43 defaults to --batch_size 2): the mirror walks the batch one frame at a time.  The slicing helpers
    on CPU: tensors with a leading B are cut, scalars pass through, volumes arrive as one dict per frame or stacked
    frame-major along the view dim (the reference indexes them by view, model.py:363-364)."""
    batch = {"source_imgs": torch.zeros(2, 3, 3, 8, 8), "ray_o": torch.arange(6.).reshape(2, 3), "start_idx": 0,
             "meta": ["a-1-2", "b-3-4"]}
    b1 = M._frame_of(batch, 1, 2)
    assert tuple(b1["source_imgs"].shape) == (1, 3, 3, 8, 8) and b1["start_idx"] == 0 and b1["meta"] == ["b-3-4"]
    assert torch.equal(b1["ray_o"], torch.tensor([[3., 4., 5.]]))
    vol = {"stage1": {"feature_volume": torch.arange(6.).reshape(6, 1, 1, 1, 1).expand(6, 8, 2, 2, 2),
                      "weight_volume": torch.zeros(6, 1, 2, 2, 2)}}
    v1 = M._volumes_of(vol, 1, 2)
    assert tuple(v1["stage1"]["feature_volume"].shape) == (3, 8, 2, 2, 2) and float(v1["stage1"]["feature_volume"][0, 0, 0, 0, 0]) == 3.0
    assert M._volumes_of([vol, "second"], 1, 2) == "second"
    with pytest.raises(ops.UfrError, match="multiple of the batch size"):
        M._volumes_of({"stage1": {"feature_volume": torch.zeros(5, 8, 2, 2, 2), "weight_volume": torch.zeros(5, 1, 2, 2, 2)}}, 0, 2)
    assert M._match_of([torch.zeros(2, 3, 64, 4, 4)], 1)[0].shape[0] == 1
    # depth_info in the reference's (1, B*V, H, W) layout: frame b gets ITS maps; a leading-B layout is sliced like the rest
    d = torch.arange(6.0).reshape(1, 6, 1, 1).expand(1, 6, 2, 2)
    f1 = M._frame_of({"depth_info": d, "x": torch.zeros(2, 5)}, 1, 2)
    assert f1["depth_info"].shape == (1, 3, 2, 2) and f1["depth_info"][0, :, 0, 0].tolist() == [3.0, 4.0, 5.0] and f1["x"].shape == (1, 5)
    assert M._frame_of({"depth_info": torch.zeros(2, 3, 2, 2)}, 1, 2)["depth_info"].shape == (1, 3, 2, 2)
    with pytest.raises(ops.UfrError, match="multiple of the batch size"):
        M._frame_of({"depth_info": torch.zeros(1, 5, 2, 2)}, 0, 2)


@pytest.mark.gpu
def test_infer_over_a_batch_of_two_frames_equals_two_calls():
    """`infer` over B = 2 (model.py:409-427 is written over a leading B): every output equals the per-frame call's, laid out
    frame-major -- both signatures (extract_geometry 4-tuple, training 17-tuple)."""
    from uforecon_amd.scene import make_frame, sampler_uniforms

    dev = "cuda:0"
    H, W, RN = 64, 96, 48
    frames = [make_frame(H, W, 3, seed, train_layout=True).to(dev) for seed in (3, 4)]
    batch = {}
    for k, v in frames[0].batch.items():
        batch[k] = torch.cat([fr.batch[k] for fr in frames], 0) if isinstance(v, torch.Tensor) else v
    # the reference's own layout of the MVS depth guide: "(B V) H W" unsqueezed (model.py:530-531), not a leading B
    batch_ref = dict(batch, depth_info=torch.cat([fr.batch["depth_info"] for fr in frames], 1))
    assert tuple(batch_ref["depth_info"].shape) == (1, 6, H, W)
    feat = torch.cat([fr.source_imgs_feat for fr in frames], 0)
    match = [torch.cat([fr.match_feature[0] for fr in frames], 0)]
    vols_stacked = {st: {k: torch.cat([fr.feature_volume[st][k] for fr in frames], 0) for k in frames[0].feature_volume[st]}
                    for st in frames[0].feature_volume}
    vols_list = [fr.feature_volume for fr in frames]
    g = torch.Generator().manual_seed(5)
    idx = torch.stack([torch.randperm(H * W, generator=g)[:RN] for _ in range(2)]).to(dev)
    U1, U2 = sampler_uniforms(9, 64, 64, 2 * RN)
    for extract in (True, False):
        m = M.UFORecon(_args(extract_geometry=extract)).to(dev)
        m.load_state_dict(load_weights(), strict=True)
        with torch.no_grad():
            both = m.infer(batch, idx, feat, vols_stacked, extract_geometry=extract, match_feature=match, uniforms=(U1, U2))
            both_l = m.infer(batch_ref, idx, feat, vols_list, extract_geometry=extract, match_feature=match, uniforms=(U1, U2))
            single = [m.infer(fr.batch, idx[b:b + 1], fr.source_imgs_feat, fr.feature_volume, extract_geometry=extract,
                              match_feature=fr.match_feature, uniforms=(U1[:, b * RN:(b + 1) * RN], U2[:, b * RN:(b + 1) * RN]))
                      for b, fr in enumerate(frames)]
        assert len(both) == (4 if extract else 17)
        for i in range(len(both)):
            if not extract and i == 16:
                assert torch.equal(both[i], single[0][i])
                continue
            want = torch.cat([s_[i] for s_ in single], 0)
            assert both[i].shape == want.shape, i
            assert torch.equal(both[i], want) and torch.equal(both_l[i], want), i
        assert both[2].shape[0] == 2                 # depth: (B, RN)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_hier_small", "c1_coarse_only", "c2_hier_interior"])
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
    if c.get("interior"):       # no sample on an image border: RGB on every ray
        assert max_rel_elem(rgb[0], g["rgb"], floor=0.05) < REL_TOL
    else:                       # rays with a sample ON a border excepted (see test_gpu_parity)
        from oracle import ufo_oracle as O

        xy, _, mask_z = O.project(fr.batch["source_poses"][0], torch.from_numpy(g["points"]))   # the golden's own samples
        ok = ~border_degenerate_rays(dict(xy=xy, mask_z=mask_z))
        assert max_rel_elem(rgb[0][ok.to(dev)], torch.from_numpy(g["rgb"])[ok], floor=0.05) < REL_TOL
    # weights updated in place -> packed copy must follow
    with torch.no_grad():
        m.deviation_network.variance.add_(0.05)
        _, _, depth2, _ = m.infer(f.batch, idx.to(dev), f.source_imgs_feat, f.feature_volume, extract_geometry=True,
                                  match_feature=f.match_feature, is_train=False, uniforms=(U1, U2))
    assert float((depth2 - depth).abs().max()) > 1e-5


def _rows_model(dev="cuda:0"):
    fr, idx, U1, U2, g = case_inputs("rows_small")
    m = M.UFORecon(_args()).to(dev)
    m.load_state_dict(load_weights(), strict=True)
    i = idx.reshape(-1)
    ray_d = fr.batch["ray_d"][0][:, i].t().contiguous()
    cz = fr.batch["cam_ray_d"][0][2, i]
    near, far = fr.batch["near_fars"][0, 0, 0] / cz, fr.batch["near_fars"][0, 0, 1] / cz     # model.py:423-427
    ray_o = fr.batch["ray_o"][0][None].expand(i.numel(), 3).contiguous()                       # model.py:411-412
    return m, fr.to(dev), fr, idx, U1, U2, g, ray_o.to(dev), ray_d.to(dev), near.to(dev), far.to(dev)


@pytest.mark.gpu
def test_sampler_classes_match_golden_rows():
    """FixedSampler.sample_ray / ImportanceSampler.sample_ray with the reference's argument shapes (sampler.py:15, 74)."""
    from oracle import ufo_oracle as O

    m, f, fr, idx, U1, U2, g, ray_o, ray_d, near, far = _rows_model()
    pts, z, pd = m.fixed_sampler.sample_ray(ray_o, ray_d, near_z=near, far_z=far, uniforms=U1)
    assert tuple(pts.shape) == g["coarse.pts"].shape and tuple(pd.shape) == g["coarse.pts"].shape
    assert torch.equal(z.cpu(), torch.from_numpy(g["coarse.z"]))
    assert torch.equal(pts.cpu(), torch.from_numpy(g["coarse.pts"]))
    w = torch.from_numpy(g["coarse.weight"]).to(z.device)
    pts2, z2, _ = m.importance_sampler.sample_ray(ray_o, ray_d, w, z, uniforms=U2)
    _, z_ref = O.importance_sample(ray_o.cpu(), ray_d.cpu(), w.cpu(), z.cpu(), U2)     # same host: CDF rounding
    assert tuple(pts2.shape) == (z.shape[0], 64, 3)
    assert rel_err(z2, z_ref) < 5e-6
    assert rel_err(m.importance_sampler.last_merged, g["fine.z"]) < 1e-5                 # vs the reference's own merge
    assert rel_err(pts2, ray_o[:, None, :] + z2[..., None] * ray_d[:, None, :]) < 1e-6


@pytest.mark.gpu
def test_volume_renderer_class_matches_golden_rows():
    """VolumeRenderer.render(z_val, radiance, geo_value, deviation_network=...) -> 5-tuple (renderer.py:7, 48)."""
    m, f, fr, idx, U1, U2, g, *_ = _rows_model()
    dev = "cuda:0"
    RN, SN = g["coarse.z"].shape
    rgb, depth, opacity, weight, var = m.renderer.render(
        torch.from_numpy(g["coarse.z"]).to(dev), torch.from_numpy(g["coarse.radiance"]).reshape(RN, SN, 3).to(dev),
        torch.from_numpy(g["coarse.srdf"]).to(dev), deviation_network=m.deviation_network)
    assert rel_err(rgb, g["coarse.rgb"]) < 1e-5 and rel_err(depth, g["coarse.depth"]) < 1e-5
    assert rel_err(opacity, g["coarse.opacity"]) < 1e-5 and rel_err(weight, g["coarse.weight"]) < 1e-5
    assert rel_err(var, g["coarse.variance"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["coarse", "fine"])
def test_sample2rgb_and_ray_transformer_forward_match_golden_rows(tag):
    """UFORecon.sample2rgb (model.py:308-348, 7-tuple) and RayTransformer.forward (ray_transformer.py:175-322, 3-tuple)
    called like the reference calls them, against the reference's own intermediates."""
    m, f, fr, idx, U1, U2, g, ray_o, ray_d, near, far = _rows_model()
    dev = "cuda:0"
    z = torch.from_numpy(g[f"{tag}.z"]).to(dev)
    pts = torch.from_numpy(g[f"{tag}.pts"]).to(dev)
    RN, SN = z.shape
    NV = 3
    with torch.no_grad():
        rgb, depth, srdf, opacity, weight, pip, var = m.sample2rgb(f.batch, pts[None], z[None], ray_d[None], idx.to(dev),
                                                                   f.source_imgs_feat, f.feature_volume, f.match_feature)
    assert tuple(rgb.shape) == (1, RN, 3) and tuple(srdf.shape) == (RN, SN, 1) and tuple(weight.shape) == (1, RN, SN)
    assert tuple(pip.shape) == (1, NV, 2, RN, SN)
    xy_ref = torch.from_numpy(g[f"{tag}.xy"])                                   # (NV,RN,SN,2)
    assert rel_err(pip[0].permute(0, 2, 3, 1), xy_ref) < 5e-6
    assert max_rel_elem(depth[0], g[f"{tag}.depth"], 1e-3) < REL_TOL
    assert rel_err(srdf[..., 0], g[f"{tag}.srdf"]) < 1e-4
    assert rel_err(weight[0], g[f"{tag}.weight"]) < 1e-4 and rel_err(opacity[0], g[f"{tag}.opacity"]) < 1e-4
    assert rel_err(var, g[f"{tag}.variance"]) < 1e-6
    ok = ~border_degenerate_rays(dict(xy=xy_ref, mask_z=torch.from_numpy(g[f"{tag}.mask_z"])))
    assert max_rel_elem(rgb[0][ok.to(dev)], torch.from_numpy(g[f"{tag}.rgb"])[ok], floor=0.05) < REL_TOL

    # RayTransformer.forward with the frustum lookup and the pair similarity as inputs
    cond = {"feat_info": torch.from_numpy(g[f"{tag}.sim8"])[None].to(dev)}
    vol = torch.from_numpy(g[f"{tag}.vol24"]).reshape(1, RN, SN, 24).to(dev)
    with torch.no_grad():
        radiance, srdf2, pip2 = m.ray_transformer(pts[None], f.batch, f.source_imgs_feat, fea_volume=vol, cond_info=cond,
                                                  points_projected=xy_ref[None].to(dev),
                                                  mask_valid=torch.from_numpy(g[f"{tag}.mask_z"])[None].to(dev))
    assert tuple(radiance.shape) == (RN * SN, 3) and tuple(srdf2.shape) == (RN, SN, 1)
    assert rel_err(srdf2[..., 0], g[f"{tag}.srdf"]) < 1e-4
    assert rel_err(pip2[0].permute(0, 2, 3, 1), xy_ref) < 5e-6
    pt_ok = ok[:, None].expand(RN, SN).reshape(-1)
    assert rel_err(radiance[pt_ok.to(dev)], torch.from_numpy(g[f"{tag}.radiance"])[pt_ok]) < 1e-4


@pytest.mark.gpu
def test_ray_transformer_forward_is_differentiable():
    """Gradients w.r.t. fea_volume and the parameters through the standalone RayTransformer.forward == autograd through
    the oracle's dense half on the same inputs."""
    from helpers import grad_rel_err
    from oracle import ufo_oracle as O

    m, f, fr, idx, U1, U2, g, ray_o, ray_d, near, far = _rows_model()
    dev = "cuda:0"
    tag = "coarse"
    RN, SN = g[f"{tag}.z"].shape
    pts = torch.from_numpy(g[f"{tag}.pts"]).to(dev)
    vol = torch.from_numpy(g[f"{tag}.vol24"]).reshape(1, RN, SN, 24).to(dev).requires_grad_(True)
    cond = {"feat_info": torch.from_numpy(g[f"{tag}.sim8"])[None].to(dev)}
    gen = torch.Generator().manual_seed(11)
    co_r, co_s = torch.rand(RN * SN, 3, generator=gen) - 0.5, torch.rand(RN, SN, 1, generator=gen) - 0.5
    radiance, srdf, _ = m.ray_transformer(pts[None], f.batch, f.source_imgs_feat, fea_volume=vol, cond_info=cond)
    ((radiance * co_r.to(dev)).sum() + (srdf * co_s.to(dev)).sum()).backward()
    # oracle: same token inputs (from the HIP gather), vol24 / sim16 re-inserted as differentiable functions
    W = m.ray_transformer.packed_weights(torch.zeros((), device=dev))
    fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
    x, rgbm, dirs, _ = ops.project_gather(fh, W, ray_o[0].contiguous(), ray_d,
                                          torch.from_numpy(g[f"{tag}.z"]).to(dev).contiguous())
    P = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in load_weights().items()}
    volc = vol.detach().cpu().reshape(-1, 24).requires_grad_(True)
    sim16 = O.mlp3(cond["feat_info"].cpu().reshape(-1, 8), P, "ray_transformer.pre_sim_mlp.")
    xc = x.cpu().clone()
    xc = torch.cat([xc[:, :, :32], volc[:, None, :].expand(-1, 3, -1), sim16[:, None, :].expand(-1, 3, -1), xc[:, :, 72:]], -1)
    rad_o, srdf_o = O.aggregate_tokens(P, xc, rgbm.cpu()[..., :3], rgbm.cpu()[..., 3], dirs.cpu()[..., :3], RN, SN)
    ((rad_o * co_r).sum() + (srdf_o * co_s[..., 0]).sum()).backward()
    assert grad_rel_err(vol.grad.reshape(-1, 24), volc.grad) < 1e-3
    for k, p in m.ray_transformer.named_parameters():
        if k.endswith("linear_radianceweight_1_softmax.4.bias"):
            continue
        assert grad_rel_err(p.grad, P["ray_transformer." + k].grad) < 1e-3, k


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
    # the fine pass's rows sit at sample positions that differ from the reference's by the importance sampler's CDF
    # rounding (1e-6 relative), which the steep signed-distance head amplifies: 2e-4 sanity bounds there
    for k in ("srdf_2", "weight_2"):
        assert rel_err(got[k], g[k]) < 2e-4, k
    # colours: white-noise source images turn a 1-ulp coordinate difference into 1e-5 of colour; rays of this fixture are
    # not selected away from image borders, where the reference's inclusive mask is a step function of the last ulp
    from helpers import border_degenerate_rays
    from oracle import ufo_oracle as O

    want = {}
    with torch.no_grad():
        O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2,
                extract_geometry=False, want=want)
    ok = ~border_degenerate_rays(want["fine"])
    assert float(ok.float().mean()) > 0.85
    for k in ("rgb", "rgb_2"):
        assert rel_err(got[k][0][ok.to(dev)], torch.from_numpy(g[k])[0][ok]) < REL_TOL, k
    # points_in_pixel (B,NV,2,RN,SN): the projected sample positions, bit-level arithmetic of torch.bmm
    assert got["points_in_pixel"].shape == g["points_in_pixel"].shape
    assert rel_err(got["points_in_pixel"], g["points_in_pixel"]) < 1e-6
    assert got["points_in_pixel_2"].shape == g["points_in_pixel_2"].shape
    assert rel_err(got["points_in_pixel_2"], g["points_in_pixel_2"]) < 1e-5     # at the merged positions (1e-6 apart)
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
