"""Generate the golden vectors of the per-ray path by RUNNING THE REFERENCE (CPU).

Run in the build container only:  ``python tests/golden/make_golden.py``.
Writes ``tests/golden/*.npz``: the reference's default-init per-ray weights (state_dict
keys, seed 0) and, per case, the reference outputs (+ intermediates for the small case).
Inputs are not stored: they are regenerated bit-identically from seeds by
``uforecon_amd.scene`` (each file carries a digest of them).
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from ref_harness import build_reference_model, ray_path_state_dict  # noqa: E402
from uforecon_amd.scene import frame_digest, make_frame  # noqa: E402

# name -> dict(scene=(H,W,NV,seed), RN, ray stride, samples, mode)
CASES = {
    # BASELINE.json configs[0]: 3 views, 64 coarse only, 256 rays
    "c1_coarse_only": dict(H=64, W=96, NV=3, seed=0, RN=256, coarse=64, fine=64, coarse_only=True),
    # configs[1] shapes at a small frame: 64+64 hierarchical
    "c2_hier_small": dict(H=64, W=96, NV=3, seed=0, RN=256, coarse=64, fine=64),
    # configs[1] at the full 512x640 frame, 256 rays spread over the image
    "c2_hier_512x640": dict(H=512, W=640, NV=3, seed=0, RN=256, coarse=64, fine=64),
    # configs[3]-like: 5 views, 128+128
    "c4_nv5_128": dict(H=48, W=64, NV=5, seed=3, RN=32, coarse=128, fine=128),
    # intermediates of every section-8(a) row, small RN
    "rows_small": dict(H=64, W=96, NV=3, seed=0, RN=8, coarse=64, fine=64, rows=True),
    # rays strictly inside the image: every output, RGB included, is compared on 100 % of the rays
    "c2_hier_interior": dict(H=64, W=96, NV=3, seed=7, RN=256, coarse=64, fine=64, interior=True),
    "c4_nv5_interior": dict(H=48, W=64, NV=5, seed=8, RN=48, coarse=64, fine=64, interior=True),
    # training layout (s_idx=1, no near/far division), forward only
    "c5_train_fwd": dict(H=64, W=96, NV=3, seed=5, RN=64, coarse=64, fine=64, train=True),
    # configs[1] at the full 512x640 frame on rays strictly inside every source image: depth AND RGB on 100 % of the rays
    # `srdf64`: the dense half of the path (view / ray transformer, DensityMLP) evaluated in float64 on the fp32 token inputs at
    # the reference's own sample positions -- a host-independent yardstick for the kernels' srdf rows (the fp32 reference is
    # itself one rounding realisation: two CPUs differ by 6e-5 here)
    "c2_hier_512x640_interior": dict(H=512, W=640, NV=3, seed=0, RN=256, coarse=64, fine=64, interior=True, srdf64=True),
    # BASELINE.json configs[3] AT FULL SIZE: 5 views, 600x800, 128+128 samples, rays strictly inside every source image
    "c4_full_interior": dict(H=600, W=800, NV=5, seed=3, RN=256, coarse=128, fine=128, interior=True, srdf64=True),
    # statistics of a TRAINED checkpoint instead of the default init (the real checkpoint is absent: .MISSING_LARGE_BLOBS):
    # every weight matrix x 8, LayerNorm gains up to 10 and biases in [-1, 1], 2-D feature maps x 30 -- dense-layer inputs
    # reach ~1e3.  The modified weights travel in the fixture.
    "c2_trained_like": dict(H=64, W=96, NV=3, seed=9, RN=128, coarse=64, fine=64, interior=True, render64=True,
                            trained_like=dict(w=8.0, gamma=10.0, feat=30.0)),
    # the same far beyond what fixed plane exponents could hold: every matrix x 64, feature maps x 300 (dense-layer inputs
    # reach ~1e6; round 4's per-matrix / per-layer exponents, ufr_weights_pack_for)
    "c2_trained_like_x64": dict(H=64, W=96, NV=3, seed=9, RN=128, coarse=64, fine=64, interior=True, render64=True,
                                trained_like=dict(w=64.0, gamma=10.0, feat=300.0)),
}


def apply_trained_like(model, fr, t, seed=11):
    """In-place: the 'trained-like' statistics of a case (see CASES['c2_trained_like']); the same function serves the tests
    (on the oracle's parameter dict via `params=`)."""
    g = torch.Generator().manual_seed(seed)
    named = model if isinstance(model, dict) else dict(model.named_parameters())
    with torch.no_grad():
        for k in sorted(named):
            p = named[k]
            if not k.startswith("ray_transformer.") or "view_token" in k:
                continue
            if p.dim() > 1:
                p.mul_(t["w"])
            elif "norm" in k and k.endswith("weight"):
                p.copy_(0.5 + (t["gamma"] - 0.5) * torch.rand(p.shape, generator=g))
            elif "norm" in k and k.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=g) * 2 - 1)
        fr.source_imgs_feat.mul_(t["feat"])
        fr.match_feature[0].mul_(t["feat"])

# reference autograd of the training loss (model.py:552-566) through infer(extract_geometry=False): gradients of every
# per-ray parameter and of the six sampled volumes -- BASELINE.json configs[4]
GRAD_CASES = {
    "c5_train_grads": dict(H=64, W=96, NV=3, seed=5, RN=64, coarse=64, fine=64),
    "c5_train_grads_nv4": dict(H=32, W=48, NV=4, seed=6, RN=24, coarse=32, fine=32),
}


def ray_indices(H, W, RN):
    step = max(1, (H * W) // RN)
    return (torch.arange(RN) * step + step // 3).clamp(max=H * W - 1)[None]


def interior_ray_indices(H, W, RN, margin=2, seed=0):
    """Distinct pixels at least `margin` pixels inside the image, scattered over rows AND columns.  The render view is
    source view 0 shifted along its camera x axis (dtu_test_sparse.py:269-272), so its first / last pixel ROW projects
    onto y = -1 / +1 of that view exactly, where the reference's inclusive in-bounds mask (grid_sample.py:13-17) is a
    step function of the last ulp; interior rays keep every comparison well-posed."""
    g = torch.Generator().manual_seed(seed)
    rows = torch.randint(margin, H - margin, (4 * RN,), generator=g)
    cols = torch.randint(margin, W - margin, (4 * RN,), generator=g)
    idx = torch.unique(rows * W + cols, sorted=False)
    assert idx.numel() >= RN
    return idx[torch.randperm(idx.numel(), generator=g)[:RN]][None]


def probe_rays(model, fr, idx, sampler_seed, train):
    """Run the reference on the rays `idx` and return, per ray, (a) the smallest distance of any sample's projection to
    an image border |x| = 1 / |y| = 1 of a source view it lies in front of, and (b) the smallest |input| of any ReLU of
    the per-ray modules.  Both are places where the reference itself is discontinuous in the last ulp (inclusive mask,
    grid_sample.py:13-17; ReLU derivative), so fixtures meant for tight comparisons avoid rays that sit on them."""
    RN = idx.numel()
    border = torch.full((RN,), float("inf"))
    margin = torch.full((RN,), float("inf"))
    hooks = []

    def relu_hook(mod, inp):
        t = inp[0].detach()
        margin.copy_(torch.minimum(margin, t.reshape(RN, -1).abs().min(1)[0]))

    for mod in model.ray_transformer.modules():
        if isinstance(mod, torch.nn.ReLU):
            hooks.append(mod.register_forward_pre_hook(relu_hook))
    orig = model.query_cond_info

    def qci(*a, **k):
        r = orig(*a, **k)
        xy, mz = r[1][0].detach(), r[2][0].detach()          # (NV,RN,SN,2), (NV,RN,SN)
        d = (xy.abs() - 1.0).abs().min(-1)[0]
        d = torch.where(mz > 0, d, torch.full_like(d, float("inf")))
        border.copy_(torch.minimum(border, d.permute(1, 0, 2).reshape(RN, -1).min(1)[0]))
        return r

    model.query_cond_info = qci
    with torch.no_grad():
        torch.manual_seed(sampler_seed)
        model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat, feature_volume=fr.feature_volume,
                    match_feature=fr.match_feature, extract_geometry=not train, is_train=train)
    model.query_cond_info = orig
    for h in hooks:
        h.remove()
    return border, margin


def clean_ray_indices(model, fr, H, W, RN, sampler_seed, train, relu_margin=0.0, border_margin=1e-4):
    """Interior rays none of whose samples projects within `border_margin` of a source-image border and (for gradient
    fixtures) none of whose ReLU inputs lies within `relu_margin` of zero.  Offending rays are replaced one for one by
    fresh candidates -- the other rays keep their column of the sampler's uniform draws -- until the set is clean."""
    pool = interior_ray_indices(H, W, min(4 * RN, (H - 4) * (W - 4) // 2))[0]
    idx, nxt = pool[:RN].clone(), RN
    for it in range(12):
        border, margin = probe_rays(model, fr, idx[None], sampler_seed, train)
        bad = torch.nonzero((border < border_margin) | (margin < relu_margin))[:, 0]
        print(f"  ray selection pass {it}: {bad.numel()} of {RN} rays replaced "
              f"(min border distance {float(border.min()):.2e}, min ReLU margin {float(margin.min()):.2e})")
        if bad.numel() == 0:
            return idx[None]
        for b in bad:
            idx[b] = pool[nxt]
            nxt += 1
    raise RuntimeError("could not find a clean ray set")


def srdf_in_float64(model, fr, pts, want_render=False):
    """srdf (RN,SN) at the sample positions `pts` (RN,SN,3) with EVERYTHING after the positions in float64: projection,
    gathers, pair similarity, frustum lookup, depth code, both transformers, DensityMLP -- through the functional
    restatement oracle/ufo_oracle.py (which reproduces the reference's fp32 rows to 2e-6) on float64 copies of the frame
    and the weights.  This is what the fp32 evaluations (the reference on any CPU, the kernels) are rounding realisations of."""
    from oracle import ufo_oracle as O

    def d(t):
        if torch.is_tensor(t):
            return t.double() if t.is_floating_point() else t
        if isinstance(t, dict):
            return {k: d(v) for k, v in t.items()}
        if isinstance(t, list):
            return [d(v) for v in t]
        return t

    P = d({k: v.detach() for k, v in ray_path_state_dict(model).items()})
    batch, feat, vols, match, pts = d(fr.batch), d(fr.source_imgs_feat), d(fr.feature_volume), d(fr.match_feature), pts.double()
    RN, SN, _ = pts.shape
    with torch.no_grad():
        s_idx = batch["start_idx"] if "start_idx" in batch else 1
        poses = batch["source_poses"][0]
        NV = poses.shape[0]
        xy, _, mask_z = O.project(poses, pts)
        sim8 = O.pair_similarity(xy, match[0][0], NV)
        vol24 = O.volume_lookup(poses, pts, vols, batch["near_fars"][0][0])
        x, rgb, dirs, mask = O.gather_inputs(P, pts, batch, feat[0], vol24, sim8, xy, mask_z.double(), s_idx)
        dirp = dirs.permute(1, 2, 0, 3).reshape(RN * SN, NV, 3).double()
        maskp = mask.permute(1, 2, 0).reshape(RN * SN, NV).double()
        rgbp = rgb.permute(2, 3, 0, 1).reshape(RN * SN, NV, 3).double()
        radiance, srdf = O.aggregate_tokens(P, x.double(), rgbp, maskp, dirp, RN, SN)
        if want_render:     # ... and the compositor on top: depth / colour of the rays at these positions
            z = (pts - batch["ray_o"][0]).norm(dim=-1)
            rgb, depth = O.composite(z, radiance.reshape(RN, SN, 3), srdf.reshape(RN, SN), P["deviation_network.variance"])[:2]
            return srdf.reshape(RN, SN), depth, rgb
    return srdf.reshape(RN, SN)


def run_case(name, c, weight_seed=0, sampler_seed=1):
    train = c.get("train", False)
    model = build_reference_model(weight_seed, test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                                  coarse_sample=c["coarse"], fine_sample=c["fine"],
                                  test_coarse_only=c.get("coarse_only", False), test_n_view=c["NV"],
                                  extract_geometry=not train)
    fr = make_frame(c["H"], c["W"], c["NV"], c["seed"], train_layout=train)
    dig = frame_digest(fr)          # of the seeded frame, before any trained-like scaling
    if c.get("trained_like"):
        apply_trained_like(model, fr, c["trained_like"])
    idx = (clean_ray_indices(model, fr, c["H"], c["W"], c["RN"], sampler_seed, train) if c.get("interior")
           else ray_indices(c["H"], c["W"], c["RN"]))
    out = {"input_digest": np.float64(dig), "ray_idx": idx.numpy(),
           "sampler_seed": np.int64(sampler_seed), "weight_seed": np.int64(weight_seed)}
    if c.get("trained_like"):
        for k, v in ray_path_state_dict(model).items():
            out["weights." + k] = v.numpy()
    cap = {}
    hooks = []
    if c.get("rows"):
        rt = model.ray_transformer
        calls = {"n": 0}

        def tag():
            return "coarse" if calls["n"] == 0 else "fine"

        def h_view(mod, inp, o):
            cap[f"{tag()}.x_tokens"] = inp[0].detach().clone()
            cap[f"{tag()}.view_out"] = o.detach().clone()

        def h_ray(mod, inp, o):
            cap[f"{tag()}.ray_out"] = o.detach().clone()

        def h_rw(mod, inp, o):
            cap[f"{tag()}.logit"] = o.detach().clone()

        hooks.append(rt.density_view_transformer.register_forward_hook(h_view))
        hooks.append(rt.density_ray_transformer.register_forward_hook(h_ray))
        hooks.append(rt.linear_radianceweight_1_softmax.register_forward_hook(h_rw))
        orig_s2r = model.sample2rgb
        orig_qci = model.query_cond_info
        orig_qdv = model.query_depth_from_volume
        orig_rt = rt.forward

        def qci(*a, **k):
            r = orig_qci(*a, **k)
            cap[f"{tag()}.sim8"] = r[0]["feat_info"][0].detach().clone()
            cap[f"{tag()}.xy"] = r[1][0].detach().clone()
            cap[f"{tag()}.mask_z"] = r[2][0].detach().clone()
            return r

        def qdv(*a, **k):
            r = orig_qdv(*a, **k)
            cap[f"{tag()}.vol24"] = r[0].detach().clone()
            return r

        def rtf(*a, **k):
            r = orig_rt(*a, **k)
            cap[f"{tag()}.radiance"] = r[0].detach().clone()
            cap[f"{tag()}.srdf"] = r[1].detach().clone()[..., 0]
            return r

        def s2r(batch, points_x, z_val, *a, **k):
            cap[f"{tag()}.z"] = z_val[0].detach().clone()
            cap[f"{tag()}.pts"] = points_x[0].detach().clone()
            r = orig_s2r(batch, points_x, z_val, *a, **k)
            cap[f"{tag()}.rgb"] = r[0][0].detach().clone()
            cap[f"{tag()}.depth"] = r[1][0].detach().clone()
            cap[f"{tag()}.opacity"] = r[3][0].detach().clone()
            cap[f"{tag()}.weight"] = r[4][0].detach().clone()
            cap[f"{tag()}.variance"] = r[6].detach().clone()
            calls["n"] += 1
            return r

        model.query_cond_info, model.query_depth_from_volume = qci, qdv
        rt.forward, model.sample2rgb = rtf, s2r

    with torch.no_grad():
        torch.manual_seed(sampler_seed)
        r = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat,
                        feature_volume=fr.feature_volume, match_feature=fr.match_feature,
                        extract_geometry=not train, is_train=train)
    for h in hooks:
        h.remove()
    if train:
        names = ["rgb_gt", "rgb", "depth", "depth_gt", "srdf", "opacity", "weight", "points_in_pixel",
                 "rgb_2", "depth_2", "srdf_2", "opacity_2", "weight_2", "points_in_pixel_2",
                 "z_val", "z_val_all", "variance"]
        for n, v in zip(names, r):
            out[n] = v.detach().numpy()
    else:
        srdf, pts, depth, rgb = r
        out.update(srdf=srdf[0].numpy(), points=pts[0].numpy(), depth=depth[0].numpy(), rgb=rgb[0].numpy())
    for k, v in cap.items():
        out[k] = v.numpy()
    if c.get("srdf64"):
        out["srdf64"] = srdf_in_float64(model, fr, torch.from_numpy(out["points"])).numpy().astype(np.float32)
        e = np.abs(out["srdf64"] - out["srdf"]).max() / np.abs(out["srdf"]).max()
        print(f"  float64 evaluation vs the reference's fp32 srdf at its own positions: {e:.2e} of scale")
    if c.get("render64"):
        # `depth64` / `rgb64`: the rays rendered in float64 from the reference's own sample positions on -- the yardstick of
        # a network that amplifies rounding (trained-like statistics).  The reference's fp32 run is one rounding realisation
        # of it; how far it sits from it is the scale of what any fp32 evaluation may deviate (tests/test_gpu_round3.py).
        s64, d64, c64 = srdf_in_float64(model, fr, torch.from_numpy(out["points"]), want_render=True)
        out["depth64"], out["rgb64"] = d64.numpy(), c64.numpy()
        e_d = float(np.abs(out["depth64"] - out["depth"]).max() / np.abs(out["depth64"]).max())
        e_c = float(np.abs(out["rgb64"] - out["rgb"]).max())
        print(f"  float64 rendering vs the reference's fp32 outputs at its own positions: depth {e_d:.2e} of scale, rgb {e_c:.2e}")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB")
    return model


def training_loss(r, batch, weight_rgb=1.0, weight_depth=1.0):
    """model.py:552-566 on the 17-tuple of infer (model.py:480-482)."""
    rgb_gt, rgb, depth, depth_gt, rgb_2, depth_2 = r[0], r[1], r[2], r[3], r[8], r[9]
    loss_rgb = torch.nn.functional.mse_loss(rgb, rgb_gt)
    loss_rgb2 = torch.nn.functional.mse_loss(rgb_2, rgb_gt)
    nf = batch["near_fars"]
    mask = (depth_gt != 0) & (depth_gt >= nf[:, 0, 0:1]) & (depth_gt <= nf[:, 0, 1:2])
    if torch.sum(mask) > 0:
        l1 = torch.nn.functional.l1_loss(depth[mask], depth_gt[mask])
        l2 = torch.nn.functional.l1_loss(depth_2[mask], depth_gt[mask])
    else:
        l1 = l2 = 0.0
    return weight_rgb * (loss_rgb + loss_rgb2) + weight_depth * (l1 + l2)


def run_grad_case(name, c, weight_seed=0, sampler_seed=1):
    model = build_reference_model(weight_seed, test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                                  coarse_sample=c["coarse"], fine_sample=c["fine"], test_n_view=c["NV"],
                                  extract_geometry=False)
    model.train()
    fr = make_frame(c["H"], c["W"], c["NV"], c["seed"], train_layout=True)
    # a ReLU input within ~1e-6 of zero has its derivative decided by the last ulp of the forward evaluation: any two
    # fp32 implementations (the reference on another CPU included) disagree there by the whole contribution of that unit
    idx = clean_ray_indices(model, fr, c["H"], c["W"], c["RN"], sampler_seed, True, relu_margin=5e-6)
    vols = []
    for st in ("stage1", "stage2", "stage3"):
        for k in ("feature_volume", "weight_volume"):
            fr.feature_volume[st][k].requires_grad_(True)
            vols.append((f"{st}.{k}", fr.feature_volume[st][k]))
    torch.manual_seed(sampler_seed)
    r = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat,
                    feature_volume=fr.feature_volume, match_feature=fr.match_feature)
    loss = training_loss(r, fr.batch)
    loss.backward()
    out = {"input_digest": np.float64(frame_digest(fr)), "ray_idx": idx.numpy(), "sampler_seed": np.int64(sampler_seed),
           "weight_seed": np.int64(weight_seed), "loss": loss.detach().numpy()}
    for n in ("rgb", "depth", "rgb_2", "depth_2"):
        out[n] = r[dict(rgb=1, depth=2, rgb_2=8, depth_2=9)[n]].detach().numpy()
    for k, p in model.named_parameters():
        if k.startswith(("ray_transformer.", "deviation_network.")):
            assert p.grad is not None, k
            out["grad." + k] = p.grad.numpy()
    for k, v in vols:
        g = v.grad
        nz = torch.nonzero(g.reshape(-1))[:, 0]
        out[f"grad_idx.{k}"] = nz.numpy().astype(np.int64)           # sparse: flat index + value
        out[f"grad_val.{k}"] = g.reshape(-1)[nz].numpy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: loss {float(loss):.6f}, {len(out)} arrays, {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB")


COSTREG_CASES = {
    # the training step WITH the one producer the reference trains in front of it (model.py:72-87, 517-524): seeded cost
    # volumes -> MVSVolume (cost_reg_2) -> frustums -> infer -> loss; gradients of feature_volume.cost_reg_2.*
    "c5_train_grads_costreg": dict(H=64, W=96, NV=3, seed=5, RN=64, coarse=64, fine=64, costreg_seed=31),
}
COSTREG_SAMPLE = 4096     # tensors beyond 20 000 elements are stored as a strided sample + their sum and squared norm


def costreg_sample_index(n):
    return torch.arange(n) if n <= 20_000 else (torch.arange(COSTREG_SAMPLE) * (n // COSTREG_SAMPLE))


def run_costreg_grad_case(name, c, weight_seed=0, sampler_seed=1):
    from uforecon_amd.scene import fill_state_dict, make_cost_volumes

    model = build_reference_model(weight_seed, test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                                  coarse_sample=c["coarse"], fine_sample=c["fine"], test_n_view=c["NV"],
                                  extract_geometry=False)
    model.train()
    fill_state_dict(model.feature_volume, c["costreg_seed"])
    fr = make_frame(c["H"], c["W"], c["NV"], c["seed"], train_layout=True)
    cost = make_cost_volumes(c["H"], c["W"], c["NV"], c["seed"])

    def build():
        vols = {}
        for st in ("stage1", "stage2", "stage3"):
            f, w = model.feature_volume(fr.batch, cost[st])                      # model.py:517-519: build_mvs_volume
            vols[st] = {"feature_volume": f, "weight_volume": w}
        return vols

    with torch.no_grad():
        fr.feature_volume = build()
    idx = clean_ray_indices(model, fr, c["H"], c["W"], c["RN"], sampler_seed, True, relu_margin=5e-6)
    fr.feature_volume = build()
    torch.manual_seed(sampler_seed)
    r = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat,
                    feature_volume=fr.feature_volume, match_feature=fr.match_feature)
    loss = training_loss(r, fr.batch)
    loss.backward()
    out = {"ray_idx": idx.numpy(), "sampler_seed": np.int64(sampler_seed), "weight_seed": np.int64(weight_seed),
           "costreg_seed": np.int64(c["costreg_seed"]), "loss": loss.detach().numpy()}
    for k, p in model.named_parameters():
        if k.startswith("feature_volume.cost_reg_2."):
            assert p.grad is not None, k
            g = p.grad.reshape(-1)
            out["grad." + k] = g[costreg_sample_index(g.numel())].numpy()
            out["gsum." + k] = np.float64(g.double().sum())
            out["gsq." + k] = np.float64((g.double() ** 2).sum())
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: loss {float(loss):.6f}, {len(out)} arrays, {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB")


def main():
    model = None
    only = sys.argv[1:]
    for name, c in COSTREG_CASES.items():
        if only and name not in only:
            continue
        run_costreg_grad_case(name, c)
    for name, c in GRAD_CASES.items():
        if only and name not in only:
            continue
        run_grad_case(name, c)
    for name, c in CASES.items():
        if only and name not in only:
            continue
        model = run_case(name, c)
    if not only:
        sd = ray_path_state_dict(model)
        np.savez_compressed(os.path.join(HERE, "ray_path_weights_seed0.npz"), **{k: v.numpy() for k, v in sd.items()})
        print("weights:", sum(v.numel() for v in sd.values()), "floats")


if __name__ == "__main__":
    main()
