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
    # training layout (s_idx=1, no near/far division), forward only
    "c5_train_fwd": dict(H=64, W=96, NV=3, seed=5, RN=64, coarse=64, fine=64, train=True),
}

# reference autograd of the training loss (model.py:552-566) through infer(extract_geometry=False): gradients of every
# per-ray parameter and of the six sampled volumes -- BASELINE.json configs[4]
GRAD_CASES = {
    "c5_train_grads": dict(H=64, W=96, NV=3, seed=5, RN=64, coarse=64, fine=64),
    "c5_train_grads_nv4": dict(H=32, W=48, NV=4, seed=6, RN=24, coarse=32, fine=32),
}


def ray_indices(H, W, RN):
    step = max(1, (H * W) // RN)
    return (torch.arange(RN) * step + step // 3).clamp(max=H * W - 1)[None]


def run_case(name, c, weight_seed=0, sampler_seed=1):
    train = c.get("train", False)
    model = build_reference_model(weight_seed, test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                                  coarse_sample=c["coarse"], fine_sample=c["fine"],
                                  test_coarse_only=c.get("coarse_only", False), test_n_view=c["NV"],
                                  extract_geometry=not train)
    fr = make_frame(c["H"], c["W"], c["NV"], c["seed"], train_layout=train)
    idx = ray_indices(c["H"], c["W"], c["RN"])
    out = {"input_digest": np.float64(frame_digest(fr)), "ray_idx": idx.numpy(),
           "sampler_seed": np.int64(sampler_seed), "weight_seed": np.int64(weight_seed)}
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
            if n.startswith("points_in_pixel"):
                continue
            out[n] = v.detach().numpy()
    else:
        srdf, pts, depth, rgb = r
        out.update(srdf=srdf[0].numpy(), points=pts[0].numpy(), depth=depth[0].numpy(), rgb=rgb[0].numpy())
    for k, v in cap.items():
        out[k] = v.numpy()
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
    idx = ray_indices(c["H"], c["W"], c["RN"])
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


def main():
    model = None
    only = sys.argv[1:]
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
