"""Golden vectors of the frustum cascade by RUNNING THE REFERENCE's modules (CPU, build container only):
``python tests/golden/make_golden_cascade.py`` -> tests/golden/cascade_*.npz.

The reference's own DepthNet / PixelwiseNet / CostRegNet / get_depth_range_samples (code1/encoder_utils/fmt) and MVSVolume
(code1/feature_volume.py) are instantiated under the reference's state_dict names, given deterministic weights
(uforecon_amd.scene.fill_state_dict) and driven through the stage loop of TransMVSNet.forward (TransMVSNet.py:183-236,
re-typed here because the method itself starts with the DCN feature extractor, which needs torchvision) and the frustum
dict of model.py:517-524.  Volumes are stored sub-sampled (every second voxel in H and W) to keep the fixture small.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from ref_harness import import_reference  # noqa: E402
from uforecon_amd.scene import CASCADE_CASES, fill_state_dict, make_cascade_case  # noqa: E402


def build_reference():
    import_reference()
    from code1.encoder_utils.fmt.FMT import FMT_with_pathway
    from code1.encoder_utils.fmt.module import CostRegNet
    from code1.encoder_utils.fmt.TransMVSNet import DepthNet
    from code1.feature_volume import MVSVolume

    class T(nn.Module):
        def __init__(self):
            super().__init__()
            self.FMT_with_pathway = FMT_with_pathway()
            self.cost_regularization = nn.ModuleList([CostRegNet(in_channels=1, base_channels=8) for _ in range(3)])
            self.DepthNet = DepthNet()

    class R(nn.Module):
        def __init__(self):
            super().__init__()
            self.transmvsnet = T()
            self.feature_volume = MVSVolume(in_channels=1, base_channels=8)

    return R().eval()


def run(name):
    c = make_cascade_case(name)
    R = fill_state_dict(build_reference(), c["weight_seed"]).eval()
    from code1.encoder_utils.fmt.module import get_depth_range_samples

    net = R.transmvsnet
    H, W = c["img_hw"]
    depth_values, features, proj_matrices = c["depth_values"], c["features"], c["proj_matrices"]
    ndepths, ratios, scales = [48, 32, 8], [4, 2, 1], [4, 2, 1]
    out = {}
    with torch.no_grad():
        depth_min, depth_max = float(depth_values[0, 0]), float(depth_values[0, -1])
        depth_interval = (depth_max - depth_min) / depth_values.size(1)
        depth, view_weights = None, None
        for k in range(3):                                                    # TransMVSNet.py:186-236
            st = "stage%d" % (k + 1)
            features_stage = [f[st] for f in features]
            if depth is not None:
                cur_depth = F.interpolate(depth.detach().unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
            else:
                cur_depth = depth_values
            samples = get_depth_range_samples(cur_depth=cur_depth, ndepth=ndepths[k], depth_inteval_pixel=ratios[k] * depth_interval,
                                              dtype=torch.float32, device="cpu", shape=[depth_values.shape[0], H, W],
                                              max_depth=depth_max, min_depth=depth_min, use_inverse_depth=False)
            if k > 0:
                view_weights = F.interpolate(view_weights, scale_factor=2, mode="nearest")
            dv = F.interpolate(samples.unsqueeze(1), [ndepths[k], H // scales[k], W // scales[k]], mode="trilinear",
                               align_corners=False).squeeze(1)
            res = net.DepthNet(features_stage, proj_matrices[st], depth_values=dv, num_depth=ndepths[k],
                               cost_regularization=net.cost_regularization[k], view_weights=view_weights)
            if k == 0:
                o, view_weights = res
            else:
                o = res
            wta = torch.argmax(o["prob_volume"], dim=1, keepdim=True).type(torch.long)
            depth = torch.gather(o["depth_values"], 1, wta).squeeze(1)
            feat, weight = R.feature_volume(None, o["cost_volume"])            # model.py:518-524
            out[st + ".depth"] = depth.numpy()
            out[st + ".photometric_confidence"] = o["photometric_confidence"].numpy()
            out[st + ".cost_volume"] = o["cost_volume"][..., ::2, ::2].numpy()
            out[st + ".feature_volume"] = feat[..., ::2, ::2].numpy()
            out[st + ".weight_volume"] = weight[..., ::2, ::2].numpy()
            if k == 0:
                out["view_weights"] = view_weights.numpy()
    np.savez_compressed(os.path.join(HERE, f"cascade_{name}.npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k.endswith("depth") or k.endswith("feature_volume")})


def run_fmt(name):
    """FMT_with_pathway.forward + the lines of TransMVSNet.get_match_feat (TransMVSNet.py:341-375) on the reference's module."""
    from uforecon_amd.scene import make_fmt_case

    c = make_fmt_case(name)
    R = fill_state_dict(build_reference(), c["weight_seed"]).eval()
    F_ = R.transmvsnet.FMT_with_pathway
    out = {}
    with torch.no_grad():
        feats = F_([dict(f) for f in c["features"]], ref_idx=0)
        for v, f in enumerate(feats):
            for st in ("stage1", "stage2", "stage3"):
                out[f"view{v}.{st}"] = f[st].numpy()
        # only stage1 of the first batch element goes on (model.py:782-783), then get_match_feat
        for f in feats:
            f["stage1"] = f["stage1"][0:1]
        od = F_.extract_cross_features(feats)
        n = c["NV"]
        index_lists = [(a, b) for a in range(n - 1) for b in range(a + 1, n)]
        img_feat = [[] for _ in range(n)]
        for k, (i, j) in enumerate(index_lists):
            img_feat[i].append(od["aug_feat0s"][0][:, k])
            img_feat[j].append(od["aug_feat1s"][0][:, k])
        out["match_feature"] = torch.stack([torch.cat(v, dim=1) for v in img_feat], dim=1).numpy()
    np.savez_compressed(os.path.join(HERE, f"fmt_{name}.npz"), **out)
    print("fmt", name, out["match_feature"].shape)


if __name__ == "__main__":
    from uforecon_amd.scene import FMT_CASES

    for n in CASCADE_CASES:
        run(n)
    for n in FMT_CASES:
        run_fmt(n)
