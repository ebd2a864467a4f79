"""Correlation-frustum construction, second part: the cascade that turns per-view
feature maps into the three correlation frustums the ray path gathers from.

Mirrors, with the reference's class names, forward signatures and state_dict keys,
  PixelwiseNet, DepthNet            code1/encoder_utils/fmt/TransMVSNet.py:23-121
  the stage loop of TransMVSNet     code1/encoder_utils/fmt/TransMVSNet.py:183-236   (after feature extraction)
  Conv3d, Deconv3d, ConvBnReLU3D, CostRegNet, CostRegNetWeight, get_depth_range_samples
                                    code1/encoder_utils/fmt/module.py:110-223, 469-543, 678-707
  MVSVolume                         code1/feature_volume.py:101-121
  the frustum dict of UFORecon      code1/model.py:517-524
Step 2 of DepthNet.forward (warp + correlate + view weighting) is the fused HIP kernel of csrc/frustum.hip
(`uforecon_amd.frustum.correlate`); the two 3-D U-Nets run on the HIP convolution kernel of csrc/conv3d.hip
(`uforecon_amd.unet3d`); the modules below own the parameters under the reference's names.  Inference only.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fmt, frustum, unet3d

Align_Corners_Range = False      # TransMVSNet.py:21


class Conv3d(nn.Module):
    """Parameters of conv + BatchNorm (+ ReLU) (module.py:110-143): `conv.weight`, `bn.*`.  Executed fused by
    `unet3d.cost_reg_net` (one ufr_conv3d launch per block); the block has no forward of its own."""

    def __init__(self, in_channels, out_channels, stride=1, transposed=False):
        super().__init__()
        conv = nn.ConvTranspose3d if transposed else nn.Conv3d
        extra = dict(output_padding=1) if transposed else {}
        self.conv = conv(in_channels, out_channels, 3, stride=stride, padding=1, bias=False, **extra)
        self.bn = nn.BatchNorm3d(out_channels)


def Deconv3d(in_channels, out_channels):
    """transposed conv (stride 2, output_padding 1) + BatchNorm (+ ReLU) (module.py:152-187)"""
    return Conv3d(in_channels, out_channels, stride=2, transposed=True)


class ConvBnReLU3D(nn.Module):
    """module.py:216-223"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class CostRegNet(nn.Module):
    """3-D U-Net over the aggregated similarity volume -> 1-channel cost volume (module.py:469-500)."""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        ch = [base_channels * f for f in (1, 2, 4, 8)]
        self.conv0 = Conv3d(in_channels, ch[0])
        for lvl in (1, 2, 3):                                    # conv1/conv2, conv3/conv4, conv5/conv6
            setattr(self, f"conv{2 * lvl - 1}", Conv3d(ch[lvl - 1], ch[lvl], stride=2))
            setattr(self, f"conv{2 * lvl}", Conv3d(ch[lvl], ch[lvl]))
        for name, lvl in (("conv7", 3), ("conv9", 2), ("conv11", 1)):
            setattr(self, name, Deconv3d(ch[lvl], ch[lvl - 1]))
        self.prob = nn.Conv3d(ch[0], 1, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return unet3d.cost_reg_net(self, x)


class CostRegNetWeight(nn.Module):
    """Same U-Net shape without normalisation, two heads: 8-channel feature frustum and sigmoid weight frustum
    (module.py:502-543)."""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        ch = [base_channels * f for f in (1, 2, 4, 8)]
        self.conv0 = nn.Conv3d(in_channels, ch[0], 3, padding=1)
        for lvl in (1, 2, 3):
            setattr(self, f"conv{2 * lvl - 1}", nn.Conv3d(ch[lvl - 1], ch[lvl], 3, stride=2, padding=1))
            setattr(self, f"conv{2 * lvl}", nn.Conv3d(ch[lvl], ch[lvl], 3, padding=1))
        for name, lvl in (("conv7", 3), ("conv9", 2), ("conv11", 1)):
            setattr(self, name, nn.ConvTranspose3d(ch[lvl], ch[lvl - 1], 3, stride=2, padding=1, output_padding=1))
        self.features = nn.Conv3d(ch[0], 8, 3, stride=1, padding=1, bias=False)
        self.weights = nn.Conv3d(ch[0], 1, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return unet3d.cost_reg_net_weight(self, x)


class PixelwiseNet(nn.Module):
    """similarity volume -> pixel-wise view weight = max over depth of a sigmoid (TransMVSNet.py:23-41)."""

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBnReLU3D(in_channels=1, out_channels=16, kernel_size=1, stride=1, pad=0)
        self.conv1 = ConvBnReLU3D(in_channels=16, out_channels=8, kernel_size=1, stride=1, pad=0)
        self.conv2 = nn.Conv3d(in_channels=8, out_channels=1, kernel_size=1, stride=1, padding=0)
        self.output = nn.Sigmoid()

    def forward(self, x1):
        x1 = self.conv2(self.conv1(self.conv0(x1))).squeeze(1)
        return torch.max(self.output(x1), dim=1, keepdim=True)[0]


def depth_wta(p, depth_values):
    """module.py:561-565"""
    return torch.gather(depth_values, 1, torch.argmax(p, dim=1, keepdim=True).type(torch.long)).squeeze(1)


class DepthNet(nn.Module):
    """One cascade stage (TransMVSNet.py:44-121).  Same arguments and return convention as the reference."""

    def __init__(self):
        super().__init__()
        self.pixel_wise_net = PixelwiseNet()

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, prob_volume_init=None,
                view_weights=None, mvs_volume_only=False):
        proj_matrices = torch.unbind(proj_matrices, 1)
        assert len(features) == len(proj_matrices), "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth
        ref_feature, src_features = features[0], features[1:]
        ref_proj, src_projs = proj_matrices[0], proj_matrices[1:]
        B = ref_feature.shape[0]
        first_stage = view_weights is None
        sims, new_weights = [], []
        for b in range(B):          # step 2: fused warp x correlate (x view weighting) kernel, one frame at a time
            src = torch.stack([s[b] for s in src_features]).contiguous()
            sim, agg = frustum.correlate(ref_feature[b].contiguous(), src, ref_proj[b], [p[b] for p in src_projs],
                                         depth_values[b].contiguous(), None if first_stage else view_weights[b].contiguous(),
                                         want_similarity=first_stage)
            if first_stage and sim.is_cuda and not self.training and not torch.is_grad_enabled():
                # the weights come out of the similarity itself (TransMVSNet.py:80-97): PixelwiseNet + the weighted aggregate
                # as one pass over the volume (ufr_pixelwise_view_weights)
                vw, agg = frustum.view_weights(self.pixel_wise_net, sim)
                new_weights.append(vw)
            elif first_stage:       # the same, layer by layer (library ops: training, and the statement of what the kernel computes)
                vw = torch.cat([self.pixel_wise_net(sim[i][None, None]) for i in range(sim.shape[0])], dim=1)[0]   # (NS,H,W)
                s_sum = torch.zeros_like(sim[0])
                w_sum = torch.full_like(vw[0], 1e-5)
                for i in range(sim.shape[0]):                        # :86-97, the reference's order
                    s_sum = s_sum + sim[i] * vw[i].unsqueeze(0)
                    w_sum = w_sum + vw[i]
                agg = s_sum / w_sum.unsqueeze(0)
                new_weights.append(vw)
            sims.append(agg)
        similarity = torch.stack(sims).unsqueeze(1)                  # (B,1,D,H,W)
        cost_reg = cost_regularization(similarity)                   # step 3
        if mvs_volume_only:
            return None
        prob_volume_pre = cost_reg.squeeze(1)
        if prob_volume_init is not None:
            prob_volume_pre = prob_volume_pre + prob_volume_init
        prob_volume = torch.exp(F.log_softmax(prob_volume_pre, dim=1))
        depth = depth_wta(prob_volume, depth_values=depth_values)
        photometric_confidence = torch.max(prob_volume, dim=1)[0]
        out = {"depth": depth, "photometric_confidence": photometric_confidence, "prob_volume": prob_volume,
               "depth_values": depth_values, "cost_volume": cost_reg}
        if first_stage:
            return out, torch.stack(new_weights).detach()
        return out


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth=192.0, min_depth=0.0):
    """module.py:678-687"""
    cur_depth_min = cur_depth - ndepth / 2 * depth_inteval_pixel
    cur_depth_max = cur_depth + ndepth / 2 * depth_inteval_pixel
    assert cur_depth.shape == torch.Size(shape)
    new_interval = (cur_depth_max - cur_depth_min) / (ndepth - 1)
    return cur_depth_min.unsqueeze(1) + (torch.arange(0, ndepth, device=cur_depth.device, dtype=cur_depth.dtype)
                                         .reshape(1, -1, 1, 1) * new_interval.unsqueeze(1))


def get_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, device, dtype, shape, max_depth=192.0, min_depth=0.0):
    """module.py:690-707 (use_inverse_depth=False, the only mode the reference calls)."""
    if cur_depth.dim() == 2:
        cur_depth_min, cur_depth_max = cur_depth[:, 0], cur_depth[:, -1]
        new_interval = (cur_depth_max - cur_depth_min) / (ndepth - 1)
        s = cur_depth_min.unsqueeze(1) + (torch.arange(0, ndepth, device=device, dtype=dtype).reshape(1, -1) * new_interval.unsqueeze(1))
        return s.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, shape[1], shape[2])
    return get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth, min_depth)


class TransMVSNetCascade(nn.Module):
    """The part of TransMVSNet after the FeatureNet backbone: FMT_with_pathway (TransMVSNet.py:181), the stage loop
    (:183-236) and get_match_feat (:341-375).  State_dict keys (`FMT_with_pathway.*`, `cost_regularization.{0,1,2}.*`,
    `DepthNet.pixel_wise_net.*`) are the reference's under `transmvsnet.`."""

    def __init__(self, ndepths=(48, 32, 8), depth_interals_ratio=(4, 2, 1), cr_base_chs=(8, 8, 8)):
        super().__init__()
        self.ndepths, self.depth_interals_ratio = list(ndepths), list(depth_interals_ratio)
        self.num_stage = len(ndepths)
        self.stage_scale = [4.0, 2.0, 1.0]
        self.FMT_with_pathway = fmt.FMT_with_pathway()
        self.cost_regularization = nn.ModuleList([CostRegNet(in_channels=1, base_channels=c) for c in cr_base_chs])
        self.DepthNet = DepthNet()

    def encode(self, features_backbone, ref_idx=0):
        """backbone pyramids -> FMT features (TransMVSNet.py:181)"""
        return self.FMT_with_pathway(features_backbone, ref_idx=ref_idx)

    def get_match_feat(self, features, cur_n_src_views=3):
        """TransMVSNet.py:341-375"""
        return fmt.get_match_feat(self.FMT_with_pathway, features, cur_n_src_views)

    def forward(self, features, proj_matrices, depth_values, img_hw):
        """features: list over views of {"stage1".."stage3": (B,C,h,w)} (view 0 = reference); proj_matrices:
        {"stageK": (B,V,2,4,4)}; depth_values (B,D0) initial hypotheses; img_hw = (H, W) of the images."""
        H, W = img_hw
        B = depth_values.shape[0]
        depth_min, depth_max = float(depth_values[0, 0].cpu()), float(depth_values[0, -1].cpu())
        # the projection matrices go to the host ONCE here (the correlate kernel takes the 12 numbers of a relative projection
        # by value: frustum.relative_projections): fetched per stage and frame they were nine device-to-host round trips per
        # frame, each draining the queue in the middle of the cascade
        proj_matrices = {k: v.detach().float().cpu() for k, v in proj_matrices.items()}
        depth_interval = (depth_max - depth_min) / depth_values.size(1)
        outputs, depth, view_weights = {}, None, None
        dev, dt = features[0]["stage1"].device, features[0]["stage1"].dtype
        for k in range(self.num_stage):
            st = "stage{}".format(k + 1)
            features_stage = [f[st] for f in features]
            scale = int(self.stage_scale[k])
            if depth is not None:
                cur_depth = F.interpolate(depth.detach().unsqueeze(1), [H, W], mode="bilinear",
                                          align_corners=Align_Corners_Range).squeeze(1)
            else:
                cur_depth = depth_values
            samples = get_depth_range_samples(cur_depth=cur_depth, ndepth=self.ndepths[k],
                                              depth_inteval_pixel=self.depth_interals_ratio[k] * depth_interval,
                                              dtype=dt, device=dev, shape=[B, H, W], max_depth=depth_max, min_depth=depth_min)
            if k > 0:
                view_weights = F.interpolate(view_weights, scale_factor=2, mode="nearest")
            dv = F.interpolate(samples.unsqueeze(1), [self.ndepths[k], H // scale, W // scale], mode="trilinear",
                               align_corners=Align_Corners_Range).squeeze(1)
            res = self.DepthNet(features_stage, proj_matrices[st], depth_values=dv, num_depth=self.ndepths[k],
                                cost_regularization=self.cost_regularization[k], view_weights=view_weights)
            if k == 0:
                out, view_weights = res
            else:
                out = res
            wta = torch.argmax(out["prob_volume"], dim=1, keepdim=True).type(torch.long)
            depth = torch.gather(out["depth_values"], 1, wta).squeeze(1)
            out["depth"] = depth
            outputs[st] = out
            outputs.update(out)
        return outputs


class MVSVolume(nn.Module):
    """feature_volume.py:101-121"""

    def __init__(self, in_channels=1, base_channels=8):
        super().__init__()
        self.cost_reg_2 = CostRegNetWeight(in_channels=in_channels, base_channels=base_channels)

    def forward(self, batch, volume):
        return self.cost_reg_2(volume)


class FrustumBuilder(nn.Module):
    """features -> the `feature_volume` dict the ray path consumes (model.py:502, 517-524).  Submodule names follow the
    reference (`transmvsnet.*`, `feature_volume.*`), so its checkpoint loads with strict=False."""

    def __init__(self):
        super().__init__()
        self.transmvsnet = TransMVSNetCascade()
        self.feature_volume = MVSVolume(in_channels=1, base_channels=8)

    @torch.no_grad()
    def forward(self, features, proj_matrices, depth_values, img_hw):
        volume_info = self.transmvsnet(features, proj_matrices, depth_values, img_hw)
        out = {}
        for st in ("stage1", "stage2", "stage3"):
            feat, weight = self.feature_volume(None, volume_info[st]["cost_volume"])
            out[st] = {"feature_volume": feat, "weight_volume": weight}
        return out, volume_info
