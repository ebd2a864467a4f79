"""The reference's inference flow on this code base, under the reference's parameter names.

`UFOReconInference` = the per-ray mirror (`uforecon_amd.model.UFORecon`) + the per-frame producers
(`transmvsnet.feature` FeatureNet, `transmvsnet.FMT_with_pathway`, `transmvsnet.cost_regularization`,
`transmvsnet.DepthNet`, `feature_volume.cost_reg_2`) + the reference's dead `pre_conv` parameter: its state_dict has
exactly the reference model's 530 entries (tests/golden/reference_state_dict_shapes.json), so a reference checkpoint
loads with strict=True.  `extract_geometry` follows code1/model.py:760-842 end to end: images -> depth map files.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import cascade, featurenet
from . import model as M


class UFOReconInference(M.UFORecon):
    def __init__(self, args, tune_convolutions: bool = True):
        super().__init__(args)
        if tune_convolutions:
            # the 3-D U-Nets are library convolutions: letting MIOpen search its algorithms once per shape halves them
            # (512x640: cascade 29 -> 19 ms, CostRegNetWeight 32 -> 17 ms); the first frame pays the search
            torch.backends.cudnn.benchmark = True
        self.transmvsnet = cascade.TransMVSNetCascade()
        self.transmvsnet.feature = featurenet.FeatureNet(base_channels=8)        # TransMVSNet.py:152
        self.feature_volume = cascade.MVSVolume(in_channels=1, base_channels=8)   # model.py:64
        self.pre_conv = nn.Conv2d(128, 32, 1, bias=False)                         # dead parameter of the reference
        self.eval()

    def train(self, mode: bool = True):
        """Inference only: the encoder's BatchNorms must run on their running statistics (in training mode they would
        normalise with batch statistics AND overwrite the loaded running_mean / running_var on every frame)."""
        if mode:
            raise M.UfrError("UFOReconInference is inference-only: it stays in eval mode (BatchNorm running statistics)")
        return super().train(False)

    # ---- model.py:139-160
    @staticmethod
    def build_pairs(imgs, proj_mats, depth_values):
        """imgs (B,N,3,H,W), proj_mats {stage: (B,N,2,4,4)}, depth_values (B,D) -> the N view rotations stacked on the batch
        axis: every source view is the reference view of one batch element."""
        N = imgs.shape[1]
        comb = np.array([list(range(i, N)) + list(range(0, i)) for i in range(N)])
        imgs = imgs[:, comb].reshape(-1, N, *imgs.shape[2:])                                   # (B*N, V, 3, H, W)
        pm = {st: proj_mats[st][:, comb].reshape(-1, N, 2, 4, 4) for st in ("stage1", "stage2", "stage3")}
        return imgs, pm, depth_values.expand(imgs.shape[0], -1)

    @torch.no_grad()
    def encode_frame(self, batch):
        """model.py:775-806: everything `infer` needs besides the batch itself.  Returns (source_imgs_feat (B,V,32,h,w),
        feature_volume dict, match_feature list) and sets batch['depth_info']."""
        imgs, pm, dv = self.build_pairs(batch["source_imgs"], batch["proj_matrices"], batch["depth_values_org_scale"])
        H, W = imgs.shape[-2:]
        # TransMVSNet.py:175-178 runs the backbone on view v of every rotation, i.e. on each source image N times; one
        # pass over the N distinct images and a gather by the rotation table gives the same tensors
        B, N = batch["source_imgs"].shape[:2]
        comb = np.array([list(range(i, N)) + list(range(0, i)) for i in range(N)])
        base = self.transmvsnet.feature(batch["source_imgs"].reshape(B * N, *batch["source_imgs"].shape[2:]))
        feats = []
        for v in range(N):
            idx = torch.as_tensor([b * N + comb[r][v] for b in range(B) for r in range(N)], device=imgs.device)
            feats.append({st: base[st][idx] for st in ("stage1", "stage2", "stage3")})
        feats = self.transmvsnet.encode(feats, ref_idx=0)                                         # :181
        volume_info = self.transmvsnet(feats, pm, dv, (H, W))                                     # :183-236
        frustums = {}
        for st in ("stage1", "stage2", "stage3"):                                                 # model.py:795-802
            f, w = self.feature_volume(batch, volume_info[st]["cost_volume"])
            frustums[st] = {"feature_volume": f, "weight_volume": w}
        for f in feats:
            f["stage1"] = f["stage1"][0:1]                                                        # :782-783
        match_feature = self.transmvsnet.get_match_feat(feats, cur_n_src_views=self.args.test_n_view)   # :785
        source_imgs_feat = torch.stack([f["stage1"] for f in feats], dim=1)                       # :787-790
        if getattr(self.args, "mvs_depth_guide", 1) > 0:                                          # :804-806
            batch["depth_info"] = (volume_info["stage3"]["depth"] * batch["scale_factor"]).unsqueeze(0)
        return source_imgs_feat, frustums, match_feature

    @torch.no_grad()
    def extract_geometry(self, batch, batch_idx=0, out_dir=None, uniforms=None):
        """code1/model.py:760-842 (same name and role as the reference's hook; `out_dir` defaults to args.out_dir)."""
        source_imgs_feat, frustums, match_feature = self.encode_frame(batch)
        out_dir = out_dir if out_dir is not None else getattr(self.args, "out_dir", None)
        return M.UFORecon.extract_geometry(self, batch, source_imgs_feat, frustums, match_feature, out_dir=out_dir,
                                           uniforms=uniforms)
