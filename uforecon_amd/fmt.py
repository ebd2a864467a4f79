"""Feature Matching Transformer and the pair-wise matching features.

Mirrors, with the reference's class names, forward signatures and state_dict keys,
  LinearAttention, AttentionLayer, EncoderLayer, FMT, FMT_with_pathway   code1/encoder_utils/fmt/FMT.py:17-316
  PositionEncodingSine                                                  code1/encoder_utils/fmt/position_encoding.py:24-60
  TransMVSNet.get_match_feat                                            code1/encoder_utils/fmt/TransMVSNet.py:341-375
These are 8 encoder layers of width 32 over h*w tokens (8 GFLOP per 512x640 3-view frame): plain library ops
(torch -> rocBLAS / elementwise kernels) are the right tool; there is nothing here worth a hand-written kernel, unlike the
per-ray path, whose transformers see 63 M tokens per frame.  Inference only.  The FeatureNet / DCN backbone that produces
the inputs is not mirrored (it needs torchvision's deformable convolution, absent here, so it could not be pinned).
"""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class LinearAttention(nn.Module):
    """FMT.py:17-39"""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, queries, keys, values):
        Q = F.elu(queries) + 1
        K = F.elu(keys) + 1
        KV = torch.einsum("nshd,nshm->nhmd", K, values)
        Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + self.eps)
        return torch.einsum("nlhd,nhmd,nlh->nlhm", Q, KV, Z).contiguous()


class AttentionLayer(nn.Module):
    """FMT.py:42-79"""

    def __init__(self, attention, d_model, n_heads, d_keys=None, d_values=None):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = attention
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads

    def forward(self, queries, keys, values):
        N, L, _ = queries.shape
        _, S, _ = keys.shape
        H = self.n_heads
        q = self.query_projection(queries).view(N, L, H, -1)
        k = self.key_projection(keys).view(N, S, H, -1)
        v = self.value_projection(values).view(N, S, H, -1)
        return self.out_projection(self.inner_attention(q, k, v).view(N, L, -1))


class EncoderLayer(nn.Module):
    """FMT.py:82-113 (dropout 0)"""

    def __init__(self, d_model, n_heads, d_keys=None, d_values=None, d_ff=None, dropout=0.0, activation="relu"):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        self.attention = AttentionLayer(LinearAttention(), d_model, n_heads, d_keys, d_values)
        d_ff = d_ff or 2 * d_model
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = getattr(F, activation)

    def forward(self, x, source):
        x = x + self.dropout(self.attention(x, source, source))
        y = x = self.norm1(x)
        y = self.dropout(self.activation(self.linear1(y)))
        y = self.dropout(self.linear2(y))
        return self.norm2(x + y)


class PositionEncodingSine(nn.Module):
    """position_encoding.py:24-60 (temp_bug_fix=True); the table is a non-persistent buffer, as in the reference."""

    def __init__(self, d_model, max_shape=(600, 600)):
        super().__init__()
        pe = torch.zeros((d_model, *max_shape))
        y_position = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
        x_position = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
        div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
        pe[0::4] = torch.sin(x_position * div_term)
        pe[1::4] = torch.cos(x_position * div_term)
        pe[2::4] = torch.sin(y_position * div_term)
        pe[3::4] = torch.cos(y_position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0), persistent=False)

    def forward(self, x):
        return x + self.pe[:, :, :x.size(2), :x.size(3)]


def _tokens(x):
    return x.flatten(2).transpose(1, 2)                       # 'n c h w -> n (h w) c'


def _image(t, H):
    n, hw, c = t.shape
    return t.transpose(1, 2).reshape(n, c, H, hw // H)         # 'n (h w) c -> n c h w'


class FMT(nn.Module):
    """FMT.py:116-201: self-attention on the reference view, self + cross on a source view, pair mode for matching."""

    def __init__(self, config):
        super().__init__()
        self.d_model, self.nhead, self.layer_names = config["d_model"], config["nhead"], config["layer_names"]
        layer = EncoderLayer(config["d_model"], config["nhead"])
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(len(self.layer_names))])
        self.pos_encoding = PositionEncodingSine(config["d_model"])

    def forward(self, ref_feature=None, src_feature=None, feat="ref", self_features=None):
        assert ref_feature is not None
        if feat == "ref":
            assert self.d_model == ref_feature.size(1)
            H = ref_feature.shape[2]
            x = _tokens(self.pos_encoding(ref_feature))
            outs = []
            for layer, name in zip(self.layers, self.layer_names):
                if name == "self":
                    x = layer(x, x)
                    outs.append(_image(x, H))
            return outs
        if feat == "src":
            assert self.d_model == ref_feature[0].size(1)
            H = ref_feature[0].shape[2]
            refs = [_tokens(r) for r in ref_feature]
            x = _tokens(self.pos_encoding(src_feature))
            for i, (layer, name) in enumerate(zip(self.layers, self.layer_names)):
                if name == "self":
                    x = layer(x, x)
                elif name == "cross":
                    x = layer(x, refs[i // 2])
                else:
                    raise KeyError(name)
            return _image(x, H)
        if feat == "cross":
            H = ref_feature.shape[2]
            f0, f1 = _tokens(self.pos_encoding(ref_feature)), _tokens(self.pos_encoding(src_feature))
            p1, p2 = torch.cat([f0, f1], dim=0), torch.cat([f1, f0], dim=0)
            for layer, name in zip(self.layers, self.layer_names):
                if name == "self":
                    p1 = layer(p1, p1)
                elif name == "cross":
                    p1 = layer(p1, p2)        # p2 is never updated in the reference (FMT.py:186-193): kept
                else:
                    raise KeyError(name)
            return _image(p1, H), _image(p1, H)   # both returns are pair_feat1, as in the reference (:196)
        raise ValueError("Wrong feature name")


class FMT_with_pathway(nn.Module):
    """FMT.py:204-316"""

    def __init__(self, base_channels=8, FMT_config=None):
        super().__init__()
        FMT_config = FMT_config or {"d_model": 32, "nhead": 8, "layer_names": ["self", "cross"] * 4}
        self.FMT = FMT(FMT_config)
        self.dim_reduction_1 = nn.Conv2d(base_channels * 4, base_channels * 2, 1, bias=False)
        self.dim_reduction_2 = nn.Conv2d(base_channels * 2, base_channels * 1, 1, bias=False)
        self.smooth_1 = nn.Conv2d(base_channels * 2, base_channels * 2, 3, padding=1, bias=False)
        self.smooth_2 = nn.Conv2d(base_channels * 1, base_channels * 1, 3, padding=1, bias=False)

    def _upsample_add(self, x, y):
        _, _, H, W = y.size()
        return F.interpolate(x, size=(H, W), mode="bilinear") + y

    def _pathway(self, f):
        f["stage2"] = self.smooth_1(self._upsample_add(self.dim_reduction_1(f["stage1"]), f["stage2"]))
        f["stage3"] = self.smooth_2(self._upsample_add(self.dim_reduction_2(f["stage2"]), f["stage3"]))

    def forward(self, features, ref_idx=0):
        """features: list over views of {"stage1","stage2","stage3"} backbone maps; updated in place and returned."""
        ref_list = None
        for v, f in enumerate(features):
            if v == ref_idx:
                ref_list = self.FMT(f["stage1"].clone(), feat="ref")
                f["stage1"] = ref_list[-1]
            else:
                f["stage1"] = self.FMT([r.clone() for r in ref_list], f["stage1"].clone(), feat="src")
            self._pathway(f)
        return features

    def extract_pair_feature(self, features, stages=("stage1",)):
        n_views = len(features)
        index_lists = [(a, b) for a in range(n_views - 1) for b in range(a + 1, n_views)]
        f0s, f1s = [], []
        for stage in stages:
            c0 = torch.stack([features[i][stage] for i, _ in index_lists], dim=1)
            c1 = torch.stack([features[j][stage] for _, j in index_lists], dim=1)
            f0s.append(c0.reshape(-1, *c0.shape[-3:]))
            f1s.append(c1.reshape(-1, *c1.shape[-3:]))
        return f0s, f1s

    def extract_cross_features(self, features, ref_idx=0):
        f0s, f1s = self.extract_pair_feature(features)
        batch_size = features[0]["stage1"].shape[0]
        a0, a1 = [], []
        for f0, f1 in zip(f0s, f1s):
            g0, g1 = self.FMT(f0, f1, feat="cross")
            a0.append(g0.reshape(batch_size, g0.shape[0] // batch_size, *g0.shape[-3:]))
            a1.append(g1.reshape(batch_size, g1.shape[0] // batch_size, *g1.shape[-3:]))
        return {"aug_feat0s": a0, "aug_feat1s": a1}


def get_match_feat(fmt_with_pathway: FMT_with_pathway, features, cur_n_src_views=3):
    """TransMVSNet.get_match_feat (TransMVSNet.py:341-375): list over scales of (B, V, 32*(V-1), h, w) -- the
    `match_feature` argument of UFORecon.infer."""
    out = fmt_with_pathway.extract_cross_features(features)
    index_lists = [(a, b) for a in range(cur_n_src_views - 1) for b in range(a + 1, cur_n_src_views)]
    result = []
    for scale_idx in range(len(out["aug_feat0s"])):
        per_view = [[] for _ in range(cur_n_src_views)]
        f0, f1 = out["aug_feat0s"][scale_idx], out["aug_feat1s"][scale_idx]
        for k, (i, j) in enumerate(index_lists):
            per_view[i].append(f0[:, k])
            per_view[j].append(f1[:, k])
        result.append(torch.stack([torch.cat(v, dim=1) for v in per_view], dim=1))
    return result
