"""Feature-matching transformer stage of the per-frame encoder and the pair-wise matching features it feeds to the ray
path (out of the per-ray hot path; 8 GFLOP per 512x640 3-view frame).  On the GPU every layer is the HIP kernel pair of csrc/fmt.hip (`ufr_fmt_layer`).

What the reference computes (code1/encoder_utils/fmt/FMT.py:17-316, position_encoding.py:24-60,
TransMVSNet.py:341-375), restated here around ONE token engine instead of a module per concept:

* a stack of eight width-32 post-norm layers, alternately "self" and "cross", each = linear attention (feature map
  elu(.)+1, 8 heads of 4) + out-projection + residual, LayerNorm, a 32-64-32 ReLU MLP + residual, LayerNorm;
* three ways of walking that stack: the reference view (self layers only, every intermediate kept), a source view (cross
  layers attend to the reference view's intermediate of the same depth), and "pair" mode for the matching features;
* a top-down pathway that pushes the 1/4-resolution result into the 1/2 and full resolution backbone maps.

Only the PARAMETER TREE follows the reference (a checkpoint must load: ``FMT.layers.<i>.attention.query_projection.weight``
...); the computation lives in the functions below: the three projections of a self layer are one fused GEMM, attention is
a per-sample 4 x 4 state per head (csrc/fmt.hip), and the three walks are one loop over a per-layer plan.
Inference only.
"""
from __future__ import annotations

import functools
import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

HEADS = 8


# ------------------------------------------------------------------ parameter tree (state_dict keys of the reference)
class _Projections(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.query_projection = nn.Linear(d, d)
        self.key_projection = nn.Linear(d, d)
        self.value_projection = nn.Linear(d, d)
        self.out_projection = nn.Linear(d, d)


class _LayerParams(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.attention = _Projections(d)
        self.linear1 = nn.Linear(d, 2 * d)
        self.linear2 = nn.Linear(2 * d, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


# ------------------------------------------------------------------ token engine
def _layer(p: _LayerParams, x: torch.Tensor, src: Optional[torch.Tensor]) -> torch.Tensor:
    """One post-norm layer (FMT.py:99-113, dropout 0) = the HIP kernel pair behind ``ufr_fmt_layer`` (csrc/fmt.hip).
    ``src is None``: self-attention.  GPU only, like every other module of the package: a CPU tensor or a missing library
    raises (the torch expression of the same layer that the kernel is checked against lives in ``oracle/fmt_oracle.py``;
    the CPU test-suite swaps it in)."""
    from . import ops

    a = p.attention
    return ops.fmt_layer([a.query_projection.weight, a.query_projection.bias, a.key_projection.weight,
                          a.key_projection.bias, a.value_projection.weight, a.value_projection.bias,
                          a.out_projection.weight, a.out_projection.bias, p.linear1.weight, p.linear1.bias,
                          p.linear2.weight, p.linear2.bias, p.norm1.weight, p.norm1.bias, p.norm2.weight, p.norm2.bias],
                         x, src)


def _to_tokens(img: torch.Tensor) -> torch.Tensor:
    return img.flatten(2).transpose(1, 2)                          # (N,C,H,W) -> (N,HW,C)


def _to_image(tok: torch.Tensor, height: int) -> torch.Tensor:
    n, t, c = tok.shape
    return tok.transpose(1, 2).reshape(n, c, height, t // height)


@functools.lru_cache(maxsize=8)
def _sine_table(d_model: int, height: int, width: int, device, dtype) -> torch.Tensor:
    """2-D sinusoidal position code of position_encoding.py:33-50 (the reference's `temp_bug_fix` variant: frequencies
    exp(-2i ln(1e4) / (d/2))), positions counted from 1; channels cycle [sin x, cos x, sin y, cos y].  Depends on the map
    size only: built once per (size, device) on the host like upstream (same values bit for bit) and kept on the device --
    rebuilding it in every call cost 20 ms of host time per frame, five times the FMT's kernels."""
    ys = torch.arange(1, height + 1, dtype=torch.float32)[:, None].expand(height, width)
    xs = torch.arange(1, width + 1, dtype=torch.float32)[None, :].expand(height, width)
    freq = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    table = torch.empty(d_model, height, width)
    table[0::4], table[1::4] = torch.sin(xs * freq), torch.cos(xs * freq)
    table[2::4], table[3::4] = torch.sin(ys * freq), torch.cos(ys * freq)
    return table.to(device=device, dtype=dtype)[None]


class FMT(nn.Module):
    """Owner of the eight layers (`layers.<i>.*` keys).  ``walk`` is the one loop behind the reference's three modes."""

    def __init__(self, config):
        super().__init__()
        self.d_model, self.nhead, self.layer_names = config["d_model"], config["nhead"], list(config["layer_names"])
        if self.nhead != HEADS or any(n not in ("self", "cross") for n in self.layer_names):
            raise ValueError(f"unsupported FMT configuration {config}")
        self.layers = nn.ModuleList([_LayerParams(self.d_model) for _ in self.layer_names])

    def _embed(self, img: torch.Tensor) -> torch.Tensor:
        return _to_tokens(img + _sine_table(self.d_model, img.shape[2], img.shape[3], img.device, img.dtype))

    def walk(self, x: torch.Tensor, cross_sources: Optional[Sequence[torch.Tensor]], skip_cross: bool = False,
             keep: bool = False) -> List[torch.Tensor]:
        """Run tokens ``x`` down the stack.  ``cross_sources[i]`` feeds cross layer i (tokens); ``skip_cross`` drops the
        cross layers (reference-view walk, FMT.py:141-146); ``keep`` returns the output of every self layer."""
        kept = []
        for i, (p, name) in enumerate(zip(self.layers, self.layer_names)):
            if name == "self":
                x = _layer(p, x, None)
                if keep:
                    kept.append(x)
            elif not skip_cross:
                x = _layer(p, x, cross_sources[i])
        return kept if keep else [x]

    # the reference's call convention, kept for callers that use it
    def forward(self, ref_feature=None, src_feature=None, feat="ref", self_features=None):
        if feat == "ref":                                             # FMT.py:137-150
            h = ref_feature.shape[2]
            return [_to_image(t, h) for t in self.walk(self._embed(ref_feature), None, skip_cross=True, keep=True)]
        if feat == "src":                                             # FMT.py:152-170: cross layer i <- ref_feature[i // 2]
            h = src_feature.shape[2]
            refs = [_to_tokens(r) for r in ref_feature]
            sources = [refs[i // 2] if n == "cross" else None for i, n in enumerate(self.layer_names)]
            return _to_image(self.walk(self._embed(src_feature), sources)[0], h)
        if feat == "cross":                                           # FMT.py:172-197
            h = ref_feature.shape[2]
            a, b = self._embed(ref_feature), self._embed(src_feature)
            # pair mode stacks (a, b) on the batch axis and lets every cross layer attend to the SWAPPED stack (b, a) of
            # the *embedded inputs*: the reference never updates the second stack (FMT.py:186-193) and returns the first
            # one twice (:196).  Both quirks are part of what `match_feature` means.
            first, second = torch.cat([a, b], 0), torch.cat([b, a], 0)
            out = _to_image(self.walk(first, [second] * len(self.layers))[0], h)
            return out, out
        raise ValueError("Wrong feature name")


class FMT_with_pathway(nn.Module):
    """FMT + the top-down pathway (`dim_reduction_*`, `smooth_*` keys; FMT.py:204-255)."""

    def __init__(self, base_channels=8, FMT_config=None):
        super().__init__()
        self.FMT = FMT(FMT_config or {"d_model": 32, "nhead": 8, "layer_names": ["self", "cross"] * 4})
        c = base_channels
        self.dim_reduction_1 = nn.Conv2d(4 * c, 2 * c, 1, bias=False)
        self.dim_reduction_2 = nn.Conv2d(2 * c, c, 1, bias=False)
        self.smooth_1 = nn.Conv2d(2 * c, 2 * c, 3, padding=1, bias=False)
        self.smooth_2 = nn.Conv2d(c, c, 3, padding=1, bias=False)

    def _push_down(self, coarse, fine, reduce, smooth):
        up = F.interpolate(reduce(coarse), size=fine.shape[-2:], mode="bilinear")
        return smooth(up + fine)

    def _pathway_hip(self, s1, f2, f3):
        """The two `_push_down` steps of one view on HIP kernels (ufr_conv2d, ufr_upsample_add): 1x1 reduction, bilinear 2x +
        the backbone's map, 3x3 smoothing -- channel-last in between, the reference's (B,C,H,W) maps out.  Needs the levels to
        be exact factors of two apart (they are: the backbone's stages)."""
        from . import ops
        w = lambda c: c.weight.detach().float().contiguous()
        s1_cl = s1.detach().float().permute(0, 2, 3, 1).contiguous()
        t2 = ops.upsample_add(ops.conv2d(s1_cl, w(self.dim_reduction_1)), f2.detach().float().contiguous())
        s2_cl = ops.conv2d(t2, w(self.smooth_1))
        t3 = ops.upsample_add(ops.conv2d(s2_cl, w(self.dim_reduction_2)), f3.detach().float().contiguous())
        return s2_cl.permute(0, 3, 1, 2).contiguous(), ops.conv2d(t3, w(self.smooth_2), out_planar=True)

    def forward(self, features, ref_idx=0):
        """``features``: per view {"stage1","stage2","stage3"} backbone maps (updated in place, FMT.py:237-255).  The
        reference view goes first because the source views attend to its intermediates."""
        order = [ref_idx] + [v for v in range(len(features)) if v != ref_idx]
        ref_levels = None
        for v in order:
            f = features[v]
            if v == ref_idx:
                ref_levels = self.FMT(f["stage1"], feat="ref")
                f["stage1"] = ref_levels[-1]
            else:
                f["stage1"] = self.FMT(ref_levels, f["stage1"], feat="src")
            s1, f2, f3 = f["stage1"], f["stage2"], f["stage3"]
            if (s1.is_cuda and not torch.is_grad_enabled() and f2.shape[-2:] == (2 * s1.shape[-2], 2 * s1.shape[-1])
                    and f3.shape[-2:] == (2 * f2.shape[-2], 2 * f2.shape[-1]) and self.dim_reduction_1.weight.shape[:2] == (16, 32)):
                f["stage2"], f["stage3"] = self._pathway_hip(s1, f2, f3)
            else:       # layer by layer (library ops): the statement of what the kernels compute; other channel counts
                f["stage2"] = self._push_down(s1, f2, self.dim_reduction_1, self.smooth_1)
                f["stage3"] = self._push_down(f["stage2"], f3, self.dim_reduction_2, self.smooth_2)
        return features

    def pair_features(self, features):
        """Pair-mode output for every view pair (a < b) of the 1/4-resolution maps: (B, n_pairs, 32, h, w), twice
        (FMT.py:257-315 `extract_cross_features`: 'aug_feat0s' and 'aug_feat1s' are the same tensor upstream)."""
        n = len(features)
        pairs = [(a, b) for a in range(n - 1) for b in range(a + 1, n)]
        batch = features[0]["stage1"].shape[0]
        first = torch.stack([features[a]["stage1"] for a, _ in pairs], 1).flatten(0, 1)
        second = torch.stack([features[b]["stage1"] for _, b in pairs], 1).flatten(0, 1)
        out, _ = self.FMT(first, second, feat="cross")
        return out.reshape(batch, out.shape[0] // batch, *out.shape[1:]), pairs


def get_match_feat(fmt_with_pathway: FMT_with_pathway, features, cur_n_src_views=3):
    """The `match_feature` argument of UFORecon.infer: [ (B, V, 32*(V-1), h, w) ] (TransMVSNet.py:341-375).

    Pair mode doubles the batch ((a, b) stacked); reshaped to (B, 2*n_pairs, ...) the reference then reads entry k < n_pairs
    for BOTH views of pair k -- so both receive view a's cross-attended map.  View v's channels are
    the pairs it takes part in, in pair order."""
    pair_maps, pairs = fmt_with_pathway.pair_features(features)
    per_view = [[] for _ in range(cur_n_src_views)]
    for k, (a, b) in enumerate(pairs):
        if a < cur_n_src_views and b < cur_n_src_views:
            per_view[a].append(pair_maps[:, k])
            per_view[b].append(pair_maps[:, k])
    return [torch.stack([torch.cat(chunks, dim=1) for chunks in per_view], dim=1)]
