"""CPU restatement of modulated deformable convolution (DCNv2) as the reference's FeatureNet uses it.

TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in a third-party dependency that is absent from /root/reference and from this
image -- `torchvision.ops.deform_conv2d` (reference requirements.txt: `torchvision`, unpinned, next to torch==1.13.1,
i.e. torchvision 0.14.x), called from code1/encoder_utils/fmt/dcn.py:66-80.  This file restates its published
algorithm (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp: `bilinear_interpolate` + `deformable_im2col` followed by
a GEMM with the flattened weight) and is checked in tests/test_dcn.py by the properties that algorithm must have
(zero offsets + unit mask == F.conv2d; integer offsets == convolution of the shifted image; linearity in the mask;
zero contribution from samples at h <= -1 or h >= H), not against torchvision outputs.
"""
from __future__ import annotations

import torch


def bilinear_zero(inp: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """torchvision's `bilinear_interpolate`: inp (B,C,H,W); y, x (B,Ho,Wo) absolute positions -> (B,C,Ho,Wo).
    Zero outside the open interval (-1, H) x (-1, W); corners outside the image contribute zero."""
    B, C, H, W = inp.shape
    inside = ~((y <= -1) | (y >= H) | (x <= -1) | (x >= W))
    y_low, x_low = torch.floor(y), torch.floor(x)
    y_high, x_high = y_low + 1, x_low + 1
    lh, lw = y - y_low, x - x_low
    hh, hw = 1 - lh, 1 - lw
    flat = inp.reshape(B, C, H * W)

    def corner(yy, xx, ok):
        ok = ok & inside
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *y.shape[1:])
        return v * ok.unsqueeze(1).to(v.dtype)

    v1 = corner(y_low, x_low, (y_low >= 0) & (x_low >= 0))
    v2 = corner(y_low, x_high, (y_low >= 0) & (x_high <= W - 1))
    v3 = corner(y_high, x_low, (y_high <= H - 1) & (x_low >= 0))
    v4 = corner(y_high, x_high, (y_high <= H - 1) & (x_high <= W - 1))
    w1, w2, w3, w4 = (hh * hw).unsqueeze(1), (hh * lw).unsqueeze(1), (lh * hw).unsqueeze(1), (lh * lw).unsqueeze(1)
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    """Same signature as torchvision.ops.deform_conv2d (one offset group, groups = 1).
    offset (B, 2*kh*kw, Ho, Wo): channel 2k = dy, 2k+1 = dx of kernel tap k = i*kw + j; mask (B, kh*kw, Ho, Wo)."""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(padding), pair(dilation)
    B, C, H, W = input.shape
    Cout, Cin, kh, kw = weight.shape
    assert Cin == C and offset.shape[1] == 2 * kh * kw
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    ys = (torch.arange(Ho, dtype=input.dtype) * sh - ph).reshape(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=input.dtype) * sw - pw).reshape(1, 1, Wo)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            val = bilinear_zero(input, ys + i * dh + offset[:, 2 * k], xs + j * dw + offset[:, 2 * k + 1])
            if mask is not None:
                val = val * mask[:, k:k + 1]
            cols.append(val)
    cols = torch.stack(cols, dim=2)                                         # (B, C, kh*kw, Ho, Wo): im2col columns
    out = torch.einsum("ock,bckhw->bohw", weight.reshape(Cout, C, kh * kw), cols)
    return out if bias is None else out + bias.reshape(1, -1, 1, 1)
