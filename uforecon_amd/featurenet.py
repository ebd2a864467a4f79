"""Feature backbone (first producer): FeatureNet with its deformable convolutions.

Mirrors, with the reference's class names and state_dict keys,
  Conv2d, FeatureNet     code1/encoder_utils/fmt/module.py:26-62, 388-468
  DCN (DCNv2)            code1/encoder_utils/fmt/dcn.py:15-80
The plain convolutions / batch norms are library ops; `deform_conv2d` -- torchvision's operator in the reference, a
dependency that does not exist in this image -- is the HIP kernel of csrc/dcn.hip behind `ufr_deform_conv2d`.
No CPU fallback: CPU tensors raise UfrError.  Inference only.  Parity of the deformable part is UNPINNED (no torchvision
to produce reference outputs): see oracle/dcn_oracle.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .ops import UfrError, _dev, _opt, _stream


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    """torchvision.ops.deform_conv2d's signature, restricted to what FeatureNet uses (3x3, stride 1, padding 1,
    dilation 1, one offset group)."""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    if pair(stride) != (1, 1) or pair(padding) != (1, 1) or pair(dilation) != (1, 1) or tuple(weight.shape[2:]) != (3, 3):
        raise UfrError("deform_conv2d: only 3x3, stride 1, padding 1, dilation 1 is built (all the reference uses)")
    lib = _lib.load()
    B, C, H, W = input.shape
    Cout = weight.shape[0]
    if offset.shape != (B, 18, H, W) or (mask is not None and mask.shape != (B, 9, H, W)) or weight.shape[1] != C:
        raise UfrError("deform_conv2d: inconsistent shapes")
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=input.device)
    nbytes = lib.ufr_deform_conv2d_workspace_bytes(B, C, H, W)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=input.device)
    _lib.check(lib.ufr_deform_conv2d(_dev(input.contiguous(), "input"), _dev(offset.contiguous(), "offset"),
                                     _opt(None if mask is None else mask.contiguous(), "mask"),
                                     _dev(weight.contiguous(), "weight"), _opt(bias, "bias"), out.data_ptr(), B, C, Cout, H, W,
                                     ws.data_ptr(), nbytes, _stream()), "ufr_deform_conv2d")
    return out


class Conv2d(nn.Module):
    """conv + BatchNorm + ReLU (module.py:26-62)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x, inplace=True) if self.relu else x


class DCN(nn.Module):
    """Modulated deformable 3x3 convolution whose offsets and masks come from a plain convolution of the same input
    (dcn.py:15-80).  Parameters: weight, bias, conv_offset_mask.{weight,bias} -- the reference's names."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1, bias=True):
        super().__init__()
        if (kernel_size, stride, padding, dilation, deformable_groups) != (3, 1, 1, 1, 1):
            raise UfrError("DCN: only kernel 3, stride 1, padding 1, dilation 1, one deformable group is built")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset_mask = nn.Conv2d(in_channels, 27, kernel_size=3, stride=1, padding=1, bias=True)
        stdv = 1.0 / (in_channels * 9) ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        out = self.conv_offset_mask(input)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)                       # dcn.py:68-69, consumed as (dy, dx) pairs per tap
        return deform_conv2d(input, offset, self.weight, self.bias, 1, 1, 1, mask=torch.sigmoid(mask))


class FeatureNet(nn.Module):
    """FPN backbone: image (B,3,H,W) -> {"stage1": (B,32,H/4,W/4), "stage2": (B,16,H/2,W/2), "stage3": (B,8,H,W)}
    (module.py:388-468)."""

    def __init__(self, base_channels=8):
        super().__init__()
        b = base_channels
        self.base_channels = b
        self.conv0 = nn.Sequential(Conv2d(3, b, 3, 1, padding=1), Conv2d(b, b, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(b, b * 2, 5, stride=2, padding=2), Conv2d(b * 2, b * 2, 3, 1, padding=1),
                                   Conv2d(b * 2, b * 2, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(b * 2, b * 4, 5, stride=2, padding=2), Conv2d(b * 4, b * 4, 3, 1, padding=1),
                                   Conv2d(b * 4, b * 4, 3, 1, padding=1))
        f = b * 4

        def head(first, out_ch):
            return nn.Sequential(first, DCN(f, f, 3, 1, 1), nn.BatchNorm2d(f), nn.ReLU(inplace=True),
                                 DCN(f, f, 3, 1, 1), nn.BatchNorm2d(f), nn.ReLU(inplace=True), DCN(f, out_ch, 3, 1, 1))

        self.out1 = head(Conv2d(f, f, 1), f)
        self.inner1 = nn.Conv2d(b * 2, f, 1, bias=True)
        self.inner2 = nn.Conv2d(b, f, 1, bias=True)
        self.out2 = head(Conv2d(f, f, 3, 1, padding=1), b * 2)
        self.out3 = head(Conv2d(f, f, 3, 1, padding=1), b)
        self.out_channels = [4 * b, b * 2, b]

    def forward(self, x):
        conv0 = self.conv0(x)
        conv1 = self.conv1(conv0)
        conv2 = self.conv2(conv1)
        intra = conv2
        outputs = {"stage1": self.out1(intra)}
        intra = F.interpolate(intra, scale_factor=2, mode="nearest") + self.inner1(conv1)
        outputs["stage2"] = self.out2(intra)
        intra = F.interpolate(intra, scale_factor=2, mode="nearest") + self.inner2(conv0)
        outputs["stage3"] = self.out3(intra)
        return outputs
