"""Feature backbone (first producer): FeatureNet with its deformable convolutions.

Mirrors, with the reference's class names and state_dict keys,
  Conv2d, FeatureNet     code1/encoder_utils/fmt/module.py:26-62, 388-468
  DCN (DCNv2)            code1/encoder_utils/fmt/dcn.py:15-80
`FeatureNet.forward` runs the whole backbone on HIP kernels over channel-last tensors (`feature_net`, below: the plan):
`ufr_conv2d` (csrc/conv2d.hip: every plain convolution with its BatchNorm / bias / ReLU / lateral addition folded into the
store, the offset-and-mask convolution of a deformable layer writing what the deformable kernel reads) and
`ufr_deform_conv2d_cl` (csrc/dcn.hip) -- `deform_conv2d` is torchvision's operator in the reference, a dependency that does
not exist in this image.  No library convolution, no layout change between layers.  The modules keep the reference's
layer-by-layer `forward`s (library ops + `deform_conv2d`) as the readable statement of what the plan computes; the CPU
test-suite and the GPU parity test (tests/test_gpu_featurenet.py) run them against the plan.
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
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            return feature_net(self, x)
        return self.forward_layers(x)

    def forward_layers(self, x):
        """module.py:443-468 layer by layer (library convolutions + `deform_conv2d`): what `feature_net` computes."""
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


# ---- the execution plan on HIP kernels ------------------------------------------------------------------------------
def _fold(conv: nn.Conv2d, bn):
    """(scale, shift) of `bn(conv(x) + bias)` in eval mode, or of the bias alone."""
    dev = conv.weight.device
    if bn is None:
        return None, (None if conv.bias is None else conv.bias.detach().float().contiguous())
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    if conv.bias is not None:
        shift = shift + conv.bias.detach() * scale
    return scale.float().contiguous().to(dev), shift.float().contiguous().to(dev)


def _plan_params(m: "FeatureNet"):
    """Folded parameters of every layer, cached on the module until a parameter or buffer changes."""
    key = tuple((t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))
    cached = getattr(m, "_ufr_plan", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    P = {}

    def block(name, blk: Conv2d):
        sc, sh = _fold(blk.conv, blk.bn)
        P[name] = dict(w=blk.conv.weight.detach().float().contiguous(), scale=sc, shift=sh, relu=blk.relu,
                       stride=blk.conv.stride[0])

    for seq in ("conv0", "conv1", "conv2"):
        for i, blk in enumerate(getattr(m, seq)):
            block(f"{seq}.{i}", blk)
    for name in ("inner1", "inner2"):
        c = getattr(m, name)
        P[name] = dict(w=c.weight.detach().float().contiguous(), scale=None, shift=c.bias.detach().float().contiguous(),
                       relu=False, stride=1)
    for name in ("out1", "out2", "out3"):
        head = getattr(m, name)                      # [Conv2d, DCN, BN, ReLU, DCN, BN, ReLU, DCN]
        block(f"{name}.0", head[0])
        for i, bn_i in ((1, 2), (4, 5), (7, None)):
            d = head[i]
            om = d.conv_offset_mask
            ent = dict(w_om=om.weight.detach().float().contiguous(), b_om=om.bias.detach().float().contiguous(),
                       w=d.weight.detach().float().contiguous(), bias=None if d.bias is None else d.bias.detach().float().contiguous(),
                       scale=None, shift=None, relu=False)
            if bn_i is not None:
                bn = head[bn_i]
                sc = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                ent.update(scale=sc.float().contiguous(), shift=(bn.bias.detach() - bn.running_mean * sc).float().contiguous(), relu=True)
            P[f"{name}.{i}"] = ent
    m._ufr_plan = (key, P)
    return P


@torch.no_grad()
def feature_net(m: "FeatureNet", x: torch.Tensor):
    """FeatureNet.forward (module.py:443-468) on the HIP kernels: image (B,3,H,W) -> the reference's dict of (B,C,h,w) maps.
    Every intermediate map is channel-last; BatchNorm (eval) / bias / ReLU ride in the producing kernel's store, the FPN's
    `interpolate(intra, 2, 'nearest') + inner(conv)` is the 1x1 convolution's fused skip, a deformable layer is two launches
    (offsets | sigmoid(masks) planar, then the sampler + contraction)."""
    from . import ops
    if m.training:
        raise UfrError("FeatureNet: inference only (BatchNorm in eval mode)")
    if not x.is_cuda:
        raise UfrError("FeatureNet runs on the GPU only (no CPU implementation)")
    B, C, H, W = x.shape
    if C != 3 or H % 4 or W % 4:
        raise UfrError(f"FeatureNet: image {tuple(x.shape)}: 3 channels, extents multiples of 4 (two stride-2 levels)")
    P = _plan_params(m)

    def conv(name, t, skip=None, in_planar=False):
        p = P[name]
        return ops.conv2d(t, p["w"], p["stride"], p["scale"], p["shift"], p["relu"], skip=skip, in_planar=in_planar)

    def dcn(name, t, out_planar=False):
        p = P[name]
        om = ops.conv2d(t, p["w_om"], 1, None, p["b_om"], False, out_planar=True, sigmoid_from=18)
        return ops.deform_conv2d_cl(t, om, p["w"], p["bias"], p["scale"], p["shift"], p["relu"], out_planar=out_planar)

    def head(name, t):
        t = conv(f"{name}.0", t)
        t = dcn(f"{name}.1", t)
        t = dcn(f"{name}.4", t)
        return dcn(f"{name}.7", t, out_planar=True)

    c0 = conv("conv0.1", conv("conv0.0", x.detach().float().contiguous(), in_planar=True))
    c1 = conv("conv1.2", conv("conv1.1", conv("conv1.0", c0)))
    c2 = conv("conv2.2", conv("conv2.1", conv("conv2.0", c1)))
    out = {"stage1": head("out1", c2)}
    intra = conv("inner1", c1, skip=c2)                  # interpolate(conv2, 2x nearest) + inner1(conv1)
    out["stage2"] = head("out2", intra)
    intra = conv("inner2", c0, skip=intra)
    out["stage3"] = head("out3", intra)
    return out
