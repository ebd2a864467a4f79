"""The two 3-D U-Nets of the frustum construction on the HIP convolution kernel (csrc/conv3d.hip, `ufr_conv3d`):

  cost_reg_net         CostRegNet.forward        code1/encoder_utils/fmt/module.py:469-500
  cost_reg_net_weight  CostRegNetWeight.forward  code1/encoder_utils/fmt/module.py:502-543 (via MVSVolume, feature_volume.py:114-121)

The modules in `uforecon_amd.cascade` only own the parameters (reference names / state_dict keys); this file is the
execution plan: which layer reads which, what is fused where.  Volumes stay channel-last between layers; the network
inputs (B,1,D,H,W) and the 1-channel cost volume are both layouts at once, the two heads of CostRegNetWeight come out of
one pass in the reference's (B,C,D,H,W) layout.  GPU only -- a CPU tensor raises (the CPU test-suite swaps these two
functions for torch expressions, `oracle/cascade_oracle.py`).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import UfrError

S1, S2, T2 = ops.CONV3D_S1, ops.CONV3D_S2, ops.CONV3D_T2


def _bn_fold(bn: torch.nn.BatchNorm3d):
    """Eval-mode BatchNorm as y = x * scale + shift."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias.detach() - bn.running_mean * scale).contiguous()


def _input_cl(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise UfrError("the frustum U-Nets run on the GPU only (no CPU implementation)")
    B, C, D, H, W = x.shape
    if C != 1:
        raise UfrError(f"the frustum U-Nets take a 1-channel volume (got {C})")
    if D % 8 or H % 8 or W % 8:
        raise UfrError(f"volume {D}x{H}x{W}: every extent must be a multiple of 8 (three stride-2 levels), as upstream")
    return x.detach().float().contiguous().view(B, D, H, W, 1)


def _unet(x_cl, layer):
    """The shared body: three stride-2 levels down, three transposed convolutions up with skip additions.
    `layer(name, x, mode, skip)` runs one named layer."""
    c0 = layer("conv0", x_cl, S1, None)
    c2 = layer("conv2", layer("conv1", c0, S2, None), S1, None)
    c4 = layer("conv4", layer("conv3", c2, S2, None), S1, None)
    x = layer("conv6", layer("conv5", c4, S2, None), S1, None)
    x = layer("conv7", x, T2, c4)      # conv4 + conv7(x): the addition rides in the kernel's store
    x = layer("conv9", x, T2, c2)
    return layer("conv11", x, T2, c0)


@torch.no_grad()
def cost_reg_net(m, x: torch.Tensor) -> torch.Tensor:
    """(B,1,D,H,W) similarity volume -> (B,1,D,H,W) cost volume.  Every inner layer = convolution + BatchNorm (eval mode,
    folded to one fma after the sum) + ReLU in one kernel."""
    if m.training:
        raise UfrError("CostRegNet: inference only (BatchNorm in eval mode)")

    def layer(name, t, mode, skip):
        blk = getattr(m, name)
        scale, shift = _bn_fold(blk.bn)
        return ops.conv3d(t, blk.conv.weight, mode, bn_scale=scale, bn_shift=shift, relu=True, skip=skip)

    x = _unet(_input_cl(x), layer)
    return ops.conv3d(x, m.prob.weight, S1, out_ncdhw=True)


def _needs_grad(m, x: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m.parameters()))


def _cost_reg_net_weight_trainable(m, x: torch.Tensor):
    """The same network through the module's own layers (library convolutions on the GPU): differentiable, which the
    forward-only `ufr_conv3d` plan is not.  `feature_volume.cost_reg_2` is the one producer the reference trains
    (model.py:75-83: everything but `transmvsnet.*`); its gradients arrive through the frustum scatter of
    `ufr_project_gather_bwd`, so this path must keep the autograd graph (module.py:530-543)."""
    if not x.is_cuda:
        raise UfrError("the frustum U-Nets run on the GPU only (no CPU implementation)")
    c0 = m.conv0(x)
    c2 = m.conv2(m.conv1(c0))
    c4 = m.conv4(m.conv3(c2))
    y = m.conv6(m.conv5(c4))
    y = c4 + m.conv7(y)
    y = c2 + m.conv9(y)
    y = c0 + m.conv11(y)
    return m.features(y), torch.sigmoid(m.weights(y))


def cost_reg_net_weight(m, x: torch.Tensor):
    """(B,1,D,H,W) cost volume -> feature frustum (B,8,D,H,W), weight frustum (B,1,D,H,W) = sigmoid.  Plain convolutions with
    bias, no activation between them (as upstream); the two heads share one pass over the last feature map.
    When a gradient is wanted (grad mode on and the input or a parameter requires grad) the differentiable
    library-convolution expression runs instead: the HIP plan is forward-only and would silently cut the graph."""
    if _needs_grad(m, x):
        return _cost_reg_net_weight_trainable(m, x)
    with torch.no_grad():
        return _cost_reg_net_weight_hip(m, x)


def _cost_reg_net_weight_hip(m, x: torch.Tensor):
    def layer(name, t, mode, skip):
        conv = getattr(m, name)
        return ops.conv3d(t, conv.weight, mode, bias=conv.bias, skip=skip)

    x = _unet(_input_cl(x), layer)
    return ops.conv3d(x, m.features.weight, S1, out_ncdhw=True, weight2=m.weights.weight)
