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
    """Eval-mode BatchNorm as y = x * scale + shift -- cached on the module until one of its tensors changes (five tiny
    launches per layer and call otherwise: 150 per frame, 1 ms of an encode_frame with the GPU idle in between)."""
    key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cached = getattr(bn, "_ufr_fold", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
    scale = scale.contiguous()
    bn._ufr_fold = (key, scale, shift)
    return scale, shift


def _input_cl(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise UfrError("the frustum U-Nets run on the GPU only (no CPU implementation)")
    B, C, D, H, W = x.shape
    if C != 1:
        raise UfrError(f"the frustum U-Nets take a 1-channel volume (got {C})")
    if D % 8 or H % 8 or W % 8:
        raise UfrError(f"volume {D}x{H}x{W}: every extent must be a multiple of 8 (three stride-2 levels), as upstream")
    return x.detach().float().contiguous().view(B, D, H, W, 1)


class _Bounds:
    """max |t| of the activations that feed a plane kernel, as one-element device tensors: a plane layer hands its output's
    bound on (ufr_conv3d_planes raises it in its store), so only tensors that come from an fp32 kernel cost a pass
    (ufr_absmax: 251 MB at full resolution = 0.07 ms; six of them per U-Net before the bounds were chained)."""

    def __init__(self):
        self._b = {}
        self._pool, self._next = None, 0

    def slot(self, like: torch.Tensor):
        """A zeroed one-element device tensor for a layer's out_absmax: slices of ONE buffer zeroed once per U-Net call (a
        `torch.zeros(1)` per layer was 60 fill launches per training step)."""
        if self._pool is None or self._next >= self._pool.numel():
            self._pool, self._next = torch.zeros(64, dtype=torch.float32, device=like.device), 0
        self._next += 1
        return self._pool[self._next - 1:self._next]

    def of(self, t):
        hit = self._b.get(id(t))
        if hit is not None and hit[0] is t:
            return hit[1]
        return ops.absmax(t)

    def put(self, t, bound):
        if bound is not None:
            self._b[id(t)] = (t, bound)


def _s1(t, weight, planes: bool, bounds: "_Bounds" = None, **kw):
    """One stride-1 layer: on the 16-bit matrix cores (ufr_conv3d_planes: fp16 plane products, fp32 accumulate, the input
    brick staged through LDS -- 2 .. 5 x the fp32 kernels, csrc/conv3d_planes.hip) when `planes` and the kernel family has
    the layer, else the fp32 kernel.  The planes' scale comes from the input's |max| (`_Bounds`).  Returns what ops.conv3d
    returns."""
    cin = t.shape[-1]
    flip = kw.pop("flip", False)
    w2 = kw.get("weight2")
    cout = weight.shape[1] if flip else weight.shape[0]
    if planes and ops.conv3d_planes_supported(cin, cout, 0 if w2 is None else w2.shape[0], S1):
        b = bounds if bounds is not None else _Bounds()
        r = ops.conv3d_planes(t, b.of(t), weight, flip=flip, want_absmax=(b.slot(t) if bounds is not None and not kw.get("out_ncdhw") else False), **kw)
        if w2 is not None:
            return r[0], r[1]
        b.put(r[0], r[1])
        return r[0]
    if flip:
        return ops.conv3d_bwd_data(t, weight, S1, (*t.shape[:4], cout), accumulate=kw.get("skip"))
    if planes and bounds is not None and cout <= 16 and w2 is None and not kw.get("out_ncdhw"):
        # an fp32-kernel layer in front of plane layers (conv0): its store takes the bound the next layer wants
        out, omax = ops.conv3d(t, weight, S1, want_absmax=bounds.slot(t), **kw)
        bounds.put(out, omax)
        return out
    return ops.conv3d(t, weight, S1, **kw)


def _s2(t, weight, planes: bool, bounds: "_Bounds" = None, **kw):
    """One stride-2 layer (conv1 / conv3 / conv5), the same way.  ``weight`` (cout, cin, 3,3,3)."""
    if planes and ops.conv3d_planes_supported(t.shape[-1], weight.shape[0], 0, S2):
        b = bounds if bounds is not None else _Bounds()
        out, omax = ops.conv3d_planes(t, b.of(t), weight, want_absmax=(b.slot(t) if bounds is not None else False), mode=S2, **kw)
        b.put(out, omax)
        return out
    return ops.conv3d(t, weight, S2, **kw)


def _t2(t, weight, planes: bool, bounds: "_Bounds" = None, **kw):
    """One transposed stride-2 layer (conv7 / conv9 / conv11); ``weight`` (cin, cout, 3,3,3)."""
    if planes and ops.conv3d_planes_supported(t.shape[-1], weight.shape[1], 0, T2):
        b = bounds if bounds is not None else _Bounds()
        out, omax = ops.conv3d_planes(t, b.of(t), weight, want_absmax=(b.slot(t) if bounds is not None else False), mode=T2, **kw)
        b.put(out, omax)
        return out
    return ops.conv3d(t, weight, T2, **kw)


def _layer(t, weight, mode, planes: bool, bounds: "_Bounds" = None, **kw):
    if mode == S1:
        return _s1(t, weight, planes, bounds, **kw)
    if mode == S2:
        return _s2(t, weight, planes, bounds, **kw)
    return _t2(t, weight, planes, bounds, **kw)


def _s2_bwd_data(d_out, weight, in_shape, accumulate=None, bounds: "_Bounds" = None):
    """Data gradient of a stride-2 layer = the transposed stride-2 convolution of d_out with the layer's forward weight
    (cout, cin, 3,3,3) read as a transposed-convolution weight (its layout as it stands)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if ops.conv3d_planes_supported(cout, cin, 0, T2) and tuple(in_shape[1:4]) == tuple(2 * n for n in d_out.shape[1:4]):
        b = bounds if bounds is not None else _Bounds()
        out, omax = ops.conv3d_planes(d_out, b.of(d_out), weight, skip=accumulate, want_absmax=(b.slot(d_out) if bounds is not None else False), mode=T2)
        b.put(out, omax)
        return out
    return ops.conv3d_bwd_data(d_out, weight, S2, in_shape, accumulate=accumulate)


def _t2_bwd_data(d_out, weight, in_shape, accumulate=None, bounds: "_Bounds" = None):
    """Data gradient of a transposed stride-2 layer = the stride-2 convolution of d_out with the layer's forward weight
    (cin, cout, 3,3,3) read as a convolution weight (rows = cin): on the plane kernels where they have the shape."""
    cin, cout = weight.shape[0], weight.shape[1]
    if ops.conv3d_planes_supported(cout, cin, 0, S2):
        b = bounds if bounds is not None else _Bounds()
        out, omax = ops.conv3d_planes(d_out, b.of(d_out), weight, skip=accumulate, want_absmax=(b.slot(d_out) if bounds is not None else False), mode=S2)
        b.put(out, omax)
        return out
    return ops.conv3d_bwd_data(d_out, weight, T2, in_shape, accumulate=accumulate)


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
def cost_reg_net(m, x: torch.Tensor, planes: bool = True) -> torch.Tensor:
    """(B,1,D,H,W) similarity volume -> (B,1,D,H,W) cost volume.  Every inner layer = convolution + BatchNorm (eval mode,
    folded to one fma after the sum) + ReLU in one kernel."""
    if m.training:
        raise UfrError("CostRegNet: inference only (BatchNorm in eval mode)")

    bounds = _Bounds()

    def layer(name, t, mode, skip):
        blk = getattr(m, name)
        scale, shift = _bn_fold(blk.bn)
        return _layer(t, blk.conv.weight, mode, planes, bounds, bn_scale=scale, bn_shift=shift, relu=True, skip=skip)

    x = _unet(_input_cl(x), layer)
    return _s1(x, m.prob.weight, planes, bounds, out_ncdhw=True)


def _needs_grad(m, x: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m.parameters()))


# the layers of CostRegNetWeight in execution order: (name, mode, skip source or None)
_LAYERS = (("conv0", S1), ("conv1", S2), ("conv2", S1), ("conv3", S2), ("conv4", S1), ("conv5", S2), ("conv6", S1),
           ("conv7", T2), ("conv9", T2), ("conv11", T2))
# ... their parameters, then the two bias-free heads
_PARAM_NAMES = tuple(f"{n}.{k}" for n, _ in _LAYERS for k in ("weight", "bias")) + ("features.weight", "weights.weight")


class CostRegNetWeightFn(torch.autograd.Function):
    """CostRegNetWeight.forward (module.py:530-543) with its adjoint on the HIP kernels: `feature_volume.cost_reg_2` is the
    one producer the reference trains (model.py:72-87); its gradients arrive through the frustum scatter of
    ufr_project_gather_bwd as d feature_volume / d weight_volume.  ``apply(x, *parameters in _PARAM_NAMES order)`` ->
    ``(features (B,8,D,H,W), sigmoid(weights) (B,1,D,H,W))``.
    Forward = the inference plan, keeping every layer's input; backward = per layer ufr_conv3d_bwd_data (the forward kernel
    on the same weights: mirrored taps / the strided-transposed twin) and ufr_conv3d_bwd_weight, the skip additions riding
    in the data-gradient launches as fused accumulations."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.set_materialize_grads(False)     # a head nobody differentiates arrives as None (backward handles it), not as zeros
        P = dict(zip(_PARAM_NAMES, params))
        x_cl = _input_cl(x)
        acts = {"x": x_cl}

        bounds = _Bounds()

        def layer(name, t, mode, skip):
            acts["in." + name] = t
            return _layer(t, P[name + ".weight"], mode, True, bounds, bias=P[name + ".bias"], skip=skip)

        y = _unet(x_cl, layer)
        feat, wsig = _s1(y, P["features.weight"], True, bounds, out_ncdhw=True, weight2=P["weights.weight"])
        ctx.acts, ctx.y = acts, y
        # the parameters through save_for_backward: an in-place update between this forward and its backward (an optimizer
        # step in between) is then an error, not a backward on new weights with old activations
        ctx.save_for_backward(wsig, *params)
        ctx.x_needs_grad = x.requires_grad
        return feat, wsig

    @staticmethod
    def backward(ctx, d_feat, d_wsig):
        wsig, *params = ctx.saved_tensors
        P = dict(zip(_PARAM_NAMES, params))
        acts, y = ctx.acts, ctx.y
        B, D, H, W, _ = y.shape
        grads = {}
        cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()       # (B,C,D,H,W) -> channel-last
        zeros = lambda c: torch.zeros(B, D, H, W, c, dtype=torch.float32, device=y.device)
        d_f = cl(d_feat.float()) if d_feat is not None else zeros(8)
        # sigmoid'(z) = s (1 - s)
        d_w = cl((d_wsig.float() * wsig * (1.0 - wsig))) if d_wsig is not None else zeros(1)
        # every weight / bias gradient of the net as a slice of ONE buffer zeroed once (the kernels accumulate into them)
        sizes = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=y.device)
        gview, off = {}, 0
        for n, p, sz in zip(_PARAM_NAMES, params, sizes):
            gview[n] = flat[off:off + sz].view(p.shape)
            off += sz
        grads["features.weight"], grads["weights.weight"] = ops.conv3d_bwd_weight_heads(
            y, d_f, d_w, out=(gview["features.weight"], gview["weights.weight"]))                      # one pass over y
        # the 1-channel head's adjoint on the fp32 kernel, then the 8-channel one on the matrix cores with the sum fused
        bounds = _Bounds()
        d_y = ops.conv3d_bwd_data(d_w, P["weights.weight"], S1, tuple(y.shape))
        d_y = _s1(d_f, P["features.weight"], True, bounds, flip=True, skip=d_y)

        def back(name, mode, d_out, accumulate=None, need_data=True):
            t = acts["in." + name]
            grads[name + ".weight"], grads[name + ".bias"] = ops.conv3d_bwd_weight(t, d_out, mode, P[name + ".weight"].shape,
                                                                                   out=(gview[name + ".weight"], gview[name + ".bias"]))
            if not need_data:
                return None
            if mode == S1 and t.shape[-1] > 1:
                return _s1(d_out, P[name + ".weight"], True, bounds, flip=True, skip=accumulate)
            if mode == T2:
                return _t2_bwd_data(d_out, P[name + ".weight"], tuple(t.shape), accumulate, bounds)
            if mode == S2:
                return _s2_bwd_data(d_out, P[name + ".weight"], tuple(t.shape), accumulate, bounds)
            return ops.conv3d_bwd_data(d_out, P[name + ".weight"], mode, tuple(t.shape), accumulate=accumulate)

        # y = c0 + conv11(x9), x9 = c2 + conv9(x7), x7 = c4 + conv7(x6), x6 = conv6(conv5(c4)), c4 = conv4(conv3(c2)), ...
        d_x9 = back("conv11", T2, d_y)
        d_x7 = back("conv9", T2, d_x9)
        d_x6 = back("conv7", T2, d_x7)
        d_x5 = back("conv6", S1, d_x6)
        d_c4 = back("conv5", S2, d_x5, accumulate=d_x7)          # c4 feeds conv5 and the skip into x7
        d_x3 = back("conv4", S1, d_c4)
        d_c2 = back("conv3", S2, d_x3, accumulate=d_x9)          # c2 feeds conv3 and the skip into x9
        d_x1 = back("conv2", S1, d_c2)
        d_c0 = back("conv1", S2, d_x1, accumulate=d_y)           # c0 feeds conv1 and the skip into y
        d_x = back("conv0", S1, d_c0, need_data=ctx.x_needs_grad)
        if d_x is not None:
            d_x = d_x.view(B, D, H, W, 1).permute(0, 4, 1, 2, 3)
        need = ctx.needs_input_grad[1:]
        return (d_x, *[(grads[n].reshape(p.shape) if nd else None) for n, p, nd in zip(_PARAM_NAMES, params, need)])


def cost_reg_net_weight(m, x: torch.Tensor):
    """(B,1,D,H,W) cost volume -> feature frustum (B,8,D,H,W), weight frustum (B,1,D,H,W) = sigmoid.  Plain convolutions with
    bias, no activation between them (as upstream); the two heads share one pass over the last feature map.
    With a gradient wanted (grad mode on and the input or a parameter requires grad) the same kernels run under
    CostRegNetWeightFn, whose backward is ufr_conv3d_bwd_data / ufr_conv3d_bwd_weight: one implementation, no library
    convolutions in either direction."""
    if _needs_grad(m, x):
        sd = dict(m.named_parameters())
        return CostRegNetWeightFn.apply(x, *[sd[n] for n in _PARAM_NAMES])
    with torch.no_grad():
        return _cost_reg_net_weight_hip(m, x)


def _cost_reg_net_weight_hip(m, x: torch.Tensor, planes: bool = True):
    bounds = _Bounds()

    def layer(name, t, mode, skip):
        conv = getattr(m, name)
        return _layer(t, conv.weight, mode, planes, bounds, bias=conv.bias, skip=skip)

    x = _unet(_input_cl(x), layer)
    return _s1(x, m.features.weight, planes, bounds, out_ncdhw=True, weight2=m.weights.weight)
