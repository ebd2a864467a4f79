"""Backward of the trainable frustum U-Net (CostRegNetWeight = `feature_volume.cost_reg_2`, the one producer the reference
trains: code1/model.py:72-87, encoder_utils/fmt/module.py:502-543) on the HIP kernels, against autograd of the
torch restatement (oracle/cascade_oracle.py: the reference's forward expression on the module's own layers)."""
import pytest
import torch

from oracle import cascade_oracle as CO
from uforecon_amd import cascade, ops, unet3d

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("mode,cin,cout", [(ops.CONV3D_S1, 1, 8), (ops.CONV3D_S1, 16, 16), (ops.CONV3D_S1, 64, 64), (ops.CONV3D_S1, 8, 1),
                                           (ops.CONV3D_S2, 8, 16), (ops.CONV3D_S2, 32, 64), (ops.CONV3D_T2, 64, 32),
                                           (ops.CONV3D_T2, 16, 8), (ops.CONV3D_S1, 8, 8), (ops.CONV3D_S1, 32, 32),
                                           (ops.CONV3D_S2, 16, 32), (ops.CONV3D_T2, 32, 16)])
def test_single_layer_gradients(mode, cin, cout):
    """ufr_conv3d_bwd_data / ufr_conv3d_bwd_weight of every layer shape of the network against torch's autograd of the same
    convolution (fp32 sums in another order: 2e-5 of each tensor's scale)."""
    g = torch.Generator().manual_seed(cin * 100 + cout + mode)
    B, D, H, W = 2, 8, 8, 16
    x = torch.randn(B, cin, D, H, W, generator=g).to(DEV).requires_grad_(True)
    if mode == ops.CONV3D_T2:
        conv = torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1)
    else:
        conv = torch.nn.Conv3d(cin, cout, 3, stride=2 if mode == ops.CONV3D_S2 else 1, padding=1)
    conv = conv.to(DEV)
    y = conv(x)
    d_out = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(d_out)
    cl = lambda t: t.detach().permute(0, 2, 3, 4, 1).contiguous()
    x_cl, d_cl = cl(x), cl(d_out)
    dw, db = ops.conv3d_bwd_weight(x_cl, d_cl, mode, conv.weight.shape)
    assert _rel(dw, conv.weight.grad) < 2e-5 and _rel(db, conv.bias.grad) < 2e-5
    d_in = ops.conv3d_bwd_data(d_cl, conv.weight, mode, tuple(x_cl.shape))
    assert _rel(d_in, cl(x.grad)) < 2e-5
    if cin > 1:     # the fused second path
        extra = torch.randn(x_cl.shape, generator=g).to(DEV)
        d_in2 = ops.conv3d_bwd_data(d_cl, conv.weight, mode, tuple(x_cl.shape), accumulate=extra)
        assert _rel(d_in2, cl(x.grad) + extra) < 2e-5


# volumes large enough for the matrix-core weight-gradient kernels (conv3d.hip launch_wgrad_t: >= 100 000 voxel pairs for
# the 16-channel tiles, >= 500 000 for the 8-channel layers); widths that are not whole segments / not multiples of four
@pytest.mark.parametrize("mode,cin,cout,dims", [
    (ops.CONV3D_S1, 16, 16, (16, 64, 104)), (ops.CONV3D_S1, 32, 32, (16, 64, 104)), (ops.CONV3D_S1, 64, 64, (16, 64, 104)),
    (ops.CONV3D_S2, 16, 32, (32, 128, 208)), (ops.CONV3D_S2, 32, 64, (32, 128, 208)),
    (ops.CONV3D_T2, 32, 16, (16, 64, 104)), (ops.CONV3D_T2, 64, 32, (16, 64, 104)),
    (ops.CONV3D_S1, 8, 8, (16, 128, 250)), (ops.CONV3D_S2, 8, 16, (32, 256, 500)), (ops.CONV3D_T2, 16, 8, (16, 128, 250))])
def test_weight_gradients_at_matrix_core_sizes(mode, cin, cout, dims):
    """ufr_conv3d_bwd_weight where it runs on the fp32 MFMA kernels, against torch's autograd of the same convolution (two
    fp32 sums of 1e5 .. 5e5 terms in different orders: 1e-4 of the tensor's scale)."""
    g = torch.Generator().manual_seed(cin * 100 + cout + mode)
    D, H, W = dims
    x = torch.randn(1, cin, D, H, W, generator=g).to(DEV)
    if mode == ops.CONV3D_T2:
        conv = torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1)
    else:
        conv = torch.nn.Conv3d(cin, cout, 3, stride=2 if mode == ops.CONV3D_S2 else 1, padding=1)
    conv = conv.to(DEV)
    y = conv(x)
    d_out = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(d_out)
    cl = lambda t: t.detach().permute(0, 2, 3, 4, 1).contiguous()
    dw, db = ops.conv3d_bwd_weight(cl(x), cl(d_out), mode, conv.weight.shape)
    assert _rel(dw, conv.weight.grad) < 1e-4 and _rel(db, conv.bias.grad) < 1e-4


@pytest.mark.parametrize("shape", [(3, 8, 16, 24), (1, 16, 8, 8)])
def test_cost_reg_net_weight_gradients_match_autograd_of_the_reference_expression(shape):
    """The whole network: parameters' and input's gradients of a scalar function of both heads, HIP forward + backward
    (unet3d.CostRegNetWeightFn) against autograd through the reference's forward expression on library convolutions."""
    B, D, H, W = shape
    torch.manual_seed(5)
    m = cascade.CostRegNetWeight(1, 8).to(DEV)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 1, D, H, W, generator=g).to(DEV)
    cf = torch.randn(B, 8, D, H, W, generator=g).to(DEV)
    cw = torch.randn(B, 1, D, H, W, generator=g).to(DEV)

    def loss_of(feat, w):
        return (feat * cf).sum() + (w * cw).sum() + 0.1 * (feat * feat).mean()

    xr = x.clone().requires_grad_(True)
    f_ref, w_ref = CO.cost_reg_net_weight(m, xr)
    loss_of(f_ref, w_ref).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    ref_x = xr.grad.clone()
    m.zero_grad()
    xh = x.clone().requires_grad_(True)
    f_hip, w_hip = m(xh)                       # grad mode on, parameters require grad -> CostRegNetWeightFn
    assert f_hip.grad_fn is not None and type(f_hip.grad_fn).__name__.startswith("CostRegNetWeightFn")
    assert _rel(f_hip, f_ref) < 1e-5 and _rel(w_hip, w_ref) < 1e-5
    loss_of(f_hip, w_hip).backward()
    worst = {k: _rel(p.grad, ref[k]) for k, p in m.named_parameters()}
    bad = {k: e for k, e in worst.items() if not e < 1e-4}
    assert not bad, bad
    assert _rel(xh.grad, ref_x) < 1e-4
    # parameters only (the training step's case: the cost volume comes from the frozen cascade)
    m.zero_grad()
    f2, w2 = m(x)
    loss_of(f2, w2).backward()
    assert all(_rel(p.grad, ref[k]) < 1e-4 for k, p in m.named_parameters())
    # and without a gradient wanted the inference plan runs (no graph)
    with torch.no_grad():
        f3, w3 = m(x)
    assert f3.grad_fn is None and torch.equal(f3, f2.detach()) and torch.equal(w3, w2.detach())


def test_cost_reg_2_gradients_match_the_references_autograd():
    """The training step with the one producer the reference trains in front of it (model.py:72-87, 517-524): seeded cost
    volumes -> MVSVolume (cost_reg_2) -> frustums -> infer -> loss.  Gradients of every `feature_volume.cost_reg_2.*`
    parameter against the REFERENCE's own autograd (tests/golden/make_golden.py:run_costreg_grad_case ran the reference's
    MVSVolume and infer): they pass through ufr_project_gather_bwd (frustum scatter) and ufr_conv3d_bwd_*."""
    import argparse
    import os
    import sys

    import numpy as np

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import CASES, case_frame, load_golden, load_weights
    from test_gpu_backward import _args, _loss_from_tuple
    from uforecon_amd import model as M
    from uforecon_amd.scene import fill_state_dict, make_cost_volumes, sampler_uniforms

    name = "c5_train_grads_costreg"
    c, g = CASES[name], load_golden(name)
    fr = case_frame(name)
    f = fr.to(DEV)
    mvs = cascade.MVSVolume(1, 8)
    fill_state_dict(mvs, c["costreg_seed"])
    mvs = mvs.to(DEV).train()
    m = M.UFORecon(_args(c)).to(DEV)
    m.load_state_dict(load_weights(), strict=True)
    m.train()
    cost = {st: v.to(DEV) for st, v in make_cost_volumes(c["H"], c["W"], c["NV"], c["seed"]).items()}
    vols = {}
    for st in ("stage1", "stage2", "stage3"):
        feat, w = mvs(f.batch, cost[st])
        vols[st] = {"feature_volume": feat, "weight_volume": w}
    idx = torch.from_numpy(g["ray_idx"]).to(DEV)
    U1, U2 = sampler_uniforms(int(g["sampler_seed"]), c["coarse"], c["fine"], c["RN"])
    r = m.infer(f.batch, idx, f.source_imgs_feat, vols, match_feature=f.match_feature, uniforms=(U1, U2))
    loss = _loss_from_tuple(r, f.batch, idx)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    worst = {}
    for k, p in mvs.named_parameters():
        key = "feature_volume." + k
        got = p.grad.reshape(-1).double().cpu()
        n = got.numel()
        sel = torch.arange(n) if n <= 20_000 else torch.arange(4096) * (n // 4096)          # make_golden.costreg_sample_index
        want = torch.from_numpy(g["grad." + key]).double()
        scale = max(float(np.sqrt(float(g["gsq." + key]) / n)) * 10, float(want.abs().max()), 1e-12)
        worst[k] = float((got[sel] - want).abs().max()) / scale
        # the whole tensor, through its stored sum and squared norm
        assert abs(float(got.sum()) - float(g["gsum." + key])) < 1e-3 * (abs(float(g["gsum." + key])) + scale * n ** 0.5), k
        assert abs(float((got ** 2).sum()) - float(g["gsq." + key])) < 1e-3 * float(g["gsq." + key]) + 1e-30, k
    # measured round 6 (printed below, per tensor): worst 9.7e-5 (conv7.weight), most 1e-5 .. 6e-5 -- the bound was 1e-3 while
    # the step's matrix precision was the only thing known about it; 3e-4 leaves the float atomics' reordering its room
    bad = {k: e for k, e in worst.items() if not e < 3e-4}
    print("cost_reg_2 gradients vs the reference's autograd, worst of each tensor:", max(worst.values()), max(worst, key=worst.get))
    for k, e in sorted(worst.items(), key=lambda kv: -kv[1]):
        print(f"   {k:24s} {e:.2e}")
    assert not bad, bad

