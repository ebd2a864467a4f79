"""Frustum cascade (second part): mirror modules vs the reference's outputs."""
import os

import numpy as np
import pytest
import torch

from uforecon_amd import cascade
from uforecon_amd.scene import CASCADE_CASES, fill_state_dict, make_cascade_case

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(name):
    return np.load(os.path.join(HERE, "golden", f"cascade_{name}.npz"))


def _compare(frustums, info, g, tol, max_outlier_frac):
    """Winner-take-all depth feeds the next stage's hypotheses: a probability tie resolved differently moves a few
    pixels' hypotheses, so agreement is asserted on all but a small fraction of the elements."""
    worst = 0.0
    for st in ("stage1", "stage2", "stage3"):
        pairs = [(info[st]["depth"], g[st + ".depth"]), (info[st]["photometric_confidence"], g[st + ".photometric_confidence"]),
                 (info[st]["cost_volume"][..., ::2, ::2], g[st + ".cost_volume"]),
                 (frustums[st]["feature_volume"][..., ::2, ::2], g[st + ".feature_volume"]),
                 (frustums[st]["weight_volume"][..., ::2, ::2], g[st + ".weight_volume"])]
        for got, ref in pairs:
            ref = torch.from_numpy(ref)
            got = got.detach().cpu()
            assert got.shape == ref.shape, (st, got.shape, ref.shape)
            bad = ((got - ref).abs() > tol * (ref.abs() + ref.abs().mean())).float().mean().item()
            worst = max(worst, bad)
    assert worst <= max_outlier_frac, worst


def test_state_dict_keys_are_the_references():
    m = cascade.FrustumBuilder()
    keys = set(m.state_dict())
    assert "transmvsnet.cost_regularization.2.conv11.bn.running_var" in keys
    assert "transmvsnet.DepthNet.pixel_wise_net.conv2.bias" in keys
    assert "feature_volume.cost_reg_2.features.weight" in keys and "feature_volume.cost_reg_2.weights.weight" in keys
    assert sum(p.numel() for p in m.feature_volume.parameters()) == 292_752          # SURVEY 8a parameter inventory


@pytest.mark.parametrize("name", list(CASCADE_CASES))
def test_mirror_with_cpu_correlate_matches_reference_golden(name):
    from oracle import cascade_oracle as CO

    c = make_cascade_case(name)
    m = fill_state_dict(cascade.FrustumBuilder(), c["weight_seed"]).eval()
    frustums, info = CO.run_cascade_cpu(m, c)
    g = _golden(name)
    assert np.allclose(info["stage1"]["depth"].numpy(), g["stage1.depth"], rtol=1e-5)
    _compare(frustums, info, g, tol=1e-4, max_outlier_frac=0.002)


def test_product_cascade_has_no_cpu_path():
    from uforecon_amd.ops import UfrError

    c = make_cascade_case("small3")
    m = fill_state_dict(cascade.FrustumBuilder(), c["weight_seed"]).eval()
    with pytest.raises(UfrError):
        m(c["features"], c["proj_matrices"], c["depth_values"], c["img_hw"])


def test_trainable_frustum_unet_refuses_cpu_tensors_too():
    """cost_reg_2 is trained by the reference: with gradients wanted the differentiable expression runs -- on the GPU only."""
    from uforecon_amd.ops import UfrError

    m = cascade.CostRegNetWeight(1, 8)
    x = torch.randn(1, 1, 8, 8, 8)
    with pytest.raises(UfrError):
        m(x)                                    # parameters require grad -> trainable path -> CPU tensor refused
    with torch.no_grad(), pytest.raises(UfrError):
        m(x)                                    # forward-only HIP plan -> CPU tensor refused


@pytest.mark.gpu
def test_gpu_frustum_unet_stays_differentiable_when_gradients_are_wanted():
    """MVSVolume's U-Net (`feature_volume.cost_reg_2`, module.py:530-543) trains in the reference: with grad mode on and
    parameters that require grad its outputs carry a graph and every parameter receives a gradient; without, the
    forward-only HIP plan runs and gives the same numbers."""
    dev = "cuda:0"
    torch.manual_seed(3)
    m = cascade.CostRegNetWeight(1, 8).to(dev)
    x = torch.randn(2, 1, 8, 16, 24, device=dev)
    f, w = m(x)
    assert f.requires_grad and w.requires_grad
    (f.square().mean() + w.mean()).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0 for p in m.parameters())
    with torch.no_grad():
        f2, w2 = m(x)
    assert not f2.requires_grad
    assert float((f2 - f).abs().max()) < 1e-4 * float(f.abs().max()) and float((w2 - w).abs().max()) < 1e-5
    for p in m.parameters():
        p.requires_grad_(False)
    f3, _ = m(x)                                # nothing to differentiate -> HIP plan even in grad mode
    assert not f3.requires_grad and torch.equal(f3, f2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASCADE_CASES))
def test_gpu_cascade_matches_reference_golden(name):
    dev = "cuda:0"
    c = make_cascade_case(name)
    m = fill_state_dict(cascade.FrustumBuilder(), c["weight_seed"]).eval().to(dev)
    feats = [{k: v.to(dev) for k, v in f.items()} for f in c["features"]]
    frustums, info = m(feats, c["proj_matrices"], c["depth_values"].to(dev), c["img_hw"])
    assert frustums["stage3"]["feature_volume"].shape == (3, 8, 8, 32, 64)
    # the HIP correlate + convolution kernels vs the reference's CPU modules (the outlier budget is for winner-take-all
    # ties that move a pixel's depth hypotheses)
    _compare(frustums, info, _golden(name), tol=1e-4, max_outlier_frac=0.002)


LAYERS = [(1, 8, 0), (16, 16, 0), (32, 32, 0), (64, 64, 0), (8, 8, 0), (8, 16, 1), (16, 32, 1), (32, 64, 1),
          (64, 32, 2), (32, 16, 2), (16, 8, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,mode", LAYERS)
def test_gpu_conv3d_layer_matches_torch(cin, cout, mode):
    """ufr_conv3d, every (cin, cout, stride / transposed) of the two U-Nets, against torch's fp32 CPU convolution of the
    same layer: plain, with bias, with folded BatchNorm + ReLU + skip; odd extents exercise the zero padding."""
    import torch.nn.functional as F

    from uforecon_amd import ops

    torch.manual_seed(100 * cin + cout + mode)
    dev = "cuda:0"
    B, D, H, W = 2, 6, 10, 36
    x = torch.randn(B, cin, D, H, W)
    w = torch.randn((cin, cout, 3, 3, 3) if mode == 2 else (cout, cin, 3, 3, 3)) / (27 * cin) ** 0.5
    bias, scale, shift = torch.randn(cout), torch.rand(cout) + 0.5, torch.randn(cout)
    ref = (F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1) if mode == 2
           else F.conv3d(x, w, stride=1 + mode, padding=1))
    skip = torch.randn_like(ref)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    cl = lambda t: t.permute(0, 2, 3, 4, 1)
    out = ops.conv3d(x_cl, w.to(dev), mode)
    assert out.shape == cl(ref).shape
    tol = 2e-6 * float(ref.abs().max()) * (27 * cin) ** 0.5
    assert float((out.cpu() - cl(ref)).abs().max()) < tol
    out = ops.conv3d(x_cl, w.to(dev), mode, bias=bias.to(dev), skip=cl(skip).contiguous().to(dev))
    assert float((out.cpu() - cl(ref + bias.view(1, -1, 1, 1, 1) + skip)).abs().max()) < tol
    want = torch.relu(ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)) + skip
    out = ops.conv3d(x_cl, w.to(dev), mode, bn_scale=scale.to(dev), bn_shift=shift.to(dev), relu=True,
                     skip=cl(skip).contiguous().to(dev))
    assert float((out.cpu() - cl(want)).abs().max()) < 2 * tol


@pytest.mark.gpu
def test_gpu_conv3d_heads_and_errors():
    """The heads: prob (8 -> 1) and features (8 -> 8) + sigmoid(weights (8 -> 1)) in one pass, reference layout out."""
    import torch.nn.functional as F

    from uforecon_amd import ops
    from uforecon_amd.ops import UfrError

    torch.manual_seed(5)
    dev = "cuda:0"
    x = torch.randn(3, 8, 8, 16, 24)
    wf, ww = torch.randn(8, 8, 3, 3, 3) / 15, torch.randn(1, 8, 3, 3, 3) / 15
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    f, s = ops.conv3d(x_cl, wf.to(dev), out_ncdhw=True, weight2=ww.to(dev))
    assert float((f.cpu() - F.conv3d(x, wf, padding=1)).abs().max()) < 1e-5
    assert float((s.cpu() - torch.sigmoid(F.conv3d(x, ww, padding=1))).abs().max()) < 1e-5
    p = ops.conv3d(x_cl, ww.to(dev), out_ncdhw=True)
    assert p.shape == (3, 1, 8, 16, 24) and float((p.cpu() - F.conv3d(x, ww, padding=1)).abs().max()) < 1e-5
    with pytest.raises(UfrError, match="not a layer"):
        ops.conv3d(torch.zeros(1, 8, 8, 8, 24, device=dev), torch.zeros(8, 24, 3, 3, 3, device=dev))
    with pytest.raises(UfrError, match="GPU"):
        ops.conv3d(torch.zeros(1, 8, 8, 8, 8), torch.zeros(8, 8, 3, 3, 3))
