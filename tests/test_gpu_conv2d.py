"""ufr_conv2d / ufr_deform_conv2d_cl and the FeatureNet plan on them (uforecon_amd/featurenet.py:feature_net) against torch:
every layer shape of the backbone against F.conv2d + BatchNorm + ReLU (+ the FPN's upsampled addition), the
offset | sigmoid(mask) layout, and the whole backbone against its own layer-by-layer forward (library convolutions + the
planar deformable kernel, which tests/test_dcn.py pins against the oracle)."""
import pytest
import torch
import torch.nn.functional as F

from uforecon_amd import featurenet as FN
from uforecon_amd import ops
from uforecon_amd.scene import fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# (cin, cout, k, stride) of every plain convolution of FeatureNet; odd extents: ragged pixel tiles and borders
@pytest.mark.parametrize("cin,cout,k,stride", [(8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1),
                                               (32, 27, 3, 1), (32, 32, 1, 1), (16, 32, 1, 1), (8, 32, 1, 1)])
@pytest.mark.parametrize("B,H,W", [(1, 36, 52), (2, 64, 80)])
def test_conv2d_layers(cin, cout, k, stride, B, H, W):
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + k)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
    scale, shift = (0.5 + torch.rand(cout, generator=g)).to(DEV), torch.randn(cout, generator=g).to(DEV)
    want = F.relu(F.conv2d(x, w, None, stride, k // 2) * scale[None, :, None, None] + shift[None, :, None, None])
    x_cl = x.permute(0, 2, 3, 1).contiguous()
    got = ops.conv2d(x_cl, w, stride, scale, shift, relu=True)
    assert got.shape == (B, want.shape[2], want.shape[3], cout)
    assert _rel(got.permute(0, 3, 1, 2), want) < 2e-5
    # planar output, no epilogue
    got2 = ops.conv2d(x_cl, w, stride, out_planar=True)
    assert _rel(got2, F.conv2d(x, w, None, stride, k // 2)) < 2e-5


def test_conv2d_stem_offsets_layout_and_upsampled_skip():
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 40, 56
    img = torch.rand(B, 3, H, W, generator=g).to(DEV)
    w = (torch.randn(8, 3, 3, 3, generator=g) / 27 ** 0.5).to(DEV)
    sc, sh = (0.5 + torch.rand(8, generator=g)).to(DEV), torch.randn(8, generator=g).to(DEV)
    got = ops.conv2d(img, w, 1, sc, sh, relu=True, in_planar=True)
    want = F.relu(F.conv2d(img, w, None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    assert _rel(got.permute(0, 3, 1, 2), want) < 2e-5
    # offsets | sigmoid(masks), planar (dcn.py:66-70)
    x = torch.randn(B, 32, H, W, generator=g).to(DEV)
    wom, bom = (torch.randn(27, 32, 3, 3, generator=g) * 0.05).to(DEV), torch.randn(27, generator=g).to(DEV)
    om = ops.conv2d(x.permute(0, 2, 3, 1).contiguous(), wom, 1, None, bom, out_planar=True, sigmoid_from=18)
    ref = F.conv2d(x, wom, bom, 1, 1)
    o1, o2, mask = torch.chunk(ref, 3, dim=1)
    assert _rel(om[:, :18], torch.cat((o1, o2), 1)) < 2e-5 and _rel(om[:, 18:], torch.sigmoid(mask)) < 2e-5
    # the FPN's lateral connection: interpolate(coarse, 2, 'nearest') + conv1x1(fine) + bias
    fine = torch.randn(B, 16, H, W, generator=g).to(DEV)
    coarse = torch.randn(B, 32, H // 2, W // 2, generator=g).to(DEV)
    w1, b1 = (torch.randn(32, 16, 1, 1, generator=g) * 0.2).to(DEV), torch.randn(32, generator=g).to(DEV)
    got = ops.conv2d(fine.permute(0, 2, 3, 1).contiguous(), w1, 1, None, b1, skip=coarse.permute(0, 2, 3, 1).contiguous())
    want = F.interpolate(coarse, scale_factor=2, mode="nearest") + F.conv2d(fine, w1, b1)
    assert _rel(got.permute(0, 3, 1, 2), want) < 2e-5
    with pytest.raises(Exception):
        ops.conv2d(fine.permute(0, 2, 3, 1).contiguous(), torch.randn(32, 16, 7, 7, device=DEV))      # not a FeatureNet shape


@pytest.mark.parametrize("cout,bn", [(32, True), (16, False), (8, False)])
def test_deform_conv2d_cl_equals_the_planar_kernel_with_the_epilogue(cout, bn):
    g = torch.Generator().manual_seed(cout)
    B, H, W = 2, 24, 40
    x = torch.randn(B, 32, H, W, generator=g).to(DEV)
    om = torch.randn(B, 27, H, W, generator=g).to(DEV)
    om[:, 18:] = torch.sigmoid(om[:, 18:])
    w = (torch.randn(cout, 32, 3, 3, generator=g) / 288 ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    want = FN.deform_conv2d(x, om[:, :18].contiguous(), w, bias, 1, 1, 1, mask=om[:, 18:].contiguous())
    sc = sh = None
    if bn:
        sc, sh = (0.5 + torch.rand(cout, generator=g)).to(DEV), torch.randn(cout, generator=g).to(DEV)
        want = F.relu(want * sc[None, :, None, None] + sh[None, :, None, None])
    x_cl = x.permute(0, 2, 3, 1).contiguous()
    got = ops.deform_conv2d_cl(x_cl, om, w, bias, sc, sh, relu=bn)
    assert _rel(got.permute(0, 3, 1, 2), want) < 1e-6
    got_p = ops.deform_conv2d_cl(x_cl, om, w, bias, sc, sh, relu=bn, out_planar=True)
    assert _rel(got_p, want) < 1e-6


@pytest.mark.parametrize("B,H,W", [(1, 64, 80), (2, 128, 160)])
def test_featurenet_plan_equals_its_layer_by_layer_forward(B, H, W):
    m = fill_state_dict(FN.FeatureNet(8), 12).eval().to(DEV)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():                                        # non-zero offset / mask convolutions (they are zero-init)
        for mod in m.modules():
            if isinstance(mod, FN.DCN):
                mod.conv_offset_mask.weight.copy_((torch.rand(mod.conv_offset_mask.weight.shape, generator=g) - 0.5) * 0.1)
                mod.conv_offset_mask.bias.copy_((torch.rand(27, generator=g) - 0.5) * 0.6)
    x = torch.rand(B, 3, H, W, generator=g).to(DEV)
    with torch.no_grad():
        want = m.forward_layers(x)
        got = m(x)
        assert getattr(m, "_ufr_plan", None) is not None        # the plan ran (and cached its folded parameters)
        again = m(x)
    for st, c, s in (("stage1", 32, 4), ("stage2", 16, 2), ("stage3", 8, 1)):
        assert got[st].shape == (B, c, H // s, W // s) and got[st].is_contiguous()
        assert _rel(got[st], want[st]) < 2e-4, st
        assert torch.equal(got[st], again[st])
    # a parameter update invalidates the folded cache
    with torch.no_grad():
        m.conv0[0].bn.weight.mul_(1.5)
        got2 = m(x)
    assert not torch.equal(got2["stage3"], got["stage3"])


@pytest.mark.parametrize("NS,D,H,W", [(2, 48, 32, 40), (4, 8, 17, 23)])
def test_pixelwise_view_weights_equal_the_module_and_the_reference_aggregate(NS, D, H, W):
    """ufr_pixelwise_view_weights against PixelwiseNet's library forward (TransMVSNet.py:23-41) and the weighted aggregate of
    DepthNet.forward (:86-97) in the reference's order."""
    from uforecon_amd import cascade, frustum

    net = fill_state_dict(cascade.PixelwiseNet(), 4).eval().to(DEV)
    sim = (torch.rand(NS, D, H, W, generator=torch.Generator().manual_seed(NS)) * 2 - 0.5).to(DEV)
    with torch.no_grad():
        want_vw = torch.cat([net(sim[i][None, None]) for i in range(NS)], dim=1)[0]
        s_sum = torch.zeros_like(sim[0])
        w_sum = torch.full_like(want_vw[0], 1e-5)
        for i in range(NS):
            s_sum = s_sum + sim[i] * want_vw[i].unsqueeze(0)
            w_sum = w_sum + want_vw[i]
        want_agg = s_sum / w_sum.unsqueeze(0)
        vw, agg = frustum.view_weights(net, sim)
    assert vw.shape == (NS, H, W) and agg.shape == (D, H, W)
    assert float((vw - want_vw).abs().max()) < 2e-6 and _rel(agg, want_agg) < 1e-5
    assert frustum.view_weights(net, sim, want_aggregate=False)[1] is None


def test_fmt_pathway_on_hip_kernels_equals_the_library_expression():
    """FMT_with_pathway's two `_push_down` steps (FMT.py:226-255) through ufr_conv2d + ufr_upsample_add against the
    layer-by-layer torch expression (1x1 conv, F.interpolate bilinear, add, 3x3 conv) -- odd extents at the coarse level."""
    from uforecon_amd import fmt

    m = fill_state_dict(fmt.FMT_with_pathway(8), 5).eval().to(DEV)
    g = torch.Generator().manual_seed(9)
    B, h, w = 2, 17, 23
    s1 = torch.randn(B, 32, h, w, generator=g).to(DEV)
    f2 = torch.randn(B, 16, 2 * h, 2 * w, generator=g).to(DEV)
    f3 = torch.randn(B, 8, 4 * h, 4 * w, generator=g).to(DEV)
    with torch.no_grad():
        want2 = m._push_down(s1, f2, m.dim_reduction_1, m.smooth_1)
        want3 = m._push_down(want2, f3, m.dim_reduction_2, m.smooth_2)
        got2, got3 = m._pathway_hip(s1, f2, f3)
    assert got2.shape == want2.shape and got3.shape == want3.shape and got2.is_contiguous() and got3.is_contiguous()
    assert _rel(got2, want2) < 2e-5 and _rel(got3, want3) < 2e-5
    # the sum the smoothing convolution reads
    r = torch.randn(B, h, w, 16, generator=g).to(DEV)
    want = F.interpolate(r.permute(0, 3, 1, 2), size=(2 * h, 2 * w), mode="bilinear") + f2
    assert _rel(ops.upsample_add(r, f2).permute(0, 3, 1, 2), want) < 1e-6
