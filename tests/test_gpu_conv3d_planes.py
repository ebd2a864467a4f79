"""ufr_conv3d_planes (csrc/conv3d_planes.hip): the stride-1 8- / 16-channel layers of the frustum U-Nets on the 16-bit matrix
cores against the fp32 kernels (ufr_conv3d / ufr_conv3d_bwd_data) AND against a float64 torch convolution -- the layers of
CostRegNetWeight, code1/encoder_utils/fmt/module.py:502-543.  The plane products carry 22 significand bits: the bound
asserted here is the one the fp32 kernel itself meets against float64 (x 4), not a loosened one."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ref64(x_cl, w, bias=None, skip=None, flip=False):
    x = x_cl.double().permute(0, 4, 1, 2, 3)
    if flip:       # data gradient of a stride-1 layer with forward weight w (cout_fwd = cin here, cin_fwd, 3,3,3)
        y = torch.nn.functional.conv_transpose3d(x, w.double(), padding=1)
    else:
        y = torch.nn.functional.conv3d(x, w.double(), None if bias is None else bias.double(), padding=1)
    y = y.permute(0, 2, 3, 4, 1)
    return y if skip is None else y + skip.double()


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("cin,cout,shape", [(8, 8, (2, 4, 12, 70)), (16, 16, (1, 3, 9, 37)), (8, 8, (1, 8, 32, 128)), (16, 16, (3, 2, 16, 64)),
                                            (32, 32, (2, 3, 10, 21)), (64, 64, (1, 2, 9, 20)), (32, 32, (3, 2, 16, 32)), (64, 64, (3, 1, 8, 16))])
@pytest.mark.parametrize("flip", [False, True])
def test_planes_layer_matches_fp32_kernel_and_float64(cin, cout, shape, flip):
    from uforecon_amd import ops

    torch.manual_seed(cin * 100 + cout + int(flip))
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, device=DEV) * 3.0
    x[0, 0, 0, :5] *= 40.0                                        # a few large values: the scale follows the measured maximum
    w = torch.randn((cin, cout, 3, 3, 3) if flip else (cout, cin, 3, 3, 3), device=DEV) * 0.2
    bias = None if flip else torch.randn(cout, device=DEV)
    skip = torch.randn(B, D, H, W, cout, device=DEV)
    amax = ops.absmax(x)
    assert float(amax) == float(x.abs().max())
    y, ymax = ops.conv3d_planes(x, amax, w, bias=bias, skip=skip, flip=flip)
    if flip:
        y32 = ops.conv3d_bwd_data(x, w, ops.CONV3D_S1, (B, D, H, W, cout), accumulate=skip)
    else:
        y32 = ops.conv3d(x, w, ops.CONV3D_S1, bias=bias, skip=skip)
    ref = _ref64(x, w, bias, skip, flip)
    e16, e32 = _err(y, ref), _err(y32, ref)
    print(f"cin {cin} cout {cout} flip {flip} {shape}: planes {e16:.2e}  fp32 kernel {e32:.2e} of the output scale")
    assert e16 < max(4 * e32, 1e-6)
    assert float(ymax) == float(y.abs().max())                    # the bound handed to the next layer is the true maximum


@pytest.mark.parametrize("cin,cout,shape", [(8, 16, (2, 4, 12, 70)), (16, 32, (1, 3, 9, 37)), (32, 64, (2, 2, 10, 22)), (8, 16, (1, 8, 32, 128)),
                                            (16, 32, (3, 4, 16, 64)), (32, 64, (3, 2, 8, 32))])
def test_planes_stride2_layer_matches_fp32_kernel_and_float64(cin, cout, shape):
    """conv1 / conv3 / conv5 (and the data gradients of conv11 / conv9 / conv7, which ARE these convolutions)."""
    from uforecon_amd import ops

    torch.manual_seed(cin + cout)
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, device=DEV) * 2.0
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.15
    bias = torch.randn(cout, device=DEV)
    ref = torch.nn.functional.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), bias.double(), stride=2, padding=1).permute(0, 2, 3, 4, 1)
    skip = torch.randn(ref.shape, device=DEV)
    y, ymax = ops.conv3d_planes(x, ops.absmax(x), w, bias=bias, skip=skip, mode=ops.CONV3D_S2)
    y32 = ops.conv3d(x, w, ops.CONV3D_S2, bias=bias, skip=skip)
    assert y.shape == y32.shape == ref.shape
    ref = ref + skip.double()
    e16, e32 = _err(y, ref), _err(y32, ref)
    print(f"stride 2, cin {cin} cout {cout} {shape}: planes {e16:.2e}  fp32 kernel {e32:.2e} of the output scale")
    assert e16 < max(4 * e32, 1e-6) and float(ymax) == float(y.abs().max())
    if shape[1] % 2 == 0 and shape[2] % 2 == 0 and shape[3] % 2 == 0:
        # as the data gradient of the transposed layer with forward weight (cin_fwd = cout here, cout_fwd = cin here)
        d = ops.conv3d_bwd_data(x, w, ops.CONV3D_T2, (B, D // 2, H // 2, W // 2, cout))
        y0, _ = ops.conv3d_planes(x, ops.absmax(x), w, mode=ops.CONV3D_S2)
        assert float((d - y0).abs().max()) < 2e-5 * float(d.abs().max())


@pytest.mark.parametrize("cin,cout,shape", [(16, 8, (2, 3, 6, 35)), (32, 16, (1, 3, 5, 18)), (64, 32, (2, 2, 5, 17)), (16, 8, (1, 4, 16, 64)),
                                            (32, 16, (3, 2, 8, 32)), (64, 32, (3, 1, 4, 16))])
def test_planes_transposed_layer_matches_fp32_kernel_and_float64(cin, cout, shape):
    """conv7 / conv9 / conv11 (and the data gradients of conv5 / conv3 / conv1)."""
    from uforecon_amd import ops

    torch.manual_seed(3 * cin + cout)
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, device=DEV) * 2.0
    w = torch.randn(cin, cout, 3, 3, 3, device=DEV) * 0.15
    bias = torch.randn(cout, device=DEV)
    ref = torch.nn.functional.conv_transpose3d(x.double().permute(0, 4, 1, 2, 3), w.double(), bias.double(), stride=2, padding=1,
                                               output_padding=1).permute(0, 2, 3, 4, 1)
    skip = torch.randn(ref.shape, device=DEV)
    y, ymax = ops.conv3d_planes(x, ops.absmax(x), w, bias=bias, skip=skip, mode=ops.CONV3D_T2)
    y32 = ops.conv3d(x, w, ops.CONV3D_T2, bias=bias, skip=skip)
    assert y.shape == y32.shape == ref.shape == (B, 2 * D, 2 * H, 2 * W, cout)
    ref = ref + skip.double()
    e16, e32 = _err(y, ref), _err(y32, ref)
    print(f"transposed, cin {cin} cout {cout} {shape}: planes {e16:.2e}  fp32 kernel {e32:.2e} of the output scale")
    assert e16 < max(4 * e32, 1e-6) and float(ymax) == float(y.abs().max())
    # as the data gradient of the strided layer whose forward weight this is (cout_fwd = cin here)
    d = ops.conv3d_bwd_data(x, w, ops.CONV3D_S2, (B, 2 * D, 2 * H, 2 * W, cout))
    y0, _ = ops.conv3d_planes(x, ops.absmax(x), w, mode=ops.CONV3D_T2)
    assert float((d - y0).abs().max()) < 2e-5 * float(d.abs().max())


def test_planes_heads_write_the_reference_layout():
    """features (8) + sigmoid(weights (1)) in one pass, (B,C,D,H,W) outputs (module.py:541-543)."""
    from uforecon_amd import ops

    torch.manual_seed(3)
    B, D, H, W = 2, 3, 10, 66
    x = torch.randn(B, D, H, W, 8, device=DEV)
    wf, ww = torch.randn(8, 8, 3, 3, 3, device=DEV) * 0.2, torch.randn(1, 8, 3, 3, 3, device=DEV) * 0.2
    f, s, _ = ops.conv3d_planes(x, ops.absmax(x), wf, out_ncdhw=True, weight2=ww)
    f32, s32 = ops.conv3d(x, wf, ops.CONV3D_S1, out_ncdhw=True, weight2=ww)
    assert f.shape == (B, 8, D, H, W) and s.shape == (B, 1, D, H, W)
    rf = _ref64(x, wf).permute(0, 4, 1, 2, 3)
    rs = torch.sigmoid(_ref64(x, ww).permute(0, 4, 1, 2, 3))
    assert _err(f, rf) < max(4 * _err(f32, rf), 1e-6) and _err(s, rs) < max(4 * _err(s32, rs), 1e-6)


def test_planes_bn_relu_and_zero_input():
    from uforecon_amd import ops

    torch.manual_seed(4)
    x = torch.randn(1, 2, 8, 32, 16, device=DEV)
    w = torch.randn(16, 16, 3, 3, 3, device=DEV) * 0.1
    sc, sh = torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV)
    y, _ = ops.conv3d_planes(x, ops.absmax(x), w, bn_scale=sc, bn_shift=sh, relu=True)
    y32 = ops.conv3d(x, w, ops.CONV3D_S1, bn_scale=sc, bn_shift=sh, relu=True)
    assert float((y - y32).abs().max()) < 2e-5 * float(y32.abs().max())
    z = torch.zeros_like(x)
    yz, zmax = ops.conv3d_planes(z, ops.absmax(z), w, bias=sh)
    assert torch.equal(yz, sh.expand_as(yz)) and float(zmax) == float(sh.abs().max())


def test_planes_refuses_other_layers():
    from uforecon_amd import ops

    assert ops.conv3d_planes_supported(8, 8, 1) and ops.conv3d_planes_supported(16, 16) and ops.conv3d_planes_supported(64, 64)
    assert not ops.conv3d_planes_supported(32, 64) and not ops.conv3d_planes_supported(1, 8)
    x = torch.randn(1, 2, 8, 8, 32, device=DEV)
    with pytest.raises(ops.UfrError, match="not a layer of this kernel family"):
        ops.conv3d_planes(x, ops.absmax(x), torch.randn(64, 32, 3, 3, 3, device=DEV))


def test_planes_cache_follows_the_weight_tensor():
    """The weights' planes are kept per weight TENSOR and version: a second call reuses them, an in-place update makes them
    again, and a different tensor that happens to get the same storage address never sees the old ones."""
    from uforecon_amd import ops

    torch.manual_seed(5)
    x = torch.randn(1, 2, 8, 32, 16, device=DEV)
    am = ops.absmax(x)
    w = torch.randn(16, 16, 3, 3, 3, device=DEV) * 0.1
    y1, _ = ops.conv3d_planes(x, am, w)
    y2, _ = ops.conv3d_planes(x, am, w)                       # planes_ready = 1
    assert torch.equal(y1, y2)
    with torch.no_grad():
        w.mul_(2.0)                                           # version bump: planes re-made
    y3, _ = ops.conv3d_planes(x, am, w)
    assert float((y3 - 2.0 * y1).abs().max()) < 1e-5 * float(y1.abs().max())
    yf, _ = ops.conv3d_planes(x, am, w, flip=True)            # the mirrored planes of the same tensor are another entry state
    ref = ops.conv3d_bwd_data(x, w, ops.CONV3D_S1, tuple(x.shape))
    assert float((yf - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    ptr = w.data_ptr()
    del w
    w2 = torch.randn(16, 16, 3, 3, 3, device=DEV) * 0.1        # usually the freed block again
    y4, _ = ops.conv3d_planes(x, am, w2)
    ref4 = ops.conv3d(x, w2, ops.CONV3D_S1)
    assert float((y4 - ref4).abs().max()) < 1e-5 * float(ref4.abs().max()), (ptr, w2.data_ptr())


@pytest.mark.parametrize("cin,cout,mode,shape", [(8, 8, 0, (2, 3, 10, 70)), (16, 16, 0, (1, 4, 9, 37)), (32, 32, 0, (2, 2, 6, 20)), (64, 64, 0, (1, 2, 5, 18)),
                                                 (8, 16, 1, (1, 4, 8, 64)), (16, 32, 1, (2, 2, 8, 36)), (32, 64, 1, (1, 2, 4, 20)),
                                                 (16, 8, 2, (1, 2, 4, 32)), (32, 16, 2, (2, 1, 4, 18)), (64, 32, 2, (1, 1, 4, 16)), (8, 1, 0, (2, 3, 10, 70)),
                                                 (1, 8, 0, (2, 3, 10, 70)), (1, 8, 0, (1, 8, 16, 64))])
def test_weight_gradient_on_the_matrix_cores_matches_float64(cin, cout, mode, shape):
    """ufr_conv3d_bwd_weight now runs csrc/conv3d_wgrad_planes.hip for every layer shape but conv0: d weight (and d bias)
    against a float64 autograd of the same layer; operands carry 16 significand bits (bf16 hi + lo), the sum is fp32."""
    from uforecon_amd import ops

    torch.manual_seed(7 * cin + cout + mode)
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, device=DEV)
    wshape = (cin, cout, 3, 3, 3) if mode == 2 else (cout, cin, 3, 3, 3)
    w = torch.zeros(wshape, dtype=torch.float64, device=DEV, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, device=DEV, requires_grad=True)
    xp = x.double().permute(0, 4, 1, 2, 3)
    F = torch.nn.functional
    y = (F.conv_transpose3d(xp, w, b, stride=2, padding=1, output_padding=1) if mode == 2 else F.conv3d(xp, w, b, stride=1 + mode, padding=1))
    dy = torch.randn(y.shape, device=DEV).float()
    (y * dy.double()).sum().backward()
    dw, db = ops.conv3d_bwd_weight(x, dy.permute(0, 2, 3, 4, 1).contiguous(), mode, wshape)
    ew = float((dw.double() - w.grad).abs().max() / w.grad.abs().max())
    eb = float((db.double() - b.grad).abs().max() / b.grad.abs().max())
    print(f"wgrad cin {cin} cout {cout} mode {mode} {shape}: d weight {ew:.2e}, d bias {eb:.2e} of scale")
    assert ew < 2e-5 and eb < 2e-5


def test_heads_weight_gradients_in_one_pass():
    from uforecon_amd import ops

    torch.manual_seed(11)
    B, D, H, W = 2, 3, 9, 40
    x = torch.randn(B, D, H, W, 8, device=DEV)
    df, dw1 = torch.randn(B, D, H, W, 8, device=DEV), torch.randn(B, D, H, W, 1, device=DEV)
    gf, gw = ops.conv3d_bwd_weight_heads(x, df, dw1)
    rf = ops.conv3d_bwd_weight(x, df, ops.CONV3D_S1, (8, 8, 3, 3, 3), want_bias=False)[0]
    rw = ops.conv3d_bwd_weight(x, dw1, ops.CONV3D_S1, (1, 8, 3, 3, 3), want_bias=False)[0]
    assert float((gf - rf).abs().max()) < 1e-5 * float(rf.abs().max()) and float((gw - rw).abs().max()) < 1e-5 * float(rw.abs().max())
