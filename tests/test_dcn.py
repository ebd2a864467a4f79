"""Deformable convolution + FeatureNet (the backbone).  torchvision is absent, so the oracle
(oracle/dcn_oracle.py, "parity unpinned") is checked by the properties its published algorithm must have, the HIP kernel
against the oracle, and the plain-convolution part of the mirror against the reference's own FeatureNet modules."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_oracle as D
from uforecon_amd.scene import _unit_uniform, fill_state_dict


def _rand(shape, seed, scale=1.0):
    return _unit_uniform(shape, torch.Generator().manual_seed(seed)) * scale


def test_oracle_zero_offset_unit_mask_is_conv2d():
    x, w, b = _rand((2, 8, 9, 11), 1), _rand((16, 8, 3, 3), 2, 0.2), _rand((16,), 3)
    got = D.deform_conv2d(x, torch.zeros(2, 18, 9, 11), w, b, 1, 1, 1, torch.ones(2, 9, 9, 11))
    assert float((got - F.conv2d(x, w, b, padding=1)).abs().max()) < 5e-6


def test_oracle_integer_offsets_are_a_shift_and_mask_is_linear():
    x, w = _rand((1, 4, 10, 12), 4), _rand((8, 4, 3, 3), 5, 0.3)
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2], off[:, 1::2] = 1.0, -2.0                      # dy = +1, dx = -2 on every tap
    shifted = torch.zeros_like(x)
    shifted[:, :, :-1, 2:] = x[:, :, 1:, :-2]                   # shifted[y, x] = x[y + 1, x - 2], zero outside
    got = D.deform_conv2d(x, off, w, None, 1, 1, 1, torch.ones(1, 9, 10, 12))
    # interior only: at the border conv2d pads the SHIFTED image, the deformable taps still see the original one
    assert float((got - F.conv2d(shifted, w, padding=1))[:, :, 1:-2, 3:-1].abs().max()) < 5e-6
    m = torch.rand(1, 9, 10, 12, generator=torch.Generator().manual_seed(6))
    a = D.deform_conv2d(x, off, w, None, 1, 1, 1, m)
    b = D.deform_conv2d(x, off, w, None, 1, 1, 1, 2.0 * m)
    assert float((2 * a - b).abs().max()) < 1e-5


def test_oracle_out_of_image_samples_contribute_zero():
    x, w = torch.ones(1, 4, 6, 6), torch.ones(8, 4, 3, 3)
    far = torch.full((1, 18, 6, 6), 100.0)
    assert float(D.deform_conv2d(x, far, w, None, 1, 1, 1, None).abs().max()) == 0.0
    edge = torch.zeros(1, 18, 6, 6)
    edge[:, 0::2] = -0.5                                         # half a pixel up: the top row of taps straddles y = -1 .. 0
    out = D.deform_conv2d(x, edge, w, None, 1, 1, 1, None)
    assert 0 < float(out[0, 0, 0, 3]) < float(out[0, 0, 3, 3])  # partially outside: partial weight, not zero, not full


def test_mirror_keys_and_plain_convolutions_match_the_reference():
    """The reference FeatureNet constructs without torchvision (only its forward needs deform_conv2d): same state_dict
    keys, and the plain-convolution trunk (conv0/conv1/conv2, inner1/inner2) gives the reference's numbers."""
    ref_dir = "/root/reference"
    if not os.path.isdir(os.path.join(ref_dir, "code1")):
        pytest.skip("reference tree not present (GPU box)")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from ref_harness import import_reference
    import_reference()
    from code1.encoder_utils.fmt.module import FeatureNet as Ref
    from uforecon_amd import featurenet as FN

    r, m = fill_state_dict(Ref(base_channels=8), 11).eval(), fill_state_dict(FN.FeatureNet(8), 11).eval()
    assert set(r.state_dict()) == set(m.state_dict())
    x = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        for name in ("conv0", "conv1", "conv2"):
            x_r = getattr(r, name)(x)
            assert torch.equal(x_r, getattr(m, name)(x)), name
            x = x_r
        assert torch.equal(r.inner2(r.conv0(torch.ones(1, 3, 8, 8))), m.inner2(m.conv0(torch.ones(1, 3, 8, 8))))


@pytest.mark.gpu
@pytest.mark.parametrize("C,Cout,H,W", [(32, 32, 17, 23), (32, 16, 8, 40), (32, 8, 33, 9), (8, 8, 5, 7)])
def test_hip_deform_conv2d_matches_oracle(C, Cout, H, W):
    from uforecon_amd import featurenet as FN

    dev = "cuda:0"
    x, w, b = _rand((2, C, H, W), 10), _rand((Cout, C, 3, 3), 11, 0.2), _rand((Cout,), 12)
    off = _rand((2, 18, H, W), 13, 2.5)                         # many taps leave the image at these sizes
    off[0, :, 0, 0] = 0.0                                        # exact integer positions too
    off[0, :, 1, 1] = -1.0                                       # y, x = -1 .. +1: the "h <= -1" boundary itself
    m = torch.rand(2, 9, H, W, generator=torch.Generator().manual_seed(14))
    want = D.deform_conv2d(x, off, w, b, 1, 1, 1, m)
    got = FN.deform_conv2d(x.to(dev), off.to(dev), w.to(dev), b.to(dev), 1, 1, 1, mask=m.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    got2 = FN.deform_conv2d(x.to(dev), off.to(dev), w.to(dev), None, 1, 1, 1, mask=None).cpu()
    want2 = D.deform_conv2d(x, off, w, None, 1, 1, 1, None)
    assert float((got2 - want2).abs().max()) <= 2e-5 * float(want2.abs().max())


@pytest.mark.gpu
def test_featurenet_gpu_matches_cpu_with_oracle_dcn():
    """Whole backbone on the GPU (library convolutions + the HIP deformable convolution) against the same modules on the
    CPU with the oracle's deform_conv2d swapped in (test harness only)."""
    from uforecon_amd import featurenet as FN

    m = fill_state_dict(FN.FeatureNet(8), 12).eval()
    with torch.no_grad():                                        # non-zero offset/mask convolutions (they are zero-init)
        for mod in m.modules():
            if isinstance(mod, FN.DCN):
                mod.conv_offset_mask.weight.copy_(_rand(tuple(mod.conv_offset_mask.weight.shape), 7, 0.05))
                mod.conv_offset_mask.bias.copy_(_rand((27,), 8, 0.3))
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(2))
    saved = FN.deform_conv2d
    FN.deform_conv2d = lambda i, o, w, b, s, p, d, mask=None: D.deform_conv2d(i, o, w, b, s, p, d, mask)
    try:
        with torch.no_grad():
            want = m(x)
    finally:
        FN.deform_conv2d = saved
    with torch.no_grad():
        got = m.to("cuda:0")(x.to("cuda:0"))
    for st, c in (("stage1", 32), ("stage2", 16), ("stage3", 8)):
        assert got[st].shape == want[st].shape and got[st].shape[1] == c
        assert float((got[st].cpu() - want[st]).abs().max()) <= 1e-3 * float(want[st].abs().max()), st
    with pytest.raises(FN.UfrError):
        m.cpu()(x)                                               # no CPU path in the product
