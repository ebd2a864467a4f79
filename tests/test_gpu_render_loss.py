"""ufr_render_loss / autograd.RenderLoss / UFORecon.training_loss (code1/model.py:552-566) against the torch restatement of
the reference's expression (oracle/ufo_oracle.py:training_loss_terms): value, the four logged terms, and the gradients
autograd sends into both passes' colours and depths."""
import pytest
import torch

from oracle import ufo_oracle as O
from uforecon_amd import autograd as ag
from uforecon_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, RN, seed, valid="some"):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    nf = torch.stack([torch.tensor([[2.0, 6.0]] * 3) + 0.1 * b for b in range(B)])          # (B,V,2)
    depth_gt = 1.0 + 6.0 * r(B, RN)                     # some below near, some beyond far
    depth_gt[:, ::7] = 0.0                              # holes of the GT depth map
    if valid == "none":
        depth_gt = torch.zeros(B, RN)
    if valid == "ties":                                 # a rendered depth that equals the GT: sign(0) = 0
        depth_gt[:] = 3.0
    depth = 2.0 + 4.0 * r(B, RN)
    depth2 = 2.0 + 4.0 * r(B, RN)
    if valid == "ties":
        depth[:, ::3] = 3.0
    t = dict(rgb=r(B, RN, 3), depth=depth, rgb2=r(B, RN, 3), depth2=depth2, rgb_gt=r(B, RN, 3), depth_gt=depth_gt, nf=nf)
    return {k: v.to(DEV) for k, v in t.items()}


@pytest.mark.parametrize("B,RN,valid,w", [(1, 1024, "some", (1.0, 1.0)), (2, 700, "some", (0.5, 2.0)), (1, 333, "none", (1.0, 1.0)),
                                          (1, 4096, "ties", (1.0, 0.25)), (3, 5, "some", (1.0, 1.0))])
def test_render_loss_matches_the_reference_expression(B, RN, valid, w):
    t = _case(B, RN, 7 * B + RN, valid)
    leaves = [t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "rgb2", "depth2")]
    ref, ref_parts = O.training_loss_terms(*leaves, t["rgb_gt"], t["depth_gt"], t["nf"], *w)
    # upstream gradient != 1: the node scales its stored cotangents
    gref = torch.autograd.grad(3.0 * ref, leaves, allow_unused=True)
    mine = [t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "rgb2", "depth2")]
    loss, parts = ag.RenderLoss.apply(*mine, t["rgb_gt"], t["depth_gt"], t["nf"], *w)
    assert loss.shape == () and parts.shape == (4,) and not parts.requires_grad
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    for a, b in zip(parts.tolist(), ref_parts):
        assert abs(a - float(b)) <= 2e-6 * max(1.0, abs(float(b)))
    gm = torch.autograd.grad(3.0 * loss, mine)
    for a, b, k in zip(gm, gref, ("rgb", "depth", "rgb2", "depth2")):
        b = torch.zeros_like(a) if b is None else b
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6), k


def test_render_loss_is_deterministic_and_rejects_bad_shapes():
    t = _case(1, 1024, 3)
    a = ops.render_loss(t["rgb"], t["depth"], t["rgb2"], t["depth2"], t["rgb_gt"], t["depth_gt"], t["nf"])
    b = ops.render_loss(t["rgb"], t["depth"], t["rgb2"], t["depth2"], t["rgb_gt"], t["depth_gt"], t["nf"])
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    with pytest.raises(Exception):
        ops.render_loss(t["rgb"][:, :5], t["depth"], t["rgb2"], t["depth2"], t["rgb_gt"], t["depth_gt"], t["nf"])
    with pytest.raises(Exception):
        ops.render_loss(t["rgb"].cpu(), t["depth"].cpu(), t["rgb2"].cpu(), t["depth2"].cpu(), t["rgb_gt"].cpu(), t["depth_gt"].cpu(), t["nf"].cpu())


def test_render_loss_second_backward_is_an_error_not_a_type_error():
    """The node frees its stored cotangents in the first backward (like any saved buffer): differentiating the same loss
    again says so."""
    from uforecon_amd import autograd as ag

    t = _case(1, 64, 5)
    ins = [t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "rgb2", "depth2")]
    loss, _ = ag.RenderLoss.apply(*ins, t["rgb_gt"], t["depth_gt"], t["nf"], 1.0, 1.0)
    loss.backward(retain_graph=True)
    assert all(x.grad is not None for x in ins)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
