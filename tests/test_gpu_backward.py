"""GPU parity of the backward kernels (BASELINE configs[4]: training step through the HIP path).

Three levels, all through the C ABI:
  * each ``ufr_*_bwd`` entry point against autograd through the CPU oracle on the same inputs (this host),
  * ``loss.backward()`` through ``UFORecon.infer(extract_geometry=False)`` -- the call of the reference's
    ``training_step`` (code1/model.py:540-566) -- against the REFERENCE's own autograd gradients
    (tests/golden/c5_train_grads*.npz: every per-ray parameter and the six sampled volumes),
  * structural properties: linearity of the adjoints, gradient accumulation, chunk invariance.
Tolerance: 1e-3 of each gradient tensor's scale (fp32 path; the forward runs the fp16x3 matrix path, the backward
recomputes it on the fp32 MFMA).
"""
import argparse
import os

import pytest
import torch

from helpers import CASES, VOLUME_KEYS, case_inputs, golden_volume_grad, grad_rel_err, load_weights, rel_err
from oracle import ufo_oracle as O
from uforecon_amd import model as M
from uforecon_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_TOL = 1e-3
SHIFT_BIAS = "ray_transformer.linear_radianceweight_1_softmax.4.bias"


def _args(c):
    return argparse.Namespace(extract_geometry=False, test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                              coarse_sample=c["coarse"], fine_sample=c["fine"], volume_type="correlation", volume_reso=96,
                              mvs_depth_guide=1, depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
                              test_coarse_only=False, test_ray_num=800)


def _loss_from_tuple(r, batch, ray_idx):
    """model.py:552-566 on the 17-tuple infer returns (caller-side torch ops, like the Lightning hook)."""
    d = dict(rgb=r[1][0], depth=r[2][0], rgb_2=r[8][0], depth_2=r[9][0])
    return O.training_loss(d, batch, ray_idx)


@pytest.mark.parametrize("SN", [64, 128, 96])
def test_composite_bwd_matches_oracle_autograd(SN):
    g = torch.Generator().manual_seed(SN)
    RN = 37
    z = torch.sort(torch.rand(RN, SN, generator=g) * 2 + 2, dim=1)[0]
    srdf = (torch.rand(RN, SN, generator=g) - 0.5) * 0.5 - (z - 3.0) * 0.3
    rad = torch.rand(RN, SN, 3, generator=g)
    var = torch.tensor(0.3)
    co = [torch.rand(RN, 3, generator=g), torch.rand(RN, generator=g), torch.rand(RN, generator=g), torch.rand(RN, SN, generator=g)]
    srdf_r, rad_r, var_r = srdf.clone().requires_grad_(True), rad.clone().requires_grad_(True), var.clone().requires_grad_(True)
    rgb, depth, opacity, w, _ = O.composite(z, rad_r, srdf_r, var_r)
    ((rgb * co[0]).sum() + (depth * co[1]).sum() + (opacity * co[2]).sum() + (w * co[3]).sum()).backward()
    d_rad, d_srdf, d_var = ops.composite_bwd(z.to(DEV), rad.to(DEV), srdf.to(DEV), var.reshape(1).to(DEV),
                                             *[t.to(DEV) for t in co])
    assert grad_rel_err(d_rad, rad_r.grad) < 1e-5
    assert grad_rel_err(d_srdf, srdf_r.grad) < 1e-4
    assert abs(float(d_var) - float(var_r.grad)) < 1e-4 * abs(float(var_r.grad))
    # optional cotangents: NULL means zero
    d_rad2, d_srdf2, _ = ops.composite_bwd(z.to(DEV), rad.to(DEV), srdf.to(DEV), var.reshape(1).to(DEV), None,
                                           co[1].to(DEV), None, None)
    rgb, depth, opacity, w, _ = O.composite(z, rad, srdf_r, var)
    srdf_r.grad = None
    (depth * co[1]).sum().backward()
    assert grad_rel_err(d_srdf2, srdf_r.grad) < 1e-4
    assert float(d_rad2.abs().max()) == 0.0


def _token_inputs(name, tag="coarse"):
    """Token inputs of one pass from the HIP gather (so both sides see identical x / colours / masks / dirs)."""
    fr, idx, U1, U2, g = case_inputs(name)
    P = load_weights()
    W = ops.PackedWeights({k: v.to(DEV) for k, v in P.items()})
    f = fr.to(DEV)
    fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
    i = idx.reshape(-1)
    ray_d = fr.batch["ray_d"][0][:, i].t().contiguous().to(DEV)
    ray_o = fr.batch["ray_o"][0].contiguous().to(DEV)
    RN = i.numel()
    near = fr.batch["near_fars"][0, 0, 0].expand(RN).contiguous().to(DEV)
    far = fr.batch["near_fars"][0, 0, 1].expand(RN).contiguous().to(DEV)
    z = ops.sample_fixed(near, far, U1.to(DEV))
    x, rgbm, dirs, dbg = ops.project_gather(fh, W, ray_o, ray_d, z, debug=True)
    return fr, P, W, fh, ray_o, ray_d, z, x, rgbm, dirs, dbg


@pytest.mark.parametrize("name", ["c5_train_grads", "c5_train_grads_nv4"])
def test_aggregate_bwd_matches_oracle_autograd(name):
    fr, P, W, fh, ray_o, ray_d, z, x, rgbm, dirs, dbg = _token_inputs(name)
    RN, SN = z.shape
    NV = x.shape[1]
    radiance, srdf, agg = ops.aggregate(W, x, rgbm, dirs, RN, SN, keep_workspace=True)
    g = torch.Generator().manual_seed(3)
    co_rad, co_srdf = torch.rand(RN * SN, 3, generator=g) - 0.5, torch.rand(RN, SN, generator=g) - 0.5
    # oracle autograd on the same token inputs
    Pg = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in P.items()}
    xr = x.cpu().clone().requires_grad_(True)
    rad_o, srdf_o = O.aggregate_tokens(Pg, xr, rgbm.cpu()[..., :3], rgbm.cpu()[..., 3], dirs.cpu()[..., :3], RN, SN)
    assert rel_err(radiance, rad_o) < 1e-4 and rel_err(srdf, srdf_o) < 1e-4
    ((rad_o * co_rad).sum() + (srdf_o * co_srdf).sum()).backward()
    grads = ops.GradBuffer(DEV)
    d_pv, _ = ops.aggregate_bwd(W, grads, x, rgbm, dirs, agg["token0"], RN, SN, co_rad.to(DEV), co_srdf.to(DEV))
    torch.cuda.synchronize()
    for k in ops.RAW_WEIGHT_KEYS:
        if "pre_sim_mlp" in k or k == "deviation_network.variance":
            continue
        if k == SHIFT_BIAS:   # true gradient is zero (softmax over views is shift-invariant): both sides hold rounding noise
            scale = float(Pg[k.replace("bias", "weight")].grad.abs().max())
            assert float(grads.grad(k).abs().max()) < 1e-2 * scale
            continue
        assert grad_rel_err(grads.grad(k), Pg[k].grad) < GRAD_TOL, k
    ref_pv = xr.grad[:, :, 32:72].sum(1)
    assert grad_rel_err(d_pv, ref_pv) < GRAD_TOL
    # gradients accumulate: a second call doubles them
    d_pv2, _ = ops.aggregate_bwd(W, grads, x, rgbm, dirs, agg["token0"], RN, SN, co_rad.to(DEV), co_srdf.to(DEV))
    k = "ray_transformer.density_view_transformer.layers.0.mlp.0.weight"
    assert grad_rel_err(grads.grad(k), 2 * Pg[k].grad) < GRAD_TOL
    assert torch.equal(d_pv2, d_pv)


def _aggregate_bwd_against_autograd(NV, SN, P, f64=False):
    from uforecon_amd.scene import make_frame

    W = ops.PackedWeights({k: v.to(DEV) for k, v in P.items()})
    fr = make_frame(48, 64, NV, seed=40 + NV, train_layout=True)
    f = fr.to(DEV)
    fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
    RN = 6
    idx = torch.arange(RN) * 397 + 411
    gen = torch.Generator().manual_seed(NV)
    ray_d = fr.batch["ray_d"][0][:, idx].t().contiguous().to(DEV)
    ray_o = fr.batch["ray_o"][0].contiguous().to(DEV)
    near = fr.batch["near_fars"][0, 0, 0].expand(RN).contiguous().to(DEV)
    far = fr.batch["near_fars"][0, 0, 1].expand(RN).contiguous().to(DEV)
    z = ops.sample_fixed(near, far, torch.rand(SN, RN, generator=gen).to(DEV))
    x, rgbm, dirs, _ = ops.project_gather(fh, W, ray_o, ray_d, z)
    radiance, srdf, agg = ops.aggregate(W, x, rgbm, dirs, RN, SN, keep_workspace=True)
    co_rad, co_srdf = torch.rand(RN * SN, 3, generator=gen) - 0.5, torch.rand(RN, SN, generator=gen) - 0.5
    dt = torch.float64 if f64 else torch.float32
    Pg = {k: v.clone().to(dt).requires_grad_("depthcode" not in k) for k, v in P.items()}
    xr = x.cpu().clone().to(dt).requires_grad_(True)
    rad_o, srdf_o = O.aggregate_tokens(Pg, xr, rgbm.cpu()[..., :3].to(dt), rgbm.cpu()[..., 3].to(dt), dirs.cpu()[..., :3].to(dt), RN, SN)
    ((rad_o * co_rad.to(dt)).sum() + (srdf_o * co_srdf.to(dt)).sum()).backward()
    grads = ops.GradBuffer(DEV)
    d_pv, _ = ops.aggregate_bwd(W, grads, x, rgbm, dirs, agg["token0"], RN, SN, co_rad.to(DEV), co_srdf.to(DEV))
    assert ops.status_poll(True) == 0
    worst = {}
    for k in ops.RAW_WEIGHT_KEYS:
        if "pre_sim_mlp" in k or k in ("deviation_network.variance", SHIFT_BIAS):
            continue
        worst[k] = grad_rel_err(grads.grad(k), Pg[k].grad)
    worst["d_pv"] = grad_rel_err(d_pv, xr.grad[:, :, 32:72].sum(1))
    return worst


@pytest.mark.parametrize("NV,SN", [(2, 48), (5, 96), (7, 32)])
def test_aggregate_bwd_other_view_counts_and_lengths(NV, SN):
    """The backward tiles hold 16 // (NV+1) points per 16 token columns and the ray kernel walks SN / 16 tiles per sweep:
    view counts with idle columns and sample totals that are not a power of two, against autograd through the oracle."""
    worst = _aggregate_bwd_against_autograd(NV, SN, load_weights())
    # a ReLU unit within rounding of zero may flip between two fp32 evaluations: allow it on a few tensors.
    # The median sits at the arithmetic's own floor: since round 4 the data-gradient chain and the weight-gradient
    # contraction run as three bf16 plane products (16 significand bits per operand: ~1e-5 per tensor; GRAD_TOL is 1e-3)
    assert sorted(worst.values())[len(worst) // 2] < 5e-5, worst
    assert max(worst.values()) < 2e-2, worst


def test_aggregate_bwd_with_checkpoint_like_weights():
    """The backward under plane exponents that differ from layer to layer: every dense matrix x 8, LayerNorm
    gains up to 10 -- the tape build runs on exponents chosen for THESE weights, the data-gradient chain on the unscaled
    bf16 planes of the same matrices.  Against autograd through the oracle in float64 (the float32 evaluation of such a
    network amplifies its own rounding)."""
    g = torch.Generator().manual_seed(3)
    P = {k: v.clone() for k, v in load_weights().items()}
    for k in P:
        if not k.startswith("ray_transformer.") or "view_token" in k or "pre_sim" in k or "depthcode" in k:
            continue
        if P[k].dim() == 2:
            P[k] *= 8.0
        elif "norm" in k and k.endswith("weight"):
            P[k] = 0.5 + 9.5 * torch.rand(P[k].shape, generator=g)
    worst = _aggregate_bwd_against_autograd(3, 64, P, f64=True)
    print({k.split(".")[-3] + "." + k.split(".")[-2] if k.count(".") > 2 else k: f"{v:.1e}" for k, v in worst.items()})
    assert sorted(worst.values())[len(worst) // 2] < 1e-4, worst
    assert max(worst.values()) < 1e-3, worst


def test_project_gather_bwd_matches_oracle_autograd():
    name = "c5_train_grads"
    fr, P, W, fh, ray_o, ray_d, z, x, rgbm, dirs, dbg = _token_inputs(name)
    RN, SN = z.shape
    g = torch.Generator().manual_seed(4)
    d_pv = (torch.rand(RN * SN, 40, generator=g) - 0.5)
    # oracle: vol24 and pre_sim_mlp(sim8) as functions of the volumes / the MLP weights
    Pg = {k: v.clone().requires_grad_("pre_sim_mlp" in k) for k, v in P.items()}
    vols = {st: {k: v.clone().requires_grad_(True) for k, v in fr.feature_volume[st].items()} for st in fr.feature_volume}
    pts = (ray_o.cpu()[None, None, :] + z.cpu()[..., None] * ray_d.cpu()[:, None, :])
    vol24 = O.volume_lookup(fr.batch["source_poses"][0], pts, vols, fr.batch["near_fars"][0][0])
    sim16 = O.mlp3(dbg["sim8"].cpu(), Pg, "ray_transformer.pre_sim_mlp.")
    ((vol24.reshape(-1, 24) * d_pv[:, :24]).sum() + (sim16 * d_pv[:, 24:]).sum()).backward()
    grads = ops.GradBuffer(DEV)
    gf = [torch.zeros_like(fr.feature_volume[st]["feature_volume"], device=DEV) for st in ("stage1", "stage2", "stage3")]
    gw = [torch.zeros_like(fr.feature_volume[st]["weight_volume"], device=DEV) for st in ("stage1", "stage2", "stage3")]
    ops.project_gather_bwd(fh, W, grads, ray_o, ray_d, z, dbg["sim8"], d_pv.to(DEV), gf, gw)
    torch.cuda.synchronize()
    for i, st in enumerate(("stage1", "stage2", "stage3")):
        assert grad_rel_err(gf[i], vols[st]["feature_volume"].grad) < 1e-4, st
        assert grad_rel_err(gw[i], vols[st]["weight_volume"].grad) < 1e-4, st
    for k in ops.RAW_WEIGHT_KEYS:
        if "pre_sim_mlp" in k:
            assert grad_rel_err(grads.grad(k), Pg[k].grad) < 1e-4, k


def _train_step(name, **model_options):
    c = CASES[name]
    fr, idx, U1, U2, g = case_inputs(name)
    m = M.UFORecon(_args(c), **model_options).to(DEV)
    m.load_state_dict(load_weights(), strict=True)
    m.train()
    f = fr.to(DEV)
    for st in f.feature_volume:
        for k in f.feature_volume[st]:
            f.feature_volume[st][k].requires_grad_(True)
    r = m.infer(f.batch, idx.to(DEV), f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature,
                uniforms=(U1, U2))
    loss = _loss_from_tuple(r, f.batch, idx.to(DEV))
    return m, f, r, loss, g


@pytest.mark.parametrize("overlap,tape_in_forward", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("name", ["c5_train_grads", "c5_train_grads_nv4"])
def test_training_step_gradients_match_reference_autograd(name, overlap, tape_in_forward):
    """loss.backward() through UFORecon.infer == the reference's autograd (golden), every parameter and volume -- with the
    backward's independent stages side by side on three streams (UFORecon(overlap=True), the default) and all on one; with
    the tape recorded by the forward (the default) and by the backward (what unaligned pools fall back to).  Both are
    arguments of the model (autograd.RenderOptions): nothing process-wide is flipped."""
    m, f, r, loss, g = _train_step(name, overlap=overlap, tape_in_forward=tape_in_forward)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    torch.cuda.synchronize()
    worst = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        worst[k] = grad_rel_err(p.grad, g["grad." + k])
    for key in VOLUME_KEYS:
        st, k = key.split(".")
        v = f.feature_volume[st][k]
        worst[key] = grad_rel_err(v.grad, golden_volume_grad(g, key, v.shape))
    bad = {k: e for k, e in worst.items() if not e < GRAD_TOL}
    assert not bad, bad
    assert r[16].requires_grad                       # variance output stays differentiable (train/variance log)


@pytest.mark.parametrize("second_taped", [True, False])
def test_two_forwards_before_one_backward_keep_their_own_tapes(second_taped):
    """The training forward records the backward's tape into a workspace that is kept between steps (autograd._acquire):
    a second forward BEFORE the first one's backward (gradient accumulation over two ray batches, one loss) must not
    overwrite it -- it gets a private buffer.  (a + b).backward() == a.backward() then b.backward(), up to the order of
    the float atomics.  second_taped = False: the second forward does NOT record its tape (what a pool that does not end
    on a tape block falls back to), so ITS backward records one -- into a private buffer as well while the first
    forward's tape is alive in the kept one (round-4 advisor finding: it used to take the kept buffer unasked)."""
    name = "c5_train_grads"
    c = CASES[name]
    fr, idx, U1, U2, g = case_inputs(name)
    f = fr.to(DEV)
    for st in f.feature_volume:
        for k in f.feature_volume[st]:
            f.feature_volume[st][k].requires_grad_(True)
    idx2 = (idx + 7).clamp_max(c["H"] * c["W"] - 1)

    def run(joint: bool):
        m = M.UFORecon(_args(c)).to(DEV)
        m.load_state_dict(load_weights(), strict=True)
        m.train()
        for st in f.feature_volume:
            for k in f.feature_volume[st]:
                f.feature_volume[st][k].grad = None
        losses = []
        for n, ix in enumerate((idx, idx2)):
            m.tape_in_forward = second_taped or n == 0
            r = m.infer(f.batch, ix.to(DEV), f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature, uniforms=(U1, U2))
            losses.append(_loss_from_tuple(r, f.batch, ix.to(DEV)))
            if not joint:
                losses[-1].backward()
        if joint:
            (losses[0] + losses[1]).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters()}

    ga, gb = run(True), run(False)
    for k in ga:
        assert grad_rel_err(ga[k], gb[k]) < 2e-4, k


def test_parameter_only_backward_matches_and_skips_the_volume_scatter():
    """Frustums without requires_grad (e.g. a frozen cost_reg_2): parameter gradients are unchanged and no volume gradient
    is produced."""
    name = "c5_train_grads_nv4"
    m, f, r, loss, g = _train_step(name)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    c = CASES[name]
    fr, idx, U1, U2, _ = case_inputs(name)
    m2 = M.UFORecon(_args(c)).to(DEV)
    m2.load_state_dict(load_weights(), strict=True)
    f2 = fr.to(DEV)
    r2 = m2.infer(f2.batch, idx.to(DEV), f2.source_imgs_feat, f2.feature_volume, match_feature=f2.match_feature, uniforms=(U1, U2))
    _loss_from_tuple(r2, f2.batch, idx.to(DEV)).backward()
    for k, p in m2.named_parameters():
        if k != SHIFT_BIAS:
            assert grad_rel_err(p.grad, ref[k]) < 1e-4, k
    assert all(v.grad is None for st in f2.feature_volume.values() for v in st.values())


def test_optimizer_step_follows_the_gradients():
    """One optimizer step through the HIP backward (plain gradient descent, small enough for the first-order decrease
    to hold; the reference trains with Adam, model.py:72-87): every per-ray parameter moves, the packed copy follows,
    and the loss of the same batch goes down."""
    name = "c5_train_grads"
    m, f, r, loss, g = _train_step(name)
    opt = torch.optim.SGD(m.parameters(), lr=1e-4)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    loss.backward()
    opt.step()
    changed = {k for k, p in m.named_parameters() if not torch.equal(p.detach(), before[k])}
    assert set(before) - changed <= {SHIFT_BIAS}      # that bias has no gradient (softmax over views is shift-invariant)
    c = CASES[name]
    fr, idx, U1, U2, _ = case_inputs(name)
    with torch.no_grad():
        r2 = m.infer(f.batch, idx.to(DEV), f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature,
                     uniforms=(U1, U2))
    loss2 = _loss_from_tuple(r2, f.batch, idx.to(DEV))
    assert float(loss2) < float(loss)


def test_full_size_training_step_properties():
    """BASELINE configs[4] at its real size (1024 random rays of a 512x640 frame, 3 views, 64+64 samples), checked through
    size-independent properties of the adjoints: (a) the backward is LINEAR in the output cotangents -- gradients of
    a L1 + b L2 equal a grad(L1) + b grad(L2); (b) a directional finite difference of the loss along a random parameter
    direction agrees with <grad, direction>; (c) volume gradients are exactly zero outside the voxels the rays touch."""
    from uforecon_amd.scene import make_frame

    H, W, NV, RN = 512, 640, 3, 1024
    c = dict(coarse=64, fine=64)
    fr = make_frame(H, W, NV, seed=9, train_layout=True)
    f = fr.to(DEV)
    vols = [f.feature_volume[st][k] for st in f.feature_volume for k in f.feature_volume[st]]
    for v in vols:
        v.requires_grad_(True)
    m = M.UFORecon(_args(c)).to(DEV)
    m.load_state_dict(load_weights(), strict=True)
    g = torch.Generator().manual_seed(17)
    idx = torch.randperm(H * W, generator=g)[:RN][None].to(DEV)
    U = (torch.rand(64, RN, generator=g), torch.rand(64, RN, generator=g))
    params = list(m.parameters())

    def outputs():
        r = m.infer(f.batch, idx, f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature, uniforms=U)
        return r[1][0], r[2][0], r[8][0], r[9][0], r[0][0], r[3][0]      # rgb, depth, rgb_2, depth_2, rgb_gt, depth_gt

    def grads_of(loss):
        gs = torch.autograd.grad(loss, params + vols, allow_unused=True)
        return [torch.zeros_like(p) if g_ is None else g_ for g_, p in zip(gs, params + vols)]

    rgb, depth, rgb2, depth2, rgb_gt, depth_gt = outputs()
    L1 = ((rgb - rgb_gt) ** 2).mean() + ((rgb2 - rgb_gt) ** 2).mean()
    L2 = (depth - depth_gt).abs().mean() + (depth2 - depth_gt).abs().mean()
    g1 = grads_of(L1)
    rgb, depth, rgb2, depth2, rgb_gt, depth_gt = outputs()
    L2 = (depth - depth_gt).abs().mean() + (depth2 - depth_gt).abs().mean()
    g2 = grads_of(L2)
    rgb, depth, rgb2, depth2, rgb_gt, depth_gt = outputs()
    L12 = 0.7 * (((rgb - rgb_gt) ** 2).mean() + ((rgb2 - rgb_gt) ** 2).mean()) \
        - 1.3 * ((depth - depth_gt).abs().mean() + (depth2 - depth_gt).abs().mean())
    g12 = grads_of(L12)
    names = [k for k, _ in m.named_parameters()]
    shift_w = g12[names.index(SHIFT_BIAS.replace("bias", "weight"))]
    for n, (a, b, ab) in enumerate(zip(g1, g2, g12)):                   # (a) linearity (atomics reorder sums: 1e-4)
        if n < len(names) and names[n] == SHIFT_BIAS:
            # true gradient zero (softmax over views is shift-invariant): all three hold rounding noise, not a linear map
            assert float(ab.abs().max()) < 1e-2 * float(shift_w.abs().max())
            continue
        want = 0.7 * a - 1.3 * b
        assert float((ab - want).abs().max()) <= 2e-4 * max(float(want.abs().max()), 1e-6)
    # (c) sparsity: a frustum voxel no sample touches receives exactly zero
    touched = sum(int((gv != 0).sum()) for gv in g12[len(params):])
    total = sum(gv.numel() for gv in g12[len(params):])
    assert 0 < touched < 0.2 * total
    # (b) directional derivative along a random parameter direction (central difference, fp32 loss) for a smooth loss of
    # the COARSE pass: the fine pass's sample positions follow the coarse weights but are detached (model.py:456-457), so
    # the gradient of a fine-pass loss is by design not its total derivative
    def smooth_coarse_loss():
        rgb, depth, _, _, rgb_gt, depth_gt = outputs()
        return ((rgb - rgb_gt) ** 2).mean() + ((depth - depth_gt) ** 2).mean()

    gc = grads_of(smooth_coarse_loss())
    with torch.no_grad():
        direction = [torch.randn(p.shape, generator=g).to(DEV) * p.abs().mean().clamp_min(1e-3) for p in params]
        slope = sum(float((gi * di).sum()) for gi, di in zip(gc[:len(params)], direction))

        def loss_at(eps):
            for p, d in zip(params, direction):
                p.add_(eps * d)
            out = float(smooth_coarse_loss())
            for p, d in zip(params, direction):
                p.sub_(eps * d)
            return out

        eps = 1e-3
        fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    assert abs(fd - slope) <= 0.03 * abs(slope) + 1e-4, (fd, slope)


# ------------------------------------------------------------------ reduced-precision ("bf16") training mode
GRAD_TOL_16BIT = 1e-1   # of each gradient tensor's scale (bf16 operands: 8 significand bits) ...
GRAD_COS_16BIT = 0.999  # ... and the whole parameter gradient must keep its direction
FWD_TOL_16BIT = 2e-3    # loss / forward rows


@pytest.fixture
def sixteen_bit_mode():
    ops.set_matrix_precision(ops.PRECISION_16BIT)
    try:
        yield
    finally:
        ops.set_matrix_precision(ops.PRECISION_FP32)


@pytest.mark.parametrize("name", ["c5_train_grads", "c5_train_grads_nv4"])
def test_training_step_16bit_mode(name, sixteen_bit_mode):
    """BASELINE configs[4] asks for the mixed-precision ("bf16") training step: ufr_set_matrix_precision(UFR_PRECISION_16BIT)
    runs every dense layer with one 16-bit plane per operand (fp16 hi planes forward, bf16 operands in the backward GEMMs
    and weight gradients, fp32 accumulation; LayerNorm / attention / softmax / compositor stay fp32).  Same golden
    gradients of the reference's fp32 autograd, looser stated tolerance."""
    assert ops.get_matrix_precision() == ops.PRECISION_16BIT
    m, f, r, loss, g = _train_step(name)
    assert abs(float(loss) - float(g["loss"])) < FWD_TOL_16BIT * abs(float(g["loss"]))
    loss.backward()
    torch.cuda.synchronize()
    worst = {}
    dot = na = nb = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        if k == SHIFT_BIAS:   # true gradient zero: rounding noise on both sides (bf16 tiles here), bounded by its layer's scale
            assert float(p.grad.abs().max()) < 1e-2 * float(dict(m.named_parameters())[k.replace("bias", "weight")].grad.abs().max())
            continue
        worst[k] = grad_rel_err(p.grad, g["grad." + k])
        ref = torch.from_numpy(g["grad." + k]).to(p.grad).reshape(p.grad.shape)
        dot += float((p.grad * ref).sum()); na += float((p.grad * p.grad).sum()); nb += float((ref * ref).sum())
    # the six volume gradients are per-point quantities (nothing averages the bf16 rounding of the data-gradient GEMMs away
    # as the sum over all tokens does for a weight gradient): direction per tensor + a loose bound on the worst entry
    vol_cos, vol_worst = {}, {}
    for key in VOLUME_KEYS:
        st, k = key.split(".")
        v = f.feature_volume[st][k]
        ref = torch.as_tensor(golden_volume_grad(g, key, v.shape)).to(v.grad).reshape(v.grad.shape)
        vol_worst[key] = grad_rel_err(v.grad, ref)
        vol_cos[key] = float((v.grad * ref).sum() / ((v.grad * v.grad).sum() * (ref * ref).sum()).sqrt())
    cos = dot / (na * nb) ** 0.5
    print("16-bit mode: loss rel", abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])), "parameter cosine", cos,
          "worst parameter", max(worst.values()), max(worst, key=worst.get), "volumes", vol_cos, vol_worst)
    assert cos > GRAD_COS_16BIT
    bad = {k: e for k, e in worst.items() if not e < GRAD_TOL_16BIT}
    assert not bad, bad
    assert min(vol_cos.values()) > 0.995 and max(vol_worst.values()) < 0.3, (vol_cos, vol_worst)
    # and it is a different arithmetic, not the fp32 path under another name
    assert max(worst.values()) > 10 * GRAD_TOL
    # Anchor (tests/golden/make_bf16_anchor.py): the REFERENCE's own training step under torch.autocast(bfloat16) -- what
    # its Lightning trainer runs with precision="bf16" -- measured against the same fp32 goldens with the same metrics.
    # This mode must be no further from the reference's fp32 autograd than the reference's own 16-bit run, in every metric.
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_train_bf16_anchor.json")) as fh:
        anchor = json.load(fh)[name]
    ours = {"loss_rel": abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])),
            "forward_rows": max(grad_rel_err(r[i].detach(), g[n]) for n, i in dict(rgb=1, depth=2, rgb_2=8, depth_2=9).items()),
            "one_minus_parameter_cosine": 1.0 - cos, "parameter_worst": max(worst.values()),
            "one_minus_volume_cosine": 1.0 - min(vol_cos.values()), "volume_worst": max(vol_worst.values())}
    theirs = {"loss_rel": anchor["loss_rel"], "forward_rows": max(anchor["forward_rows"].values()),
              "one_minus_parameter_cosine": 1.0 - anchor["parameter_cosine"], "parameter_worst": anchor["parameter_worst"],
              "one_minus_volume_cosine": 1.0 - anchor["volume_cosine_min"], "volume_worst": anchor["volume_worst"]}
    print("16-bit mode vs the reference's bf16-autocast run (errors against the fp32 goldens):",
          {k: (ours[k], theirs[k]) for k in ours})
    worse = {k: (ours[k], theirs[k]) for k in ours if not ours[k] <= theirs[k]}
    assert not worse, worse


def test_precision_mode_is_restored_and_validated():
    assert ops.get_matrix_precision() == ops.PRECISION_FP32
    from uforecon_amd._lib import UfrError
    with pytest.raises(UfrError):
        ops.set_matrix_precision(5)
    assert ops.get_matrix_precision() == ops.PRECISION_FP32
