"""GPU parity tests: every C-ABI entry point against the CPU oracle and the reference's golden
vectors, on the same seeded inputs.  Tolerance for the rendered depth / RGB is the north-star
1e-4 relative (REL_TOL); row-level intermediates get the tighter bounds written next to them.
"""
import pytest
import torch

from helpers import CASES, REL_TOL, border_degenerate_rays, case_inputs, load_weights, max_rel_elem, rel_err
from oracle import ufo_oracle as O
from uforecon_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def weights():
    P = load_weights()
    return ops.PackedWeights({k: v.to(DEV) for k, v in P.items()})


def _oracle_rows(name):
    fr, idx, U1, U2, g = case_inputs(name)
    want = {}
    with torch.no_grad():
        O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2,
                want=want, coarse_only=CASES[name].get("coarse_only", False))
    return fr, idx, U1, U2, g, want


def _frame_handle(fr):
    f = fr.to(DEV)
    return ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)


def _ray_setup(fr, idx):
    i = idx.reshape(-1)
    ray_d = fr.batch["ray_d"][0][:, i].t().contiguous()
    cz = fr.batch["cam_ray_d"][0][2, i]
    near = fr.batch["near_fars"][0, 0, 0] / cz
    far = fr.batch["near_fars"][0, 0, 1] / cz
    return fr.batch["ray_o"][0].contiguous(), ray_d, near.contiguous(), far.contiguous()


def test_fixed_sampler_bit_exact():
    fr, idx, U1, U2, g, want = _oracle_rows("rows_small")
    ray_o, ray_d, near, far = _ray_setup(fr, idx)
    z = ops.sample_fixed(near.to(DEV), far.to(DEV), U1.to(DEV))
    assert torch.equal(z.cpu(), want["coarse"]["z"])
    assert torch.equal(z.cpu(), torch.from_numpy(g["coarse.z"]))
    pts = ops.points(ray_o.to(DEV), ray_d.to(DEV), z)
    assert torch.equal(pts.cpu(), want["coarse"]["pts"])


@pytest.mark.parametrize("name", ["rows_small", "c2_hier_small", "c4_nv5_128"])
def test_importance_sampler_and_merge(name):
    fr, idx, U1, U2, g, want = _oracle_rows(name)
    c = want["coarse"]
    z_fine, z_all = ops.sample_importance_merge(c["weight"].to(DEV).contiguous(), c["z"].to(DEV).contiguous(), U2.to(DEV))
    # same inputs -> positions agree to float rounding of the CDF (the reference's own cumsum is double)
    assert rel_err(z_fine, want["fine"]["z_fine"]) < 5e-6
    assert rel_err(z_all, want["fine"]["z"]) < 5e-6
    assert bool((z_all[:, 1:] >= z_all[:, :-1]).all())


@pytest.mark.parametrize("tag", ["coarse", "fine"])
def test_compositor(tag, weights):
    fr, idx, U1, U2, g, want = _oracle_rows("rows_small")
    w = want[tag]
    RN, SN = w["z"].shape
    rgb, depth, opacity, weight = ops.composite(w["z"].to(DEV).contiguous(), w["radiance"].reshape(RN, SN, 3).to(DEV).contiguous(),
                                                w["srdf"].to(DEV).contiguous(), weights.variance)
    # inputs are the oracle's rows computed on THIS host, so compare with its outputs (the golden fine
    # pass was sampled on the build host, whose torch CPU kernels round the CDF differently)
    assert rel_err(weight, w["weight"]) < 5e-6
    assert rel_err(depth, w["depth"]) < 5e-6
    assert rel_err(rgb, w["rgb"]) < 5e-6
    if tag == "coarse":
        assert rel_err(weight, g["coarse.weight"]) < 5e-6
        assert rel_err(opacity, g["coarse.opacity"]) < 5e-6


@pytest.mark.parametrize("name,tag", [("rows_small", "coarse"), ("rows_small", "fine"), ("c4_nv5_128", "coarse")])
def test_project_gather_rows(name, tag, weights):
    fr, idx, U1, U2, g, want = _oracle_rows(name)
    w = want[tag]
    fh = _frame_handle(fr)
    ray_o, ray_d, _, _ = _ray_setup(fr, idx)
    x, rgbm, dirs, dbg = ops.project_gather(fh, weights, ray_o.to(DEV), ray_d.to(DEV), w["z"].to(DEV).contiguous(), debug=True)
    RN, SN = w["z"].shape
    NV = fh.NV
    assert rel_err(dbg["xy"].reshape(NV, RN, SN, 2), w["xy"]) < 5e-6
    assert torch.equal(dbg["mask_z"].reshape(NV, RN, SN).cpu(), w["mask_z"])
    assert rel_err(dbg["sim8"].reshape(RN, SN, 8), w["sim8"]) < 1e-5
    assert rel_err(dbg["vol24"].reshape(RN, SN, 24), w["vol24"]) < 1e-5
    assert rel_err(x, w["x"]) < 1e-5                                   # (P,NV,80) token inputs
    assert rel_err(rgbm[..., :3], w["rgb_s"]) < 1e-5
    assert torch.equal(rgbm[..., 3].cpu(), w["mask"].permute(1, 2, 0).reshape(-1, NV))
    assert rel_err(dirs[..., :3], w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3)) < 1e-5


@pytest.mark.parametrize("name,tag", [("rows_small", "coarse"), ("rows_small", "fine"), ("c4_nv5_128", "fine")])
def test_aggregate_rows(name, tag, weights):
    """View transformer, ray transformer, SRDF and radiance heads from the ORACLE's token inputs."""
    fr, idx, U1, U2, g, want = _oracle_rows(name)
    w = want[tag]
    RN, SN = w["z"].shape
    NV = w["x"].shape[1]
    x = w["x"].to(DEV).contiguous()
    rgbm = torch.cat([w["rgb_s"], w["mask"].permute(1, 2, 0).reshape(-1, NV, 1)], -1).to(DEV).contiguous()
    dirs = torch.cat([w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3), torch.zeros(RN * SN, NV, 1)], -1).to(DEV).contiguous()
    radiance, srdf, dbg = ops.aggregate(weights, x, rgbm, dirs, RN, SN, debug=True)
    assert rel_err(dbg["view_out"], w["view_out"]) < 2e-5
    assert rel_err(dbg["ray_out"].reshape(RN, SN, 88), w["ray_out"]) < 2e-5
    assert rel_err(srdf, w["srdf"]) < 5e-5
    assert rel_err(radiance, w["radiance"]) < 2e-5


@pytest.mark.parametrize("name", ["c1_coarse_only", "c2_hier_small", "c4_nv5_128", "c2_hier_512x640"])
def test_render_rays_matches_reference_golden(name, weights):
    """Whole path through ufr_render_rays vs the reference's own outputs (tests/golden)."""
    c = CASES[name]
    fr, idx, U1, U2, g, want = _oracle_rows(name)
    fh = _frame_handle(fr)
    out = ops.render_rays(fh, weights, idx.to(DEV), U1.to(DEV), U2.to(DEV), coarse_only=c.get("coarse_only", False))
    torch.cuda.synchronize()
    depth_ref = torch.from_numpy(g["depth"])
    rgb_ref = torch.from_numpy(g["rgb"])
    # north-star bound: per-pixel depth within 1e-4 relative of the reference, every ray
    assert max_rel_elem(out["depth"], depth_ref, floor=1e-3) < REL_TOL
    cz = fr.batch["cam_ray_d"][0][2, idx.reshape(-1)]
    assert rel_err(out["depth_z"], depth_ref * cz) < REL_TOL
    # ... and RGB within 1e-4 on every ray whose samples do not sit ON an image border of a source
    # view: there the reference's inclusive in-bounds mask (grid_sample.py:13-17) flips with the last
    # ulp of the sample position (rows of the render view are aligned with source view 0), so the
    # reference is discontinuous and no implementation can track it; such rays must stay rare.
    degenerate = border_degenerate_rays(want["coarse" if c.get("coarse_only") else "fine"])
    frac = float(degenerate.float().mean())
    print(f"{name}: {frac:.1%} of the rays have a sample within 2e-5 of a source-image border (RGB not asserted on them)")
    # the small frames' ray grids include whole first / last pixel rows, which project onto y = -+1 of source view 0 exactly;
    # at the benchmark's 512x640 the excluded rays must stay below 5 % (and the *_interior fixtures assert RGB on 100 %)
    assert frac < (0.05 if name == "c2_hier_512x640" else 0.15)
    ok = ~degenerate
    assert max_rel_elem(out["rgb"][ok.to(DEV)], rgb_ref[ok], floor=0.05) < REL_TOL
    assert rel_err(out["z_all"], (torch.from_numpy(g["points"]) - fr.batch["ray_o"][0]).norm(dim=-1)) < 1e-5
    # measured 2e-5 .. 9e-5 on the small frames, 5.4e-4 on the 512x640 one: the fine pass's rows are compared at sample
    # positions that differ by the importance sampler's CDF rounding, which white-noise maps and the signed-distance head
    # amplify (the oracle on this host is 6e-5 from the golden itself on the small frames);
    # test_render_rays_interior_rays_full_coverage pins srdf at the golden's own positions to 1e-4
    assert rel_err(out["srdf"], g["srdf"]) < 1e-3


@pytest.mark.parametrize("name", ["c1_coarse_only", "c2_hier_small", "c4_nv5_128", "c2_hier_512x640"])
def test_render_rays_rgb_on_every_ray_with_the_border_decision_passed_in(name, weights):
    """RGB on 100 % of the rays of every golden fixture, border-degenerate ones included.  A sample that projects ONTO an
    image border of a source view is in or out of the reference's inclusive mask (grid_sample.py:13-17) by the last ulp of
    its position: the reference is a step function there, and the only thing the mask does is send that view's blend
    logit to -1e9 (ray_transformer.py:315-317) -- so a border sample has exactly two admissible colours.  Checked here:
    (a) the kernels' in-image decision differs from the reference arithmetic's (the oracle on this host, at the kernels'
    own sample positions) ONLY at samples within 2e-5 of a border; (b) with that decision passed in, the reference
    arithmetic reproduces the kernels' RGB within 1e-4 on EVERY ray -- i.e. every sample's colour is one of its two
    admissible values; depth does not depend on the decision and is asserted against the golden elsewhere."""
    c = CASES[name]
    fr, idx, U1, U2, g = case_inputs(name)
    fh = _frame_handle(fr)
    out = ops.render_rays(fh, weights, idx.to(DEV), U1.to(DEV), U2.to(DEV), coarse_only=c.get("coarse_only", False))
    ray_o, ray_d, _, _ = _ray_setup(fr, idx)
    z = out["z_all"].cpu()
    RN, S = z.shape
    NV = fh.NV
    # the kernels' own decision at these positions (the same gather kernel, one call: test_chunk_invariance pins that)
    _, rgbm, _, dbg = ops.project_gather(fh, weights, ray_o.to(DEV), ray_d.to(DEV), out["z_all"].contiguous(), debug=True)
    ours = rgbm[..., 3].reshape(RN, S, NV).permute(2, 0, 1).cpu()
    pts = ray_o[None, None, :] + z[..., None] * ray_d[:, None, :]
    rows = {}
    with torch.no_grad():
        O.render_pass(load_weights(), fr.batch, pts, z, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, want=rows)
        rgb_o, depth_o = O.render_pass(load_weights(), fr.batch, pts, z, fr.source_imgs_feat, fr.feature_volume,
                                       fr.match_feature, mask_override=ours)[:2]
    near = ((rows["xy"].abs() - 1.0).abs() < 2e-5).any(-1)                  # (NV,RN,S): within 2e-5 of a border
    differs = ours != rows["mask"]
    assert not bool((differs & ~near).any()), "in-image decisions differ away from the borders"
    print(f"{name}: {int(near.any(0).any(-1).sum())} of {RN} rays touch a border; the decision differs on {int(differs.sum())} samples")
    assert max_rel_elem(out["rgb"], rgb_o, floor=0.05) < REL_TOL             # every ray
    assert max_rel_elem(out["depth"], depth_o, floor=1e-3) < REL_TOL


@pytest.mark.parametrize("name", ["c2_hier_interior", "c4_nv5_interior"])
def test_render_rays_interior_rays_full_coverage(name, weights):
    """Rays strictly inside the image (no sample projects onto an image border of a source view, where the
    reference's inclusive mask is a step function of the last ulp): depth AND RGB within 1e-4 on 100 % of the rays,
    sample positions (coarse + importance samples, merged) against the golden, srdf within 1e-4 of its scale at the
    golden's own sample positions.  (End to end the srdf rows are compared at positions that differ by the CDF rounding
    of the importance sampler -- 1e-6 relative -- which the steep signed-distance head amplifies: the oracle on this
    host is already 6e-5 away from the golden there, so that comparison only gets a 2e-4 sanity bound.)"""
    fr, idx, U1, U2, g, want = _oracle_rows(name)
    assert not bool(border_degenerate_rays(want["fine"]).any())
    out = ops.render_rays(_frame_handle(fr), weights, idx.to(DEV), U1.to(DEV), U2.to(DEV))
    torch.cuda.synchronize()
    assert max_rel_elem(out["depth"], g["depth"], floor=1e-3) < REL_TOL
    assert max_rel_elem(out["rgb"], g["rgb"], floor=0.05) < REL_TOL
    z_ref = (torch.from_numpy(g["points"]) - fr.batch["ray_o"][0]).norm(dim=-1)
    assert rel_err(out["z_all"], z_ref) < 1e-5                    # importance sampler + merge vs the reference's own
    assert rel_err(out["srdf"], g["srdf"]) < 2e-4
    ray_o, ray_d, _, _ = _ray_setup(fr, idx)
    RN, SN = z_ref.shape
    x, rgbm, dirs = ops.project_gather(_frame_handle(fr), weights, ray_o.to(DEV), ray_d.to(DEV), z_ref.to(DEV).contiguous())[:3]
    srdf = ops.aggregate(weights, x, rgbm, dirs, RN, SN)[1]
    assert rel_err(srdf.reshape(RN, SN), g["srdf"]) < 1e-4


@pytest.mark.parametrize("NV", [2, 4, 6, 7])
def test_render_rays_other_view_counts(NV, weights):
    """The view transformer is instantiated per token count L = NV + 1 (16 // L points per MFMA column tile, DPP or
    ds_bpermute token exchanges, different pair counts in the gather): every supported NV against the oracle."""
    from uforecon_amd.scene import make_frame, sampler_uniforms

    fr = make_frame(48, 64, NV, seed=20 + NV)
    RN = 24
    idx = (torch.arange(RN) * 120 + 37)[None]
    U1, U2 = sampler_uniforms(3, 64, 64, RN)
    want = {}
    with torch.no_grad():
        _, _, depth_ref, rgb_ref = O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume,
                                           fr.match_feature, U1, U2, want=want)
    out = ops.render_rays(_frame_handle(fr), weights, idx.reshape(-1).to(DEV), U1.to(DEV), U2.to(DEV))
    assert max_rel_elem(out["depth"], depth_ref.reshape(-1), floor=1e-3) < REL_TOL
    ok = ~border_degenerate_rays(want["fine"])
    assert max_rel_elem(out["rgb"][ok.to(DEV)], rgb_ref.reshape(-1, 3)[ok], floor=0.05) < REL_TOL


def test_render_rays_non_power_of_two_sample_total(weights):
    """64 + 32 samples: the ray transformer divides the values by the sequence length (linear_attention.py:41), which
    is a true division for totals that are not a power of two."""
    from uforecon_amd.scene import make_frame, sampler_uniforms

    fr = make_frame(48, 64, 3, seed=31)
    RN = 24
    idx = (torch.arange(RN) * 110 + 211)[None]
    U1, U2 = sampler_uniforms(6, 64, 32, RN)
    want = {}
    with torch.no_grad():
        srdf_ref, _, depth_ref, rgb_ref = O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume,
                                                  fr.match_feature, U1, U2, want=want)
    out = ops.render_rays(_frame_handle(fr), weights, idx.reshape(-1).to(DEV), U1.to(DEV), U2.to(DEV))
    assert tuple(out["srdf"].shape) == (RN, 96)
    assert max_rel_elem(out["depth"], depth_ref.reshape(-1), floor=1e-3) < REL_TOL
    assert rel_err(out["srdf"], srdf_ref) < 1e-4
    ok = ~border_degenerate_rays(want["fine"])
    assert max_rel_elem(out["rgb"][ok.to(DEV)], rgb_ref.reshape(-1, 3)[ok], floor=0.05) < REL_TOL


def test_render_rays_chunking_is_invisible(weights):
    """Rays are independent: rendering in chunks of 64 equals one launch group, bit for bit."""
    fr, idx, U1, U2, g = case_inputs("c2_hier_small")
    fh = _frame_handle(fr)
    a = ops.render_rays(fh, weights, idx.to(DEV), U1.to(DEV), U2.to(DEV))
    ws = ops.RenderWorkspace(DEV, 64, 64, 3, chunk_rays=64)
    b = ops.render_rays(fh, weights, idx.to(DEV), U1.to(DEV), U2.to(DEV), workspace=ws)
    assert torch.equal(a["depth"], b["depth"]) and torch.equal(a["rgb"], b["rgb"])


def test_side_streams_are_invisible(weights):
    """Chunks issued round-robin on library-owned side streams run beside each other (a gather next to another
    chunk's transformers on the same CU); the result must not depend on that.  Regression: the view transformer
    once read its view-token fragment from LDS before the workgroup had finished filling it -- visible only when
    co-resident gather workgroups staggered the start of its waves."""
    fr, idx, U1, U2, g = case_inputs("c2_hier_small")
    f = fr.to(DEV)
    fh = _frame_handle(fr)
    H, W = f.batch["source_imgs"].shape[-2:]
    HW = H * W
    gen = torch.Generator().manual_seed(9)
    U1f, U2f = torch.rand(64, HW, generator=gen).to(DEV), torch.rand(64, HW, generator=gen).to(DEV)
    ridx = torch.arange(HW, device=DEV)
    base = ops.render_rays(fh, weights, ridx, U1f, U2f, workspace=ops.RenderWorkspace(DEV, 64, 64, 3, chunk_rays=8192, n_streams=1))
    base = {k: v.clone() for k, v in base.items() if torch.is_tensor(v)}
    for chunk, streams in ((1024, 3), (512, 4), (1024, 2)):
        for _ in range(3):
            ws = ops.RenderWorkspace(DEV, 64, 64, 3, chunk_rays=chunk, n_streams=streams)
            out = ops.render_rays(fh, weights, ridx, U1f, U2f, workspace=ws)
            torch.cuda.synchronize()
            for k in ("depth", "rgb", "srdf", "z_all"):
                assert torch.equal(out[k], base[k]), (chunk, streams, k)


def test_fine_pass_pool_equals_full_reevaluation(weights):
    """ufr_render_rays keeps the coarse samples' per-point results and evaluates only the new points in the
    fine pass; the reference (model.py:466-472) re-evaluates all merged samples.  Walking the reference's order
    through the stepwise entry points gives the same bits: a point's gathers and view-transformer output depend
    on that point alone."""
    fr, idx, U1, U2, g = case_inputs("c2_hier_small")
    f = fr.to(DEV)
    fh = _frame_handle(fr)
    ray_o, ray_d, near, far = (t.to(DEV) for t in _ray_setup(fr, idx))
    var = weights.variance
    z1 = ops.sample_fixed(near, far, U1.to(DEV))
    RN, SN = z1.shape
    x, rgbm, dirs, _ = ops.project_gather(fh, weights, ray_o, ray_d, z1)
    rad1, srdf1, _ = ops.aggregate(weights, x, rgbm, dirs, RN, SN)
    _, _, _, w1 = ops.composite(z1, rad1, srdf1, var)
    _, z2 = ops.sample_importance_merge(w1, z1, U2.to(DEV), want_fine=False)
    S2 = z2.shape[1]
    x, rgbm, dirs, _ = ops.project_gather(fh, weights, ray_o, ray_d, z2)         # all SN+PN merged samples
    rad2, srdf2, _ = ops.aggregate(weights, x, rgbm, dirs, RN, S2)
    rgb2, depth2, _, _ = ops.composite(z2, rad2, srdf2, var)
    out = ops.render_rays(fh, weights, idx.to(DEV), U1.to(DEV), U2.to(DEV))
    assert torch.equal(out["z_all"], z2)
    assert torch.equal(out["srdf"], srdf2)
    assert torch.equal(out["depth"], depth2) and torch.equal(out["rgb"], rgb2)


def test_edge_cases(weights):
    fr, idx, U1, U2, g = case_inputs("c2_hier_small")
    fh = _frame_handle(fr)
    # a single ray, and a ray count that is not a multiple of any tile size
    for n in (1, 37):
        out = ops.render_rays(fh, weights, idx[:, :n].to(DEV).contiguous(), U1[:, :n].to(DEV).contiguous(),
                              U2[:, :n].to(DEV).contiguous())
        assert max_rel_elem(out["depth"], torch.from_numpy(g["depth"][:n]), floor=1e-3) < REL_TOL
    with pytest.raises(ops.UfrError, match="multiple of 16"):
        ops.render_rays(fh, weights, idx.to(DEV), U1[:60].to(DEV).contiguous(), U2.to(DEV))
    with pytest.raises(ops.UfrError, match="GPU"):
        ops.sample_fixed(torch.zeros(4), torch.ones(4), torch.rand(64, 4))


def test_only_non_finite_weights_are_refused():
    """The dense layers are packed as fp16 planes of 2^s_M w with the exponent chosen per matrix from max |w|
    (ufr_layout_f16.h): every finite weight packs; a parameter that is not finite must fail ufr_weights_pack loudly -- it
    must never render."""
    from uforecon_amd._lib import UfrError

    key = "ray_transformer.density_view_transformer.layers.0.mlp.0.weight"
    for bad in (float("nan"), float("inf")):
        P = {k: v.clone().to(DEV) for k, v in load_weights().items()}
        P[key][3, 5] = bad
        with pytest.raises(UfrError, match="not finite"):
            ops.PackedWeights(P)
    for big in (250.0, 300.0, 1e4, 3e7):
        P = {k: v.clone().to(DEV) for k, v in load_weights().items()}
        P[key][3, 5] = big
        W = ops.PackedWeights(P)
        s, _ = W.scale_exponents()["vt_mlp0"]
        assert 2.0 ** 14 <= big * 2.0 ** s <= 2.0 ** 15      # the matrix's planes sit at the top of fp16's range
    with pytest.raises(UfrError, match="input_abs_max"):
        ops.PackedWeights({k: v.to(DEV) for k, v in load_weights().items()}, input_abs_max=float("inf"))


def _rows_against_the_oracle(P, x, w, RN, SN, NV, input_abs_max=None, precision=None):
    """ufr_aggregate's rows against the oracle IN FLOAT64 on the same parameters and tokens; beside each error the distance
    of the oracle's own float32 evaluation from that yardstick (what the reference's arithmetic itself loses)."""
    mask = w["mask"].permute(1, 2, 0).reshape(-1, NV)
    dirs3 = w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3)
    ref, ref32 = {}, {}
    with torch.no_grad():
        rad32, srdf32 = O.aggregate_tokens(P, x, w["rgb_s"], mask, dirs3, RN, SN, want=ref32)
        P64 = {k: v.double() for k, v in P.items()}
        rad_ref, srdf_ref = O.aggregate_tokens(P64, x.double(), w["rgb_s"].double(), mask.double(), dirs3.double(), RN, SN, want=ref)
    rgbm = torch.cat([w["rgb_s"], mask[..., None]], -1).to(DEV).contiguous()
    dirs = torch.cat([dirs3, torch.zeros(RN * SN, NV, 1)], -1).to(DEV).contiguous()
    W = ops.PackedWeights({k: v.to(DEV) for k, v in P.items()}, input_abs_max=input_abs_max)
    radiance, srdf, dbg = ops.aggregate(W, x.to(DEV), rgbm, dirs, RN, SN, debug=True, precision=precision)
    assert ops.status_poll(True) == 0
    err = dict(view_out=rel_err(dbg["view_out"], ref["view_out"]), ray_out=rel_err(dbg["ray_out"].reshape(RN, SN, 88), ref["ray_out"]),
               srdf=rel_err(srdf, srdf_ref), radiance=rel_err(radiance, rad_ref))
    fp32 = dict(view_out=rel_err(ref32["view_out"], ref["view_out"]), ray_out=rel_err(ref32["ray_out"], ref["ray_out"]),
                srdf=rel_err(srdf32, srdf_ref), radiance=rel_err(rad32, rad_ref))
    return err, fp32, W


@pytest.mark.parametrize("top", [1e4, 1e-4])
def test_aggregate_rows_weight_magnitudes(top):
    """Weights far outside fp16's own range (largest element of a matrix = 1e4 or 1e-4) render like any others: the planes'
    exponents follow each matrix (prep.hip: weight_scale_kernel).  Scaled here: the matrices whose output feeds a
    LayerNorm (merge, mlp.0, mlp.2 of both transformers) -- the network's function is unchanged up to the epsilon, so
    the comparison with the fp32 oracle on the same weights stays well conditioned."""
    fr, idx, U1, U2, g, want = _oracle_rows("rows_small")
    w = want["coarse"]
    RN, SN = w["z"].shape
    NV = w["x"].shape[1]
    P = {k: v.clone() for k, v in load_weights().items()}
    for tr in ("density_view_transformer", "density_ray_transformer"):
        for m in ("merge", "mlp.0", "mlp.2"):
            k = f"ray_transformer.{tr}.layers.0.{m}.weight"
            P[k] *= top / float(P[k].abs().max())
    err, fp32, W = _rows_against_the_oracle(P, w["x"].contiguous(), w, RN, SN, NV)
    print(f"weights with max |w| = {top:g}: {err} (the fp32 oracle: {fp32}); exponents {W.scale_exponents()}")
    for k, tol in (("view_out", 2e-5), ("ray_out", 2e-5), ("radiance", 2e-5), ("srdf", 1e-4)):
        assert err[k] < max(tol, 2 * fp32[k]), k


def test_aggregate_rows_checkpoint_like_magnitudes():
    """Every dense matrix x 64 and token features x 300 (what a trained checkpoint on un-normalised backbone features could
    look like; round 3's fixed exponents held x 8 / x 30): with the input bound stated at pack time the analytic per-layer
    exponents keep every layer in range -- status clear, every row within 1e-5 of the float64 yardstick.  (The reference's
    own float32 arithmetic is 7e-3 off on the ray-transformer rows of this network: torch forms elu(q) + 1 as
    (exp(q) - 1) + 1, which rounds the small K' of strongly negative q away; the kernels evaluate exp(q) directly.)"""
    fr, idx, U1, U2, g, want = _oracle_rows("rows_small")
    w = want["coarse"]
    RN, SN = w["z"].shape
    NV = w["x"].shape[1]
    P = {k: v.clone() for k, v in load_weights().items()}
    for k in P:
        if k.startswith("ray_transformer.") and P[k].dim() == 2 and "view_token" not in k and "pre_sim" not in k:
            P[k] *= 64.0
    x = (w["x"] * 300.0).contiguous()
    err, fp32, W = _rows_against_the_oracle(P, x, w, RN, SN, NV, input_abs_max=float(x.abs().max()))
    print(f"x64 weights, x300 tokens: {err} (the fp32 oracle: {fp32}); exponents {W.scale_exponents()}")
    for k in err:      # measured: 3e-8 / 4e-7 / 6e-7 / 0 -- while the float32 oracle is 7e-3 off on ray_out (elu(q) + 1 of q << 0)
        assert err[k] < 1e-5, k


@pytest.mark.parametrize("scale", [1e-3, 1e-2, 40.0])
def test_aggregate_rows_token_magnitudes(scale, weights):
    """The dense layers run as fp16 plane products (ufr_layout_f16.h): fp16's exponent range must not show.  Token inputs
    far smaller / larger than the synthetic scene's (|x| up to 1.7) against the oracle on the same scaled tokens."""
    fr, idx, U1, U2, g, want = _oracle_rows("rows_small")
    w = want["coarse"]
    RN, SN = w["z"].shape
    NV = w["x"].shape[1]
    x = (w["x"] * scale).contiguous()
    mask = w["mask"].permute(1, 2, 0).reshape(-1, NV)
    dirs3 = w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3)
    ref = {}
    with torch.no_grad():
        rad_ref, srdf_ref = O.aggregate_tokens(load_weights(), x, w["rgb_s"], mask, dirs3, RN, SN, want=ref)
    rgbm = torch.cat([w["rgb_s"], mask[..., None]], -1).to(DEV).contiguous()
    dirs = torch.cat([dirs3, torch.zeros(RN * SN, NV, 1)], -1).to(DEV).contiguous()
    radiance, srdf, dbg = ops.aggregate(weights, x.to(DEV), rgbm, dirs, RN, SN, debug=True)
    assert rel_err(dbg["view_out"], ref["view_out"]) < 2e-5
    assert rel_err(dbg["ray_out"].reshape(RN, SN, 88), ref["ray_out"]) < 2e-5
    assert rel_err(srdf, srdf_ref) < 5e-5
    assert rel_err(radiance, rad_ref) < 2e-5


def test_render_rays_16bit_matrix_mode(weights):
    """ufr_set_matrix_precision(UFR_PRECISION_16BIT) (the mixed-precision training mode) through the whole forward path:
    one fp16 plane per operand instead of the fp32-grade split -- depth within 2e-3 of the reference golden (measured
    ~2e-4), clearly different from the default mode, and the default mode is back afterwards bit for bit."""
    fr, idx, U1, U2, g, want = _oracle_rows("c2_hier_interior")
    fh = _frame_handle(fr)
    args = (idx.to(DEV), U1.to(DEV), U2.to(DEV))
    ref = ops.render_rays(fh, weights, *args)["depth"].clone()
    ops.set_matrix_precision(ops.PRECISION_16BIT)
    try:
        low = ops.render_rays(fh, weights, *args)["depth"].clone()
    finally:
        ops.set_matrix_precision(ops.PRECISION_FP32)
    again = ops.render_rays(fh, weights, *args)["depth"]
    torch.cuda.synchronize()
    e_low = max_rel_elem(low, g["depth"], floor=1e-3)
    assert e_low < 2e-3
    assert e_low > 10 * max_rel_elem(ref, g["depth"], floor=1e-3)
    assert torch.equal(again, ref)
