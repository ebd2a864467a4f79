"""torch.autograd glue of the training step (BASELINE configs[4]): the HIP forward entry points paired with their
``ufr_*_bwd`` adjoints, so ``loss.backward()`` through ``UFORecon.infer(extract_geometry=False)`` (the call of
``training_step``, code1/model.py:540-548) runs on the kernels.

What the reference's autograd differentiates on this path: every ``ray_transformer.*``
parameter, ``deviation_network.variance`` and the six sampled volumes.  Sample positions are detached
(model.py:456-457) and the 2-D maps come from a frozen producer (model.py:82-83): no gradient, exactly as upstream.
"""
from __future__ import annotations

import dataclasses

import torch

from . import ops

STAGES = ("stage1", "stage2", "stage3")


class RenderPass(torch.autograd.Function):
    """One ``sample2rgb`` pass (model.py:308-348): gather -> aggregate -> composite.

    ``apply(frame, weights, ray_o, ray_d, z, *params, *volumes)`` with ``params`` = the 40 tensors of
    ``ops.RAW_WEIGHT_KEYS`` (live nn.Parameters) and ``volumes`` = feature / weight volume of the three stages.
    Returns ``rgb (RN,3), depth (RN), opacity (RN), weight (RN,SN), srdf (RN,SN), xy (NV,P,2)``.
    Activations kept for the backward: the token inputs (x, rgb/mask, dir), the pair similarity, the view transformer's
    token-0 rows, radiance and srdf -- everything else is recomputed inside the backward kernels."""

    @staticmethod
    def forward(ctx, frame, weights, ray_o, ray_d, z, *tensors):
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not as a zero tensor (a fill launch each)
        n_par = len(ops.RAW_WEIGHT_KEYS)
        RN, SN = z.shape
        prec = weights.mode()     # resolved ONCE: the backward of this node runs in the mode its forward ran in
        x, rgbm, dirs, dbg = ops.project_gather(frame, weights, ray_o, ray_d, z, want_sim8=True, want_xy=True)
        radiance, srdf, agg = ops.aggregate(weights, x, rgbm, dirs, RN, SN, keep_workspace=True, precision=prec)
        variance = weights.variance.reshape(1)
        rgb, depth, opacity, weight = ops.composite(z, radiance.view(RN, SN, 3), srdf, variance)
        ctx.frame, ctx.weights, ctx.n_par, ctx.precision = frame, weights, n_par, prec
        ctx.vol_shapes = [tuple(t.shape) for t in tensors[n_par:]]
        ctx.save_for_backward(ray_o, ray_d, z, x, rgbm, dirs, dbg["sim8"], agg["token0"], radiance, srdf)
        ctx.mark_non_differentiable(dbg["xy"])
        return rgb, depth, opacity, weight, srdf, dbg["xy"]

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_opacity, d_weight, d_srdf_out, _d_xy):
        ray_o, ray_d, z, x, rgbm, dirs, sim8, token0, radiance, srdf = ctx.saved_tensors
        frame, W = ctx.frame, ctx.weights
        RN, SN = z.shape
        dev = z.device
        d_radiance, d_srdf, d_var = ops.composite_bwd(z, radiance.view(RN, SN, 3), srdf, W.variance.reshape(1), d_rgb, d_depth,
                                                      d_opacity, d_weight)
        if d_srdf_out is not None:
            d_srdf = d_srdf + d_srdf_out
        grads = ops.GradBuffer(dev)
        prec = ctx.precision
        d_pv, _ = ops.aggregate_bwd(W, grads, x, rgbm, dirs, token0, RN, SN, d_radiance.view(RN * SN, 3), d_srdf, precision=prec)
        need = ctx.needs_input_grad[5:]
        if any(need[ctx.n_par:]):       # frustum gradients wanted (feature_volume.cost_reg_2 trains through them)
            gvol = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.vol_shapes]      # written whole: no zero-fill
            ops.project_gather_bwd(frame, W, grads, ray_o, ray_d, z, sim8, d_pv, gvol[0::2], gvol[1::2], precision=prec,
                                   accumulate=False)
        else:                           # parameters only: skip the scatter-add (and 0.7 GB of zeroed gradient volumes)
            gvol = [None] * len(ctx.vol_shapes)
            ops.project_gather_bwd(frame, W, grads, ray_o, ray_d, z, sim8, d_pv, None, None, precision=prec)
        gpar = [grads.grad(k) for k in ops.RAW_WEIGHT_KEYS]
        gpar[-1] = d_var.reshape(gpar[-1].shape)                      # deviation_network.variance
        out = [g if (n and g is not None) else None for g, n in zip(gpar + gvol, need)]
        return (None, None, None, None, None, *out)


_SIDE = {}


@dataclasses.dataclass(frozen=True)
class RenderOptions:
    """How ONE RenderTwoPass node schedules its work -- an argument of the node (UFORecon(...) owns the values), so two
    models in a process, or a test next to a trainer, cannot change each other's backward.
    overlap: the backward's independent stages on three streams (False: all on the caller's stream -- per-kernel timing).
    tape_in_forward: the forward kernels run in their tape instantiation and record the backward's activations (False:
    plain forward kernels, the backward records the tape itself).
    record_tape: the caller's statement that a backward will follow this forward (model.py derives it from the grad mode)."""
    overlap: bool = True
    tape_in_forward: bool = True
    record_tape: bool = True


_WS = {}


def _workspace(kind: str, dev, numel_of) -> torch.Tensor:
    """Tile buffers of the backward kernels, kept between steps (a few GB at the training size).  As per-step
    `torch.empty` they went through the caching allocator with `record_stream` marks from the side streams: the host runs
    a step ahead of the GPU, the freed multi-GB blocks were still "in use" when the next step asked for them, and the
    allocator fell back to hipMalloc / hipFree (which synchronises) -- steps of 9..15 ms instead of 5, now and then.
    Ordering between steps: backward() ends with main waiting for the side streams, and starts with them waiting for main."""
    n = int(numel_of())
    key = (kind, torch.device(dev).index or 0)
    t = _WS.get(key)
    if t is None or t.numel() < n:
        t = _WS[key] = torch.empty(n, dtype=torch.float32, device=dev)
        _GWS_STATE.pop(key, None)        # a fresh allocation is uninitialised, whatever address the allocator handed back
    return t


# kept frustum-scatter workspace, per _WS key: (the tensor OBJECT the state describes, "zero" as the last call left it |
# "in use").  Never keyed on a raw address: the caching allocator hands a freed block's address to the next, larger,
# uninitialised workspace, and a stale "zero" entry would skip its memset.
_GWS_STATE = {}


def _gws_is_zero(key, t) -> bool:
    held = _GWS_STATE.get(key)
    return held is not None and held[0] is t and held[1] == "zero"


_BUSY = {}          # key -> weak reference to the autograd context that holds the kept buffer between its forward and backward


def _acquire(kind: str, dev, numel_of, ctx):
    """A tile buffer that must survive from a forward to its backward: the kept one (_workspace) when no live forward
    holds it, else a private allocation (two forwards before a backward, e.g. gradient accumulation over frames).
    Returns (tensor, key or None); _release(key) in the backward.  A holder that died without a backward (a graph that
    was dropped) is detected through the weak reference."""
    import weakref

    key = (kind, torch.device(dev).index or 0)
    holder = _BUSY.get(key)
    if holder is not None and holder() is not None and holder() is not ctx:
        return torch.empty(int(numel_of()), dtype=torch.float32, device=dev), None
    _BUSY[key] = weakref.ref(ctx)
    return _workspace(kind, dev, numel_of), key


def _release(key) -> None:
    if key is not None:
        _BUSY.pop(key, None)


def _side_stream(dev, i: int = 0) -> "torch.cuda.Stream":
    """Side streams per device for the multi-stream backward of RenderTwoPass (created once: stream creation is slow)."""
    key = (torch.device(dev).index or 0, i)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(dev)
    return _SIDE[key]


class RenderTwoPass(torch.autograd.Function):
    """Both passes of ``infer`` (model.py:429-473) with the coarse samples' per-point work shared: the fine pass gathers
    and view-transforms only its PN new samples; the ray transformer and the compositor -- which do couple the samples of a
    ray -- see all SN+PN merged samples through the slot -> pool-row table of ``ufr_sample_importance_pool``.  Same numbers
    as two ``RenderPass`` applications (a sample's gathers and view-transformer output depend on its own position only), one
    third less gather / view-transformer work forwards AND backwards: the cotangents that the fine pass sends to the coarse
    samples' token-0 rows and colours are added to the coarse pass's own before ONE view-transformer backward over them.

    ``apply(frame, weights, ray_o, ray_d, z1, U2, options, *params, *volumes)`` (``options``: RenderOptions or None) ->
    ``rgb, depth, opacity, weight, srdf, xy   (coarse)   rgb2, depth2, opacity2, weight2, srdf2, xy2, z2   (fine)``."""

    @staticmethod
    def forward(ctx, frame, weights, ray_o, ray_d, z1, U2, options, *tensors):
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not as a zero tensor (a fill launch each)
        opt = ctx.options = options if options is not None else RenderOptions()
        n_par = len(ops.RAW_WEIGHT_KEYS)
        RN, SN = z1.shape
        PN = U2.shape[0]
        S2, P1, P2 = SN + PN, RN * SN, RN * PN
        dev = z1.device
        prec = weights.mode()     # resolved ONCE: the backward of this node runs in the mode its forward ran in
        var = weights.variance.reshape(1)
        # the sample pool [P1 coarse rows (ray-major) | P2 new rows]: the view transformer writes its rows in place, the ray
        # transformer and the compositor of the fine pass read them through the slot -> row table (no copies, no gathers)
        pool_tok = torch.empty(P1 + P2, ops._lib.TOKEN_DIM, dtype=torch.float32, device=dev)
        pool_rad = torch.empty(P1 + P2, 3, dtype=torch.float32, device=dev)
        sim8_pool = torch.empty(P1 + P2, 8, dtype=torch.float32, device=dev)      # pre_sim_mlp inputs, pool rows
        # ... and the gathered view-transformer inputs of both passes, so that ONE backward launch walks all pool rows
        NV = frame.NV
        pool_x = torch.empty(P1 + P2, NV, ops._lib.TOKEN_DIM, dtype=torch.float32, device=dev)
        pool_rgbm = torch.empty(P1 + P2, NV, 4, dtype=torch.float32, device=dev)
        pool_dirs = torch.empty(P1 + P2, NV, 4, dtype=torch.float32, device=dev)
        x1, rgbm1, dirs1, g1 = ops.project_gather(frame, weights, ray_o, ray_d, z1, want_xy=True, sim8_out=sim8_pool[:P1],
                                                  out=(pool_x[:P1], pool_rgbm[:P1], pool_dirs[:P1]))
        # The forward kernels run in their TAPE instantiation and record the activations straight into the backward's
        # workspaces (one view tape over the whole pool, one ray tape per pass): the backward starts at its data-gradient
        # stage, nothing is computed twice.  (Needs the coarse rows to end on a tape block: else the backward records.)
        lib = ops._lib.load()
        taped = opt.record_tape and opt.tape_in_forward and P1 % ops.view_tape_block_points(NV) == 0
        ctx.ws_keys = []
        if taped:
            vws, k = _acquire("view", dev, lambda: lib.ufr_view_transform_bwd_workspace_bytes(P1 + P2, NV) // 4, ctx)
            ctx.ws_keys.append(k)
            ws_c, k = _acquire("ray_c", dev, lambda: lib.ufr_ray_transform_bwd_workspace_bytes(RN, SN) // 4, ctx)
            ctx.ws_keys.append(k)
            ws_f, k = _acquire("ray_f", dev, lambda: lib.ufr_ray_transform_bwd_workspace_bytes(RN, S2) // 4, ctx)
            ctx.ws_keys.append(k)
            ctx.tapes = (vws, ws_c, ws_f)
            ops.view_transform_tape(weights, x1, rgbm1, dirs1, pool_tok[:P1], pool_rad[:P1], vws, 0, P1 + P2, precision=prec)
            srdf1 = ops.ray_transform_tape(weights, pool_tok[:P1], RN, SN, ws_c, precision=prec)
        else:
            ctx.tapes = None
            ops.view_transform(weights, x1, rgbm1, dirs1, token0=pool_tok[:P1], radiance=pool_rad[:P1], precision=prec)
            srdf1 = ops.ray_transform(weights, pool_tok[:P1], RN, SN, precision=prec)
        rgb, depth, opacity, weight = ops.composite(z1, pool_rad[:P1].view(RN, SN, 3), srdf1, var)
        z2, z_new, row = ops.sample_importance_pool(weight, z1, U2)                  # model.py:455-470 (weights detached)
        x2, rgbm2, dirs2, g2 = ops.project_gather(frame, weights, ray_o, ray_d, z_new, want_xy=True, sim8_out=sim8_pool[P1:],
                                                  out=(pool_x[P1:], pool_rgbm[P1:], pool_dirs[P1:]))
        if taped:
            ops.view_transform_tape(weights, x2, rgbm2, dirs2, pool_tok[P1:], pool_rad[P1:], vws, P1, P1 + P2, precision=prec)
            srdf2 = ops.ray_transform_tape(weights, pool_tok, RN, S2, ws_f, row=row, precision=prec)
        else:
            ops.view_transform(weights, x2, rgbm2, dirs2, token0=pool_tok[P1:], radiance=pool_rad[P1:], precision=prec)
            srdf2 = ops.ray_transform(weights, pool_tok, RN, S2, row=row, precision=prec)
        rgb2, depth2, opacity2, weight2 = ops.composite(z2, pool_rad, srdf2, var, row=row)
        xy2 = torch.cat([g1["xy"], g2["xy"]], 1)[:, row.reshape(-1).long()]         # (NV, RN*S2, 2), a returned value only
        ctx.frame, ctx.weights, ctx.n_par, ctx.precision = frame, weights, n_par, prec
        ctx.vol_shapes = [tuple(t.shape) for t in tensors[n_par:]]
        ctx.save_for_backward(ray_o, ray_d, z1, z2, row, pool_x, pool_rgbm, pool_dirs, sim8_pool, srdf1,
                              pool_tok, pool_rad, srdf2)
        ctx.mark_non_differentiable(g1["xy"], xy2, z2)
        return rgb, depth, opacity, weight, srdf1, g1["xy"], rgb2, depth2, opacity2, weight2, srdf2, xy2, z2

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_opacity, d_weight, d_srdf, _dxy, d_rgb2, d_depth2, d_opacity2, d_weight2, d_srdf2,
                 _dxy2, _dz2):
        (ray_o, ray_d, z1, z2, row, pool_x, pool_rgbm, pool_dirs, sim8_pool, srdf1,
         pool_tok, pool_rad, srdf2) = ctx.saved_tensors
        frame, W, prec = ctx.frame, ctx.weights, ctx.precision
        RN, SN = z1.shape
        S2 = z2.shape[1]
        P1 = RN * SN
        dev = z1.device
        var = W.variance.reshape(1)
        grads = ops.GradBuffer(dev)
        d_var = torch.zeros((), dtype=torch.float32, device=dev)
        # cotangent pools, same rows as the forward's.  The fine pass writes every row of pool_a once (each pool row is exactly
        # one merged slot); the coarse pass writes its cotangents for the coarse rows into pool_b (whose other rows are zero);
        # the view transformer's backward adds the two.  So the two ray-transformer backwards are INDEPENDENT and run on two
        # streams: a launch group has one wave per ray -- 1 024 rays fill half of the GPU's 2 048 wave slots -- and side by
        # side the coarse pass (half the work) hides behind the fine one.
        pool_a = torch.empty_like(pool_tok)
        pool_b = torch.empty_like(pool_tok)
        pool_b[P1:].zero_()
        pool_drad = torch.empty_like(pool_rad)
        # ---- compositors (the coarse one ADDS its d radiance onto the fine one's rows: same stream, in order)
        _, d_srdf_s, _ = ops.composite_bwd(z2, pool_rad, srdf2, var, d_rgb2, d_depth2, d_opacity2, d_weight2, row=row,
                                           d_radiance=pool_drad, accumulate=False, d_variance=d_var)
        if d_srdf2 is not None:
            d_srdf_s = d_srdf_s + d_srdf2
        # (the coarse pass's weights feed the importance sampler detached: model.py:456-457)
        _, d_srdf_c, _ = ops.composite_bwd(z1, pool_rad[:P1].view(RN, SN, 3), srdf1, var, d_rgb, d_depth, d_opacity, d_weight,
                                           d_radiance=pool_drad[:P1], accumulate=True, d_variance=d_var)
        if d_srdf is not None:
            d_srdf_c = d_srdf_c + d_srdf
        # ---- ray transformers backwards: fine on this stream, coarse beside it; and beside both, on a third stream, the TAPE
        # stage of the view transformer's backward (the forward again, recording its activations: it needs only forward
        # tensors, and it is bound by its 2 GB of stores while the ray kernels leave half of the GPU's wave slots empty)
        main = torch.cuda.current_stream(dev)
        side, side2 = (_side_stream(dev, 0), _side_stream(dev, 1)) if ctx.options.overlap else (main, main)
        lib = ops._lib.load()
        taped = ctx.tapes is not None
        if taped:
            vws, ws_c, ws_f = ctx.tapes
        else:
            # the same ownership rule as the forward's: a kept buffer that a LIVE forward of another node recorded its tape
            # into must not be overwritten by this node's tape stage (two frames with different view counts under one
            # (l0 + l1).backward(): one taped, one not) -- a private buffer then.  Released with the others below.
            ctx.ws_keys = list(getattr(ctx, "ws_keys", ()))
            vws, k = _acquire("view", dev, lambda: lib.ufr_view_transform_bwd_workspace_bytes(pool_tok.shape[0], pool_x.shape[1]) // 4, ctx)
            ctx.ws_keys.append(k)
            ws_c, k = _acquire("ray_c", dev, lambda: lib.ufr_ray_transform_bwd_workspace_bytes(RN, SN) // 4, ctx)
            ctx.ws_keys.append(k)
            ws_f, k = _acquire("ray_f", dev, lambda: lib.ufr_ray_transform_bwd_workspace_bytes(RN, S2) // 4, ctx)
            ctx.ws_keys.append(k)
        ray_first = ops.STAGE_DGRAD if taped else (ops.STAGE_TAPE | ops.STAGE_DGRAD)
        side.wait_stream(main)
        side2.wait_stream(main)
        need = ctx.needs_input_grad[7:]
        want_vol = any(need[ctx.n_par:])
        gws = _workspace("gather_bwd", dev, lambda: ops.project_gather_bwd_workspace_floats(frame)) if want_vol else None
        gkey = ("gather_bwd", torch.device(dev).index or 0)
        with torch.cuda.stream(side2):
            if not taped:
                ops.view_transform_bwd(W, grads, pool_x, pool_rgbm, pool_dirs, None, None, None, precision=prec,
                                       stages=ops.STAGE_TAPE, workspace=vws)
            if gws is not None and not _gws_is_zero(gkey, gws):
                # the frustum scatter's record volume: ufr_project_gather_bwd leaves it zero again (it reads and re-zeroes
                # only what the scatter touched), so it is filled once per allocation -- or after a call that did not return
                gws.zero_()
        # (the ray weight-gradient contractions feed nothing downstream either: tape + data gradients first, on both
        # streams; the contractions afterwards on the side stream, beside the view transformer's data gradients)
        with torch.cuda.stream(side):
            ops.ray_transform_bwd(W, grads, pool_tok[:P1], RN, SN, d_srdf_c, out=(pool_b[:P1], None), precision=prec,
                                  stages=ray_first, workspace=ws_c)
        ops.ray_transform_bwd(W, grads, pool_tok, RN, S2, d_srdf_s, row=row, out=(pool_a, None), precision=prec,
                              stages=ray_first, workspace=ws_f)
        main.wait_stream(side)
        main.wait_stream(side2)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.ray_transform_bwd(W, grads, pool_tok[:P1], RN, SN, d_srdf_c, out=(pool_b[:P1], None), precision=prec,
                                  stages=ops.STAGE_WGRAD, workspace=ws_c)
            ops.ray_transform_bwd(W, grads, pool_tok, RN, S2, d_srdf_s, row=row, out=(pool_a, None), precision=prec,
                                  stages=ops.STAGE_WGRAD, workspace=ws_f)
        # ---- view transformer backwards, ONE launch group over the pool: coarse samples once, with the cotangents of both
        # passes, and the new samples.  Data gradients here; the weight-gradient contractions -- nothing downstream waits
        # for them -- beside the frustum scatter (one bound by HBM, the other by the L2's atomic rate)
        d_pv = torch.empty(pool_tok.shape[0], 40, dtype=torch.float32, device=dev)   # pool rows again
        ops.view_transform_bwd(W, grads, pool_x, pool_rgbm, pool_dirs, pool_a, pool_b, pool_drad, precision=prec, d_pv=d_pv,
                               stages=ops.STAGE_DGRAD, workspace=vws)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.view_transform_bwd(W, grads, pool_x, pool_rgbm, pool_dirs, None, None, None, precision=prec,
                                   stages=ops.STAGE_WGRAD, workspace=vws)
        if want_vol:
            # written whole by the unpack: no zero-fill.  (Zero-filling them on the side stream and letting the unpack ADD only the
            # voxel groups the rays reached was measured: 4.31 vs 4.24 ms per step -- the fill contends with the ray backwards)
            gvol = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.vol_shapes]
            gf, gw = gvol[0::2], gvol[1::2]
        else:
            gvol = [None] * len(ctx.vol_shapes)
            gf = gw = None
        # ONE frustum scatter over all merged samples of a ray (z2: sorted, twice the density of either pass -- the run
        # folding of gather_bwd.hip removes more corner records), d_pv / sim8 addressed through the slot -> row table
        if gws is not None:
            _GWS_STATE[gkey] = (gws, "in use")
        if want_vol and ctx.options.overlap:
            # the two halves share no output: pre_sim_mlp's weight gradients (0.09 ms) beside the scatter, not behind it
            side2.wait_stream(main)
            with torch.cuda.stream(side2):
                ops.project_gather_bwd(frame, W, grads, ray_o, ray_d, z2, sim8_pool, d_pv, None, None, precision=prec, row=row)
        ops.project_gather_bwd(frame, W, grads, ray_o, ray_d, z2, sim8_pool, d_pv, gf, gw, precision=prec, row=row,
                               accumulate=False, zeroed_workspace=gws, presim=not (want_vol and ctx.options.overlap))
        main.wait_stream(side2)
        if gws is not None:
            _GWS_STATE[gkey] = (gws, "zero")
        main.wait_stream(side)
        # (no record_stream marks: every tensor the side streams touch stays referenced until this function returns, i.e.
        # until after the join above is enqueued -- whatever reuses its memory later on this stream is ordered behind it;
        # the marks would only keep the allocator from reusing the blocks while the host runs ahead of the GPU)
        for k in getattr(ctx, "ws_keys", ()):
            _release(k)
        ctx.tapes = None
        gpar = [grads.grad(k) for k in ops.RAW_WEIGHT_KEYS]
        gpar[-1] = d_var.reshape(gpar[-1].shape)
        out = [g if (n and g is not None) else None for g, n in zip(gpar + gvol, need)]
        return (None, None, None, None, None, None, None, *out)


class RenderLoss(torch.autograd.Function):
    """The training loss of a ray batch (code1/model.py:552-566) as ONE node on one kernel (ufr_render_loss): the forward
    launch also writes d loss / d (rgb, depth, rgb_2, depth_2), the backward scales them by the upstream gradient.
    ``apply(rgb, depth, rgb_2, depth_2, rgb_gt, depth_gt, near_fars, weight_rgb, weight_depth)`` -> ``loss ()``,
    ``parts (4,)`` = [rgb coarse, rgb fine, depth coarse, depth fine] (what the reference logs; not differentiable).
    As torch expressions the same loss is ~25 launches and ~30 autograd nodes over 1 024-ray tensors -- 0.9 ms of a 4.7 ms
    step with the GPU idle in between (tools/dev/step_timeline.py)."""

    @staticmethod
    def forward(ctx, rgb, depth, rgb2, depth2, rgb_gt, depth_gt, near_fars, weight_rgb, weight_depth):
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not as a zero tensor (a fill launch each)
        loss, *cot = ops.render_loss(rgb, depth, rgb2, depth2, rgb_gt, depth_gt, near_fars, weight_rgb, weight_depth)
        ctx.cot = cot
        parts = loss[1:]
        ctx.mark_non_differentiable(parts)
        return loss[0], parts

    @staticmethod
    def backward(ctx, g, _gparts):
        cot, ctx.cot = ctx.cot, None
        if cot is None:      # the stored cotangents are released by the first backward, like any saved buffer
            raise RuntimeError("RenderLoss: backward through this loss a second time -- its stored cotangents were freed by the "
                               "first call (compute the loss again, or differentiate it once)")
        if g is None:
            return (None,) * 9
        need = ctx.needs_input_grad[:4]
        out = torch._foreach_mul([c for c, n in zip(cot, need) if n], g)      # one launch
        it = iter(out)
        return (*[(next(it) if n else None) for n in need], None, None, None, None, None)


class Aggregate(torch.autograd.Function):
    """RayTransformer.forward (ray_transformer.py:175-322) as the reference exposes it: the frustum lookup `fea_volume`
    and the pair similarity cond_info['feat_info'] are INPUTS.  ``apply(frame, weights, points (P,3), RN, SN,
    vol24 (P,24), sim8 (P,8), *params)`` -> ``radiance (P,3), srdf (RN,SN), xy (NV,P,2)``.  Differentiable w.r.t. vol24
    and the parameters (sim8 descends from the frozen matching features)."""

    @staticmethod
    def forward(ctx, frame, weights, points, RN, SN, vol24, sim8, *params):
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not as a zero tensor (a fill launch each)
        P = RN * SN
        zeros3 = torch.zeros(P, 3, dtype=torch.float32, device=points.device)
        # every point is its own "ray" of one sample: position = o + 0 * d, exact
        x, rgbm, dirs, dbg = ops.project_gather(frame, weights, points, zeros3, zeros3[:, :1].contiguous(), want_xy=True,
                                                vol24_in=vol24, sim8_in=sim8)
        prec = weights.mode()
        radiance, srdf, agg = ops.aggregate(weights, x, rgbm, dirs, RN, SN, keep_workspace=True, precision=prec)
        ctx.frame, ctx.weights, ctx.dims, ctx.precision = frame, weights, (RN, SN), prec
        ctx.save_for_backward(points, zeros3, x, rgbm, dirs, sim8, agg["token0"])
        ctx.mark_non_differentiable(dbg["xy"])
        return radiance, srdf, dbg["xy"]

    @staticmethod
    def backward(ctx, d_radiance, d_srdf, _d_xy):
        points, zeros3, x, rgbm, dirs, sim8, token0 = ctx.saved_tensors
        RN, SN = ctx.dims
        dev = x.device
        W = ctx.weights
        if d_radiance is None:
            d_radiance = torch.zeros(RN * SN, 3, device=dev)
        if d_srdf is None:
            d_srdf = torch.zeros(RN, SN, device=dev)
        grads = ops.GradBuffer(dev)
        d_pv, _ = ops.aggregate_bwd(W, grads, x, rgbm, dirs, token0, RN, SN, d_radiance, d_srdf, precision=ctx.precision)
        ops.project_gather_bwd(ctx.frame, W, grads, points, zeros3, zeros3[:, :1].contiguous(), sim8, d_pv, None, None,
                               precision=ctx.precision)
        gpar = [grads.grad(k) for k in ops.RAW_WEIGHT_KEYS]
        need = ctx.needs_input_grad
        d_vol24 = d_pv[:, :24].contiguous() if need[5] else None
        return (None, None, None, None, None, d_vol24, None, *[g if n else None for g, n in zip(gpar, need[7:])])


class Composite(torch.autograd.Function):
    """VolumeRenderer.render (renderer.py:7-48) with its adjoint: ``apply(z, radiance (RN,SN,3), srdf, variance)``."""

    @staticmethod
    def forward(ctx, z, radiance, srdf, variance):
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not as a zero tensor (a fill launch each)
        rgb, depth, opacity, weight = ops.composite(z, radiance, srdf, variance.detach().reshape(1))
        ctx.save_for_backward(z, radiance, srdf, variance)
        return rgb, depth, opacity, weight

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_opacity, d_weight):
        z, radiance, srdf, variance = ctx.saved_tensors
        d_rad, d_srdf, d_var = ops.composite_bwd(z, radiance, srdf, variance.detach().reshape(1), d_rgb, d_depth, d_opacity,
                                                 d_weight)
        return None, d_rad, d_srdf, d_var.reshape(variance.shape)


def flat_volumes(feature_volume: dict):
    """The six sampled volumes in the order RenderPass expects."""
    out = []
    for st in STAGES:
        out += [feature_volume[st]["feature_volume"], feature_volume[st]["weight_volume"]]
    return out
