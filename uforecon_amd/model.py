"""Host-side mirror of the reference's per-ray API (code1/model.py and friends) on top of libufr.so.

Same class names, call signatures, return conventions and ``state_dict`` keys as the reference for
this path, so the caller (the Lightning hooks in code1/model.py:492-842) can swap them in
(INTEGRATION.md).  The modules only OWN the parameters; every computation is a HIP kernel reached
through the C ABI -- there is no eager/CPU fallback, a missing library or a CPU tensor raises.

Inference (``extract_geometry``) runs the fused whole-path entry point; the training / validation signature
(``infer(extract_geometry=False)``, what ``training_step`` calls, code1/model.py:540-548) runs the stepwise entry
points under ``uforecon_amd.autograd`` so that ``loss.backward()`` reaches every ``ray_transformer.*`` parameter,
``deviation_network.variance`` and the six sampled volumes through the ``ufr_*_bwd`` kernels.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from ._lib import UfrError


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


# ---- B > 1: one frame per batch element.  The reference writes `infer` over a leading B (model.py:409-427) and ships
# `--batch_size 2` as the default (main.py:43); its volume lookup indexes the cost volumes by VIEW along dim 0
# (model.py:363-364), so the volumes of several frames can only arrive stacked frame-major along that dim, or as one dict
# per frame.  Every element is an independent frame: the mirror walks them with one FrameHandle each and concatenates the
# results the way the reference's `(B RN) ...` rearranges lay them out.
def _frame_of(batch: dict, b: int, B: int) -> dict:
    out = {}
    for k, v in batch.items():
        if k == "depth_info" and isinstance(v, torch.Tensor) and v.dim() == 4 and v.shape[0] == 1 and B > 1:
            # the reference's layout: stage-3 depth "(B V) H W" unsqueezed to (1, B*V, H, W) (model.py:530-531): view-major
            # per frame, like the volumes (_volumes_of)
            if v.shape[1] % B:
                raise UfrError(f"depth_info: {v.shape[1]} maps are not a multiple of the batch size {B}")
            nv = v.shape[1] // B
            out[k] = v[:, b * nv:(b + 1) * nv]
        elif isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == B:
            out[k] = v[b:b + 1]
        elif isinstance(v, (list, tuple)) and len(v) == B and k != "start_idx":
            out[k] = type(v)(v[b:b + 1])
        else:
            out[k] = v
    return out


def _volumes_of(feature_volume, b: int, B: int):
    if feature_volume is None:
        return None
    if isinstance(feature_volume, (list, tuple)):
        if len(feature_volume) != B:
            raise UfrError(f"feature_volume: {len(feature_volume)} frames for a batch of {B}")
        return feature_volume[b]
    out = {}
    for st, d in feature_volume.items():
        n = d["feature_volume"].shape[0]
        if n % B:
            raise UfrError(f"feature_volume[{st}]: leading dim {n} is not a multiple of the batch size {B}")
        nv = n // B
        out[st] = {k: v[b * nv:(b + 1) * nv] for k, v in d.items()}
    return out


def _match_of(match_feature, b: int):
    return None if match_feature is None else [t[b:b + 1] for t in match_feature]


# --------------------------------------------------------------------------- small modules
class SingleVarianceNetwork(nn.Module):
    """code1/encoder_utils/single_variance_network.py:5-11 (state_dict key: ``variance``)."""

    def __init__(self, init_val: float):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        return torch.ones([len(x), 1]).type_as(x) * torch.exp(self.variance * 10.0)


class FixedSampler:
    """code1/encoder_utils/sampler.py:7-50.  ``uniforms`` (SN,RN) may be passed explicitly; by default
    they are drawn exactly like the reference does (CPU generator, shape (SN,RN))."""

    def __init__(self, point_num: int = 64, sample_radius: float = 1.3):
        self.sample_radius = sample_radius
        self.point_num = point_num

    def sample_ray(self, ray_o, ray_d, jitter=True, near_z=None, far_z=None, uniforms: Optional[torch.Tensor] = None):
        RN = ray_d.shape[0]
        if near_z is None:
            mid = -(ray_o * ray_d).sum(-1)
            near_z, far_z = mid - self.sample_radius, mid + self.sample_radius
        if uniforms is None:
            uniforms = torch.rand(self.point_num, RN) if jitter else torch.full((self.point_num, RN), 0.5)
        U = uniforms.to(ray_d.device, torch.float32).contiguous()
        z = ops.sample_fixed(near_z.reshape(-1).float().contiguous(), far_z.reshape(-1).float().contiguous(), U)
        ray_o = ray_o.float().contiguous()
        ray_d = ray_d.float().contiguous()
        points_x = ops.points(ray_o if ray_o.numel() == 3 else ray_o, ray_d, z)
        points_d = ray_d[:, None, :].expand(RN, self.point_num, 3)
        return points_x, z.clone(), points_d


class ImportanceSampler:
    """code1/encoder_utils/sampler.py:53-108."""

    def __init__(self, point_num: int = 128):
        self.point_num = point_num

    def sample_ray(self, ray_o, ray_d, weight, z_val, uniforms: Optional[torch.Tensor] = None):
        RN = z_val.shape[0]
        if uniforms is None:
            uniforms = torch.rand(self.point_num, RN)
        U2 = uniforms.to(z_val.device, torch.float32).contiguous()
        z_fine, z_all = ops.sample_importance_merge(weight.float().contiguous(), z_val.float().contiguous(), U2)
        self.last_merged = z_all  # coarse+fine sorted (model.py:466-470), reused by UFORecon.infer
        ray_d = ray_d.float().contiguous()
        points_x = ops.points(ray_o.float().contiguous(), ray_d, z_fine)
        points_d = ray_d[:, None, :].expand(RN, self.point_num, 3)
        return points_x, z_fine, points_d


class VolumeRenderer:
    """code1/encoder_utils/renderer.py:3-48."""

    def __init__(self, args=None):
        self.args = args

    def render(self, z_val, radiance, geo_value, cos_anneal_ratio=1.0, deviation_network=None):
        if cos_anneal_ratio != 1.0:
            raise UfrError("cos_anneal_ratio != 1.0 is never used by the reference (model.py:338-341)")
        var = deviation_network.variance
        z, rad, geo = z_val.float().contiguous(), radiance.float().contiguous(), geo_value.float().contiguous()
        if _wants_grad(rad, geo, var):
            rgb, depth, opacity, weight = ag.Composite.apply(z, rad, geo, var)
        else:
            rgb, depth, opacity, weight = ops.composite(z, rad, geo, var.detach().reshape(1).float().contiguous())
        inv_s = torch.exp(var * 10.0).clip(1e-6, 1e6).reshape(1, 1)                  # renderer.py:25, 48
        return rgb, depth, opacity, weight, 1.0 / inv_s


# --------------------------------------------------------------------------- parameter containers
class _LoFTRLayer(nn.Module):
    """Parameter layout of attention/transformer.py:7-33 (all Linear bias-free)."""

    def __init__(self, d_model: int):
        super().__init__()
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(),
                                 nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _LocalFeatureTransformer(nn.Module):
    """attention/transformer.py:61-77: one 'self' layer, xavier-uniform on every matrix."""

    def __init__(self, d_model: int):
        super().__init__()
        self.layers = nn.ModuleList([_LoFTRLayer(d_model)])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class _ViewToken(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.register_parameter("view_token", nn.Parameter(torch.randn([1, dim])))


class _DepthCode(nn.Module):
    """Buffers of PositionalEncoding_NeRF(num_freqs=4, d_in=1) (ray_transformer.py:29-51) -- kept so the
    state_dict matches; the kernel has them as constants."""

    def __init__(self):
        super().__init__()
        freqs = np.pi * 2.0 ** torch.arange(0, 4)
        self.register_buffer("_freqs", torch.repeat_interleave(freqs, 2).view(1, -1, 1))
        ph = torch.zeros(8)
        ph[1::2] = np.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))


def _mlp3(i, h1, h2, o):
    return nn.Sequential(nn.Linear(i, h1), nn.ReLU(inplace=True), nn.Linear(h1, h2), nn.ReLU(inplace=True), nn.Linear(h2, o))


class RayTransformer(nn.Module):
    """code1/ray_transformer.py:86-322 for the shipped configuration (correlation volumes, explicit
    similarity, MVS depth guide with positional encoding, no direction-SRDF)."""

    def __init__(self, args=None, img_feat_dim=32, fea_volume_dim=24, sim_feat_dim=26):
        super().__init__()
        self.args = args
        for flag, want in (("volume_type", "correlation"), ("explicit_similarity", True), ("depth_pos_encoding", True),
                           ("use_dir_srdf", False)):
            if args is not None and hasattr(args, flag) and getattr(args, flag) != want:
                raise UfrError(f"RayTransformer: args.{flag}={getattr(args, flag)!r} is outside the HIP path "
                               f"(every shipped script uses {want!r})")
        if img_feat_dim != 32 or fea_volume_dim != 24:
            raise UfrError("RayTransformer: feature dims are fixed to 32 / 24 by the kernels")
        self.depthcode = _DepthCode()
        self.pre_sim_mlp = _mlp3(8, 32, 32, 16)
        self.density_view_transformer = _LocalFeatureTransformer(80)
        self.density_ray_transformer = _LocalFeatureTransformer(88)
        self.DensityMLP = _mlp3(88, 32, 16, 1)
        self.viewToken = _ViewToken(80)
        self.linear_radianceweight_1_softmax = _mlp3(83, 16, 8, 1)
        self._packed = {}        # storage identity of the 40 tensors -> [PackedWeights, version counters]

    # the packed copy follows the parameters.  Same storage, new version counters (an optimizer step, load_state_dict):
    # re-pack IN PLACE, asynchronously (no allocation, no host synchronisation per training step); other storage (moved to
    # another device, a different `variance` tensor): a new packed copy, at most two are kept (UFORecon's real variance and
    # the stub of RayTransformer.forward).  An update that bypasses the version counter (`p.data = ...`, `set_`) needs
    # invalidate_packed().
    def packed_weights(self, variance: torch.Tensor, precision: Optional[int] = None) -> ops.PackedWeights:
        params = {"ray_transformer." + k: v for k, v in self.state_dict(keep_vars=True).items()}
        params["deviation_network.variance"] = variance
        ptrs = tuple((v.data_ptr(), str(v.device)) for v in params.values())
        vers = tuple(v._version for v in params.values())
        ent = self._packed.get(ptrs)
        if ent is None:
            if len(self._packed) >= 2:
                self._packed.clear()
            # args.ufr_input_abs_max (optional): the bound of the token features the fp16 planes are laid out for
            bound = getattr(self.args, "ufr_input_abs_max", None) if self.args is not None else None
            ent = self._packed[ptrs] = [ops.PackedWeights(params, precision, input_abs_max=bound), vers]
        elif ent[1] != vers:
            ent[0].repack()
            ent[1] = vers
        ent[0].precision = precision
        return ent[0]

    def invalidate_packed(self) -> None:
        """Forget the packed copies (after parameter updates that do not bump the tensors' version counters)."""
        self._packed.clear()

    def forward(self, point3D, batch, source_imgs_feat, fea_volume=None, cond_info=None, points_projected=None,
                mask_valid=None):
        """ray_transformer.py:175-322 -> ``(radiance (B*RN*SN,3), srdf (B*RN,SN,1), points_in_pixel (B,NV,2,RN,SN))``.

        ``fea_volume`` (B,RN,SN,24) is the blended frustum lookup and ``cond_info['feat_info']`` (B,RN,SN,8) the pair
        similarity, as ``sample2rgb`` hands them over (model.py:324-333).  ``points_projected`` / ``mask_valid`` must be the
        projection of ``point3D`` by ``batch['source_poses']`` (they always are, model.py:237): the kernel recomputes it
        (bit-identical arithmetic) and returns it as ``points_in_pixel``.  Differentiable w.r.t. ``fea_volume`` and the
        parameters."""
        B, RN, SN, _ = point3D.shape
        if fea_volume is None or cond_info is None or "feat_info" not in cond_info:
            raise UfrError("RayTransformer.forward needs fea_volume and cond_info['feat_info'] "
                           "(the shipped configuration: correlation volumes + explicit similarity)")
        if B != 1:      # one frame per batch element; radiance / srdf are "(B RN SN) C" / "(B RN) SN 1" upstream
            outs = [self.forward(point3D[b:b + 1], _frame_of(batch, b, B), source_imgs_feat[b:b + 1], fea_volume[b:b + 1],
                                 {"feat_info": cond_info["feat_info"][b:b + 1]}) for b in range(B)]
            return tuple(torch.cat([o[i] for o in outs], 0) for i in range(3))
        dev = source_imgs_feat.device
        keyed = [source_imgs_feat] + [batch[k] for k in ("source_imgs", "depth_info", "source_poses", "source_poses_inv",
                                                          "ref_pose_inv", "w2cs")]
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in keyed) + (batch.get("start_idx", 1),)
        lite = getattr(self, "_lite", None)
        if lite is None:
            lite = self._lite = {}
        if key not in lite:               # a few frames are kept: the elements of a batch alternate
            if len(lite) >= 4:
                lite.clear()
            lite[key] = (ops.FrameHandle(batch, source_imgs_feat, None, None), keyed)   # pins the keyed tensors (their addresses are the key)
        frame = lite[key][0]
        variance = getattr(self, "_variance_stub", None)
        if variance is None or variance.device != dev:
            variance = self._variance_stub = torch.zeros((), device=dev)       # not an input of this module
        W = self.packed_weights(variance, getattr(self, "precision", None))
        P = RN * SN
        pts = point3D.reshape(P, 3).detach().float().contiguous()
        vol24 = fea_volume.reshape(P, 24).float().contiguous()
        sim8 = cond_info["feat_info"].reshape(P, 8).detach().float().contiguous()
        sd = dict(self.named_parameters())
        params = [sd[k[len("ray_transformer."):]] for k in ops.RAW_WEIGHT_KEYS[:-1]] + [variance]
        radiance, srdf, xy = ag.Aggregate.apply(frame, W, pts, RN, SN, vol24, sim8, *params)
        points_in_pixel = xy.reshape(1, -1, RN, SN, 2).permute(0, 1, 4, 2, 3)
        return radiance, srdf.reshape(B * RN, SN, 1), points_in_pixel


# --------------------------------------------------------------------------- orchestrator
class UFORecon(nn.Module):
    """The per-ray half of code1/model.py:28-482 (``infer`` / ``sample2rgb``).  Parameters live under the
    reference's names (``ray_transformer.*``, ``deviation_network.variance``), so a reference
    checkpoint loads with ``load_state_dict(strict=False)`` (the encoder keys are not ours)."""

    def __init__(self, args, precision: Optional[int] = None, overlap: bool = True, tape_in_forward: bool = True):
        """``precision``: matrix precision of this model's dense layers (ops.PRECISION_FP32 / PRECISION_16BIT; None = the
        process default at call time, include/ufr.h).  It travels with every call -- there is no global state to flip
        between a forward and its backward.
        ``overlap`` / ``tape_in_forward``: how the training step's backward is scheduled (autograd.RenderOptions): its
        independent stages on three streams or all on the caller's; the activation tape recorded by the forward kernels
        or by the backward.  Same gradients either way (tests/test_gpu_backward.py); per model, not per process."""
        super().__init__()
        self.args = args
        self.precision = precision
        self.overlap, self.tape_in_forward = bool(overlap), bool(tape_in_forward)
        if getattr(args, "extract_geometry", False):
            self.point_num, self.point_num_2 = args.test_sample_coarse, args.test_sample_fine
        else:
            self.point_num, self.point_num_2 = args.coarse_sample, args.fine_sample
        self.fixed_sampler = FixedSampler(point_num=self.point_num)
        self.importance_sampler = ImportanceSampler(point_num=self.point_num_2)
        self.deviation_network = SingleVarianceNetwork(0.3)
        self.renderer = VolumeRenderer(args)
        self.ray_transformer = RayTransformer(args=args)
        self._frames = {}         # frame key -> (FrameHandle, the keyed tensors); a few entries: the elements of a batch alternate
        self._ws: Optional[ops.RenderWorkspace] = None

    # ---- per-frame state: channel-last copies are cached on the identity of the frame tensors
    def frame_handle(self, batch, source_imgs_feat, feature_volume, match_feature) -> ops.FrameHandle:
        # every tensor the handle snapshots (maps, volumes, and the view-dependent camera state: rays, poses, near/far)
        # enters the key with its storage address AND version counter, so a new render view of the same sources or an
        # in-place update never reuses stale state
        keyed = [source_imgs_feat, match_feature[0]]
        keyed += [batch[k] for k in ("source_imgs", "depth_info", "source_poses", "source_poses_inv", "ref_pose_inv", "w2cs",
                                     "near_fars", "ray_o", "ray_d", "cam_ray_d") if k in batch]
        keyed += ag.flat_volumes(feature_volume)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in keyed) + (batch.get("start_idx", 1),)
        ent = self._frames.get(key)
        if ent is None:
            if len(self._frames) >= max(2, int(batch["source_imgs"].shape[0])):
                self._frames.clear()
            # the entry pins every keyed tensor: its address cannot be recycled while the entry is cached
            ent = self._frames[key] = (ops.FrameHandle(batch, source_imgs_feat, feature_volume, match_feature), keyed)
        return ent[0]

    def _weights(self) -> ops.PackedWeights:
        return self.ray_transformer.packed_weights(self.deviation_network.variance, self.precision)

    def sample2rgb(self, batch, points_x, z_val, ray_d, ray_idx, source_imgs_feat, feature_volume, match_feature):
        """model.py:308-348.  ``points_x`` must be ``ray_o + z_val * ray_d`` (it always is in the reference);
        the kernels recompute the positions from ``z_val``."""
        B, RN, SN, _ = points_x.shape
        if B != 1:      # one frame per element (ray_d is "(B RN) 3" upstream, model.py:409-410)
            rd = ray_d.reshape(B, RN, 3)
            outs = [self.sample2rgb(_frame_of(batch, b, B), points_x[b:b + 1], z_val[b:b + 1], rd[b], ray_idx[b:b + 1],
                                    source_imgs_feat[b:b + 1], _volumes_of(feature_volume, b, B), _match_of(match_feature, b))
                    for b in range(B)]
            return tuple(torch.cat([o[i] for o in outs], 0) for i in range(6)) + (outs[0][6],)
        fh = self.frame_handle(batch, source_imgs_feat, feature_volume, match_feature)
        ray_o = batch["ray_o"][0].float().contiguous()
        z = z_val.reshape(RN, SN).detach().float().contiguous()
        rgb, depth, opacity, weight, srdf, pip = self._render_pass(fh, ray_o, ray_d.reshape(RN, 3).float().contiguous(), z,
                                                                   feature_volume)
        return rgb[None], depth[None], srdf.reshape(RN, SN, 1), opacity[None], weight[None], pip, self._variance_out()

    def _live_params(self):
        sd = {"ray_transformer." + k: v for k, v in self.ray_transformer.named_parameters()}
        sd.update({"ray_transformer." + k: v for k, v in self.ray_transformer.named_buffers()})
        sd["deviation_network.variance"] = self.deviation_network.variance
        return [sd[k] for k in ops.RAW_WEIGHT_KEYS]

    def _variance_out(self):
        """``1 / inv_s`` (renderer.py:25, 48) as a differentiable function of the parameter."""
        return 1.0 / torch.exp(self.deviation_network.variance.reshape(1, 1) * 10.0).clip(1e-6, 1e6)

    def _render_pass(self, fh, ray_o, ray_d, z, feature_volume):
        """gather -> aggregate -> composite for the samples ``z`` (RN,SN); differentiable when anything upstream is."""
        W = self._weights()
        RN, SN = z.shape
        params, vols = self._live_params(), ag.flat_volumes(feature_volume)
        if _wants_grad(*params, *vols):
            rgb, depth, opacity, weight, srdf, xy = ag.RenderPass.apply(fh, W, ray_o, ray_d, z, *params, *vols)
        else:
            x, rgbm, dirs, dbg = ops.project_gather(fh, W, ray_o, ray_d, z, want_xy=True)
            radiance, srdf, _ = ops.aggregate(W, x, rgbm, dirs, RN, SN)
            rgb, depth, opacity, weight = ops.composite(z, radiance.reshape(RN, SN, 3), srdf, W.variance.reshape(1))
            xy = dbg["xy"]
        # points_in_pixel (B,NV,2,RN,SN): the projected sample positions (ray_transformer.py:211-220, 322)
        return rgb, depth, opacity, weight, srdf, xy.reshape(1, -1, RN, SN, 2).permute(0, 1, 4, 2, 3)

    def infer(self, batch, ray_idx, source_imgs_feat, feature_volume=None, extract_geometry=False, match_feature=None,
              ray_idx_all=None, is_train=True, uniforms=None):
        """model.py:393-482.  ``uniforms=(U1 (SN,RN), U2 (PN,RN))`` pins the sampler randomness; by default
        they are drawn from the CPU generator in the reference's order and shapes."""
        B, RN = ray_idx.shape
        if B != 1:      # one frame per element; the sampler draws are (SN, B*RN) with "(B RN)" columns (sampler.py:42, 86)
            if uniforms is None:       # ONE draw per pass for the whole batch, in the reference's order and shapes
                only_coarse = bool(extract_geometry and getattr(self.args, "test_coarse_only", False))
                uniforms = (torch.rand(self.point_num, B * RN), None if only_coarse else torch.rand(self.point_num_2, B * RN))
            outs = []
            for b in range(B):
                u = None if uniforms is None else tuple(None if U is None else U[:, b * RN:(b + 1) * RN] for U in uniforms)
                outs.append(self.infer(_frame_of(batch, b, B), ray_idx[b:b + 1], source_imgs_feat[b:b + 1],
                                       _volumes_of(feature_volume, b, B), extract_geometry, _match_of(match_feature, b),
                                       ray_idx_all, is_train, u))
            n = len(outs[0])
            # every entry is "B ..." or "(B RN) ..." upstream: frame-major along dim 0; the variance (last of the 17) is shared
            return tuple(outs[0][i] if (n == 17 and i == 16) else torch.cat([o[i] for o in outs], 0) for i in range(n))
        dev = source_imgs_feat.device
        coarse_only = bool(extract_geometry and getattr(self.args, "test_coarse_only", False))
        if uniforms is None:
            U1 = torch.rand(self.point_num, RN)
            U2 = None if coarse_only else torch.rand(self.point_num_2, RN)
        else:
            U1, U2 = uniforms
        U1 = U1.to(dev, torch.float32).contiguous()
        U2 = None if U2 is None else U2.to(dev, torch.float32).contiguous()
        fh = self.frame_handle(batch, source_imgs_feat, feature_volume, match_feature)
        W = self._weights()
        if extract_geometry:
            if self._ws is None or self._ws.key[:3] != (self.point_num, 0 if coarse_only else self.point_num_2, fh.NV):
                self._ws = ops.RenderWorkspace(dev, self.point_num, 0 if coarse_only else self.point_num_2, fh.NV)
            out = ops.render_rays(fh, W, ray_idx.reshape(-1).to(dev), U1, U2, coarse_only=coarse_only, workspace=self._ws)
            ray_d = batch["ray_d"][0][:, ray_idx.reshape(-1)].t()
            points = batch["ray_o"][0][None, None, :] + out["z_all"][..., None] * ray_d[:, None, :]
            return out["srdf"][None], points[None], out["depth"][None], out["rgb"][None]   # model.py:475-478 / 452

        # training / validation signature: no near/far scaling (model.py:423), GT gathers (model.py:398-406);
        # differentiable through autograd.RenderPass; the importance sampler sees detached weights (model.py:456-457)
        idx = ray_idx.reshape(-1)
        ref_img = batch["ref_img"].reshape(B, 3, -1)
        rgb_gt = ref_img[:, :, idx].permute(0, 2, 1)
        depth_gt = batch["depths_h"][:, 0].reshape(B, -1)[:, idx]
        ray_d = batch["ray_d"][0][:, idx].t().float().contiguous()
        ray_o = batch["ray_o"][0].float().contiguous()
        near = batch["near_fars"][0, 0, 0].expand(RN).float().contiguous()
        far = batch["near_fars"][0, 0, 1].expand(RN).float().contiguous()
        z1 = ops.sample_fixed(near, far, U1)
        # both passes in one autograd node: the fine pass evaluates only its new samples per point and shares the coarse
        # samples' rows with the coarse pass, forwards and backwards (autograd.RenderTwoPass; same numbers as two sample2rgb
        # calls on z1 and on the merged z2, model.py:445, 472)
        params, vols = self._live_params(), ag.flat_volumes(feature_volume)
        # (the returned `variance` is five tiny launches on a scalar: queued HERE, in front of the render kernels, they run
        # while the host is still enqueueing those; behind them they sat on the step's critical path)
        variance = self._variance_out()
        # a backward will follow <=> something upstream wants a gradient: then the forward records its tape (autograd.py)
        opt = ag.RenderOptions(overlap=self.overlap, tape_in_forward=self.tape_in_forward,
                               record_tape=_wants_grad(*params, *vols))
        (rgb, depth, opacity, weight, srdf, xy1, rgb2, depth2, opacity2, weight2, srdf2, xy2, z2) = ag.RenderTwoPass.apply(
            fh, W, ray_o, ray_d, z1, U2, opt, *params, *vols)
        S2 = z2.shape[1]
        pip = xy1.reshape(1, -1, RN, self.point_num, 2).permute(0, 1, 4, 2, 3)
        pip2 = xy2.reshape(1, -1, RN, S2, 2).permute(0, 1, 4, 2, 3)
        return (rgb_gt, rgb[None], depth[None], depth_gt, srdf.reshape(RN, -1, 1), opacity[None], weight[None], pip,
                rgb2[None], depth2[None], srdf2.reshape(RN, S2, 1), opacity2[None], weight2[None], pip2,
                z1[None], z2[None], variance)                                                # model.py:480-482

    # ---- frame-level entry: the per-ray loop and post-processing of extract_geometry (model.py:810-842)
    def training_loss(self, outputs, batch):
        """The loss of ``training_step`` (code1/model.py:552-566) on the 17-tuple ``infer`` returns (model.py:480-482):
        ``weight_rgb (mse(rgb) + mse(rgb_2)) + weight_depth (l1(depth | valid gt) + l1(depth_2 | valid gt))``, as one
        autograd node on one kernel (ufr_render_loss).  -> ``loss ()``, ``parts`` = dict of the four terms the reference
        logs (train/rgb_coarse, rgb_fine, depth_ray_coarse, depth_ray_fine), detached.  GPU only."""
        rgb_gt, rgb, depth, depth_gt, rgb2, depth2 = (outputs[i] for i in (0, 1, 2, 3, 8, 9))
        if not depth_gt.is_cuda:
            raise UfrError("training_loss runs on the GPU only (no CPU implementation; oracle/ufo_oracle.py:training_loss is the checker)")
        loss, parts = ag.RenderLoss.apply(rgb, depth, rgb2, depth2, rgb_gt, depth_gt, batch["near_fars"],
                                          float(getattr(self.args, "weight_rgb", 1.0)), float(getattr(self.args, "weight_depth", 1.0)))
        return loss, {"rgb_coarse": parts[0], "rgb_fine": parts[1], "depth_ray_coarse": parts[2], "depth_ray_fine": parts[3]}

    def render_depth_map(self, batch, source_imgs_feat, feature_volume, match_feature, uniforms=None):
        """Every pixel of the render view in ONE call (the reference loops over chunks of ``test_ray_num`` = 800
        rays, model.py:815; here chunking is internal to ufr_render_rays and invisible).  Returns
        ``depths (H,W) float32`` = ray depth * cam_ray_d.z * scale_mat[0][0,0] (model.py:818-826) and
        ``rgbs (H,W,3) float32`` in [0,1], both on the device."""
        B, L, _, imgH, imgW = batch["source_imgs"].shape
        if B != 1:      # one depth map per element: (B, H, W), (B, H, W, 3)
            outs = [self.render_depth_map(_frame_of(batch, b, B), source_imgs_feat[b:b + 1], _volumes_of(feature_volume, b, B),
                                          _match_of(match_feature, b), uniforms) for b in range(B)]
            return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])
        dev = source_imgs_feat.device
        HW = imgH * imgW
        coarse_only = bool(getattr(self.args, "test_coarse_only", False))
        if uniforms is None:   # one draw per frame instead of one per 800-ray chunk: same distribution
            U1 = torch.rand(self.point_num, HW)
            U2 = None if coarse_only else torch.rand(self.point_num_2, HW)
        else:
            U1, U2 = uniforms
        U1 = U1.to(dev, torch.float32).contiguous()
        U2 = None if U2 is None else U2.to(dev, torch.float32).contiguous()
        fh = self.frame_handle(batch, source_imgs_feat, feature_volume, match_feature)
        PN = 0 if coarse_only else self.point_num_2
        if self._ws is None or self._ws.key[:3] != (self.point_num, PN, fh.NV):
            self._ws = ops.RenderWorkspace(dev, self.point_num, PN, fh.NV)
        ray_idx = torch.arange(HW, device=dev)
        out = ops.render_rays(fh, self._weights(), ray_idx, U1, U2, coarse_only=coarse_only, workspace=self._ws,
                              want_srdf=False)
        depths = out["depth_z"].view(imgH, imgW) * batch["scale_mat"][0][0, 0].to(dev)       # model.py:821, 826
        return depths, out["rgb"].view(imgH, imgW, 3)

    def extract_geometry(self, batch, source_imgs_feat, feature_volume, match_feature, out_dir=None, uniforms=None):
        """The per-ray part of model.py:760-842 plus its outputs: renders the frame and, when ``out_dir`` is given,
        writes the files ``save_depth_outputs`` describes.  (The encoder half of the reference's method -- building
        ``source_imgs_feat`` / ``feature_volume`` / ``match_feature`` -- stays with the caller.)"""
        depths, rgbs = self.render_depth_map(batch, source_imgs_feat, feature_volume, match_feature, uniforms)
        depths, rgbs = depths.cpu().numpy(), rgbs.cpu().numpy()
        if out_dir is not None:
            meta = batch["meta"][0]
            save_depth_outputs(out_dir, meta.split("-")[1], meta.split("-")[-1], depths, rgbs,
                               batch["extrinsic_render_view"][0].cpu().numpy(),
                               batch["intrinsic_render_view"][0].cpu().numpy())
        return depths, rgbs


def save_depth_outputs(out_dir, scan_name, ref_view, depths, rgbs, extrinsic, intrinsic):
    """The reference's wire format (model.py:828-842), byte for byte:
      <out_dir>/<scan>/depth/<view>.png   8-bit preview, depth / max(depth) * 255 (truncating cast)
      <out_dir>/rgb/<scan>/<view>.jpg     8-bit RGB, rgb * 255 (truncating cast)
      <out_dir>/depth/<scan>/<view>.npy   pickled dict {"depth": (H,W) f32, "extrinsic": 4x4, "intrinsic": 3x3}
    (`np.save` of a dict: readers need ``np.load(..., allow_pickle=True).item()``, as tsdf_fusion.py does)."""
    import os

    import numpy as np
    from PIL import Image

    os.makedirs(os.path.join(out_dir, scan_name, "depth"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "depth", scan_name), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "rgb", scan_name), exist_ok=True)
    depths = np.asarray(depths, np.float32)
    rgb8 = (np.asarray(rgbs).astype(np.float32) * 255).astype(np.uint8)
    depth8 = ((depths / np.max(depths)).astype(np.float32) * 255).astype(np.uint8)
    Image.fromarray(depth8).save(os.path.join(out_dir, scan_name, "depth", "%s.png" % ref_view))
    Image.fromarray(rgb8).save(os.path.join(out_dir, "rgb", scan_name, "%s.jpg" % ref_view))
    np.save(os.path.join(out_dir, "depth", scan_name, "%s.npy" % ref_view),
            {"depth": depths, "extrinsic": np.asarray(extrinsic), "intrinsic": np.asarray(intrinsic)})

