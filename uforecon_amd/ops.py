"""Torch-facing wrappers over the C ABI (include/ufr.h).

torch is plumbing here: it owns device memory (caching allocator) and the current HIP stream;
every computation is a libufr.so kernel.  All wrappers validate device / dtype / contiguity the
way TORCH_CHECK would and raise ``UfrError`` with the library's message on failure.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import UfrError

# state_dict key (reference names) for every pointer of ufr_raw_weights, in declaration order
_RT = "ray_transformer."
_VT = _RT + "density_view_transformer.layers.0."
_RY = _RT + "density_ray_transformer.layers.0."
RAW_WEIGHT_KEYS = (
    [_RT + f"pre_sim_mlp.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [_VT + k for k in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight",
                         "mlp.2.weight", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
    + [_RY + k for k in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight",
                         "mlp.2.weight", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
    + [_RT + f"DensityMLP.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [_RT + f"linear_radianceweight_1_softmax.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [_RT + "viewToken.view_token", "deviation_network.variance"]
)
RAW_WEIGHT_SHAPES = (
    [(32, 8), (32,), (32, 32), (32,), (16, 32), (16,)]
    + [(80, 80)] * 4 + [(160, 160), (80, 160)] + [(80,)] * 4
    + [(88, 88)] * 4 + [(176, 176), (88, 176)] + [(88,)] * 4
    + [(32, 88), (32,), (16, 32), (16,), (1, 16), (1,)]
    + [(16, 83), (16,), (8, 16), (8,), (1, 8), (1,)]
    + [(1, 80), ()]
)
assert len(RAW_WEIGHT_KEYS) == 40 == len(RAW_WEIGHT_SHAPES)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    if not isinstance(t, torch.Tensor):
        raise UfrError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise UfrError(f"{name}: must live on the GPU (got {t.device}); the per-ray path has no CPU implementation")
    if t.dtype != dtype:
        raise UfrError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise UfrError(f"{name}: must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str) -> Optional[int]:
    return None if t is None else _dev(t, name)


PRECISION_DEFAULT, PRECISION_FP32, PRECISION_16BIT = -1, 0, 1


def resolve_precision(precision: Optional[int]) -> int:
    """An explicit matrix precision (include/ufr.h: UFR_PRECISION_FP32 / _16BIT) for a call: None / PRECISION_DEFAULT
    resolve to the process default NOW, so that the value recorded with a forward is what its backward gets."""
    if precision is None or precision == PRECISION_DEFAULT:
        return get_matrix_precision()
    if precision not in (PRECISION_FP32, PRECISION_16BIT):
        raise UfrError(f"unknown matrix precision {precision!r}")
    return int(precision)


def status_poll(synchronize: bool = True, mask: int = 7) -> int:
    """ufr_status_poll_bits: raises UfrError when one of the ``mask`` bits of the device's sticky range status is set (1: an
    activation beyond the fp16x3 planes, 2: NaN among a transformer kernel's inputs, 4: a weight outside the planes'
    range), clearing those bits.  With ``synchronize=False`` only what an earlier launch has already delivered is reported
    (no host synchronisation)."""
    flags = C.c_int32(0)
    _lib.check(_lib.load().ufr_status_poll_bits(_stream(), int(bool(synchronize)), int(mask), C.byref(flags)), "ufr_status_poll")
    return int(flags.value)


import itertools

_PACK_SERIAL = itertools.count(1)


class PackedWeights:
    """ufr_raw_weights (pointers into the live parameters) + the MFMA-ordered packed copy.  ``precision``: the matrix
    precision the calls made with these weights use (None = the process default at call time).  ``input_abs_max``: an
    OPTIONAL floor of the bound of the feature maps and volume features (None = the library default, 4094): the exponents of
    the activations' fp16 planes follow the bound MEASURED per frame (``fit``: ufr_frame_prepare measures, ufr_weights_fit_frame
    re-derives the table when a frame exceeds what it serves), so no side information about a checkpoint's feature scale
    is needed.  Weights of any finite magnitude pack without further ado."""

    def __init__(self, params: Dict[str, torch.Tensor], precision: Optional[int] = None,
                 input_abs_max: Optional[float] = None):
        lib = _lib.load()
        self.precision = precision
        self.input_abs_max = input_abs_max
        self._keep = []
        ptrs = []
        for key, shape in zip(RAW_WEIGHT_KEYS, RAW_WEIGHT_SHAPES):
            if key not in params:
                raise UfrError(f"missing parameter {key}")
            t = params[key].detach()
            if tuple(t.shape) != tuple(shape):
                raise UfrError(f"{key}: shape {tuple(t.shape)}, expected {tuple(shape)}")
            t = t.contiguous()
            ptrs.append(_dev(t, key))
            self._keep.append(t)
        self.raw = _lib.RawWeights()
        C.memmove(C.byref(self.raw), (C.c_void_p * 40)(*ptrs), C.sizeof(self.raw))
        self.device = self._keep[0].device
        self.packed = torch.empty(lib.ufr_packed_weights_bytes() // 4, dtype=torch.float32, device=self.device)
        self.serial = next(_PACK_SERIAL)
        self.epoch = 0          # bumped by every pack: a frame is fitted once per (frame, epoch)
        self.repack(check=True)

    def repack(self, check: bool = False) -> None:
        """Call after the parameters changed in place (e.g. an optimizer step): the planes AND their exponents follow
        the parameters.  Asynchronous: a non-finite parameter raises the sticky status, which the next compute call (or
        ``status_poll``) reports; ``check=True`` synchronises and raises at once (construction does)."""
        lib = _lib.load()
        if self.input_abs_max is None:
            rc = lib.ufr_weights_pack(C.byref(self.raw), self.packed.data_ptr(), _stream())
        else:
            rc = lib.ufr_weights_pack_for(C.byref(self.raw), self.packed.data_ptr(), float(self.input_abs_max), _stream())
        _lib.check(rc, "ufr_weights_pack")
        self.epoch += 1
        if check:
            status_poll(True, mask=4)    # the pack's own bit: an unrelated, still unreported overflow must not fail a valid pack

    def fit(self, frame: "FrameHandle") -> None:
        """The activation exponents follow ``frame``'s measured feature bound (ufr_weights_fit_frame): one tiny asynchronous
        kernel, issued once per (frame, pack) -- a no-op on the device unless the frame exceeds the bound the table serves."""
        key = (self.serial, self.epoch)       # a process-wide serial number, not id(self): an id is reused after collection
        if key in frame._fitted:
            return
        _lib.check(_lib.load().ufr_weights_fit_frame(self.packed.data_ptr(), C.byref(frame.frame), _stream()), "ufr_weights_fit_frame")
        if len(frame._fitted) > 8:
            frame._fitted.clear()
        frame._fitted.add(key)

    def mode(self) -> int:
        return resolve_precision(self.precision)

    def scale_exponents(self) -> Dict[str, tuple]:
        """(s_M, a_M) per dense matrix, read back from the packed blob's scale table (synchronises; for tests and reports)."""
        import math
        lib = _lib.load()
        off, n = lib.ufr_packed_scale_table_offset(), lib.ufr_packed_scale_table_entries()
        t = self.packed[off:off + 4 * n].cpu().view(n, 4)
        names = ("vt_q", "vt_k", "vt_v", "vt_merge", "vt_mlp0", "vt_mlp2", "rt_q", "rt_k", "rt_v", "rt_merge", "rt_mlp0",
                 "rt_mlp2", "dm0", "dm2", "dm4", "rw0", "rw2", "rw4")
        return {nm: (int(round(math.log2(float(t[i, 3])))), int(round(math.log2(float(t[i, 0]))))) for i, nm in enumerate(names)}

    @property
    def variance(self) -> torch.Tensor:
        return self._keep[-1]


class FrameHandle:
    """Per-frame channel-last copies + camera constants (ufr_frame_prepare)."""

    def __init__(self, batch: dict, source_imgs_feat: torch.Tensor, feature_volume: Optional[dict], match_feature,
                 stages=("stage1", "stage2", "stage3")):
        """``feature_volume`` / ``match_feature`` may be None: such a handle only serves ``project_gather`` calls that
        pass ``vol24_in`` / ``sim8_in`` (RayTransformer.forward gets them as arguments)."""
        lib = _lib.load()
        imgs = batch["source_imgs"]
        if imgs.shape[0] != 1:
            raise UfrError("the per-ray path handles one frame per call (B=1), as the reference's test loop does")
        _, NV, _, H, W = imgs.shape
        s_idx = batch["start_idx"] if "start_idx" in batch else 1  # model.py:313
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        self._keep = dict(
            imgs=f32(imgs[0]), depth=f32(batch["depth_info"][0]), feat=f32(source_imgs_feat[0]),
        )
        if match_feature is not None:
            self._keep["match"] = f32(match_feature[0][0])
        if tuple(self._keep["depth"].shape) != (NV, H, W):
            # (1, B*V, H, W) of a B > 1 batch handed over whole would silently render every frame with frame 0's maps
            raise UfrError(f"depth_info shape {tuple(batch['depth_info'].shape)}: expected (1, {NV}, {H}, {W})")
        d = _lib.FrameDesc()
        d.NV, d.H, d.W = NV, H, W
        d.source_imgs = _dev(self._keep["imgs"], "source_imgs")
        d.depth_info = _dev(self._keep["depth"], "depth_info")
        d.feat = _dev(self._keep["feat"], "source_imgs_feat")
        if tuple(self._keep["feat"].shape) != (NV, 32, H // 4, W // 4):
            raise UfrError(f"source_imgs_feat shape {tuple(source_imgs_feat.shape)}")
        if match_feature is not None:
            d.match = _dev(self._keep["match"], "match_feature")
            if tuple(self._keep["match"].shape) != (NV, 32 * (NV - 1), H // 4, W // 4):
                raise UfrError(f"match_feature shape {tuple(match_feature[0].shape)}")
        for i, st in enumerate(stages if feature_volume is not None else ()):
            fv = f32(feature_volume[st]["feature_volume"])
            wv = f32(feature_volume[st]["weight_volume"])
            if fv.shape[0] != NV or fv.shape[1] != 8 or wv.shape[1] != 1 or fv.shape[2:] != wv.shape[2:]:
                raise UfrError(f"{st}: volume shapes {tuple(fv.shape)} / {tuple(wv.shape)}")
            self._keep[f"fv{i}"], self._keep[f"wv{i}"] = fv, wv
            d.vol_feat[i], d.vol_weight[i] = _dev(fv, st), _dev(wv, st)
            d.vol_D[i], d.vol_H[i], d.vol_W[i] = fv.shape[2], fv.shape[3], fv.shape[4]
        # small camera constants: host copies
        host = lambda t: t.detach().to("cpu", torch.float32).contiguous()
        self._host = dict(
            poses=host(batch["source_poses"][0]),
            cam_pos=host(batch["source_poses_inv"][0, :, :3, 3]),
            ref_pos=host(batch["ref_pose_inv"][0, :3, 3]),
            w2c_z=host(batch["w2cs"][0, s_idx:, 2, :]),
        )
        if self._host["w2c_z"].shape[0] != NV:
            raise UfrError("w2cs[s_idx:] does not match the number of source views")
        fp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_float))
        d.source_poses, d.source_cam_pos = fp(self._host["poses"]), fp(self._host["cam_pos"])
        d.ref_cam_pos, d.w2c_row2 = fp(self._host["ref_pos"]), fp(self._host["w2c_z"])
        if "near_fars" in batch:
            nf = batch["near_fars"][0][0].detach().cpu()
            d.vol_near, d.vol_far = float(nf[0]), float(nf[1])
        elif feature_volume is not None:
            raise UfrError("batch['near_fars'] is required to look up the frustums (model.py:328)")
        nbytes = lib.ufr_frame_workspace_bytes(C.byref(d))
        if nbytes == 0:
            raise UfrError("frame: " + lib.ufr_last_error().decode())
        self.device = self._keep["imgs"].device
        self.workspace = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        self.frame = _lib.Frame()
        _lib.check(lib.ufr_frame_prepare(C.byref(d), self.workspace.data_ptr(), nbytes, C.byref(self.frame), _stream()),
                   "ufr_frame_prepare")
        self.NV, self.H, self.W = NV, H, W
        self._fitted = set()    # (id(PackedWeights), its pack epoch) this frame's feature bound was handed to
        self.near_z, self.far_z = float(d.vol_near), float(d.vol_far)
        # view-dependent state of the whole-frame renderer (absent when the handle only serves RayTransformer.forward)
        self.ray_o = [float(v) for v in batch["ray_o"][0].detach().cpu()] if "ray_o" in batch else None
        self.ray_d = f32(batch["ray_d"][0]) if "ray_d" in batch else None
        self.cam_ray_d = f32(batch["cam_ray_d"][0]) if "cam_ray_d" in batch else None


# ----------------------------------------------------------------------------- per-op wrappers
def sample_fixed(near: torch.Tensor, far: torch.Tensor, U: torch.Tensor) -> torch.Tensor:
    SN, RN = U.shape
    z = torch.empty(RN, SN, dtype=torch.float32, device=U.device)
    _lib.check(_lib.load().ufr_sample_fixed(_dev(near, "near"), _dev(far, "far"), _dev(U, "U"), z.data_ptr(), RN, SN,
                                            _stream()), "ufr_sample_fixed")
    return z


def sample_importance_merge(weight: torch.Tensor, z: torch.Tensor, U2: torch.Tensor, want_fine: bool = True):
    RN, SN = z.shape
    PN = U2.shape[0]
    z_fine = torch.empty(RN, PN, dtype=torch.float32, device=z.device) if want_fine else None
    z_all = torch.empty(RN, SN + PN, dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().ufr_sample_importance_merge(
        _dev(weight, "weight"), _dev(z, "z"), _dev(U2, "U2"), _opt(z_fine, "z_fine"), z_all.data_ptr(), RN, SN, PN,
        _stream()), "ufr_sample_importance_merge")
    return z_fine, z_all


def points(ray_o: torch.Tensor, ray_d: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    RN, SN = z.shape
    stride = 0 if ray_o.numel() == 3 else 3
    out = torch.empty(RN, SN, 3, dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().ufr_points(_dev(ray_o, "ray_o"), stride, _dev(ray_d, "ray_d"), _dev(z, "z"), out.data_ptr(),
                                      RN, SN, _stream()), "ufr_points")
    return out


def project_gather(frame: FrameHandle, weights: PackedWeights, ray_o: torch.Tensor, ray_d: torch.Tensor,
                   z: torch.Tensor, debug: bool = False, want_sim8: bool = False, want_xy: bool = False,
                   vol24_in: Optional[torch.Tensor] = None, sim8_in: Optional[torch.Tensor] = None,
                   sim8_out: Optional[torch.Tensor] = None, out=None):
    """``out`` = (x (P,NV,80), rgb (P,NV,4), dirs (P,NV,4)): write into these (row ranges of the two-pass step's pool)."""
    RN, SN = z.shape
    P, NV, dev = RN * SN, frame.NV, z.device
    if out is not None:
        x, rgb, dirs = out
        assert x.shape == (P, NV, _lib.TOKEN_DIM) and rgb.shape == (P, NV, 4) and dirs.shape == (P, NV, 4)
        assert x.is_contiguous() and rgb.is_contiguous() and dirs.is_contiguous()
    else:
        x = torch.empty(P, NV, _lib.TOKEN_DIM, dtype=torch.float32, device=dev)
        rgb = torch.empty(P, NV, 4, dtype=torch.float32, device=dev)
        dirs = torch.empty(P, NV, 4, dtype=torch.float32, device=dev)
    dbg = {}
    if debug:
        dbg = dict(sim8=torch.empty(P, 8, device=dev), vol24=torch.empty(P, 24, device=dev),
                   xy=torch.empty(NV, P, 2, device=dev), mask_z=torch.empty(NV, P, device=dev))
    else:
        if sim8_out is not None:   # ... into a row range of the two-pass step's pool
            dbg["sim8"] = sim8_out
        elif want_sim8:   # the backward of pre_sim_mlp needs its input
            dbg["sim8"] = torch.empty(P, 8, device=dev)
        if want_xy:     # points_in_pixel of the reference's return tuples
            dbg["xy"] = torch.empty(NV, P, 2, device=dev)
    stride = 0 if ray_o.numel() == 3 else 3
    weights.fit(frame)      # the tokens gathered here meet `weights`' dense layers next: their exponents follow this frame
    _lib.check(_lib.load().ufr_project_gather(
        C.byref(frame.frame), C.byref(weights.raw), _dev(ray_o, "ray_o"), stride, _dev(ray_d, "ray_d"), _dev(z, "z"),
        RN, SN, x.data_ptr(), rgb.data_ptr(), dirs.data_ptr(), _opt(dbg.get("sim8"), "sim8"),
        _opt(dbg.get("vol24"), "vol24"), _opt(dbg.get("xy"), "xy"), _opt(dbg.get("mask_z"), "mask_z"),
        _opt(vol24_in, "vol24_in"), _opt(sim8_in, "sim8_in"), _stream()),
        "ufr_project_gather")
    return x, rgb, dirs, dbg


def _max_points(NV: int) -> int:
    """points one transformer launch addresses (32-bit offsets: P * (NV + 1) * 80 < 2^30, include/ufr.h)"""
    return ((1 << 30) - 1) // ((NV + 1) * _lib.TOKEN_DIM)


def aggregate(weights: PackedWeights, x: torch.Tensor, rgb: torch.Tensor, dirs: torch.Tensor, RN: int, SN: int,
              debug: bool = False, keep_workspace: bool = False, precision: Optional[int] = None):
    lib = _lib.load()
    NV, dev = x.shape[1], x.device
    P = RN * SN
    rays_max = _max_points(NV) // SN
    if RN > rays_max >= 1:      # beyond one launch's 32-bit addressing: rays are independent, so call per ray range
        parts = [aggregate(weights, x[r0 * SN:(r0 + rays_max) * SN], rgb[r0 * SN:(r0 + rays_max) * SN],
                           dirs[r0 * SN:(r0 + rays_max) * SN], min(rays_max, RN - r0), SN, debug, keep_workspace, precision)
                 for r0 in range(0, RN, rays_max)]
        dbg = {k: torch.cat([p[2][k] for p in parts]) for k in parts[0][2]}
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), dbg
    radiance = torch.empty(P, 3, dtype=torch.float32, device=dev)
    srdf = torch.empty(RN, SN, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ufr_aggregate_workspace_bytes(RN, SN, NV) // 4, dtype=torch.float32, device=dev)
    dbg = {}
    if debug:
        dbg = dict(view_out=torch.empty(P, NV + 1, _lib.TOKEN_DIM, device=dev), ray_out=torch.empty(P, _lib.RAY_DIM, device=dev))
    _lib.check(lib.ufr_aggregate(weights.packed.data_ptr(), _dev(x, "x_tokens"), _dev(rgb, "rgb"), _dev(dirs, "dir"),
                                 RN, SN, NV, radiance.data_ptr(), srdf.data_ptr(), ws.data_ptr(),
                                 _opt(dbg.get("view_out"), "view_out"), _opt(dbg.get("ray_out"), "ray_out"),
                                 weights.mode() if precision is None else precision, _stream()),
               "ufr_aggregate")
    if keep_workspace:   # the head of the workspace is the view transformer's token-0 output (P,80): the backward needs it
        dbg["token0"] = ws[: P * _lib.TOKEN_DIM].view(P, _lib.TOKEN_DIM)
    return radiance, srdf, dbg


def composite(z: torch.Tensor, radiance: torch.Tensor, srdf: torch.Tensor, variance: torch.Tensor,
              row: Optional[torch.Tensor] = None):
    """``row`` (RN,SN) int32: slot (ray, s) takes its colour from ``radiance.view(-1, 3)[row]`` (the sample pool)."""
    RN, SN = z.shape
    dev = z.device
    rgb = torch.empty(RN, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(RN, dtype=torch.float32, device=dev)
    opacity = torch.empty(RN, dtype=torch.float32, device=dev)
    weight = torch.empty(RN, SN, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().ufr_composite(_dev(z, "z"), _dev(radiance, "radiance"),
                                         None if row is None else _dev(row, "row", torch.int32), _dev(srdf, "srdf"),
                                         _dev(variance, "variance"), RN, SN, rgb.data_ptr(), depth.data_ptr(),
                                         opacity.data_ptr(), weight.data_ptr(), _stream()), "ufr_composite")
    return rgb, depth, opacity, weight


# ----------------------------------------------------------------------------- backward (training step)
class GradBuffer:
    """One flat fp32 buffer holding the gradients of the 40 per-ray parameters in RAW_WEIGHT_KEYS order (the kernels
    accumulate into it with atomics; one buffer = one all-reduce for data-parallel training)."""

    def __init__(self, device):
        sizes = [max(1, int(torch.Size(s).numel())) for s in RAW_WEIGHT_SHAPES]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 3) // 4 * 4)        # 16-byte aligned slots
        self.flat = torch.zeros(offs[-1], dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n].view(shape) for k, shape, o, n in zip(RAW_WEIGHT_KEYS, RAW_WEIGHT_SHAPES, offs, sizes)}
        self.raw = _lib.RawGrads()
        for i, o in enumerate(offs[:-1]):
            self.raw.p[i] = self.flat.data_ptr() + 4 * o

    def grad(self, key: str) -> torch.Tensor:
        return self.views[key]


def composite_bwd(z, radiance, srdf, variance, d_rgb, d_depth, d_opacity, d_weight, row=None, d_radiance=None,
                  accumulate: bool = False, d_variance=None):
    """``row`` / ``d_radiance`` / ``accumulate``: the pool form -- d_radiance rows are those ``row`` names inside the given
    pool-sized buffer, added to its content when ``accumulate``.  ``d_variance`` (a zeroed scalar) is accumulated."""
    RN, SN = z.shape
    dev = z.device
    if d_radiance is None:
        d_radiance = torch.empty(RN, SN, 3, dtype=torch.float32, device=dev)
    d_srdf = torch.empty(RN, SN, dtype=torch.float32, device=dev)
    if d_variance is None:
        d_variance = torch.zeros((), dtype=torch.float32, device=dev)
    keep = []

    def opt(t, n):
        if t is None:
            return None
        keep.append(t.contiguous())
        return _dev(keep[-1], n)

    _lib.check(_lib.load().ufr_composite_bwd(
        _dev(z, "z"), _dev(radiance, "radiance"), None if row is None else _dev(row, "row", torch.int32), _dev(srdf, "srdf"),
        _dev(variance, "variance"), RN, SN,
        opt(d_rgb, "d_rgb"), opt(d_depth, "d_depth"), opt(d_opacity, "d_opacity"), opt(d_weight, "d_weight"),
        _dev(d_radiance, "d_radiance"), int(accumulate), d_srdf.data_ptr(), _dev(d_variance, "d_variance"), _stream()),
        "ufr_composite_bwd")
    return d_radiance, d_srdf, d_variance


def render_loss(rgb, depth, rgb2, depth2, rgb_gt, depth_gt, near_fars, weight_rgb: float = 1.0, weight_depth: float = 1.0):
    """ufr_render_loss (code1/model.py:552-566): both passes' colours (B,RN,3) / ray depths (B,RN), the ground truth of the
    same shapes, ``near_fars`` = batch['near_fars'] (B,V,2).  -> ``loss (5,)`` = [total, rgb coarse, rgb fine, depth coarse,
    depth fine] and the cotangents ``d_rgb, d_depth, d_rgb2, d_depth2`` of the total, shaped as the inputs."""
    B, RN = depth_gt.shape
    dev = depth_gt.device
    c = lambda t: t.detach().float().contiguous()
    rgb, depth, rgb2, depth2, rgb_gt, depth_gt, nf = c(rgb), c(depth), c(rgb2), c(depth2), c(rgb_gt), c(depth_gt), c(near_fars)
    if tuple(rgb.shape) != (B, RN, 3) or tuple(rgb2.shape) != (B, RN, 3) or tuple(rgb_gt.shape) != (B, RN, 3) or \
            tuple(depth.shape) != (B, RN) or tuple(depth2.shape) != (B, RN) or nf.shape[0] != B or nf.shape[-1] != 2:
        raise UfrError(f"render_loss: shapes {tuple(rgb.shape)} {tuple(depth.shape)} {tuple(rgb2.shape)} {tuple(depth2.shape)} "
                            f"{tuple(rgb_gt.shape)} {tuple(depth_gt.shape)} {tuple(nf.shape)}")
    out = torch.empty(5 + 8 * B * RN, dtype=torch.float32, device=dev)       # one allocation: loss | d_rgb | d_depth | d_rgb2 | d_depth2
    n = B * RN
    loss, d_rgb, d_depth, d_rgb2, d_depth2 = out[:5], out[5:5 + 3 * n], out[5 + 3 * n:5 + 4 * n], out[5 + 4 * n:5 + 7 * n], out[5 + 7 * n:]
    _lib.check(_lib.load().ufr_render_loss(
        _dev(rgb, "rgb"), _dev(depth, "depth"), _dev(rgb2, "rgb2"), _dev(depth2, "depth2"), _dev(rgb_gt, "rgb_gt"),
        _dev(depth_gt, "depth_gt"), _dev(nf, "near_fars"), nf[0].numel(), B, RN, float(weight_rgb), float(weight_depth),
        loss.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_rgb2.data_ptr(), d_depth2.data_ptr(), _stream()), "ufr_render_loss")
    return loss, d_rgb.view(B, RN, 3), d_depth.view(B, RN), d_rgb2.view(B, RN, 3), d_depth2.view(B, RN)


def aggregate_bwd(weights: PackedWeights, grads: GradBuffer, x, rgb, dirs, token0, RN: int, SN: int, d_radiance, d_srdf,
                  precision: Optional[int] = None):
    """-> (d_pv (P,40), {}) -- the empty dict keeps the call sites of the former debug dump (gone with ABI 500)."""
    lib = _lib.load()
    NV, dev = x.shape[1], x.device
    P = RN * SN
    d_pv = torch.empty(P, 40, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ufr_aggregate_bwd_workspace_bytes(RN, SN, NV) // 4, dtype=torch.float32, device=dev)
    _lib.check(lib.ufr_aggregate_bwd(
        C.byref(weights.raw), C.byref(grads.raw), weights.packed.data_ptr(), _dev(x, "x_tokens"), _dev(rgb, "rgb"),
        _dev(dirs, "dir"), _dev(token0, "token0"), RN, SN, NV, _dev(d_radiance.contiguous(), "d_radiance"),
        _dev(d_srdf.contiguous(), "d_srdf"), d_pv.data_ptr(), ws.data_ptr(),
        weights.mode() if precision is None else precision, _stream()),
        "ufr_aggregate_bwd")
    return d_pv, {}


def project_gather_bwd_workspace_floats(frame: FrameHandle) -> int:
    return _lib.load().ufr_project_gather_bwd_workspace_bytes(C.byref(frame.frame)) // 4


def project_gather_bwd(frame: FrameHandle, weights: PackedWeights, grads: GradBuffer, ray_o, ray_d, z, sim8, d_pv,
                       grad_vol_feat, grad_vol_weight, precision: Optional[int] = None,
                       row: Optional[torch.Tensor] = None, accumulate: bool = True,
                       zeroed_workspace: Optional[torch.Tensor] = None, presim: bool = True) -> None:
    """Writes the frustum gradients grad_vol_feat[s] (NV,8,D,Hs,Ws) / grad_vol_weight[s] (NV,1,D,Hs,Ws) -- added to what the
    tensors hold (``accumulate``, the default: pass zeros) or overwriting them whole (``accumulate=False``: pass
    ``torch.empty``, no zero-fill needed) -- and accumulates the pre_sim_mlp gradients into `grads`.  ``row`` (RN,SN) int32:
    ``sim8`` / ``d_pv`` are pool tensors and slot (ray, s) owns ``d_pv[row[ray, s]]``.  ``zeroed_workspace``: the scatter's
    record volume (project_gather_bwd_workspace_floats), already zero-filled by the caller -- ordered before this call.
    ``presim=False``: the volume scatter only (UFR_GBWD_NO_PRESIM); the pre_sim_mlp half is the call with both lists None."""
    RN, SN = z.shape
    lib = _lib.load()
    stride = 0 if ray_o.numel() == 3 else 3
    gf = gw = ws = None        # both None: pre_sim_mlp gradients only
    if grad_vol_feat is not None:
        gf = (C.c_void_p * 3)(*[_dev(t, "grad_vol_feat") for t in grad_vol_feat])
        gw = (C.c_void_p * 3)(*[_dev(t, "grad_vol_weight") for t in grad_vol_weight])
        # channel-last record volume of the scatter (zeroed by the library unless the caller did)
        ws = zeroed_workspace if zeroed_workspace is not None else torch.empty(
            project_gather_bwd_workspace_floats(frame), dtype=torch.float32, device=z.device)
    _lib.check(lib.ufr_project_gather_bwd(
        C.byref(frame.frame), C.byref(weights.raw), C.byref(grads.raw), _dev(ray_o, "ray_o"), stride, _dev(ray_d, "ray_d"),
        _dev(z, "z"), RN, SN, _dev(sim8, "sim8"), _dev(d_pv, "d_pv"), None if row is None else _dev(row, "row", torch.int32),
        gf, gw, int(bool(accumulate)) | (2 if (zeroed_workspace is not None and ws is not None) else 0) | (0 if presim else 4),
        None if ws is None else ws.data_ptr(),
        weights.mode() if precision is None else precision, _stream()), "ufr_project_gather_bwd")


# ----------------------------------------------------------------------------- halves of aggregate / sample pool
def sample_importance_pool(weight: torch.Tensor, z: torch.Tensor, U2: torch.Tensor):
    """-> z_all (RN,SN+PN) sorted, z_new (RN,PN) sorted along the ray, row (RN,SN+PN) int32: pool row of every merged slot
    (pool = [RN*SN coarse rows | RN*PN new rows])."""
    RN, SN = z.shape
    PN = U2.shape[0]
    dev = z.device
    z_all = torch.empty(RN, SN + PN, dtype=torch.float32, device=dev)
    z_new = torch.empty(RN, PN, dtype=torch.float32, device=dev)
    row = torch.empty(RN, SN + PN, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().ufr_sample_importance_pool(_dev(weight, "weight"), _dev(z, "z"), _dev(U2, "U2"), z_all.data_ptr(),
                                                      z_new.data_ptr(), row.data_ptr(), RN, SN, PN, _stream()),
               "ufr_sample_importance_pool")
    return z_all, z_new, row


def view_transform(weights: PackedWeights, x: torch.Tensor, rgb: torch.Tensor, dirs: torch.Tensor,
                   token0: Optional[torch.Tensor] = None, radiance: Optional[torch.Tensor] = None,
                   precision: Optional[int] = None):
    """``token0`` (P,80) / ``radiance`` (P,3): optional destinations (row ranges of the two-pass step's sample pool)."""
    P, NV = x.shape[0], x.shape[1]
    if token0 is None:
        token0 = torch.empty(P, _lib.TOKEN_DIM, dtype=torch.float32, device=x.device)
    if radiance is None:
        radiance = torch.empty(P, 3, dtype=torch.float32, device=x.device)
    if token0.shape[0] != P or radiance.shape[0] != P:
        raise UfrError("view_transform: destination rows do not match the number of points")
    pmax = _max_points(NV)
    if P > pmax:                # beyond one launch's 32-bit addressing: points are independent, so call per point range
        for p0 in range(0, P, pmax):
            view_transform(weights, x[p0:p0 + pmax], rgb[p0:p0 + pmax], dirs[p0:p0 + pmax], token0[p0:p0 + pmax],
                           radiance[p0:p0 + pmax], precision)
        return token0, radiance
    _lib.check(_lib.load().ufr_view_transform(weights.packed.data_ptr(), _dev(x, "x_tokens"), _dev(rgb, "rgb"), _dev(dirs, "dir"),
                                              P, NV, _dev(token0, "token0"), _dev(radiance, "radiance"),
                                              weights.mode() if precision is None else precision, _stream()),
               "ufr_view_transform")
    return token0, radiance


def ray_transform(weights: PackedWeights, token0: torch.Tensor, RN: int, SN: int, row: Optional[torch.Tensor] = None,
                  precision: Optional[int] = None) -> torch.Tensor:
    """``row`` (RN,SN) int32: slot (ray, s) reads ``token0[row]`` (the sample pool); None = slot order."""
    lib = _lib.load()
    srdf = torch.empty(RN, SN, dtype=torch.float32, device=token0.device)
    ws = torch.empty(lib.ufr_ray_transform_workspace_bytes(SN) // 4, dtype=torch.float32, device=token0.device)
    _lib.check(lib.ufr_ray_transform(weights.packed.data_ptr(), _dev(token0, "token0"),
                                     None if row is None else _dev(row, "row", torch.int32), RN, SN, srdf.data_ptr(),
                                     ws.data_ptr(), weights.mode() if precision is None else precision, _stream()),
               "ufr_ray_transform")
    return srdf


def view_tape_block_points(NV: int) -> int:
    return int(_lib.load().ufr_view_tape_block_points(NV))


def view_transform_tape(weights: PackedWeights, x, rgb, dirs, token0, radiance, workspace: torch.Tensor, p0: int, P_total: int,
                        precision: Optional[int] = None) -> None:
    """Training forward of the view transformer for the pool rows [p0, p0 + P): writes ``token0`` / ``radiance`` (the
    range's rows) like view_transform and records the tape into ``workspace`` (view_transform_bwd_workspace(P_total, NV)),
    whose backward then runs with ``stages=STAGE_DGRAD | STAGE_WGRAD``."""
    P, NV = x.shape[0], x.shape[1]
    _lib.check(_lib.load().ufr_view_transform_tape(
        weights.packed.data_ptr(), _dev(x, "x_tokens"), _dev(rgb, "rgb"), _dev(dirs, "dir"), P, NV, _dev(token0, "token0"),
        _dev(radiance, "radiance"), workspace.data_ptr(), p0, P_total, weights.mode() if precision is None else precision,
        _stream()), "ufr_view_transform_tape")


def ray_transform_tape(weights: PackedWeights, token0: torch.Tensor, RN: int, SN: int, workspace: torch.Tensor,
                       row: Optional[torch.Tensor] = None, precision: Optional[int] = None) -> torch.Tensor:
    """Training forward of the ray transformer: srdf like ray_transform, the tape into ``workspace``
    (ray_transform_bwd_workspace(RN, SN)); its backward then runs without STAGE_TAPE."""
    srdf = torch.empty(RN, SN, dtype=torch.float32, device=token0.device)
    _lib.check(_lib.load().ufr_ray_transform_tape(
        weights.packed.data_ptr(), _dev(token0, "token0"), None if row is None else _dev(row, "row", torch.int32), RN, SN,
        srdf.data_ptr(), workspace.data_ptr(), weights.mode() if precision is None else precision, _stream()),
        "ufr_ray_transform_tape")
    return srdf


def ray_transform_bwd_workspace(RN: int, SN: int, device) -> torch.Tensor:
    return torch.empty(_lib.load().ufr_ray_transform_bwd_workspace_bytes(RN, SN) // 4, dtype=torch.float32, device=device)


def ray_transform_bwd(weights: PackedWeights, grads: GradBuffer, token0: torch.Tensor, RN: int, SN: int, d_srdf: torch.Tensor,
                      row: Optional[torch.Tensor] = None, out=None, accumulate: bool = False, precision: Optional[int] = None,
                      _workspace_out: Optional[list] = None, stages: int = 7, workspace: Optional[torch.Tensor] = None):
    """-> the two partial d token0 buffers of the ray kernel's sweeps (their sum is the gradient).  Plain form: (RN*SN,80)
    in slot order.  Pool form: ``row`` maps slots to rows of ``token0`` and of the two pool-sized buffers ``out=(a, b)``,
    which are overwritten or, with ``accumulate``, added to.  ``b`` may be None: nothing is zero-filled then (the caller owns
    the second buffer, e.g. lets another pass write it concurrently)."""
    lib = _lib.load()
    dev = token0.device
    if out is None:
        a = torch.empty(RN * SN, _lib.TOKEN_DIM, dtype=torch.float32, device=dev)
        b = torch.empty(RN * SN, _lib.TOKEN_DIM, dtype=torch.float32, device=dev)
    else:
        a, b = out
    ws = workspace if workspace is not None else ray_transform_bwd_workspace(RN, SN, dev)
    d_srdf = d_srdf.contiguous()
    # stages (include/ufr.h: ufr_ray_transform_bwd_stages): tape + data gradients (3) write ``out``; the weight gradients (4)
    # feed nothing downstream -- a caller may run them later / elsewhere on the same ``workspace``
    _lib.check(lib.ufr_ray_transform_bwd_stages(C.byref(weights.raw), C.byref(grads.raw), weights.packed.data_ptr(),
                                                _dev(token0, "token0"), None if row is None else _dev(row, "row", torch.int32),
                                                RN, SN, _dev(d_srdf, "d_srdf"), _dev(a, "d_token0_a"), _opt(b, "d_token0_b"),
                                                int(accumulate), ws.data_ptr(), stages,
                                                weights.mode() if precision is None else precision, _stream()),
               "ufr_ray_transform_bwd")
    if _workspace_out is not None:      # development: the tape / cotangent tiles (tools/dev/ray_bwd_check.py)
        _workspace_out.append(ws)
    return a, b


STAGE_TAPE, STAGE_DGRAD, STAGE_WGRAD, STAGE_ALL = 1, 2, 4, 7


def view_transform_bwd_workspace(P: int, NV: int, device) -> torch.Tensor:
    """The tile buffers of the view transformer's backward (csrc/bwd_tape.h): pass the same tensor to every stage."""
    return torch.empty(_lib.load().ufr_view_transform_bwd_workspace_bytes(P, NV) // 4, dtype=torch.float32, device=device)


def view_transform_bwd(weights: PackedWeights, grads: GradBuffer, x, rgb, dirs, d_token0_a, d_token0_b, d_radiance,
                       precision: Optional[int] = None, d_pv: Optional[torch.Tensor] = None, stages: int = STAGE_ALL,
                       workspace: Optional[torch.Tensor] = None):
    """``stages`` (include/ufr.h: ufr_view_transform_bwd_stages): STAGE_TAPE needs only x / rgb / dirs, STAGE_DGRAD the
    cotangents (writes d_pv), STAGE_WGRAD nothing but the workspace -- a caller that splits them passes ONE ``workspace``
    (view_transform_bwd_workspace) to all and orders them with stream events."""
    P, NV = x.shape[0], x.shape[1]
    if d_pv is None and (stages & STAGE_DGRAD):
        d_pv = torch.empty(P, 40, dtype=torch.float32, device=x.device)
    ta = None if d_token0_a is None else d_token0_a.contiguous()
    tb = None if d_token0_b is None else d_token0_b.contiguous()
    dr = None if d_radiance is None else d_radiance.contiguous()
    lib = _lib.load()
    ws = workspace if workspace is not None else view_transform_bwd_workspace(P, NV, x.device)
    _lib.check(lib.ufr_view_transform_bwd_stages(
        C.byref(weights.raw), C.byref(grads.raw), weights.packed.data_ptr(), _dev(x, "x_tokens"), _dev(rgb, "rgb"),
        _dev(dirs, "dir"), _opt(ta, "d_token0_a"), _opt(tb, "d_token0_b"), _opt(dr, "d_radiance"), P, NV, _opt(d_pv, "d_pv"),
        ws.data_ptr(), stages, weights.mode() if precision is None else precision, _stream()), "ufr_view_transform_bwd")
    return d_pv


class RenderWorkspace:
    """Reusable scratch of ufr_render_rays (sized for `chunk_rays`)."""

    def __init__(self, device, SN: int, PN: int, NV: int, chunk_rays: int = 0, n_streams: int = 3):
        lib = _lib.load()
        self.chunk = chunk_rays if chunk_rays > 0 else lib.ufr_default_chunk_rays()
        self.n_streams = max(1, int(n_streams))
        self.key = (SN, PN, NV, self.chunk)
        self.nbytes = lib.ufr_render_workspace_bytes(self.chunk, SN, PN, NV) * self.n_streams
        self.buf = torch.empty(self.nbytes // 4, dtype=torch.float32, device=device)


def render_rays(frame: FrameHandle, weights: PackedWeights, ray_idx: torch.Tensor, U1: torch.Tensor,
                U2: Optional[torch.Tensor], coarse_only: bool = False, workspace: Optional[RenderWorkspace] = None,
                want_srdf: bool = True, out: Optional[dict] = None):
    """UFORecon.infer(extract_geometry=True) for the rays `ray_idx` (RN,) of the prepared frame."""
    if frame.cam_ray_d is None or frame.ray_d is None or frame.ray_o is None:
        raise UfrError("batch['ray_o'], ['ray_d'] and ['cam_ray_d'] are required for extract_geometry rendering")
    dev = frame.device
    RN = ray_idx.numel()
    SN = U1.shape[0]
    PN = 0 if coarse_only else U2.shape[0]
    S = SN + PN
    if workspace is None or workspace.key[:3] != (SN, PN, frame.NV):
        workspace = RenderWorkspace(dev, SN, PN, frame.NV)
    o = out or {}
    depth = o.get("depth", torch.empty(RN, dtype=torch.float32, device=dev))
    depth_z = o.get("depth_z", torch.empty(RN, dtype=torch.float32, device=dev))
    rgb = o.get("rgb", torch.empty(RN, 3, dtype=torch.float32, device=dev))
    srdf = torch.empty(RN, S, dtype=torch.float32, device=dev) if want_srdf else None
    z_all = torch.empty(RN, S, dtype=torch.float32, device=dev) if want_srdf else None
    a = _lib.RenderArgs()
    a.frame = C.pointer(frame.frame)
    a.packed_weights = weights.packed.data_ptr()
    a.raw = C.pointer(weights.raw)
    a.ray_idx = _dev(ray_idx.reshape(-1), "ray_idx", torch.int64)
    a.ray_d, a.cam_ray_d = _dev(frame.ray_d, "ray_d"), _dev(frame.cam_ray_d, "cam_ray_d")
    a.ray_o = (C.c_float * 3)(*frame.ray_o)
    a.near_z, a.far_z = frame.near_z, frame.far_z
    a.U1 = _dev(U1, "U1")
    a.U2 = None if coarse_only else _dev(U2, "U2")
    if U1.shape[1] != RN or (not coarse_only and U2.shape[1] != RN):
        raise UfrError("U1/U2 must be (samples, RN)")
    a.RN, a.SN, a.PN, a.coarse_only = RN, SN, PN, int(coarse_only)
    a.precision = weights.mode()
    a.depth, a.depth_z, a.rgb = depth.data_ptr(), depth_z.data_ptr(), rgb.data_ptr()
    a.srdf = _opt(srdf, "srdf")
    a.z_all = _opt(z_all, "z_all")
    a.chunk_rays = workspace.chunk
    a.n_streams = workspace.n_streams
    a.workspace, a.workspace_bytes = workspace.buf.data_ptr(), workspace.nbytes
    _lib.check(_lib.load().ufr_render_rays(C.byref(a), _stream()), "ufr_render_rays")
    return dict(depth=depth, depth_z=depth_z, rgb=rgb, srdf=srdf, z_all=z_all, workspace=workspace)


def fmt_layer(params, x: torch.Tensor, src: Optional[torch.Tensor]) -> torch.Tensor:
    """One FMT encoder layer on the GPU (ufr_fmt_layer).  ``params``: the 16 tensors in ufr_fmt_layer_weights order;
    x (N,T,32); src (N,S,32) or None for self-attention."""
    lib = _lib.load()
    N, T, D = x.shape
    if D != 32:
        raise UfrError(f"fmt_layer: d_model {D}, the kernel is built for 32")
    w = _lib.FmtLayerWeights()
    keep = []
    for name, t in zip([f[0] for f in _lib.FmtLayerWeights._fields_], params):
        t = t.detach().contiguous()
        keep.append(t)
        setattr(w, name, _dev(t, "fmt." + name))
    x = x.contiguous()
    S = T
    if src is not None:
        src = src.contiguous()
        if src.shape[0] != N or src.shape[2] != D:
            raise UfrError(f"fmt_layer: source tokens {tuple(src.shape)} do not match {tuple(x.shape)}")
        S = src.shape[1]
    out = torch.empty_like(x)
    ws = torch.empty(lib.ufr_fmt_layer_workspace_bytes(N, S) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.ufr_fmt_layer(C.byref(w), _dev(x, "x"), _opt(src, "src"), N, T, S, out.data_ptr(), ws.data_ptr(),
                                 _stream()), "ufr_fmt_layer")
    return out


def set_matrix_precision(mode: int) -> None:
    """What PRECISION_DEFAULT resolves to (include/ufr.h): PRECISION_FP32 (the 1e-4 parity mode) or PRECISION_16BIT (one
    16-bit plane per operand: the mixed-precision training mode of BASELINE configs[4]).  A convenience for tools; a
    model pins its mode with ``UFORecon(args, precision=...)`` / ``PackedWeights(..., precision=...)``, and an autograd
    node always runs its backward in the mode its forward ran in."""
    _lib.check(_lib.load().ufr_set_matrix_precision(int(mode)), "ufr_set_matrix_precision")


def get_matrix_precision() -> int:
    return int(_lib.load().ufr_get_matrix_precision())


def profile_enable(on: bool) -> None:
    _lib.load().ufr_profile_enable(int(on))


def profile_read() -> dict:
    cap = 16
    names = (C.c_char_p * cap)()
    ms = (C.c_float * cap)()
    launches = (C.c_int32 * cap)()
    n = _lib.load().ufr_profile_read(names, ms, launches, cap)
    return {names[i].decode(): dict(ms=float(ms[i]), launches=int(launches[i])) for i in range(n)}


CONV3D_S1, CONV3D_S2, CONV3D_T2 = 0, 1, 2


def conv3d(x_cl: torch.Tensor, weight: torch.Tensor, mode: int = CONV3D_S1, bias: Optional[torch.Tensor] = None,
           bn_scale: Optional[torch.Tensor] = None, bn_shift: Optional[torch.Tensor] = None, relu: bool = False,
           skip: Optional[torch.Tensor] = None, out_ncdhw: bool = False, weight2: Optional[torch.Tensor] = None,
           want_absmax: bool = False):
    """One 3x3x3 layer of the frustum U-Nets (ufr_conv3d).  ``want_absmax`` (channel-last outputs of at most 16 channels):
    returns ``(out, max |out| as a one-element device tensor)`` -- the bound a following plane layer wants.  ``x_cl`` (B,D,H,W,cin) channel-last; ``weight`` in the
    checkpoint's layout (conv (cout,cin,3,3,3); transposed conv, mode CONV3D_T2, (cin,cout,3,3,3)).  Returns the
    channel-last (B,Do,Ho,Wo,cout) output, or with ``out_ncdhw`` the reference's (B,cout,Do,Ho,Wo) -- and, when ``weight2``
    names a second head, the pair (out, sigmoid(second head))."""
    B, D, H, W, cin = x_cl.shape
    transposed = mode == CONV3D_T2
    cout = weight.shape[1] if transposed else weight.shape[0]
    if tuple(weight.shape) != ((cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)):
        raise UfrError(f"conv3d: weight {tuple(weight.shape)} does not match {cin} input channels (mode {mode})")
    Do, Ho, Wo = ((2 * D, 2 * H, 2 * W) if transposed else ((D + 1) // 2, (H + 1) // 2, (W + 1) // 2) if mode == CONV3D_S2
                  else (D, H, W))
    dev = x_cl.device
    cout2 = 0 if weight2 is None else weight2.shape[0]
    out = torch.empty((B, cout, Do, Ho, Wo) if out_ncdhw else (B, Do, Ho, Wo, cout), dtype=torch.float32, device=dev)
    out2 = torch.empty((B, cout2, Do, Ho, Wo), dtype=torch.float32, device=dev) if cout2 else None
    if skip is not None and tuple(skip.shape) != tuple(out.shape):
        raise UfrError(f"conv3d: skip {tuple(skip.shape)} does not match the output {tuple(out.shape)}")
    keep = [t.detach().contiguous() for t in (weight, weight2, bias, bn_scale, bn_shift) if t is not None]
    it = iter(keep)
    ptr = lambda t, n: None if t is None else _dev(next(it), n)
    w_p, w2_p, b_p, s_p, h_p = ptr(weight, "weight"), ptr(weight2, "weight2"), ptr(bias, "bias"), ptr(bn_scale, "bn_scale"), \
        ptr(bn_shift, "bn_shift")
    # (want_absmax: True = a fresh zero word; or a caller-owned one-element ZERO tensor, e.g. a slice of a pool zeroed once)
    omax = want_absmax if isinstance(want_absmax, torch.Tensor) else (torch.zeros(1, dtype=torch.float32, device=dev) if want_absmax else None)
    _lib.check(_lib.load().ufr_conv3d(_dev(x_cl, "x"), w_p, w2_p, b_p, s_p, h_p, _opt(skip, "skip"), out.data_ptr(),
                                      _opt(out2, "out2"), B, D, H, W, cin, cout, cout2, int(mode), int(bool(relu)),
                                      int(bool(out_ncdhw)), _opt(omax, "out_absmax"), _stream()), "ufr_conv3d")
    if omax is not None:
        return out, omax
    return (out, out2) if cout2 else out


def absmax(x: torch.Tensor, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max |x| as a one-element device tensor (ufr_absmax: one pass at HBM speed, no host round trip); ``into``: a running
    maximum to raise instead of a fresh zero."""
    x = x.detach()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    out = torch.zeros(1, dtype=torch.float32, device=x.device) if into is None else into
    _lib.check(_lib.load().ufr_absmax(_dev(x, "x"), x.numel(), out.data_ptr(), _stream()), "ufr_absmax")
    return out


# the weights' planes of ufr_conv3d_planes, kept while the weight TENSOR OBJECT lives unchanged: id(weight) -> (weak reference to
# it, its version counter, the same of weight2, flip, the planes).  Never keyed on an address: the allocator hands a freed
# tensor's address -- version 0 again -- to the next one.  An in-place update (optimizer.step) bumps the version, and the
# next call makes the planes again (training: once per layer and step; inference: once per checkpoint).
_PLANES = {}


def _planes_lookup(weight, weight2, flip, nbytes, transposed=False):
    import weakref

    ent = _PLANES.get(id(weight))
    sig = (weight._version, None if weight2 is None else (id(weight2), weight2._version), bool(flip), nbytes, weight.device, transposed)
    if ent is not None and ent[0]() is weight and ent[1] == sig and (weight2 is None or ent[2]() is weight2):
        return ent[3], True
    if len(_PLANES) > 512:          # tensors that died without being looked up again
        for k in [k for k, e in _PLANES.items() if e[0]() is None]:
            del _PLANES[k]
    ws = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _PLANES[id(weight)] = (weakref.ref(weight), sig, None if weight2 is None else weakref.ref(weight2), ws)
    return ws, False


def conv3d_planes_supported(cin: int, cout: int, cout2: int = 0, mode: int = 0) -> bool:
    return _lib.load().ufr_conv3d_planes_workspace_bytes(cin, cout, cout2, int(mode)) > 0


def conv3d_planes(x_cl: torch.Tensor, x_absmax: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  bn_scale: Optional[torch.Tensor] = None, bn_shift: Optional[torch.Tensor] = None, relu: bool = False,
                  skip: Optional[torch.Tensor] = None, out_ncdhw: bool = False, weight2: Optional[torch.Tensor] = None,
                  flip: bool = False, want_absmax: bool = True, mode: int = 0):
    """A stride-1 3x3x3 layer with 8 or 16 input channels on the 16-bit matrix cores (ufr_conv3d_planes: fp16 plane
    products, fp32 accumulate, the input brick staged through LDS).  ``x_absmax``: one-element device tensor >= max |x_cl|
    (`absmax`, or the previous layer's returned bound).  ``flip``: the data gradient of a stride-1 layer -- ``weight`` is
    that layer's forward weight (cin of this call = its cout).  Returns ``(out, out_absmax or None)``, with a second head
    ``(out, sigmoid(out2), None)``."""
    B, D, H, W, cin = x_cl.shape
    if flip or int(mode) == CONV3D_T2:
        if tuple(weight.shape[2:]) != (3, 3, 3) or weight.shape[0] != cin:
            raise UfrError(f"conv3d_planes(flip / transposed): weight {tuple(weight.shape)} does not match {cin} input channels")
        cout = weight.shape[1]
    else:
        if tuple(weight.shape[1:]) != (cin, 3, 3, 3):
            raise UfrError(f"conv3d_planes: weight {tuple(weight.shape)} does not match {cin} input channels")
        cout = weight.shape[0]
    cout2 = 0 if weight2 is None else weight2.shape[0]
    lib = _lib.load()
    mode = int(mode)
    nbytes = lib.ufr_conv3d_planes_workspace_bytes(cin, cout, cout2, mode)
    if not nbytes:
        raise UfrError(f"conv3d_planes: (cin {cin}, cout {cout}+{cout2}, mode {mode}) is not a layer of this kernel family")
    dev = x_cl.device
    Do, Ho, Wo = ((D + 1) // 2, (H + 1) // 2, (W + 1) // 2) if mode == CONV3D_S2 else (2 * D, 2 * H, 2 * W) if mode == CONV3D_T2 else (D, H, W)
    out = torch.empty((B, cout, Do, Ho, Wo) if out_ncdhw else (B, Do, Ho, Wo, cout), dtype=torch.float32, device=dev)
    out2 = torch.empty((B, cout2, Do, Ho, Wo), dtype=torch.float32, device=dev) if cout2 else None
    if skip is not None and tuple(skip.shape) != tuple(out.shape):
        raise UfrError(f"conv3d_planes: skip {tuple(skip.shape)} does not match the output {tuple(out.shape)}")
    ws, ready = _planes_lookup(weight, weight2, flip, nbytes, mode == CONV3D_T2)
    omax = None
    if not out_ncdhw:
        omax = want_absmax if isinstance(want_absmax, torch.Tensor) else (torch.zeros(1, dtype=torch.float32, device=dev) if want_absmax else None)
    keep = [t.detach().contiguous() for t in (weight, weight2, bias, bn_scale, bn_shift) if t is not None]
    it = iter(keep)
    ptr = lambda t, n: None if t is None else _dev(next(it), n)
    w_p, w2_p, b_p, s_p, h_p = ptr(weight, "weight"), ptr(weight2, "weight2"), ptr(bias, "bias"), ptr(bn_scale, "bn_scale"), \
        ptr(bn_shift, "bn_shift")
    _lib.check(lib.ufr_conv3d_planes(_dev(x_cl, "x"), _dev(x_absmax, "x_absmax"), w_p, w2_p, b_p, s_p, h_p, _opt(skip, "skip"),
                                     out.data_ptr(), _opt(out2, "out2"), _opt(omax, "out_absmax"), B, D, H, W, cin, cout, cout2,
                                     mode, int(bool(relu)), int(bool(out_ncdhw)), int(bool(flip)), ws.data_ptr(), nbytes, int(ready),
                                     _stream()),
               "ufr_conv3d_planes")
    return (out, out2, None) if cout2 else (out, omax)


CONV2D_RELU, CONV2D_IN_PLANAR, CONV2D_OUT_PLANAR = 1, 2, 4


def conv2d(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, scale: Optional[torch.Tensor] = None,
           shift: Optional[torch.Tensor] = None, relu: bool = False, skip: Optional[torch.Tensor] = None,
           in_planar: bool = False, out_planar: bool = False, sigmoid_from: int = -1):
    """One plain convolution of FeatureNet (ufr_conv2d): ``x`` channel-last (B,H,W,cin) -- or, with ``in_planar``, the image
    (B,3,H,W); ``weight`` (cout,cin,k,k), zero padding k // 2; ``out = [relu](conv * scale + shift) [+ up2(skip)]``; returns
    channel-last (B,Ho,Wo,cout) or with ``out_planar`` (B,cout,Ho,Wo); ``sigmoid_from``: sigmoid on channels >= it."""
    if in_planar:
        B, cin, H, W = x.shape
    else:
        B, H, W, cin = x.shape
    cout, cin_w, k, k2 = weight.shape
    if cin_w != cin or k != k2:
        raise UfrError(f"conv2d: weight {tuple(weight.shape)} does not match {cin} input channels")
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty((B, cout, Ho, Wo) if out_planar else (B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    if skip is not None and tuple(skip.shape) != (B, Ho // 2, Wo // 2, cout):
        raise UfrError(f"conv2d: skip {tuple(skip.shape)} is not the half-resolution map {(B, Ho // 2, Wo // 2, cout)}")
    flags = (CONV2D_RELU if relu else 0) | (CONV2D_IN_PLANAR if in_planar else 0) | (CONV2D_OUT_PLANAR if out_planar else 0)
    _lib.check(_lib.load().ufr_conv2d(_dev(x, "x"), _dev(weight, "weight"), _opt(scale, "scale"), _opt(shift, "shift"),
                                      _opt(skip, "skip"), out.data_ptr(), B, cin, cout, H, W, k, int(stride), flags,
                                      int(sigmoid_from), _stream()), "ufr_conv2d")
    return out


def upsample_add(reduced_cl: torch.Tensor, fine: torch.Tensor) -> torch.Tensor:
    """``bilinear_2x(reduced_cl (B,h,w,C)) + fine (B,C,2h,2w)`` -> channel-last (B,2h,2w,C) (ufr_upsample_add: the sum the
    smoothing convolutions of the FMT pathway read)."""
    B, h, w, Cc = reduced_cl.shape
    if tuple(fine.shape) != (B, Cc, 2 * h, 2 * w):
        raise UfrError(f"upsample_add: fine {tuple(fine.shape)} is not (B,C,2h,2w) of reduced {tuple(reduced_cl.shape)}")
    out = torch.empty(B, 2 * h, 2 * w, Cc, dtype=torch.float32, device=fine.device)
    _lib.check(_lib.load().ufr_upsample_add(_dev(reduced_cl, "reduced"), _dev(fine, "fine"), out.data_ptr(), B, Cc, h, w, _stream()),
               "ufr_upsample_add")
    return out


def deform_conv2d_cl(x_cl: torch.Tensor, offset_mask: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, relu: bool = False,
                     out_planar: bool = False):
    """The deformable 3x3 layer on a channel-last input (ufr_deform_conv2d_cl): ``x_cl`` (B,H,W,32); ``offset_mask``
    (B,27,H,W) planar = DCN.conv_offset_mask's output with the sigmoid already on channels 18..26 (``conv2d(...,
    out_planar=True, sigmoid_from=18)``); ``out = [relu]((dcn + bias) * scale + shift)``, channel-last or planar."""
    B, H, W, C = x_cl.shape
    Cout = weight.shape[0]
    if tuple(offset_mask.shape) != (B, 27, H, W) or tuple(weight.shape) != (Cout, C, 3, 3):
        raise UfrError(f"deform_conv2d_cl: shapes {tuple(x_cl.shape)} {tuple(offset_mask.shape)} {tuple(weight.shape)}")
    out = torch.empty((B, Cout, H, W) if out_planar else (B, H, W, Cout), dtype=torch.float32, device=x_cl.device)
    flags = (CONV2D_RELU if relu else 0) | (CONV2D_OUT_PLANAR if out_planar else 0)
    _lib.check(_lib.load().ufr_deform_conv2d_cl(_dev(x_cl, "x"), _dev(offset_mask, "offset_mask"), _dev(weight, "weight"),
                                                _opt(bias, "bias"), _opt(scale, "scale"), _opt(shift, "shift"), out.data_ptr(),
                                                B, C, Cout, H, W, flags, _stream()), "ufr_deform_conv2d_cl")
    return out


def conv3d_bwd_data(d_out_cl: torch.Tensor, weight: torch.Tensor, mode: int, in_shape, accumulate: Optional[torch.Tensor] = None):
    """Data gradient of one plain 3x3x3 layer (ufr_conv3d_bwd_data): ``d_out_cl`` channel-last at the layer's output extent,
    ``weight`` the layer's forward weight in the checkpoint's layout, ``in_shape`` = (B,D,H,W,cin) of the layer's input.
    ``accumulate`` (same shape): added -- a gradient arriving on two paths (the U-Net's skip additions)."""
    B, D, H, W, cin = in_shape
    cout = d_out_cl.shape[-1]
    d_in = torch.empty(in_shape, dtype=torch.float32, device=d_out_cl.device)
    w = weight.detach().contiguous()
    _lib.check(_lib.load().ufr_conv3d_bwd_data(_dev(d_out_cl, "d_out"), _dev(w, "weight"), _opt(accumulate, "accumulate"),
                                               d_in.data_ptr(), B, D, H, W, cin, cout, int(mode), _stream()), "ufr_conv3d_bwd_data")
    return d_in


def conv3d_bwd_weight_heads(x_cl: torch.Tensor, d_out_cl: torch.Tensor, d_out2_cl: torch.Tensor, out=None):
    """Weight gradients of CostRegNetWeight's two heads in one pass over the shared input (ufr_conv3d_bwd_weight_heads):
    ``x_cl`` (B,D,H,W,8), ``d_out_cl`` (B,D,H,W,8), ``d_out2_cl`` (B,D,H,W,1) -> (d features.weight (8,8,3,3,3),
    d weights.weight (1,8,3,3,3))."""
    B, D, H, W, cin = x_cl.shape
    if cin != 8 or tuple(d_out_cl.shape) != (B, D, H, W, 8) or tuple(d_out2_cl.shape) != (B, D, H, W, 1):
        raise UfrError(f"conv3d_bwd_weight_heads: shapes {tuple(x_cl.shape)}, {tuple(d_out_cl.shape)}, {tuple(d_out2_cl.shape)}")
    # out = (dw, dw2): caller-owned ZEROED tensors of those shapes (slices of one buffer zeroed once per backward)
    dw, dw2 = out if out is not None else (torch.zeros(8, 8, 3, 3, 3, dtype=torch.float32, device=x_cl.device),
                                           torch.zeros(1, 8, 3, 3, 3, dtype=torch.float32, device=x_cl.device))
    _lib.check(_lib.load().ufr_conv3d_bwd_weight_heads(_dev(x_cl, "in"), _dev(d_out_cl, "d_out"), _dev(d_out2_cl, "d_out2"), dw.data_ptr(),
                                                       dw2.data_ptr(), B, D, H, W, _stream()), "ufr_conv3d_bwd_weight_heads")
    return dw, dw2


def conv3d_bwd_weight(x_cl: torch.Tensor, d_out_cl: torch.Tensor, mode: int, weight_shape, want_bias: bool = True, out=None):
    """Weight (and bias) gradient of one plain 3x3x3 layer (ufr_conv3d_bwd_weight) -> (d_weight in the checkpoint's
    layout, d_bias or None)."""
    B, D, H, W, cin = x_cl.shape
    cout = d_out_cl.shape[-1]
    if out is not None:      # (dw, db): caller-owned ZEROED tensors (slices of one buffer zeroed once per backward)
        dw, db = out
    else:
        dw = torch.zeros(weight_shape, dtype=torch.float32, device=x_cl.device)
        db = torch.zeros(cout, dtype=torch.float32, device=x_cl.device) if want_bias else None
    _lib.check(_lib.load().ufr_conv3d_bwd_weight(_dev(x_cl, "in"), _dev(d_out_cl, "d_out"), dw.data_ptr(), _opt(db, "d_bias"),
                                                 B, D, H, W, cin, cout, int(mode), _stream()), "ufr_conv3d_bwd_weight")
    return dw, db

