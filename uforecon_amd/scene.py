"""Synthetic DTU-like frames for the per-ray path (no dataset / checkpoint needed).

Builds the `batch` dict and the per-frame encoder outputs that the reference
hands to ``UFORecon.infer`` (/root/reference/code1/model.py:393), with camera
conventions derived the way the reference test dataset derives them
(/root/reference/code1/dataset/dtu_test_sparse.py:331-336, 405-429):

* ``source_poses = normalize @ K_pad @ w2c`` so projected x,y land in [-1,1],
* ``ray_d`` from ``ref_pose_inv @ homo_pixel`` on a ``linspace(-1,1)`` pixel grid,
* ``cam_ray_d`` from view-0 intrinsics,
* ``near_fars = [0.95 (dist-1), 1.05 (dist+1)]`` (unit-sphere scene),
* render pose = source pose 0 shifted along its camera-x axis
  (dtu_test_sparse.py:269-272).

Everything is produced on the CPU from seeded generators so the oracle, the
golden fixtures and the HIP path all see bit-identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

STAGES = ("stage1", "stage2", "stage3")
# (depth planes, spatial scale) of the three correlation frustums
# (/root/reference/main.py:81 ndepths="48,32,8"; model.py:796-802).
STAGE_SHAPE = {"stage1": (48, 4), "stage2": (32, 2), "stage3": (8, 1)}


def _look_at_w2c(eye: np.ndarray, target: np.ndarray) -> np.ndarray:
    """OpenCV-style world->camera (x right, y down, z forward)."""
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)  # rows = camera axes in world
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return w2c


def _unit_uniform(shape, g):
    """Zero-mean unit-variance uniform noise.  ``torch.rand`` on the CPU generator is
    exact integer->float arithmetic, so (unlike ``randn``, whose vectorised Box-Muller
    goes through libm) the values are identical on every host."""
    return (torch.rand(*shape, generator=g) - 0.5) * math.sqrt(12.0)


def frame_digest(frame: "Frame") -> float:
    """Order-independent fingerprint of a frame's tensors (float64 sums), stored in the
    golden fixtures so a test can tell 'inputs differ on this host' from 'outputs wrong'."""
    acc = 0.0
    ts = [frame.source_imgs_feat, frame.match_feature[0], frame.batch["source_imgs"],
          frame.batch["depth_info"], frame.batch["source_poses"], frame.batch["ray_d"]]
    for st in STAGES:
        ts += [frame.feature_volume[st]["feature_volume"], frame.feature_volume[st]["weight_volume"]]
    for i, t in enumerate(ts):
        acc += (i + 1) * float(t.double().abs().sum())
    return acc


@dataclass
class Frame:
    """One frame worth of inputs of the per-ray path."""

    batch: dict
    source_imgs_feat: torch.Tensor  # (1,NV,32,H/4,W/4)
    feature_volume: dict  # stage -> {'feature_volume': (NV,8,D,h,w), 'weight_volume': (NV,1,D,h,w)}
    match_feature: list  # [ (1,NV,32*(NV-1),H/4,W/4) ]
    H: int
    W: int
    NV: int

    def to(self, device) -> "Frame":
        def mv(x):
            if torch.is_tensor(x):
                return x.to(device)
            if isinstance(x, dict):
                return {k: mv(v) for k, v in x.items()}
            if isinstance(x, list):
                return [mv(v) for v in x]
            return x

        return Frame(mv(self.batch), mv(self.source_imgs_feat), mv(self.feature_volume),
                     mv(self.match_feature), self.H, self.W, self.NV)


def make_cameras(H: int, W: int, NV: int, offset_dist: float = 0.08, train_layout: bool = False):
    """Cameras on an arc of radius ~3 around the origin, looking at it."""
    w2cs, Ks, nfs = [], [], []
    for i in range(NV):
        a = (i - (NV - 1) / 2.0) * 0.35
        eye = np.array([3.0 * math.sin(a), 0.3 * i, -3.0 * math.cos(a)])
        w2c = _look_at_w2c(eye, np.zeros(3))
        K = np.array([[0.9 * W, 0.0, (W - 1) / 2.0], [0.0, 0.9 * W, (H - 1) / 2.0], [0.0, 0.0, 1.0]])
        dist = float(np.linalg.norm(eye))
        w2cs.append(w2c)
        Ks.append(K)
        nfs.append([0.95 * (dist - 1.0), 1.05 * (dist + 1.0)])
    w2cs = np.stack(w2cs)
    Ks = np.stack(Ks)
    nfs = np.array(nfs)

    # render view: source 0 shifted along its own camera x axis
    c2w0 = np.linalg.inv(w2cs[0])
    render_c2w = c2w0.copy()
    render_c2w[:3, 3] += render_c2w[:3, 0] * offset_dist
    render_w2c = np.linalg.inv(render_c2w)

    K_pad = np.tile(np.eye(4), (NV, 1, 1))
    K_pad[:, :3, :3] = Ks
    normalize = np.array([[2.0 / (W - 1), 0, -1, 0], [0, 2.0 / (H - 1), -1, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    source_poses = normalize[None] @ K_pad @ w2cs
    ref_pose = normalize @ K_pad[0] @ render_w2c
    ref_pose_inv = np.linalg.inv(ref_pose)
    source_poses_inv = np.linalg.inv(source_poses)

    h_line = np.linspace(0, H - 1, H) * 2 / (H - 1) - 1
    w_line = np.linspace(0, W - 1, W) * 2 / (W - 1) - 1
    hm, wm = np.meshgrid(h_line, w_line, indexing="ij")
    homo = np.stack([wm.reshape(-1), hm.reshape(-1), np.ones(H * W), np.ones(H * W)])
    ray_o = ref_pose_inv[:3, -1]
    rd = (ref_pose_inv @ homo)[:3] - ray_o[:, None]
    rd = rd / np.linalg.norm(rd, axis=0)
    crd = (np.linalg.inv(normalize @ K_pad[0]) @ homo)[:3]
    crd = crd / np.linalg.norm(crd, axis=0)

    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    cams = dict(
        w2cs=f32(w2cs)[None], intrinsics=f32(Ks)[None], near_fars=f32(nfs)[None],
        source_poses=f32(source_poses)[None], source_poses_inv=f32(source_poses_inv)[None],
        ref_pose_inv=f32(ref_pose_inv)[None], ray_o=f32(ray_o)[None],
        ray_d=f32(rd)[None], cam_ray_d=f32(crd)[None],
        scale_mat=torch.eye(4)[None] * 1.0, scale_factor=torch.tensor([1.0]),
    )
    if train_layout:
        # training batches carry the GT reference view at index 0 of w2cs /
        # intrinsics / near_fars and have no 'start_idx' (model.py:313 -> s_idx=1)
        cams["w2cs"] = torch.cat([f32(render_w2c)[None, None], cams["w2cs"]], 1)
        cams["intrinsics"] = torch.cat([cams["intrinsics"][:, :1], cams["intrinsics"]], 1)
        cams["near_fars"] = torch.cat([cams["near_fars"][:, :1], cams["near_fars"]], 1)
    return cams


def make_frame(H: int = 64, W: int = 96, NV: int = 3, seed: int = 0, offset_dist: float = 0.08,
               train_layout: bool = False) -> Frame:
    """Seeded synthetic frame."""
    assert H % 4 == 0 and W % 4 == 0
    g = torch.Generator().manual_seed(seed)
    batch = make_cameras(H, W, NV, offset_dist, train_layout)
    h, w = H // 4, W // 4
    batch["source_imgs"] = torch.rand(1, NV, 3, H, W, generator=g)
    batch["depth_info"] = 2.0 + 2.0 * torch.rand(1, NV, H, W, generator=g)
    if train_layout:
        batch["ref_img"] = torch.rand(1, 3, H, W, generator=g)
        batch["depths_h"] = 2.0 + 2.0 * torch.rand(1, NV + 1, H, W, generator=g)
    else:
        batch["start_idx"] = 0
    feat = _unit_uniform((1, NV, 32, h, w), g)
    match = [_unit_uniform((1, NV, 32 * (NV - 1), h, w), g)]
    vol = {}
    for st in STAGES:
        D, s = STAGE_SHAPE[st]
        vol[st] = {
            "feature_volume": _unit_uniform((NV, 8, D, H // s, W // s), g),
            "weight_volume": torch.rand(NV, 1, D, H // s, W // s, generator=g),
        }
    return Frame(batch, feat, vol, match, H, W, NV)


def make_cost_volumes(H: int, W: int, NV: int, seed: int) -> dict:
    """Seeded 1-channel cost volumes of the three cascade stages, (NV,1,D,H/s,W/s): what `MVSVolume` (feature_volume.py:
    100-121, i.e. `cost_reg_2`) turns into the feature / weight frustums -- the input of the one producer the reference
    trains.  Used by the gradient fixture of `feature_volume.cost_reg_2.*` and its test."""
    g = torch.Generator().manual_seed(90_000 + seed)
    return {st: _unit_uniform((NV, 1, STAGE_SHAPE[st][0], H // STAGE_SHAPE[st][1], W // STAGE_SHAPE[st][1]), g) for st in STAGES}


def sampler_uniforms(seed: int, point_num: int, point_num_2: int, RN: int):
    """The two uniform draws of one ``infer`` call, in the reference's order and shapes.

    FixedSampler draws ``torch.rand((SN, RN))`` (sampler.py:42) and then
    ImportanceSampler draws ``torch.rand(point_num, RN).T`` (sampler.py:86), both on the
    CPU default generator.  Drawing them here from ``torch.manual_seed(seed)`` in that
    order reproduces what the reference consumes after the same ``manual_seed``.
    """
    st = torch.random.get_rng_state()
    torch.manual_seed(seed)
    u1 = torch.rand(point_num, RN)
    u2 = torch.rand(point_num_2, RN)
    torch.random.set_rng_state(st)
    return u1, u2


# ------------------------------------------------------------------ correlation-volume step
# name -> feature channels, feature-map size, depth hypotheses, views (reference + sources), seed.
# The stage shapes follow the reference's cascade (TransMVSNet.py:125: ndepths 48/32/8 at 1/4, 1/2, 1/1 resolution with
# 32/16/8 feature channels); "edge" puts the hypotheses so close to the cameras that many samples leave the source
# images or fall behind them (the -99 path of module.py:355-360).
CORRELATE_CASES = {
    "stage1_small": dict(C=32, H=32, W=40, D=48, NV=3, seed=11),
    "stage3_small": dict(C=8, H=64, W=80, D=8, NV=3, seed=12),
    "nv5_stage2": dict(C=16, H=24, W=36, D=32, NV=5, seed=13),
    "edge": dict(C=8, H=20, W=28, D=16, NV=3, seed=14, edge=True),
}


def make_correlate_case(name: str, **over):
    """Seeded inputs of DepthNet.forward step 2 for one frame: features (C,H,W) of the reference and NS source views,
    projection pairs (2,4,4) = [extrinsic, intrinsic] as in batch['proj_matrices'] stage entries, per-pixel depth
    hypotheses (D,H,W) and pixel-wise view weights (NS,H,W)."""
    c = dict(CORRELATE_CASES[name]) if name in CORRELATE_CASES else {}
    c.update(over)
    C, H, W, D, NV = c["C"], c["H"], c["W"], c["D"], c["NV"]
    g = torch.Generator().manual_seed(c["seed"])
    cams = make_cameras(H, W, NV)
    pairs = []
    for v in range(NV):
        pp = torch.zeros(2, 4, 4)
        pp[0] = cams["w2cs"][0, v]
        pp[1, :3, :3] = cams["intrinsics"][0, v]
        pp[1, 3, 3] = 1.0
        pairs.append(pp)
    near, far = float(cams["near_fars"][0, 0, 0]), float(cams["near_fars"][0, 0, 1])
    if c.get("edge"):
        near, far = -0.4, 2.4   # from behind the reference camera to just short of the working volume
    lin = torch.linspace(0.0, 1.0, D).reshape(D, 1, 1)
    jitter = (torch.rand(D, H, W, generator=g) - 0.5) * (0.5 / D)
    depth_values = (near + (far - near) * (lin + jitter)).contiguous()
    feats = [_unit_uniform((C, H, W), g) for _ in range(NV)]
    vw = torch.rand(NV - 1, H, W, generator=g)
    return dict(name=name, ref_fea=feats[0], src_feas=feats[1:], ref_proj_pair=pairs[0], src_proj_pairs=pairs[1:],
                depth_values=depth_values, view_weights=vw, C=C, H=H, W=W, D=D, NS=NV - 1)


def correlate_digest(c) -> float:
    ts = [c["ref_fea"], *c["src_feas"], c["ref_proj_pair"], *c["src_proj_pairs"], c["depth_values"], c["view_weights"]]
    return sum((i + 1) * float(t.double().abs().sum()) for i, t in enumerate(ts))


# ------------------------------------------------------------------ TSDF fusion
TSDF_CASES = {
    "sphere3": dict(H=48, W=64, NV=3, voxel_size=0.08, margin=3, radius=0.8, seed=31),
    "sphere5_holes": dict(H=40, W=56, NV=5, voxel_size=0.1, margin=5, radius=0.7, seed=32, holes=True),
}


def make_tsdf_case(name: str):
    """Depth maps (z-depth, float32, 0 = no measurement) of a sphere at the origin seen from the arc cameras of
    `make_cameras`, 8-bit-valued colour images as float32 in [0,1] (what tsdf_fusion.read_img yields), intrinsics and
    camera-to-world poses -- the inputs of the reference's save_tsdf loop (tsdf_fusion.py:459-499)."""
    c = dict(TSDF_CASES[name])
    H, W, NV, R = c["H"], c["W"], c["NV"], c["radius"]
    g = torch.Generator().manual_seed(c["seed"])
    cams = make_cameras(H, W, NV)
    depths, colors, intrs, poses = [], [], [], []
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    for v in range(NV):
        K = cams["intrinsics"][0, v].double().numpy()
        w2c = cams["w2cs"][0, v].double().numpy()
        c2w = np.linalg.inv(w2c)
        d_cam = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)   # z = 1 rays
        d_w = d_cam @ c2w[:3, :3].T
        o = c2w[:3, 3]
        a = (d_w * d_w).sum(-1)
        b = 2.0 * (d_w @ o)
        cc = float(o @ o) - R * R
        disc = b * b - 4 * a * cc
        t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)   # z-depth of the first hit
        depth = np.where(disc > 0, t, 0.0).astype(np.float32)
        if c.get("holes"):
            mask = torch.rand(H, W, generator=g).numpy() < 0.15
            depth[mask] = 0.0
        col = torch.randint(0, 256, (H, W, 3), generator=g).float().numpy() / 255.0
        depths.append(depth)
        colors.append(col.astype(np.float32))
        intrs.append(K.astype(np.float32))
        poses.append(c2w.astype(np.float32))
    return dict(name=name, depths=depths, colors=colors, intrinsics=intrs, poses=poses, voxel_size=c["voxel_size"],
                margin=c["margin"])


# ------------------------------------------------------------------ cascade of correlation frustums
def fill_state_dict(module, seed: int):
    """Deterministic, construction-order-independent parameters: every state_dict entry is filled from its own
    generator, seeded by (seed, crc32 of the key) -- the reference-side golden scripts and the tests give a mirror and the
    reference's own modules bit-identical weights without a multi-megabyte fixture, whatever other entries the two
    modules hold.  Scales keep activations O(1)."""
    import zlib

    sd = module.state_dict()
    for k in sorted(sd):
        g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(k.encode())) % (2 ** 31))
        t = sd[k]
        if k.endswith("num_batches_tracked"):
            t.fill_(1)
        elif k.endswith("running_var"):
            t.copy_(1.0 + 0.3 * (torch.rand(t.shape, generator=g) - 0.5))
        elif k.endswith("running_mean") or k.endswith("bn.bias"):
            t.copy_(0.2 * (torch.rand(t.shape, generator=g) - 0.5))
        elif k.endswith("bn.weight") or ("norm" in k and k.endswith("weight")):
            t.copy_(1.0 + 0.4 * (torch.rand(t.shape, generator=g) - 0.5))
        elif t.dim() >= 2:
            transposed = any(n in k for n in ("conv7", "conv9", "conv11"))
            fan_in = t.shape[0] * t[0, 0].numel() if transposed else t[0].numel()
            t.copy_(_unit_uniform(tuple(t.shape), g) * (1.0 / math.sqrt(max(fan_in, 1))))
        else:
            t.copy_(0.1 * (torch.rand(t.shape, generator=g) - 0.5))
    module.load_state_dict(sd)
    return module


CASCADE_CASES = {"small3": dict(H=32, W=64, NV=3, seed=41, weight_seed=7)}
FMT_CASES = {"small3": dict(H=32, W=64, NV=3, seed=51, weight_seed=8)}


def make_fmt_case(name: str):
    """Backbone feature pyramids of NV views for a batch of NV view rotations (FeatureNet outputs: 32/16/8 channels at
    1/4, 1/2, 1/1 resolution) -- the input of FMT_with_pathway.forward (TransMVSNet.py:178-181)."""
    c = dict(FMT_CASES[name])
    H, W, NV = c["H"], c["W"], c["NV"]
    g = torch.Generator().manual_seed(c["seed"])
    feats = [{"stage1": _unit_uniform((NV, 32, H // 4, W // 4), g), "stage2": _unit_uniform((NV, 16, H // 2, W // 2), g),
              "stage3": _unit_uniform((NV, 8, H, W), g)} for _ in range(NV)]
    return dict(name=name, features=feats, NV=NV, weight_seed=c["weight_seed"])


def make_cascade_case(name: str):
    """Inputs of the cascade for one frame as UFORecon.build_pairs prepares them (model.py:139-160): B = NV rotations of
    the view list (every source view is the reference once), per-view / per-stage feature maps (32/16/8 channels at
    1/4, 1/2, 1/1 resolution), projection pairs per stage (intrinsics scaled with the stage), initial hypotheses."""
    c = dict(CASCADE_CASES[name])
    H, W, NV = c["H"], c["W"], c["NV"]
    g = torch.Generator().manual_seed(c["seed"])
    cams = make_cameras(H, W, NV)
    rot = [list(range(i, NV)) + list(range(0, i)) for i in range(NV)]          # build_pairs: all_combinations
    chans, scales = {"stage1": 32, "stage2": 16, "stage3": 8}, {"stage1": 4, "stage2": 2, "stage3": 1}
    base = {st: [_unit_uniform((chans[st], H // s, W // s), g) for _ in range(NV)] for st, s in scales.items()}
    # smooth the features a little (3x3 box) so that neighbouring hypotheses correlate like real feature maps do
    for st in base:
        base[st] = [F_avg(t) for t in base[st]]
    features = [{st: torch.stack([base[st][rot[b][v]] for b in range(NV)]) for st in scales} for v in range(NV)]
    proj = {}
    for st, s in scales.items():
        pm = torch.zeros(NV, NV, 2, 4, 4)
        for b in range(NV):
            for v in range(NV):
                src = rot[b][v]
                pm[b, v, 0] = cams["w2cs"][0, src]
                K = cams["intrinsics"][0, src].clone()
                K[:2] = K[:2] / s
                pm[b, v, 1, :3, :3] = K
                pm[b, v, 1, 3, 3] = 1.0
        proj[st] = pm
    near, far = float(cams["near_fars"][0, 0, 0]), float(cams["near_fars"][0, 0, 1])
    depth_values = torch.linspace(near, far, 48).reshape(1, -1).expand(NV, -1).contiguous()
    return dict(name=name, features=features, proj_matrices=proj, depth_values=depth_values, img_hw=(H, W), NV=NV,
                weight_seed=c["weight_seed"])


def F_avg(t):
    import torch.nn.functional as F

    return F.avg_pool2d(t[None], 3, stride=1, padding=1, count_include_pad=False)[0].contiguous()
