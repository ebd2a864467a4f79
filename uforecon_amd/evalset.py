"""The reference's evaluation loop over a set of frames, on the GPUs of one node (BASELINE.json configs[2]).

What the reference does (main.py:229-230 -> code1/model.py:760-842, once per frame of the 15-scene x 3-render-view DTU
evaluation set): run the per-frame producers (backbone, FMT, frustum cascade: `UFOReconInference.encode_frame`), render
every pixel ray of the frame in chunks, post-process and write the depth map.  Here, per frame:

  producers   `encode_frame` on a side HIP stream, ONE FRAME AHEAD of the ray path: frame k+1 is encoded while frame k's
              rays run (the ray kernels leave room: they are bound by the matrix cores, the producers by HBM / L2);
              with several ranks either every rank encodes every frame ("replicated": no communication, but the encode
              time does not shrink with the rank count) or the frames' producers are dealt round-robin over the ranks and
              the owner broadcasts the per-frame tensors (features 39 MB + frustums 672 MB at 512x640) over xGMI
              ("sharded": 1/N of the encodes per rank for one ~0.7 GB broadcast per frame, also one frame ahead);
  ray path    this rank's row tile of the frame through `ufr_render_rays` (uforecon_amd.dist.RayShard);
  exchange    one all-gather of the depth (+ RGB) tiles per frame.

`run()` returns the depth maps (optionally) and the time split the scaling discussion needs: producer ms, ray-path ms,
all-gather ms per frame, and the inclusive wall time.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .dist import RayShard, all_gather_tiles
from .scene import STAGE_SHAPE, STAGES, make_cameras


def make_eval_batch(H: int, W: int, NV: int, seed: int, device) -> dict:
    """One synthetic evaluation frame: `make_cameras` geometry with a render-view offset that
    depends on the frame, uniform source images, and the encoder's inputs (`proj_matrices` per cascade stage,
    `depth_values_org_scale`) derived from the cameras like dtu_test_sparse.py:382-436 does."""
    cams = make_cameras(H, W, NV, offset_dist=0.06 + 0.01 * (seed % 5))
    g = torch.Generator().manual_seed(7000 + seed)
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in cams.items()}
    batch["source_imgs"] = torch.rand(1, NV, 3, H, W, generator=g).to(device)
    batch["start_idx"] = 0
    pm = {}
    for st, s in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        p = torch.zeros(1, NV, 2, 4, 4, device=device)
        p[0, :, 0] = batch["w2cs"][0, :NV]
        K = batch["intrinsics"][0, :NV].clone()
        K[:, :2] = K[:, :2] / s
        p[0, :, 1, :3, :3] = K
        p[0, :, 1, 3, 3] = 1.0
        pm[st] = p
    batch["proj_matrices"] = pm
    near, far = float(batch["near_fars"][0, 0, 0]), float(batch["near_fars"][0, 0, 1])
    batch["depth_values_org_scale"] = torch.linspace(near, far, 48, device=device)[None]
    batch["meta"] = [f"dtu-scan{seed // 3}-refview{seed % 3}"]
    return batch


def _frame_tensor_shapes(H: int, W: int, NV: int):
    """Shapes of what `encode_frame` hands to the ray path, in the order they travel in the broadcast buffer."""
    h, w = H // 4, W // 4
    shapes = [("feat", (1, NV, 32, h, w)), ("match", (1, NV, 32 * (NV - 1), h, w)), ("depth_info", (1, NV, H, W))]
    for st in STAGES:
        D, s = STAGE_SHAPE[st]
        shapes += [(st + ".f", (NV, 8, D, H // s, W // s)), (st + ".w", (NV, 1, D, H // s, W // s))]
    return shapes


def _lowest_priority() -> int:
    try:
        return int(torch.cuda.Stream.priority_range()[0])       # (least, greatest): least = numerically largest = lowest
    except Exception:  # noqa: BLE001 -- older torch: no query; 0 is always valid
        return 0


class EvalLoop:
    def __init__(self, net, shard: RayShard, n_streams: int = 3, overlap: bool = True, producers: str = "replicated",
                 chunk_rays: int = 0):
        if producers not in ("replicated", "sharded"):
            raise ops.UfrError(f"producers={producers!r}")
        self.net, self.shard, self.overlap, self.producers = net, shard, overlap, producers
        self.dev = next(net.parameters()).device
        self.main = torch.cuda.current_stream(self.dev)
        # the producers' stream at the LOWEST priority the device offers (torch: larger number = lower priority): frame
        # k+1's encode kernels are dispatched into what frame k's ray kernels (highest priority: csrc/ufr_api.hip
        # side_pool_get) leave free, instead of taking turns with them
        self.enc_stream = torch.cuda.Stream(self.dev, priority=_lowest_priority()) if overlap else self.main
        # the frustum broadcast of frame k+1 on its OWN process group (= its own communicator and stream under RCCL): on the
        # default group it would sit in front of frame k's all-gather in one queue -- 0.7 GB ahead of 5 MB
        self.bcast_group = dist.new_group() if (producers == "sharded" and shard.world > 1) else None
        self.n_streams, self.chunk_rays = n_streams, chunk_rays
        self.ray_idx = shard.ray_indices(self.dev)
        self._ws = None

    # ---- producers of one frame, on the encode stream; returns (feat, frustums, match, depth_info, events)
    def _encode(self, batch, k: int):
        sh = self.shard
        with torch.cuda.stream(self.enc_stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            owner = k % sh.world if self.producers == "sharded" else sh.rank
            if self.producers == "sharded" and sh.world > 1:
                H, W = batch["source_imgs"].shape[-2:]
                NV = batch["source_imgs"].shape[1]
                shapes = _frame_tensor_shapes(H, W, NV)
                sizes = [int(torch.Size(s).numel()) for _, s in shapes]
                flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.dev)
                views, off = {}, 0
                for (name, s), n in zip(shapes, sizes):
                    views[name] = flat[off:off + n].view(s)
                    off += n
                if sh.rank == owner:
                    feat, fr, match = self.net.encode_frame(batch)
                    views["feat"].copy_(feat)
                    views["match"].copy_(match[0])
                    views["depth_info"].copy_(batch["depth_info"])
                    for st in STAGES:
                        views[st + ".f"].copy_(fr[st]["feature_volume"])
                        views[st + ".w"].copy_(fr[st]["weight_volume"])
                dist.broadcast(flat, src=owner, group=self.bcast_group)
                feat, match = views["feat"], [views["match"]]
                fr = {st: {"feature_volume": views[st + ".f"], "weight_volume": views[st + ".w"]} for st in STAGES}
                depth_info = views["depth_info"]
                keep = [flat]
            else:
                feat, fr, match = self.net.encode_frame(batch)
                depth_info = batch["depth_info"]
                keep = [feat, match[0], depth_info] + [t for st in STAGES for t in fr[st].values()]
            e1.record()
            done = torch.cuda.Event()
            done.record()
        if self.enc_stream is not self.main:
            for t in keep:            # produced on the encode stream, consumed on the main one: the allocator must not hand
                t.record_stream(self.main)   # the memory to a later encode while the ray kernels still read it
        return feat, fr, match, depth_info, (e0, e1, done), owner

    def run(self, batches: List[dict], uniforms: Optional[Callable[[int], tuple]] = None, keep_depth: bool = False,
            want_rgb: bool = True):
        """Render every frame of `batches`.  uniforms(k) -> (U1 (SN,HW), U2 (PN,HW)) pins the sampler randomness of frame k
        (sliced to this rank's rays); default: fresh GPU draws.  Returns a dict of per-frame timings (ms) and totals."""
        net, sh, dev = self.net, self.shard, self.dev
        SN, PN = net.point_num, net.point_num_2
        n = len(batches)
        NV = batches[0]["source_imgs"].shape[1]
        if self._ws is None:
            self._ws = ops.RenderWorkspace(dev, SN, PN, NV, chunk_rays=self.chunk_rays, n_streams=self.n_streams)
        RN = self.ray_idx.numel()
        out = dict(depth=torch.empty(RN, device=dev), depth_z=torch.empty(RN, device=dev), rgb=torch.empty(RN, 3, device=dev))
        W = net._weights()
        depths, ev = [], []
        torch.cuda.synchronize(dev)
        if sh.world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        pending = self._encode(batches[0], 0)
        for k in range(n):
            cur = pending
            if k + 1 < n and self.overlap:
                pending = self._encode(batches[k + 1], k + 1)     # enqueued BEFORE frame k's rays: runs beside them
            feat, fr, match, depth_info, (e0, e1, done), owner = cur
            self.main.wait_event(done)
            batch = batches[k]
            batch["depth_info"] = depth_info
            r0, r1, a0, a1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            r0.record()
            fh = net.frame_handle(batch, feat, fr, match)
            if uniforms is None:
                U1, U2 = torch.rand(SN, RN, device=dev), torch.rand(PN, RN, device=dev)
            else:
                u1, u2 = uniforms(k)
                U1, U2 = u1.to(dev)[:, self.ray_idx].contiguous(), u2.to(dev)[:, self.ray_idx].contiguous()
            ops.render_rays(fh, W, self.ray_idx, U1, U2, workspace=self._ws, want_srdf=False, out=out)
            r1.record()
            a0.record()
            d, c = all_gather_tiles(out["depth_z"], out["rgb"] if want_rgb else None, sh)
            d = d * batch["scale_mat"][0][0, 0]                                             # model.py:826
            a1.record()
            if keep_depth:
                depths.append(d.clone())
            ev.append((e0, e1, r0, r1, a0, a1, owner))
            if k + 1 < n and not self.overlap:
                pending = self._encode(batches[k + 1], k + 1)
        torch.cuda.synchronize(dev)
        if sh.world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        enc = [e0.elapsed_time(e1) for e0, e1, *_ in ev]
        ray = [r0.elapsed_time(r1) for _, _, r0, r1, *_ in ev]
        ag = [a0.elapsed_time(a1) for *_, a0, a1, _ in ev]
        HW = sh.H * sh.W
        return dict(frames=n, wall_s=wall, ms_per_frame=wall / n * 1e3, rays_per_s=HW * n / wall,
                    encode_ms=enc, ray_path_ms=ray, all_gather_ms=ag,
                    encodes_on_this_rank=sum(1 for *_, o in ev if o == sh.rank),
                    ray_path_rays_per_s_this_rank=RN * n / (sum(ray) * 1e-3), depths=depths)
