"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Rays are independent given the per-frame tensors (no cross-ray op anywhere on the path: the ray
transformer attends only within a ray, code1/ray_transformer.py:301-305), so rank r renders the
contiguous ROW TILE [row0, row1) of the H x W ray grid -- contiguous tiles keep the gather locality
and make the exchange a plain concatenation in (H W) order, which is how the reference assembles
the frame from its chunks (code1/model.py:825).  The only collective is one all-gather of the
depth (and RGB) tiles per frame: 4 B (+12 B) per ray, i.e. 1.3 MB (+3.9 MB) per 512x640 frame --
latency-bound on xGMI, far below the ~153 GB/s per link.  The per-frame inputs are replicated.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class RayShard:
    H: int
    W: int
    world: int
    rank: int

    def rows(self, rank: int | None = None):
        r = self.rank if rank is None else rank
        base, rem = divmod(self.H, self.world)
        row0 = r * base + min(r, rem)
        return row0, row0 + base + (1 if r < rem else 0)

    @property
    def n_rays(self) -> int:
        r0, r1 = self.rows()
        return (r1 - r0) * self.W

    @property
    def max_rays(self) -> int:
        return max((self.rows(r)[1] - self.rows(r)[0]) * self.W for r in range(self.world))

    def ray_indices(self, device) -> torch.Tensor:
        r0, r1 = self.rows()
        return torch.arange(r0 * self.W, r1 * self.W, device=device, dtype=torch.int64)


def all_gather_tiles(depth: torch.Tensor, rgb: torch.Tensor | None, shard: RayShard):
    """All-gather the per-rank tiles into full (H,W) depth and (H,W,3) RGB maps on every rank.

    Tiles may differ by one row when world does not divide H; they are padded to the largest tile so
    a single all_gather_into_tensor per map suffices."""
    if shard.world == 1:
        return depth.reshape(shard.H, shard.W), None if rgb is None else rgb.reshape(shard.H, shard.W, 3)
    n, m = shard.n_rays, shard.max_rays

    def gather(t: torch.Tensor, width: int):
        send = t.reshape(n, width)
        if n < m:
            send = torch.cat([send, send.new_zeros(m - n, width)], 0)
        recv = send.new_empty(shard.world * m, width)
        dist.all_gather_into_tensor(recv, send.contiguous())
        parts = []
        for r in range(shard.world):
            r0, r1 = shard.rows(r)
            parts.append(recv[r * m: r * m + (r1 - r0) * shard.W])
        return torch.cat(parts, 0)

    d = gather(depth, 1).reshape(shard.H, shard.W)
    c = None if rgb is None else gather(rgb, 3).reshape(shard.H, shard.W, 3)
    return d, c


# ------------------------------------------------------------------ data-parallel training (BASELINE configs[4])
def allreduce_gradients(params, world: int | None = None, group=None) -> int:
    """Average the gradients of `params` over the ranks with ONE collective: the gradients are packed into a single
    flat fp32 buffer (the per-ray path has 148 947 parameters = 0.6 MB; with feature_volume.cost_reg_2.* 1.8 MB), summed
    with an RCCL all-reduce (ring over xGMI: latency-bound at this size, so one bucket, not one call per tensor) and
    scattered back.  Every rank must pass the same parameters in the same order.  The bucket holds EVERY tensor that
    requires grad -- one whose ``.grad`` is None on this rank (unused this step) contributes zeros and receives the other
    ranks' mean -- so the buffer length cannot differ between ranks.  Tensors that are not parameters of the module (the
    sampled volumes, when a caller trains through them) are reduced the same way: pass them in ``params``.
    Returns the number of floats reduced."""
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return 0
    world = dist.get_world_size(group) if world is None else world
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
    off = 0
    for p in ps:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.to(p.dtype).clone()
        else:
            p.grad.copy_(g)
        off += n
    return off
