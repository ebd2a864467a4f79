"""How torch's CPU kernels round the ops of the per-ray path (the arithmetic csrc/gather.hip and
csrc/sampler.hip reproduce).  Each candidate formula is emulated in numpy float32 (fma = float64
product-sum rounded once) and compared BIT FOR BIT with torch.  Prints the mismatch count per candidate;
0 identifies torch's formula on this host.  Run: python tools/probe_torch_arith.py
"""
import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def probe_bmm():
    torch.manual_seed(0)
    P = torch.randn(3, 4, 4)
    pts = torch.randn(5000, 3)
    ph = torch.cat([pts, torch.ones(5000, 1)], 1)
    q = torch.bmm(P, ph.t()[None].expand(3, 4, -1).contiguous()).numpy()
    x, y, z = pts.numpy().T
    bad = 0
    for v in range(3):
        for i in range(4):
            M = P[v, i].numpy()
            a = (M[0] * x).astype(f32)
            a = fma(M[1], y, a)
            a = fma(M[2], z, a)
            bad += int(((a + M[3]).astype(f32) != q[v, i]).sum())
    print("bmm 4x4 == k-ordered fma chain:", bad, "mismatches")


def probe_grid_sample_2d():
    torch.manual_seed(0)
    C, H, W, N = 4, 37, 53, 20000
    inp = torch.randn(1, C, H, W)
    grid = torch.rand(1, 1, N, 2) * 2.2 - 1.1
    x, y = grid[0, 0, :, 0].numpy(), grid[0, 0, :, 1].numpy()
    I = inp[0].numpy()

    def gather(ix, iy):
        ok = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
        return np.where(ok, I[:, np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)], f32(0))

    for ac, pad in ((False, "zeros"), (True, "border")):
        ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=ac)[0, :, 0].numpy()
        if ac:
            ux = ((x + f32(1)) * f32((W - 1) / 2)).astype(f32)
            uy = ((y + f32(1)) * f32((H - 1) / 2)).astype(f32)
            ux, uy = np.clip(ux, f32(0), f32(W - 1)), np.clip(uy, f32(0), f32(H - 1))
        else:
            ux = fma((x + f32(1)).astype(f32), np.full_like(x, f32(W / 2)), np.full_like(x, f32(-0.5)))
            uy = fma((y + f32(1)).astype(f32), np.full_like(y, f32(H / 2)), np.full_like(y, f32(-0.5)))
        fx, fy = np.floor(ux), np.floor(uy)
        w_, n_ = ux - fx, uy - fy
        e_, s_ = f32(1) - w_, f32(1) - n_
        nw, ne, sw, se = (s_ * e_).astype(f32), (s_ * w_).astype(f32), (n_ * e_).astype(f32), (n_ * w_).astype(f32)
        ix0, iy0 = fx.astype(np.int64), fy.astype(np.int64)
        a = (gather(ix0, iy0) * nw).astype(f32)
        a = fma(gather(ix0 + 1, iy0), np.broadcast_to(ne, a.shape), a)
        a = fma(gather(ix0, iy0 + 1), np.broadcast_to(sw, a.shape), a)
        a = fma(gather(ix0 + 1, iy0 + 1), np.broadcast_to(se, a.shape), a)
        print(f"grid_sample 2-D align_corners={ac} {pad}:", int((a != ref).sum()), "mismatches")


def probe_grid_sample_3d():
    torch.manual_seed(0)
    C, D, H, W, N = 3, 7, 11, 13, 20000
    inp = torch.randn(1, C, D, H, W)
    grid = torch.rand(1, 1, 1, N, 3) * 2.2 - 1.1
    x, y, z = [grid[0, 0, 0, :, i].numpy() for i in range(3)]
    I = inp[0].numpy()
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, :, 0, 0].numpy()
    un = lambda c, s: (((c + f32(1)) / f32(2)).astype(f32) * f32(s - 1)).astype(f32)
    ix, iy, iz = un(x, W), un(y, H), un(z, D)
    fx, fy, fz = np.floor(ix), np.floor(iy), np.floor(iz)
    out = np.zeros((C, N), f32)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wx = (fx + f32(1)) - ix if dx == 0 else ix - fx
                wy = (fy + f32(1)) - iy if dy == 0 else iy - fy
                wz = (fz + f32(1)) - iz if dz == 0 else iz - fz
                wt = ((wx * wy).astype(f32) * wz).astype(f32)
                cx, cy, cz = (fx + dx).astype(np.int64), (fy + dy).astype(np.int64), (fz + dz).astype(np.int64)
                ok = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H) & (cz >= 0) & (cz < D)
                v = I[:, np.clip(cz, 0, D - 1), np.clip(cy, 0, H - 1), np.clip(cx, 0, W - 1)]
                out = np.where(ok, (out + (v * wt).astype(f32)).astype(f32), out)
    print("grid_sample 3-D (scalar unnormalize, unfused accumulate):", int((out != ref).sum()), "mismatches")


def probe_addcmul_cumsum_sum():
    torch.manual_seed(0)
    x = torch.rand(100000) * 2 - 1
    fr = torch.repeat_interleave(np.pi * 2.0 ** torch.arange(0, 4), 2)
    ph = torch.zeros(8)
    ph[1::2] = np.pi * 0.5
    arg = torch.addcmul(ph[None], x[:, None].repeat(1, 8), fr[None]).numpy()
    print("addcmul == single fma:", int((fma(x.numpy()[:, None], fr.numpy()[None], ph.numpy()[None]) != arg).sum()), "mismatches")
    w = torch.rand(256, 64) ** 4
    cs = torch.cumsum(w, 1).numpy()
    print("cumsum == float64 running sum:", int((np.cumsum(w.numpy().astype(np.float64), 1).astype(f32) != cs).sum()), "mismatches")
    wn = w.numpy()
    acc = np.zeros((256, 4, 8), f32)
    for c in range(2):
        for k in range(4):
            acc[:, k] = (acc[:, k] + wn[:, (c * 4 + k) * 8:(c * 4 + k + 1) * 8]).astype(f32)
    t = acc[:, 0]
    for k in range(1, 4):
        t = (t + acc[:, k]).astype(f32)
    h = np.zeros(256, f32)
    for l in range(8):
        h = (h + t[:, l]).astype(f32)
    print("sum(64) == 4 accumulators x 8 lanes, in-order combine:", int((h != w.sum(1).numpy()).sum()), "mismatches")


if __name__ == "__main__":
    print("torch", torch.__version__, torch.backends.cpu.get_cpu_capability())
    probe_bmm()
    probe_grid_sample_2d()
    probe_grid_sample_3d()
    probe_addcmul_cumsum_sum()
