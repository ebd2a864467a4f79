"""Development check of the ray transformer's backward (ufr_ray_transform_bwd) against autograd through the oracle:
d token0 and every ray-transformer / DensityMLP parameter gradient, per tensor."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_weights
from oracle import ufo_oracle as O
from uforecon_amd import ops

DEV = "cuda:0"
RN, SN = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = ops.PRECISION_16BIT if "16" in sys.argv[3:] else ops.PRECISION_FP32
torch.manual_seed(0)
P = load_weights()
tok = torch.randn(RN * SN, 80) * 0.7
co = torch.randn(RN, SN)
Pg = {k: v.clone().requires_grad_(k.startswith("ray_transformer.density_ray") or "DensityMLP" in k) for k, v in P.items()}
t = tok.clone().requires_grad_(True)
pe = O.order_posenc(8, SN).to(t)
r = torch.cat([t.reshape(RN, SN, 80), pe[None].expand(RN, SN, 8)], 2)
r = O.loftr_layer(r, Pg, O.RT)
srdf = O.mlp3(r, Pg, "ray_transformer.DensityMLP.")[..., 0]
(srdf * co).sum().backward()
W = ops.PackedWeights({k: v.to(DEV) for k, v in P.items()}, precision=prec)
srdf_k = ops.ray_transform(W, tok.to(DEV), RN, SN)
print("forward srdf err", float((srdf_k.cpu() - srdf.detach()).abs().max() / srdf.detach().abs().max()))
grads = ops.GradBuffer(DEV)
a, b = ops.ray_transform_bwd(W, grads, tok.to(DEV), RN, SN, co.to(DEV))
torch.cuda.synchronize()
got = (a + b).cpu()
ref = t.grad
print("d token0 rel err", float((got - ref).abs().max() / ref.abs().max()), "scale", float(ref.abs().max()))
e = (got - ref).abs().reshape(RN, SN, 80)
print("  per-feature-tile max err:", [float(e[..., 16 * i:16 * i + 16].max()) for i in range(5)])
print("  per-token-tile max err:", [float(e[:, 16 * i:16 * i + 16].max()) for i in range(SN // 16)])
for k in ops.RAW_WEIGHT_KEYS:
    if Pg[k].grad is None:
        continue
    g, rf = grads.grad(k).cpu(), Pg[k].grad
    print(f"{k:70s} {float((g - rf).abs().max() / rf.abs().max().clamp_min(1e-9)):.3e}  scale {float(rf.abs().max()):.2e}")

# ---- tile-level comparison (fp32 mode): decode the workspace [order code | tape | state | dY | srdf]
if prec == ops.PRECISION_FP32:
    keep = []
    grads2 = ops.GradBuffer(DEV)
    ops.ray_transform_bwd(W, grads2, tok.to(DEV), RN, SN, co.to(DEV), _workspace_out=keep)
    torch.cuda.synchronize()
    ws = keep[0].cpu()
    def al(n): return (n * 4 + 255) // 256 * 256 // 4
    nb = (SN // 16 + 1) // 2
    RT_COUNT, DR_COUNT = 59, 55
    o_tape = al(SN * 8)
    o_state = o_tape + al(RN * nb * RT_COUNT * 2 * 256)
    o_dy = o_state + al(RN * 16 * 256)
    tape = ws[o_tape:o_tape + RN * nb * RT_COUNT * 2 * 256].reshape(RN, nb, RT_COUNT, 2, 64, 4)
    dy = ws[o_dy:o_dy + RN * nb * DR_COUNT * 2 * 256].reshape(RN, nb, DR_COUNT, 2, 64, 4)
    def nat88(t, g, r): return 16 * t + 4 * g + r if t < 5 else (80 + 2 * g + r if r < 2 else -1)
    def tiles_to_feat(T, tile0, n, fmap, dim):
        """T (RN,nb,tiles,2,64,4) -> (RN, SN, dim)"""
        out = torch.zeros(RN, nb * 32, dim)
        for t in range(n):
            for lane in range(64):
                g, jj = lane >> 4, lane & 15
                for r in range(4):
                    f = fmap(t, g, r)
                    if f is None or f < 0 or f >= dim: continue
                    for c in range(2):
                        out[:, c * 16 + jj::32, f] = T[:, :, tile0 + t, c, lane, r]
        return out[:, :SN]
    nat = lambda t, g, r: 16 * t + 4 * g + r
    # oracle intermediates
    Pd = {k: v.clone() for k, v in P.items()}
    pre = O.RT
    x = torch.cat([tok.reshape(RN, SN, 80), pe[None].expand(RN, SN, 8)], 2).requires_grad_(True)
    C = 88
    q = (x @ Pd[pre + "q_proj.weight"].t()); k = (x @ Pd[pre + "k_proj.weight"].t()); v = (x @ Pd[pre + "v_proj.weight"].t())
    msg = O.linear_attention(q.view(RN, SN, 8, 11), k.view(RN, SN, 8, 11), v.view(RN, SN, 8, 11)).reshape(RN, SN, C)
    msg.retain_grad(); q.retain_grad(); k.retain_grad(); v.retain_grad()
    mpre = msg @ Pd[pre + "merge.weight"].t(); mpre.retain_grad()
    m = torch.nn.functional.layer_norm(mpre, (C,), Pd[pre + "norm1.weight"], Pd[pre + "norm1.bias"])
    hidp = torch.cat([x, m], 2) @ Pd[pre + "mlp.0.weight"].t(); hidp.retain_grad()
    hid = torch.relu(hidp)
    opre = hid @ Pd[pre + "mlp.2.weight"].t(); opre.retain_grad()
    o = x + torch.nn.functional.layer_norm(opre, (C,), Pd[pre + "norm2.weight"], Pd[pre + "norm2.bias"]); o.retain_grad()
    dm = "ray_transformer.DensityMLP."
    d1p = torch.nn.functional.linear(o, Pd[dm + "0.weight"], Pd[dm + "0.bias"]); d1p.retain_grad()
    d2p = torch.nn.functional.linear(torch.relu(d1p), Pd[dm + "2.weight"], Pd[dm + "2.bias"]); d2p.retain_grad()
    s_ = torch.nn.functional.linear(torch.relu(d2p), Pd[dm + "4.weight"], Pd[dm + "4.bias"])[..., 0]
    (s_ * co).sum().backward()
    def cmp(name, got, ref):
        print(f"  {name:10s} err {float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)):.3e}  scale {float(ref.abs().max()):.2e}")
    RT = dict(X=0, Q=6, MSG=12, ZS=18, XH1=20, M=26, HID=32, XH2=43, O=49, D1=55, D2=57, MISC=58)
    DR = dict(Q=0, K=6, V=14, MPRE=22, HID=28, OPRE=39, D1=45, D2=47, SR=48, SCR=49)
    print("tape tiles:")
    cmp("X", tiles_to_feat(tape, RT["X"], 6, nat88, 88), x.detach())
    cmp("M", tiles_to_feat(tape, RT["M"], 6, nat88, 88), m.detach())
    cmp("HID", tiles_to_feat(tape, RT["HID"], 11, nat, 176), hid.detach())
    cmp("O", tiles_to_feat(tape, RT["O"], 6, nat88, 88), o.detach())
    cmp("D1", tiles_to_feat(tape, RT["D1"], 2, nat, 32), torch.relu(d1p).detach())
    cmp("D2", tiles_to_feat(tape, RT["D2"], 1, nat, 16), torch.relu(d2p).detach())
    print("cotangent tiles:")
    cmp("dD2", tiles_to_feat(dy, DR["D2"], 1, nat, 16), d2p.grad)
    cmp("dD1", tiles_to_feat(dy, DR["D1"], 2, nat, 32), d1p.grad)
    cmp("dOPRE", tiles_to_feat(dy, DR["OPRE"], 6, nat88, 88), opre.grad)
    cmp("dHID", tiles_to_feat(dy, DR["HID"], 11, nat, 176), hidp.grad)
    cmp("dMPRE", tiles_to_feat(dy, DR["MPRE"], 6, nat88, 88), mpre.grad)
    def quad11(t, g, r):
        h, qq = divmod(4 * t + r, 3)
        return 11 * h + 3 * g + qq if 3 * g + qq < 11 else -1
    def head11(t, g, r):
        return 11 * t + 3 * g + r if (r < 3 and 3 * g + r < 11) else -1
    cmp("dQ", tiles_to_feat(dy, DR["Q"], 6, quad11, 88), q.grad)
    cmp("dK", tiles_to_feat(dy, DR["K"], 8, head11, 88), k.grad)
    cmp("dV", tiles_to_feat(dy, DR["V"], 8, head11, 88), v.grad)
    # ---- ReLU bit masks of the MISC tile
    misc = tape[:, :, RT["MISC"]]                      # (RN, nb, 2, 64, 4)
    b0 = misc[..., 2].contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    b1 = misc[..., 3].contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    def bit(bitidx):
        return ((b0 >> bitidx) & 1) if bitidx < 32 else ((b1 >> (bitidx - 32)) & 1)
    def mask_feat(bit0, n, dim):
        out = torch.zeros(RN, nb * 32, dim)
        for t in range(n):
            for lane in range(64):
                g, jj = lane >> 4, lane & 15
                for r in range(4):
                    f = 16 * t + 4 * g + r
                    if f >= dim: continue
                    for c in range(2):
                        out[:, c * 16 + jj::32, f] = bit(bit0 + 4 * t + r)[:, :, c, lane].float()
        return out[:, :SN]
    print("mask mismatches: hid", int((mask_feat(0, 11, 176) != (hidp > 0).float()).sum()), " d1", int((mask_feat(44, 2, 32) != (d1p > 0).float()).sum()),
          " d2", int((mask_feat(52, 1, 16) != (d2p > 0).float()).sum()))
    print("rstd1/2 sample", misc[0, 0, 0, :3, 0], misc[0, 0, 0, :3, 1])
    w4 = P[dm + "4.weight"][0]
    exp = (d2p > 0).float() * w4 * co[..., None]
    cmp("dD2 (own formula)", exp, d2p.grad)
    got = tiles_to_feat(dy, DR["D2"], 1, nat, 16)
    print("ratio got/expected at nonzero (first ray, first 4 tokens):")
    print(got[0, :4], exp[0, :4])
    if os.environ.get("UFR_LIB", "").endswith("rddbg.so"):
        dbg = dy[:, :, DR["SCR"] + 5]
        print("dgrad saw misc lane0:", dbg[0, 0, 0, 0], " tape misc lane0:", misc[0, 0, 0, 0])
        print("dgrad saw w4 (lane groups 0..3):", dy[0, 0, DR["SCR"] + 4, 0, ::16], " true w4:", w4)
        seen = dy[:, :, DR["SCR"] + 3]       # (RN, nb, 2, 64, 4): relu_on(52 + r) as the kernel evaluated it
        want = torch.stack([bit(52 + r) for r in range(4)], -1).float()
        print("relu_on(52+r) kernel vs tape bits: mismatches", int((seen != want).sum()), "of", seen.numel())
        print(" kernel:", seen[0, 0, 0, :8].flatten().tolist()); print(" bits:  ", want[0, 0, 0, :8].flatten().tolist())
