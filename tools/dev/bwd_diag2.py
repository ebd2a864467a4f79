"""Development probe: where do view-transformer gradients differ?  Compares the per-token d_hid dump of ufr_aggregate_bwd with
autograd through an inlined copy of the oracle's layer, and prints the hidden pre-activation at the mismatching units."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from oracle import ufo_oracle as O
from uforecon_amd import ops
import test_gpu_backward as T

name = sys.argv[1] if len(sys.argv) > 1 else "c5_train_grads"
fr, P, W, fh, ray_o, ray_d, z, x, rgbm, dirs, dbg = T._token_inputs(name)
RN, SN = z.shape
NV = x.shape[1]
radiance, srdf, agg = ops.aggregate(W, x, rgbm, dirs, RN, SN, keep_workspace=True)
g = torch.Generator().manual_seed(3)
co_rad, co_srdf = torch.rand(RN * SN, 3, generator=g) - 0.5, torch.rand(RN, SN, generator=g) - 0.5
grads = ops.GradBuffer("cuda:0")
d_pv, dd = ops.aggregate_bwd(W, grads, x, rgbm, dirs, agg["token0"], RN, SN, co_rad.cuda(), co_srdf.cuda(), debug=True)
torch.cuda.synchronize()
dv = dd["view"].cpu().reshape(RN * SN, NV + 1, 881)

# inlined oracle view layer with a handle on the hidden pre-activation
Pg = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in P.items()}
pre = O.VT
xc = x.cpu()
tok = Pg["ray_transformer.viewToken.view_token"].expand(xc.shape[0], 1, 80)
xt = torch.cat([tok, xc], 1)
N, L, C = xt.shape
q = (xt @ Pg[pre + "q_proj.weight"].t()).view(N, L, 8, 10)
k = (xt @ Pg[pre + "k_proj.weight"].t()).view(N, L, 8, 10)
v = (xt @ Pg[pre + "v_proj.weight"].t()).view(N, L, 8, 10)
m = O.linear_attention(q, k, v).reshape(N, L, C) @ Pg[pre + "merge.weight"].t()
m = F.layer_norm(m, (C,), Pg[pre + "norm1.weight"], Pg[pre + "norm1.bias"])
hpre = torch.cat([xt, m], 2) @ Pg[pre + "mlp.0.weight"].t()
hpre.retain_grad()
o = torch.relu(hpre) @ Pg[pre + "mlp.2.weight"].t()
y = xt + F.layer_norm(o, (C,), Pg[pre + "norm2.weight"], Pg[pre + "norm2.bias"])
t0 = y[:, 0].reshape(RN, SN, 80)
pe = O.order_posenc(8, SN).to(t0)
r = O.loftr_layer(torch.cat([t0, pe[None].expand(RN, SN, 8)], 2), Pg, O.RT)
srdf_o = O.mlp3(r, Pg, "ray_transformer.DensityMLP.")[..., 0]
logit = O.mlp3(torch.cat([y[:, 1:], dirs.cpu()[..., :3]], -1), Pg, "ray_transformer.linear_radianceweight_1_softmax.")
logit = torch.where(rgbm.cpu()[..., 3:4] == 0, torch.full_like(logit, -1e9), logit)
rad_o = (rgbm.cpu()[..., :3] * torch.softmax(logit, dim=-2)).sum(1)
((rad_o * co_rad).sum() + (srdf_o * co_srdf).sum()).backward()
dh_ref = hpre.grad                       # (N, L, 160)
dh_hip = dv[:, :, 160:320]
err = (dh_hip - dh_ref).abs()
scale = dh_ref.abs().max()
print("d_hid max err / scale:", float(err.max() / scale))
bad = torch.nonzero(err > 1e-4 * scale)
print("mismatching (point, token, unit):", bad.shape[0])
for b in bad[:20]:
    p, t, u = [int(i) for i in b]
    print(f"  p={p} t={t} u={u} hpre={float(hpre[p, t, u]):+.3e} ref={float(dh_ref[p, t, u]):+.3e} hip={float(dh_hip[p, t, u]):+.3e}")
