"""Development probe: per-tensor gradient errors of the training step (HIP vs golden reference autograd vs oracle autograd
on this host).  python tools/dev/bwd_diag.py [case]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import CASES, VOLUME_KEYS, case_inputs, golden_volume_grad, grad_rel_err, load_weights, rel_err
from oracle import ufo_oracle as O
import test_gpu_backward as T

name = sys.argv[1] if len(sys.argv) > 1 else "c5_train_grads"
m, f, r, loss, g = T._train_step(name)
loss.backward()
torch.cuda.synchronize()
# oracle autograd on this host
fr, idx, U1, U2, _ = case_inputs(name)
P = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in load_weights().items()}
for st in fr.feature_volume:
    for k in fr.feature_volume[st]:
        fr.feature_volume[st][k] = fr.feature_volume[st][k].clone().requires_grad_(True)
ro = O.infer(P, fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2, extract_geometry=False)
lo = O.training_loss(ro, fr.batch, idx)
lo.backward()
print(f"loss hip {float(loss):.7f} golden {float(g['loss']):.7f} oracle {float(lo):.7f}")
names = ["rgb_gt", "rgb", "depth", "depth_gt", "srdf", "opacity", "weight", "pp", "rgb_2", "depth_2", "srdf_2", "opacity_2",
         "weight_2", "pp2", "z_val", "z_val_all", "variance"]
got = dict(zip(names, r))
for k in ("rgb", "depth", "rgb_2", "depth_2", "z_val_all", "weight_2"):
    ref = ro[k]
    print(f"fwd {k:10s} vs oracle {rel_err(got[k].detach().reshape(ref.shape), ref.detach()):.2e}")
print(f"{'tensor':75s} {'hip-gold':>9s} {'hip-orc':>9s} {'orc-gold':>9s} {'scale':>9s}")
for k, p in m.named_parameters():
    gg = torch.from_numpy(g["grad." + k])
    print(f"{k:75s} {grad_rel_err(p.grad, gg):9.2e} {grad_rel_err(p.grad, P[k].grad):9.2e} {grad_rel_err(P[k].grad, gg):9.2e} {float(gg.abs().max()):9.2e}")
for key in VOLUME_KEYS:
    st, k = key.split(".")
    v = f.feature_volume[st][k]
    gg = golden_volume_grad(g, key, v.shape)
    vo = fr.feature_volume[st][k].grad
    print(f"{key:75s} {grad_rel_err(v.grad, gg):9.2e} {grad_rel_err(v.grad, vo):9.2e} {grad_rel_err(vo, gg):9.2e} {float(gg.abs().max()):9.2e}")
