"""Anchor for the 16-bit matrix mode (BASELINE configs[4] "bf16"): RUN THE REFERENCE's training step under
``torch.autocast("cpu", dtype=torch.bfloat16)`` -- what Lightning's ``precision="bf16"`` does to it -- on the rays, seeds
and weights of the fp32 gradient goldens, and record how far THAT run is from the fp32 goldens, with the metrics of
tests/test_gpu_backward.py::test_training_step_16bit_mode.  The GPU test then demands that this repo's 16-bit mode is no
further from the reference's fp32 autograd than the reference's own bf16 run.

Run in the build container only:  ``python tests/golden/make_bf16_anchor.py``  ->  tests/golden/c5_train_bf16_anchor.json
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from helpers import VOLUME_KEYS, golden_volume_grad, grad_rel_err  # noqa: E402
from make_golden import GRAD_CASES, training_loss  # noqa: E402
from ref_harness import build_reference_model  # noqa: E402
from uforecon_amd.scene import frame_digest, make_frame  # noqa: E402


def run(name, c):
    g = np.load(os.path.join(HERE, f"{name}.npz"))
    model = build_reference_model(int(g["weight_seed"]), test_sample_coarse=c["coarse"], test_sample_fine=c["fine"],
                                  coarse_sample=c["coarse"], fine_sample=c["fine"], test_n_view=c["NV"],
                                  extract_geometry=False)
    model.train()
    fr = make_frame(c["H"], c["W"], c["NV"], c["seed"], train_layout=True)
    assert float(g["input_digest"]) == frame_digest(fr)
    idx = torch.from_numpy(g["ray_idx"])
    vols = {}
    for st in ("stage1", "stage2", "stage3"):
        for k in ("feature_volume", "weight_volume"):
            fr.feature_volume[st][k].requires_grad_(True)
            vols[f"{st}.{k}"] = fr.feature_volume[st][k]
    torch.manual_seed(int(g["sampler_seed"]))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        r = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat,
                        feature_volume=fr.feature_volume, match_feature=fr.match_feature)
        loss = training_loss([t.float() if torch.is_tensor(t) else t for t in r], fr.batch)
    loss.backward()
    out = {"loss_rel": abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))}
    fwd = {}
    for n, i in dict(rgb=1, depth=2, rgb_2=8, depth_2=9).items():
        fwd[n] = grad_rel_err(r[i].detach().float(), g[n])
    out["forward_rows"] = fwd
    worst, dot, na, nb = {}, 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        if not k.startswith(("ray_transformer.", "deviation_network.")):
            continue
        ref = torch.from_numpy(g["grad." + k]).reshape(p.grad.shape)
        worst[k] = grad_rel_err(p.grad.float(), ref)
        pg = p.grad.double()
        dot += float((pg * ref.double()).sum()); na += float((pg * pg).sum()); nb += float((ref.double() ** 2).sum())
    out["parameter_cosine"] = dot / (na * nb) ** 0.5
    out["parameter_worst"] = max(worst.values())
    out["parameter_worst_name"] = max(worst, key=worst.get)
    vc, vw = {}, {}
    for key in VOLUME_KEYS:
        v = vols[key]
        ref = torch.as_tensor(golden_volume_grad(g, key, v.shape)).reshape(v.grad.shape).double()
        vg = v.grad.double()
        vw[key] = grad_rel_err(vg, ref)
        vc[key] = float((vg * ref).sum() / ((vg * vg).sum() * (ref * ref).sum()).sqrt())
    out["volume_cosine_min"] = min(vc.values())
    out["volume_worst"] = max(vw.values())
    out["volume_cosine"] = vc
    return out


def main():
    res = {"_": "errors of the reference's own bf16-autocast training step against its fp32 autograd goldens "
                "(tests/golden/make_bf16_anchor.py); torch " + torch.__version__}
    for name, c in GRAD_CASES.items():
        res[name] = run(name, c)
        print(name, json.dumps(res[name], indent=1))
    with open(os.path.join(HERE, "c5_train_bf16_anchor.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
