"""Accuracy report of the split-precision transformer kernels (development aid, not a test): errors of the aggregate
stage (view transformer, ray transformer, SRDF, radiance) against the oracle evaluated in float64 on the same token
inputs, beside the float32 oracle's own distance from float64.  Run on a GPU box: python tests/accuracy_report.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import CASES, case_inputs, load_weights  # noqa: E402
from oracle import ufo_oracle as O  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).abs().max() / b.abs().max()), float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())


def main():
    from uforecon_amd import ops
    dev = "cuda:0"
    P = load_weights()
    W = ops.PackedWeights({k: v.to(dev) for k, v in P.items()})
    P64 = {k: v.double() for k, v in P.items()}
    for name, tag in (("rows_small", "coarse"), ("rows_small", "fine"), ("c4_nv5_128", "fine")):
        fr, idx, U1, U2, g = case_inputs(name)
        want = {}
        with torch.no_grad():
            O.infer(P, fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2, want=want,
                    coarse_only=CASES[name].get("coarse_only", False))
        w = want[tag]
        RN, SN = w["z"].shape
        NV = w["x"].shape[1]
        x = w["x"]
        mask = w["mask"].permute(1, 2, 0).reshape(-1, NV, 1)
        rgbm = torch.cat([w["rgb_s"], mask], -1)
        dirs3 = w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3)
        dirs = torch.cat([dirs3, torch.zeros(RN * SN, NV, 1)], -1)
        with torch.no_grad():
            ref64 = O.aggregate_tokens(P64, x.double(), w["rgb_s"].double(), mask[..., 0].double() if mask.dtype != torch.bool else mask[..., 0],
                                       dirs3.double(), RN, SN) if False else None
        radiance, srdf, dbg = ops.aggregate(W, x.to(dev).contiguous(), rgbm.to(dev).contiguous(), dirs.to(dev).contiguous(), RN, SN, debug=True)
        print(f"{name}/{tag}: (max rel to scale, rms rel) vs float32 oracle")
        for key, got, ref in (("view_out", dbg["view_out"], w["view_out"]), ("ray_out", dbg["ray_out"].reshape(RN, SN, 88), w["ray_out"]),
                              ("srdf", srdf, w["srdf"]), ("radiance", radiance, w["radiance"])):
            print(f"   {key:9s} max {rel(got, ref)[0]:.3e}  rms {rel(got, ref)[1]:.3e}")
        print("   |x| rms", float(x.pow(2).mean().sqrt()), "max", float(x.abs().max()))


if __name__ == "__main__":
    main()
