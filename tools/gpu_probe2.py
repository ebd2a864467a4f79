"""Development probe 2: chain gather -> aggregate -> composite on the GPU and localise errors."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import CASES, case_inputs, load_weights, rel_err  # noqa: E402
from oracle import ufo_oracle as O  # noqa: E402
from uforecon_amd import ops  # noqa: E402

DEV = "cuda:0"
torch.set_printoptions(precision=6, linewidth=200)


def main():
    W = ops.PackedWeights({k: v.to(DEV) for k, v in load_weights().items()})
    for name in sys.argv[1:] or ["c1_coarse_only", "c4_nv5_128"]:
        fr, idx, U1, U2, g = case_inputs(name)
        want = {}
        with torch.no_grad():
            O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2,
                    want=want, coarse_only=CASES[name].get("coarse_only", False))
        w = want["coarse"]
        f = fr.to(DEV)
        fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
        i = idx.reshape(-1)
        ray_d = fr.batch["ray_d"][0][:, i].t().contiguous().to(DEV)
        ray_o = fr.batch["ray_o"][0].contiguous().to(DEV)
        RN, SN = w["z"].shape
        NV = fh.NV
        x, rgbm, dirs, dbg = ops.project_gather(fh, W, ray_o, ray_d, w["z"].to(DEV).contiguous(), debug=True)
        mask_ref = w["mask"].permute(1, 2, 0).reshape(-1, NV)
        bad = (rgbm[..., 3].cpu() != mask_ref)
        print(f"==== {name}: mask mismatches {int(bad.sum())} of {bad.numel()}; masked-out fraction {float((mask_ref == 0).float().mean()):.4f}")
        if bad.any():
            p, v = bad.nonzero()[0].tolist()
            xy = w["xy"].reshape(NV, -1, 2)[v, p]
            print("   first mismatch point", p, "view", v, "xy ref", xy, "xy gpu", dbg["xy"][v, p].cpu(), "mask_z", w["mask_z"].reshape(NV, -1)[v, p])
        print("   x tokens", rel_err(x, w["x"]), " rgb", rel_err(rgbm[..., :3], w["rgb_s"]))
        rad, srdf, _ = ops.aggregate(W, x, rgbm, dirs, RN, SN)
        er = (rad.cpu() - w["radiance"]).abs().max(dim=1)[0]
        print("   radiance maxabs", float(er.max()), " srdf rel", rel_err(srdf, w["srdf"]))
        worst = int(er.argmax())
        print("   worst point", worst, "gpu", rad[worst].cpu(), "ref", w["radiance"][worst], "mask", mask_ref[worst],
              "blend_w", w["blend_w"][worst].reshape(-1), "logit", w["logit"][worst].reshape(-1))
        # with the oracle's own inputs
        rgbm2 = torch.cat([w["rgb_s"], mask_ref[..., None]], -1).to(DEV).contiguous()
        dirs2 = torch.cat([w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3), torch.zeros(RN * SN, NV, 1)], -1).to(DEV).contiguous()
        rad2, srdf2, _ = ops.aggregate(W, w["x"].to(DEV).contiguous(), rgbm2, dirs2, RN, SN)
        print("   (oracle inputs) radiance maxabs", float((rad2.cpu() - w["radiance"]).abs().max()), " srdf rel", rel_err(srdf2, w["srdf"]))
        rgb, depth, op, wt = ops.composite(w["z"].to(DEV).contiguous(), rad.reshape(RN, SN, 3), srdf, W.variance)
        print("   chain: depth rel", rel_err(depth, w["depth"]), " rgb maxabs", float((rgb.cpu() - w["rgb"]).abs().max()))


if __name__ == "__main__":
    main()
