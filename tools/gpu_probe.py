"""Development probe (GPU box): per-row error table against the oracle + a first timing.

Not part of the product or the test-suite; prints instead of asserting so that one gpurun call
shows every stage's error at once.  Usage: python tools/gpu_probe.py [--time]
"""
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import CASES, case_inputs, load_weights, max_rel_elem, rel_err  # noqa: E402
from oracle import ufo_oracle as O  # noqa: E402
from uforecon_amd import ops  # noqa: E402
from uforecon_amd.scene import make_frame, sampler_uniforms  # noqa: E402

DEV = "cuda:0"


def rows(name):
    fr, idx, U1, U2, g = case_inputs(name)
    want = {}
    with torch.no_grad():
        O.infer(load_weights(), fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2,
                want=want, coarse_only=CASES[name].get("coarse_only", False))
    return fr, idx, U1, U2, g, want


def step(label, fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        print(f"[FAIL] {label}: {type(e).__name__}: {e}")
        traceback.print_exc(limit=3)


def main():
    print("device:", torch.cuda.get_device_name(0))
    W = ops.PackedWeights({k: v.to(DEV) for k, v in load_weights().items()})
    for name in ("rows_small", "c4_nv5_128"):
        fr, idx, U1, U2, g, want = rows(name)
        f = fr.to(DEV)
        fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
        i = idx.reshape(-1)
        ray_d = fr.batch["ray_d"][0][:, i].t().contiguous().to(DEV)
        ray_o = fr.batch["ray_o"][0].contiguous().to(DEV)
        cz = fr.batch["cam_ray_d"][0][2, i]
        near = (fr.batch["near_fars"][0, 0, 0] / cz).contiguous().to(DEV)
        far = (fr.batch["near_fars"][0, 0, 1] / cz).contiguous().to(DEV)
        print(f"==== {name}")

        def samp():
            z = ops.sample_fixed(near, far, U1.to(DEV))
            print("  sample_fixed  maxabs", float((z.cpu() - want["coarse"]["z"]).abs().max()))
            zf, za = ops.sample_importance_merge(want["coarse"]["weight"].to(DEV).contiguous(), want["coarse"]["z"].to(DEV).contiguous(), U2.to(DEV))
            print("  importance    z_fine", rel_err(zf, want["fine"]["z_fine"]), " z_all", rel_err(za, want["fine"]["z"]))
        step("samplers", samp)

        for tag in ("coarse", "fine"):
            w = want[tag]
            RN, SN = w["z"].shape
            NV = fh.NV

            def gath():
                x, rgbm, dirs, dbg = ops.project_gather(fh, W, ray_o, ray_d, w["z"].to(DEV).contiguous(), debug=True)
                print(f"  [{tag}] gather  xy", rel_err(dbg["xy"].reshape(NV, RN, SN, 2), w["xy"]),
                      " sim8", rel_err(dbg["sim8"].reshape(RN, SN, 8), w["sim8"]),
                      " vol24", rel_err(dbg["vol24"].reshape(RN, SN, 24), w["vol24"]))
                xe = (x.cpu() - w["x"]).abs()
                print(f"  [{tag}] tokens  feat", float(xe[..., :32].max()), " vol", float(xe[..., 32:56].max()),
                      " sim16", float(xe[..., 56:72].max()), " pe", float(xe[..., 72:].max()))
                print(f"  [{tag}] rgb", rel_err(rgbm[..., :3], w["rgb_s"]), " mask_eq",
                      bool(torch.equal(rgbm[..., 3].cpu(), w["mask"].permute(1, 2, 0).reshape(-1, NV))),
                      " dir", rel_err(dirs[..., :3], w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3)))
            step(f"gather {tag}", gath)

            def aggr():
                x = w["x"].to(DEV).contiguous()
                rgbm = torch.cat([w["rgb_s"], w["mask"].permute(1, 2, 0).reshape(-1, NV, 1)], -1).to(DEV).contiguous()
                dirs = torch.cat([w["dirs"].permute(1, 2, 0, 3).reshape(-1, NV, 3), torch.zeros(RN * SN, NV, 1)], -1).to(DEV).contiguous()
                rad, srdf, dbg = ops.aggregate(W, x, rgbm, dirs, RN, SN, debug=True)
                vo = dbg["view_out"].cpu()
                print(f"  [{tag}] view_out all", rel_err(vo, w["view_out"]), " tok0", rel_err(vo[:, 0], w["view_out"][:, 0]))
                print(f"  [{tag}] ray_out", rel_err(dbg["ray_out"].reshape(RN, SN, 88), w["ray_out"]),
                      " srdf", rel_err(srdf, w["srdf"]), " radiance", rel_err(rad, w["radiance"]))
            step(f"aggregate {tag}", aggr)

            def comp():
                rgb, depth, op, wt = ops.composite(w["z"].to(DEV).contiguous(), w["radiance"].reshape(RN, SN, 3).to(DEV).contiguous(),
                                                   w["srdf"].to(DEV).contiguous(), W.variance)
                print(f"  [{tag}] composite weight", rel_err(wt, w["weight"]), " depth", rel_err(depth, w["depth"]), " rgb", rel_err(rgb, w["rgb"]))
            step(f"composite {tag}", comp)

    for name in ("c1_coarse_only", "c2_hier_small", "c4_nv5_128", "c2_hier_512x640"):
        def e2e():
            c = CASES[name]
            fr, idx, U1, U2, g = case_inputs(name)
            f = fr.to(DEV)
            fh = ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
            out = ops.render_rays(fh, W, idx.to(DEV), U1.to(DEV), U2.to(DEV), coarse_only=c.get("coarse_only", False))
            torch.cuda.synchronize()
            print(f"==== e2e {name}: depth max-rel", max_rel_elem(out["depth"], torch.from_numpy(g["depth"]), 1e-3),
                  " rgb max-rel(floor .05)", max_rel_elem(out["rgb"], torch.from_numpy(g["rgb"]), 0.05),
                  " srdf", rel_err(out["srdf"], g["srdf"]))
        step(f"e2e {name}", e2e)

    if "--time" in sys.argv:
        H, Wd, NV = 512, 640, 3
        fr = make_frame(H, Wd, NV, 0).to(DEV)
        fh = ops.FrameHandle(fr.batch, fr.source_imgs_feat, fr.feature_volume, fr.match_feature)
        for RN in (4096, 32768):
            idx = torch.arange(RN, device=DEV, dtype=torch.int64) + 100 * Wd
            U1, U2 = sampler_uniforms(1, 64, 64, RN)
            U1, U2 = U1.to(DEV), U2.to(DEV)
            ws = ops.RenderWorkspace(DEV, 64, 64, NV)
            ops.render_rays(fh, W, idx, U1, U2, workspace=ws, want_srdf=False)
            torch.cuda.synchronize()
            ops.profile_enable(True)
            t = time.time()
            ops.render_rays(fh, W, idx, U1, U2, workspace=ws, want_srdf=False)
            torch.cuda.synchronize()
            dt = time.time() - t
            prof = ops.profile_read()
            ops.profile_enable(False)
            print(f"==== timing RN={RN}: {dt * 1e3:.2f} ms -> {RN / dt:.0f} rays/s")
            for k, v in prof.items():
                print(f"      {k:18s} {v['ms']:9.3f} ms in {v['launches']} launches")


if __name__ == "__main__":
    main()
