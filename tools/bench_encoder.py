"""Per-frame producers at 512x640, 3 views (the part of a frame that is replicated on every rank when rays are sharded):
FeatureNet (library convolutions + HIP deformable convolution), FMT, the frustum cascade (HIP correlate kernel + HIP 3-D
U-Nets, csrc/conv3d.hip) and the matching features.  Prints the time of `UFOReconInference.encode_frame` and of its parts."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uforecon_amd import pipeline  # noqa: E402
from uforecon_amd.scene import fill_state_dict, make_frame  # noqa: E402


def measure(height: int = 512, width: int = 640, reps: int = 3, tune: bool = False, verbose: bool = False):
    a = argparse.Namespace(height=height, width=width, reps=reps, tune=tune)
    torch.backends.cudnn.benchmark = a.tune
    dev, NV = "cuda:0", 3
    fr = make_frame(a.height, a.width, NV, seed=0).to(dev)
    batch = fr.batch
    pm = {}
    for st, s in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        p = torch.zeros(1, NV, 2, 4, 4, device=dev)
        p[0, :, 0] = batch["w2cs"][0, :NV]
        K = batch["intrinsics"][0, :NV].clone()
        K[:, :2] = K[:, :2] / s
        p[0, :, 1, :3, :3] = K
        p[0, :, 1, 3, 3] = 1.0
        pm[st] = p
    batch["proj_matrices"] = pm
    near, far = float(batch["near_fars"][0, 0, 0]), float(batch["near_fars"][0, 0, 1])
    batch["depth_values_org_scale"] = torch.linspace(near, far, 48, device=dev)[None]
    args = argparse.Namespace(extract_geometry=True, test_sample_coarse=64, test_sample_fine=64, coarse_sample=64, fine_sample=64,
                              volume_type="correlation", volume_reso=96, mvs_depth_guide=1, depth_pos_encoding=True,
                              use_dir_srdf=False, explicit_similarity=True, test_coarse_only=False, test_ray_num=800,
                              test_n_view=NV, out_dir=None)
    net = fill_state_dict(pipeline.UFOReconInference(args), 21).eval().to(dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3, out

    with torch.no_grad():
        imgs, pmr, dv = net.build_pairs(batch["source_imgs"], batch["proj_matrices"], batch["depth_values_org_scale"])
        t_fn, feats = timed(lambda: [net.transmvsnet.feature(imgs[:, v]) for v in range(NV)])
        from uforecon_amd import ops
        ops.profile_enable(True)
        [net.transmvsnet.feature(imgs[:, v]) for v in range(NV)]
        torch.cuda.synchronize()
        pr = ops.profile_read()
        ops.profile_enable(False)
        dcn = pr.get("deform_conv2d", dict(ms=0.0, launches=0))
        if verbose:
            print(f"   of FeatureNet: {dcn['launches']} deformable convolutions, {dcn['ms']:.1f} ms in the HIP kernel (+ re-layout)")
        t_fmt, feats2 = timed(lambda: net.transmvsnet.encode([dict(f) for f in feats], ref_idx=0))
        t_cas, info = timed(lambda: net.transmvsnet(feats2, pmr, dv, (a.height, a.width)))
        t_vol, _ = timed(lambda: [net.feature_volume(batch, info[st]["cost_volume"]) for st in ("stage1", "stage2", "stage3")])
        t_all, _ = timed(lambda: net.encode_frame(batch))
    if verbose:
        print(f"{a.height}x{a.width}, {NV} views ({NV} rotations): FeatureNet {t_fn:.1f} ms, FMT {t_fmt:.1f} ms, cascade (3 stages: "
              f"correlate + PixelwiseNet + CostRegNet) {t_cas:.1f} ms, CostRegNetWeight x3 {t_vol:.1f} ms; encode_frame total {t_all:.1f} ms")
    del net, fr, batch
    torch.cuda.empty_cache()
    return dict(height=a.height, width=a.width, views=NV, encode_frame_ms=t_all,
                parts_ms_timed_separately=dict(featurenet_with_the_references_redundancy=t_fn, fmt=t_fmt,
                                               cascade_correlate_pixelwise_costreg=t_cas, cost_reg_net_weight_x3=t_vol),
                deformable_convolution_ms=dcn["ms"], miopen_algorithm_search=bool(a.tune))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tune", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen algorithm search)")
    a = ap.parse_args()
    measure(a.height, a.width, a.reps, a.tune, verbose=True)


if __name__ == "__main__":
    main()
