"""Development probe: per-entry-point time inside UFOReconInference.encode_frame (512x640, 3 views) from the library's own
HIP-event scopes, and what is left for torch ops."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from uforecon_amd import ops, pipeline  # noqa: E402
from uforecon_amd.scene import fill_state_dict, make_frame  # noqa: E402


def main():
    dev, NV, H, W = "cuda:0", 3, 512, 640
    fr = make_frame(H, W, NV, seed=0).to(dev)
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
    with torch.no_grad():
        for _ in range(2):
            net.encode_frame(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            net.encode_frame(batch)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / 3 * 1e3
        ops.profile_enable(True)
        net.encode_frame(batch)
        torch.cuda.synchronize()
        pr = ops.profile_read()
        ops.profile_enable(False)
    print(f"encode_frame {total:.2f} ms")
    acc = 0.0
    for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"  {k:28s} {v['launches']:4d} calls {v['ms']:8.3f} ms")
        acc += v["ms"]
    print(f"  library entry points {acc:.2f} ms; the rest (torch ops, gaps) {total - acc:.2f} ms")


if __name__ == "__main__":
    main()
