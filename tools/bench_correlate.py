"""Micro-benchmark of the correlation-volume step at the three cascade stages of a 512x640, 3-view frame
(TransMVSNet.py:125: D = 48/32/8 at 1/4, 1/2, 1/1 resolution with 32/16/8 channels).  Prints one line per stage:
launch time (HIP events via ufr_profile_*), algorithmic bytes and the fraction of the HBM roof.  (The CPU side of the
comparison is timed by tests/test_gpu_frustum.py::test_correlate_cpu_vs_gpu_timing, the only place allowed to run the oracle.)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uforecon_amd import frustum, ops  # noqa: E402
from uforecon_amd.scene import make_correlate_case  # noqa: E402

DEV = "cuda:0"
STAGES = [("stage1", 32, 128, 160, 48), ("stage2", 16, 256, 320, 32), ("stage3", 8, 512, 640, 8)]


def measure(reps: int = 20, verbose: bool = False):
    """-> one dict per cascade stage: launch time, algorithmic bytes, achieved GB/s, fraction of the 8 TB/s roof"""
    rows = []
    a = argparse.Namespace(reps=reps)
    for name, C, H, W, D in STAGES:
        c = make_correlate_case("custom", C=C, H=H, W=W, D=D, NV=3, seed=7)
        args = (c["ref_fea"].to(DEV), torch.stack(c["src_feas"]).to(DEV), c["ref_proj_pair"], c["src_proj_pairs"],
                c["depth_values"].to(DEV), c["view_weights"].to(DEV))
        frustum.correlate(*args)
        torch.cuda.synchronize()
        ops.profile_enable(True)
        for _ in range(a.reps):
            frustum.correlate(*args, want_similarity=False)
        torch.cuda.synchronize()
        p = ops.profile_read()["correlate"]
        ops.profile_enable(False)
        ms = p["ms"] / p["launches"]
        NS = 2
        # algorithmic HBM bytes: read every feature map once in its original layout and once channel-last, write the
        # channel-last copy, read hypotheses + weights, write the aggregate.  The reference moves >= 2 x C*D*H*W*4 per view.
        feat = (1 + NS) * C * H * W * 4
        algo = 3 * feat + D * H * W * 4 * 2 + NS * H * W * 4
        ref_bytes = NS * 2 * C * D * H * W * 4
        rows.append(dict(stage=name, channels=C, height=H, width=W, depth_hypotheses=D, ms_per_launch=ms,
                         algorithmic_bytes=algo, achieved_gbps=algo / ms / 1e6, hbm_frac=algo / ms / 1e6 / 8000,
                         samples_per_s=NS * D * H * W / ms * 1e3, reference_warped_volume_bytes=ref_bytes))
        if verbose:
            print(f"{name}: C={C} {H}x{W} D={D}: {ms * 1e3:8.1f} us/launch  {algo / ms / 1e6:7.1f} GB/s algorithmic "
                  f"({algo / 1e6:.1f} MB; {algo / ms / 1e6 / 8000:.1%} of the 8 TB/s roof); samples/s "
                  f"{NS * D * H * W / ms / 1e6:.2f} G; warped volume the reference materialises: {ref_bytes / 1e6:.0f} MB")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    measure(ap.parse_args().reps, verbose=True)


if __name__ == "__main__":
    main()
