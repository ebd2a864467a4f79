"""Development probe: per-layer time of ufr_conv3d / _bwd_data / _bwd_weight at the three stages' sizes (512x640, 3 views)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops  # noqa: E402

S1, S2, T2 = ops.CONV3D_S1, ops.CONV3D_S2, ops.CONV3D_T2
LAYERS = [("conv0", S1, 1, 8, 1), ("conv1", S2, 8, 16, 1), ("conv2", S1, 16, 16, 2), ("conv3", S2, 16, 32, 2), ("conv4", S1, 32, 32, 4),
          ("conv5", S2, 32, 64, 4), ("conv6", S1, 64, 64, 8), ("conv7", T2, 64, 32, 8), ("conv9", T2, 32, 16, 4), ("conv11", T2, 16, 8, 2),
          ("features", S1, 8, 8, 1), ("weights", S1, 8, 1, 1)]


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = "cuda:0"
    tot = dict(f=0.0, d=0.0, w=0.0)
    for stage, (D, H, W) in (("stage1", (48, 128, 160)), ("stage2", (32, 256, 320)), ("stage3", (8, 512, 640))):
        for name, mode, cin, cout, down in LAYERS:
            d, h, w = D // down, H // down, W // down
            x = torch.randn(3, d, h, w, cin, device=dev)
            wt = torch.randn((cin, cout, 3, 3, 3) if mode == T2 else (cout, cin, 3, 3, 3), device=dev) * 0.1
            if cout % 4:
                y_shape = (3, d, h, w, cout)
                dy = torch.randn(y_shape, device=dev)
                tf = float("nan")
            else:
                y = ops.conv3d(x, wt, mode)
                dy = torch.randn_like(y)
                tf = t(lambda: ops.conv3d(x, wt, mode))
            tw = t(lambda: ops.conv3d_bwd_weight(x, dy, mode, wt.shape))
            td = t(lambda: ops.conv3d_bwd_data(dy, wt, mode, tuple(x.shape))) if cin > 1 else float("nan")
            print(f"{stage} {name:9s} {cin:3d}->{cout:3d} mode {mode} @ {d}x{h}x{w}: fwd {tf:7.3f}  dgrad {td:7.3f}  wgrad {tw:7.3f} ms")
            tot["f"] += 0 if tf != tf else tf
            tot["d"] += 0 if td != td else td
            tot["w"] += tw
    print("totals (ms):", tot)


if __name__ == "__main__":
    main()
