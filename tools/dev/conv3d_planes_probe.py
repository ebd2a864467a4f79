"""Development probe: ufr_conv3d_planes against ufr_conv3d / _bwd_data, per layer, at the three stages' sizes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops  # noqa: E402


def t(fn, reps=5):
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
    for stage, (D, H, W) in (("stage1", (48, 128, 160)), ("stage2", (32, 256, 320)), ("stage3", (8, 512, 640))):
        for name, cin, cout, down in (("conv1", 8, 16, 1), ("conv3", 16, 32, 2), ("conv5", 32, 64, 4)):
            d, h, w = D // down, H // down, W // down
            x = torch.randn(3, d, h, w, cin, device=dev)
            wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
            am = ops.absmax(x)
            t32 = t(lambda: ops.conv3d(x, wt, ops.CONV3D_S2))
            t16 = t(lambda: ops.conv3d_planes(x, am, wt, mode=ops.CONV3D_S2))
            print(f"{stage} {name:9s} {cin:3d}->{cout} s2 @ {d}x{h}x{w}: fp32 {t32:7.3f}  planes {t16:7.3f} ms")
        for name, cin, cout, down in (("conv7", 64, 32, 8), ("conv9", 32, 16, 4), ("conv11", 16, 8, 2)):
            d, h, w = D // down, H // down, W // down
            x = torch.randn(3, d, h, w, cin, device=dev)
            wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.1
            sk = torch.randn(3, 2 * d, 2 * h, 2 * w, cout, device=dev)
            am = ops.absmax(x)
            t32 = t(lambda: ops.conv3d(x, wt, ops.CONV3D_T2, skip=sk))
            t16 = t(lambda: ops.conv3d_planes(x, am, wt, skip=sk, mode=ops.CONV3D_T2))
            print(f"{stage} {name:9s} {cin:3d}->{cout} t2 @ {d}x{h}x{w}: fp32 {t32:7.3f}  planes {t16:7.3f} ms")
        for name, cin, cout, cout2, down in (("heads", 8, 8, 1, 1), ("features", 8, 8, 0, 1), ("conv2", 16, 16, 0, 2), ("conv4", 32, 32, 0, 4),
                                            ("conv6", 64, 64, 0, 8)):
            d, h, w = D // down, H // down, W // down
            x = torch.randn(3, d, h, w, cin, device=dev)
            wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
            w2 = torch.randn(cout2, cin, 3, 3, 3, device=dev) * 0.1 if cout2 else None
            am = ops.absmax(x)
            t32 = t(lambda: ops.conv3d(x, wt, ops.CONV3D_S1, out_ncdhw=bool(cout2), weight2=w2))
            t16 = t(lambda: ops.conv3d_planes(x, am, wt, out_ncdhw=bool(cout2), weight2=w2))
            tm = t(lambda: ops.absmax(x))
            print(f"{stage} {name:9s} {cin:3d}->{cout}+{cout2} @ {d}x{h}x{w}: fp32 {t32:7.3f}  planes {t16:7.3f}  absmax(x) {tm:6.3f} ms")


if __name__ == "__main__":
    main()
