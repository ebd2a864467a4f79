"""Development A/B (same box): the two frustum U-Nets with the stride-1 layers on the plane kernels vs on the fp32 kernels,
at the three stages' sizes (512x640, 3 views).  Wall time of back-to-back calls (what encode_frame pays) per net and stage."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import cascade, unet3d  # noqa: E402
from uforecon_amd.scene import fill_state_dict  # noqa: E402


def wall(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    dev = "cuda:0"
    crn = fill_state_dict(cascade.CostRegNet(1, 8), 1).eval().to(dev)
    crw = fill_state_dict(cascade.CostRegNetWeight(1, 8), 2).eval().to(dev)
    tot = {True: 0.0, False: 0.0}
    for stage, (D, H, W) in (("stage1", (48, 128, 160)), ("stage2", (32, 256, 320)), ("stage3", (8, 512, 640))):
        x = torch.rand(3, 1, D, H, W, device=dev)
        for planes in (False, True):
            with torch.no_grad():
                a = wall(lambda: unet3d.cost_reg_net(crn, x, planes=planes))
                b = wall(lambda: unet3d._cost_reg_net_weight_hip(crw, x, planes=planes))
            tot[planes] += a + b
            print(f"{stage} planes={planes!s:5}: CostRegNet {a:6.3f} ms  CostRegNetWeight {b:6.3f} ms")
    print(f"all stages, both nets: fp32 kernels {tot[False]:.2f} ms, planes {tot[True]:.2f} ms")


if __name__ == "__main__":
    main()
