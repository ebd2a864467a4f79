"""Development probe: where the frustum cascade's time goes besides the U-Nets (512x640, 3 views)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import cascade  # noqa: E402
from uforecon_amd.scene import fill_state_dict  # noqa: E402


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
    pw = fill_state_dict(cascade.PixelwiseNet(), 3).eval().to(dev)
    sim = torch.rand(2, 48, 128, 160, device=dev)
    with torch.no_grad():
        print(f"PixelwiseNet x2 (stage 1, one frame): {t(lambda: [pw(sim[i][None, None]) for i in range(2)]):.3f} ms")
        vw = torch.rand(2, 128, 160, device=dev)

        def agg():
            s_sum = torch.zeros_like(sim[0])
            w_sum = torch.full_like(vw[0], 1e-5)
            for i in range(2):
                s_sum = s_sum + sim[i] * vw[i].unsqueeze(0)
                w_sum = w_sum + vw[i]
            return s_sum / w_sum.unsqueeze(0)
        print(f"first-stage aggregation: {t(agg):.3f} ms")
        for name, shape in (("stage1", (1, 48, 128, 160)), ("stage2", (1, 32, 256, 320)), ("stage3", (1, 8, 512, 640))):
            cr = torch.randn(shape, device=dev)
            dv = torch.rand(shape, device=dev)

            def tail():
                pv = torch.exp(F.log_softmax(cr, dim=1))
                d = cascade.depth_wta(pv, depth_values=dv)
                return d, torch.max(pv, dim=1)[0]
            print(f"{name} softmax / wta / confidence: {t(tail):.3f} ms")
            cur = torch.rand(1, shape[2], shape[3], device=dev)
            print(f"{name} depth range samples: {t(lambda: cascade.get_cur_depth_range_samples(cur, shape[1], 1.0, cur.shape)):.3f} ms")


if __name__ == "__main__":
    main()
