"""Development probe: per-phase cycle counts of view_bwd_kernel (needs lib/libufr_timing.so: python -m uforecon_amd.build
--variant timing -DUFR_BWD_TIMING; run with UFR_LIB pointing at it)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from uforecon_amd import _lib, ops
import test_gpu_backward as T

fr, P, W, fh, ray_o, ray_d, z, x, rgbm, dirs, dbg = T._token_inputs("c5_train_grads")
RN, SN = z.shape
# replicate the rays to get a realistic launch (many tiles per workgroup)
rep = 32
x, rgbm, dirs = x.repeat(rep, 1, 1), rgbm.repeat(rep, 1, 1), dirs.repeat(rep, 1, 1)
RN *= rep
radiance, srdf, agg = ops.aggregate(W, x, rgbm, dirs, RN, SN, keep_workspace=True)
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
lib.ufr_debug_vb_phases(buf, 64, 1)
grads = ops.GradBuffer("cuda:0")
d_rad, d_srdf = torch.rand(RN * SN, 3, device="cuda:0"), torch.rand(RN, SN, device="cuda:0")
torch.cuda.synchronize()
import time
t = time.perf_counter()
ops.aggregate_bwd(W, grads, x, rgbm, dirs, agg["token0"], RN, SN, d_rad, d_srdf)
torch.cuda.synchronize()
print("aggregate_bwd ms", (time.perf_counter() - t) * 1e3, "points", RN * SN)
lib.ufr_debug_vb_phases(buf, 64, 0)
tot = sum(buf)
names = {0: "P0 commit", 1: "P1 qkv", 2: "P2 attn", 3: "P3 merge", 4: "P4 LN1", 5: "P5 mlp0", 6: "P6 mlp2", 7: "P7 LN2", 8: "P8 rw0",
         9: "P9 h2", 11: "softmax", 12: "B1a dh2", 13: "B1b dh1", 14: "B2 dy + wg rw0", 15: "B3 LN2b", 16: "W1 wg mlp2",
         17: "B4 dhid", 18: "B5 dcat + wg mlp0", 19: "B6 LN1b", 20: "B7 dmsg + wg merge", 21: "B8 attn q", 22: "B9 attn kv",
         23: "B10 tail (barrier)", 24: "B11 out", 30: "B10 fetch", 31: "B10 gemm q", 32: "B10 gemm k v", 33: "B10 wgrad qkv",
         34: "B5 gemm", 35: "B5 wgrad", 18: "B5 tail (barrier)"}
n_tiles = (RN * SN + 7) // 8 // 256      # 32-token tiles (8 points at NV = 3) per workgroup
for i, v in enumerate(buf):
    if v:
        print(f"{i:2d} {names.get(i, ''):22s} {v:12d} {100.0 * v / tot:5.1f}%  {v / max(n_tiles, 1):9.0f} cycles/tile")
print("total cycles per tile", tot / max(n_tiles, 1))

# ---- ray_bwd_kernel (markers are numbered in source order; the line printed is the marker's line in csrc/ray_bwd.hip)
if hasattr(lib, "ufr_debug_rb_phases"):
    import re
    rb = (C.c_ulonglong * 64)()
    lib.ufr_debug_rb_phases(rb, 64, 0)
    src = open(os.path.join(ROOT, "uforecon_amd", "csrc", "ray_bwd.hip")).read().splitlines()
    where = {int(m.group(1)): i + 1 for i, l in enumerate(src) for m in [re.search(r"UFR_BWD_PHASE\(g_rb_phase, (\d+)\)", l)] if m}
    tot = sum(rb)
    rays_per_wg = max(RN // 256, 1)
    print(f"ray_bwd: {tot / rays_per_wg:.0f} cycles per ray (SN = {SN}), workgroup 0")
    for i, v in enumerate(rb):
        if v:
            ln = where.get(i, 0)
            ctx = next((src[k].strip() for k in range(ln, min(ln + 6, len(src))) if src[k].strip() and "UFR_BWD_PHASE" not in src[k]), "")
            print(f"{i:2d} line {ln:4d} {v / rays_per_wg:10.0f} cyc/ray {100.0 * v / tot:5.1f}%   next: {ctx[:90]}")
