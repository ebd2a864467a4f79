"""Round-6 probe: ns per point and CU of the weight-stationary pipeline skeleton (tools/dev/ws_probe.hip) against the
shipped view transformer (0.365-0.38 ms per 262 144 points = 1 024 points per CU: 356-371 ns per point and CU)."""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "ws_probe.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(here, "ws_probe.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.ws_run.restype = C.c_float
out = torch.empty(256 * 512, device="cuda")
for stagger in (0, 1):
  for steps in (128, 512, 2048):
    ms = lib.ws_run(256, steps, C.c_void_p(out.data_ptr()), stagger)
    torch.cuda.synchronize()
    print(f"stagger {stagger} steps {steps}: {ms:.3f} ms -> {ms * 1e6 / steps:.0f} ns per step (8 points) = {ms * 1e6 / steps / 8:.0f} ns per point and CU")
