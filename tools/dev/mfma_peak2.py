import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_peak2.so")); lib.run2.restype = C.c_float
out = torch.empty(4096 * 512, device="cuda")
w = torch.randn(33 * 64 * 64 * 4 + 1024, device="cuda")
for blocks in (256,):
    for mode, name in ((0, "L2 shared 64KB"), (4, "L2 per-CU regions (2 MB)"), (3, "LDS"), (2, "no loads")):
        iters = 500
        ms = lib.run2(blocks, iters, mode, C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()))
        n = iters * 16 * (4 if mode == 1 else 8)
        flops = blocks * 8 * n * 2048.0
        print(f"blocks {blocks} ({blocks*8/1024:.0f} waves/SIMD) {name}: {ms:.3f} ms {flops/ms/1e9:.1f} TFLOP/s")
