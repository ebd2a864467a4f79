import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_peak.so")); lib.run.restype = C.c_float
out = torch.empty(4096 * 256, device="cuda")
for blocks in (256, 512, 1024, 2048):
    for ch in (1, 2, 4):
        iters = 4000
        ms = lib.run(blocks, iters, ch, C.c_void_p(out.data_ptr()))
        n_mfma_per_wave = iters * 8 * ch
        flops = blocks * 4 * n_mfma_per_wave * 2048.0
        waves_per_simd = blocks * 4 / 1024
        cyc = n_mfma_per_wave * max(1.0, waves_per_simd) * 32
        print(f"blocks {blocks} chains {ch}: {ms:.3f} ms {flops/ms/1e9:.1f} TFLOP/s  -> implied clock {cyc/ms/1e6:.3f} GHz if issue-bound")
