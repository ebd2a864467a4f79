import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_valu2.so")); lib.run4.restype = C.c_float
out = torch.empty(4096 * 256, device="cuda")
for blocks in (256, 512):
    for kind, kn in ((0, "valu only"), (1, "f32 mfma"), (2, "bf16 mfma")):
        for nf in (0, 2, 4, 8):
            if kind == 0 and nf == 0: continue
            ms = lib.run4(blocks, 2000, kind, nf, C.c_void_p(out.data_ptr()))
            print(f"waves/SIMD {blocks//256}  {kn:10s} + {nf} fma per 2 mfma: {ms:7.3f} ms")
