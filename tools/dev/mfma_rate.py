import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_rate.so"))
out = torch.zeros(4, dtype=torch.int64, device="cuda"); sink = torch.empty(256 * 256, device="cuda")
iters = 2000
lib.run(C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()), iters)
o = out.cpu().tolist()
for k, name in enumerate(("16x16x32_bf16", "16x16x16bf16_1k", "32x32x16_bf16")):
    print(f"{name}: {o[k] / (iters * 32):.2f} cycles per MFMA (one wave per SIMD)")
