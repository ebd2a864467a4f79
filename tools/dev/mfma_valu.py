import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_valu.so")); lib.run3.restype = C.c_float
out = torch.empty(4096 * 256, device="cuda")
for mode, name in ((1, "f32 mfma only"), (4, "bf16 mfma only"), (22, "valu 1/mfma"), (24, "valu 2/mfma"), (28, "valu 4/mfma"), (32, "f32+valu 1/mfma"), (34, "f32+valu 2/mfma"), (38, "f32+valu 4/mfma"), (62, "bf16+valu 1/mfma"), (64, "bf16+valu 2/mfma"), (68, "bf16+valu 4/mfma")):
    ms = lib.run3(512, 2000, mode, C.c_void_p(out.data_ptr()))
    print(f"{name:30s} {ms:8.3f} ms")
