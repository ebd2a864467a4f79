// dev probe 3: do MFMA and plain VALU work overlap on one SIMD?  (a) f32 16x16x4 MFMA, (b) bf16 16x16x32 MFMA,
// each alone, VALU alone, and both in the same waves (independent chains).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NV>  // bit0: mfma f32, bit1: valu, bit2: mfma bf16
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float a = seed + threadIdx.x, b = 2.f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (short)(threadIdx.x + i); hb[i] = (short)(i * 3 + 1); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE & 1) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
      }
      if (MODE & 4) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, ha, acc1, 0, 0, 0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);   // NV independent VALU fma per 2 MFMAs
      }
    }
  }
  float s = acc0[0] + acc1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" float run3(int blocks, int iters, int mode, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
#define L(M, N) hipLaunchKernelGGL((k<M, N>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f)
    switch (mode) {
      case 1: L(1, 2); break; case 4: L(4, 2); break;
      case 22: L(2, 2); break; case 24: L(2, 4); break; case 28: L(2, 8); break;
      case 32: L(3, 2); break; case 34: L(3, 4); break; case 38: L(3, 8); break;
      case 62: L(6, 2); break; case 64: L(6, 4); break; case 68: L(6, 8); break;
    }
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
