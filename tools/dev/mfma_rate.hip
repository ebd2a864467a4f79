// dev probe: cycles per instruction of bf16 MFMA variants (one wave per SIMD, independent accumulators, s_memtime)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f32x16 c0 = {}, c1 = {};
  bf16x8 x8, y8;
  s16x4 x4, y4;
  for (int i = 0; i < 8; ++i) { x8[i] = (__bf16)(1.f + i + threadIdx.x); y8[i] = (__bf16)(0.5f * i); }
  for (int i = 0; i < 4; ++i) { x4[i] = (short)(0x3f80 + i); y4[i] = (short)(0x3f80 + 2 * i); }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y8, x8, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, x8, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y8, y8, a3, 0, 0, 0);
      } else if (KIND == 1) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x4, y4, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y4, x4, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x4, x4, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y4, y4, a3, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, y8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y8, x8, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y8, y8, c1, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = (t1 - t0);
  sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + c0[0] + c1[5];
}
extern "C" void run(unsigned long long* out, float* sink, int iters) {
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, sink, iters);
  (void)hipDeviceSynchronize();
}
