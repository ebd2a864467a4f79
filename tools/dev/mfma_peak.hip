// dev probe: peak issue rate of v_mfma_f32_16x16x4_f32 with 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int c = 1; c < CH; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
extern "C" float run(int blocks, int iters, int ch, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    if (ch == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    else if (ch == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
