// dev probe 2: the gemm_stream stage pattern in isolation (8 MFMAs = 2 chains x 4 k-steps, 1 fragment load)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(512, 2) stage_loop(const f32x4* __restrict__ w, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  f32x4 b0[4], b1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { b0[i] = f32x4{1.f + lane, 2.f, 3.f, 4.f + i}; b1[i] = f32x4{0.5f * lane, 1.f, (float)i, 2.f}; }
  f32x4 ring[4];
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  if (MODE == 3) {
    for (int i = threadIdx.x; i < 64 * 64; i += 512) lds[i] = w[i];
    __syncthreads();
  }
  const f32x4* A = (MODE == 3 ? (const f32x4*)lds : (MODE == 4 ? w + (blockIdx.x % 32) * 4096 : w)) + lane;
#pragma unroll
  for (int s = 0; s < 3; ++s) ring[s] = A[s * 64];
  for (int it = 0; it < iters; ++it) {
    int zero = 0; asm volatile("" : "+s"(zero));
    const f32x4* Ai = A + zero;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) ring[(s + 3) & 3] = Ai[((s + 3) & 63) * 64];
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 a = ring[s & 3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b0[s & 3][r], acc0, 0, 0, 0);
        if (MODE != 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b1[s & 3][r], acc1, 0, 0, 0);
      }
    }
  }
  f32x4 s = acc0 + acc1;
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
extern "C" float run2(int blocks, int iters, int mode, const void* w, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    if (mode == 0) hipLaunchKernelGGL(stage_loop<0>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)w, out, iters);
    else if (mode == 1) hipLaunchKernelGGL(stage_loop<1>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)w, out, iters);
    else if (mode == 2) hipLaunchKernelGGL(stage_loop<2>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)w, out, iters);
    else if (mode == 3) hipLaunchKernelGGL(stage_loop<3>, dim3(blocks), dim3(512), 65536, 0, (const f32x4*)w, out, iters);
    else hipLaunchKernelGGL(stage_loop<4>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)w, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
