// dev probe 4 (inline asm, exact instruction stream): per loop body 2 MFMAs (independent accumulators) and
// NF independent v_fma_f32 on NF different registers.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define FMA1(x) "v_fma_f32 %" #x ", %" #x ", %" #x ", %" #x "\n\t"
template <int KIND, int NF>  // KIND 0: no mfma, 1: f32 16x16x4, 2: bf16 16x16x32
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float a = 1.f + threadIdx.x, b = 2.f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (short)(0x3f80 + i); hb[i] = (short)(0x3f80 + 2 * i); }
  float v0 = 0.5f, v1 = 0.25f, v2 = 0.125f, v3 = 0.7f, v4 = 0.3f, v5 = 0.2f, v6 = 0.1f, v7 = 0.9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %2, %1\n\t" : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b));
      if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %2, %1\n\t" : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb));
      if (NF >= 2) asm volatile(FMA1(0) FMA1(1) : "+v"(v0), "+v"(v1));
      if (NF >= 4) asm volatile(FMA1(0) FMA1(1) : "+v"(v2), "+v"(v3));
      if (NF >= 8) asm volatile(FMA1(0) FMA1(1) FMA1(2) FMA1(3) : "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
extern "C" float run4(int blocks, int iters, int kind, int nf, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
#define L(K, N) if (kind == K && nf == N) hipLaunchKernelGGL((k<K, N>), dim3(blocks), dim3(256), 0, 0, out, iters)
    L(0, 2); L(0, 4); L(0, 8); L(1, 0); L(1, 2); L(1, 4); L(1, 8); L(2, 0); L(2, 2); L(2, 4); L(2, 8);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
