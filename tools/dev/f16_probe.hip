#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// lo halves: fp16(x - hi) with one v_fma_mix each
__device__ __forceinline__ unsigned lo_pack(float a, float b, unsigned h) {
  unsigned r = 0;
  // d.lo = fp16( h.lo * -1 + a ): src0 = h (f16, low half), src1 = -1.0 (f32 const), src2 = a (f32)
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(b));
  return r;
}
__global__ void probe(float* out, const float* in) {
  const int lane = threadIdx.x;
  // 1. subnormal inputs to the MFMA
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }  // 2^-20
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];   // 32 * 2^-20 = 3.0517578125e-05 if preserved
  // 2. split of in[lane*2], in[lane*2+1]
  float x0 = in[2 * lane], x1 = in[2 * lane + 1];
  unsigned h = cvt_pk_rne(x0, x1);
  unsigned l = lo_pack(x0, x1, h);
  f16x2 hh = __builtin_bit_cast(f16x2, h), ll = __builtin_bit_cast(f16x2, l);
  out[16 + 4 * lane + 0] = (float)hh[0]; out[16 + 4 * lane + 1] = (float)ll[0];
  out[16 + 4 * lane + 2] = (float)hh[1]; out[16 + 4 * lane + 3] = (float)ll[1];
}
template <int KIND>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  f32x4 acc[4] = {};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.f + i); b[i] = (_Float16)(0.5f * i); }
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[u], 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  float *d_out, *d_in;
  hipMalloc(&d_out, 1 << 22); hipMalloc(&d_in, 4096);
  std::vector<float> in(128);
  const float vals[8] = {1.2345678f, -0.012345678f, 3.3e-4f, 7.7e-6f, 123.456f, 6.0e-8f, -65000.f, 1e-3f};
  for (int i = 0; i < 128; ++i) in[i] = vals[i % 8] * (1.f + i * 0.01f);
  hipMemcpy(d_in, in.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_in);
  std::vector<float> out(16 + 256);
  hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  printf("mfma subnormal: %g (expect 3.05176e-05 if preserved)\n", out[0]);
  for (int i = 0; i < 8; ++i) {
    const int lane = i / 2, k = i % 2;
    float h = out[16 + 4 * lane + 2 * k], l = out[16 + 4 * lane + 2 * k + 1];
    printf("x=%.9g hi=%.9g lo=%.9g  rel err of hi+lo: %.3g\n", in[i], h, l, ((double)h + l - in[i]) / in[i]);
  }
  for (int kind = 0; kind < 2; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, d_out, 20000);
      else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, d_out, 20000);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double flops = 1024.0 * 4 * 20000 * 4 * 2 * 16 * 16 * 32;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", kind == 0 ? "f16" : "bf16", ms, flops / ms * 1e-9);
  }
  return 0;
}
