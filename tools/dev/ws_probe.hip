// Development probe (round 6): the throughput skeleton of a WEIGHT-STATIONARY view transformer under the
// package's power cap, before building it.  One 512-thread workgroup per CU; wave w owns NF[w] weight fragments (hi + lo fp16
// planes, 8 registers each) for the whole launch; per step (= 8 points = 2 column tiles) it reads its B operands (activation
// planes written by the previous slice) from LDS, issues 3 plane products per (fragment, column tile), runs NV[w] filler
// VALU instructions (the slice's LayerNorm / attention / split work), writes 5 tiles x 2 planes back to LDS, and meets the
// workgroup at one barrier.  Random operands (the matrix pipes toggle like the real kernel's).  Prints ns per point and CU.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NFRAG, int NVALU, bool VALU_FIRST = false>
__device__ __forceinline__ void slice(const f16x8 (&w)[NFRAG][2], const char* in, char* out, int lane, f32x4& sink, float& vsink) {
  constexpr int C = 2;
  f32x4 acc[C][5];
  if constexpr (VALU_FIRST) {     // staggered variant: this slice's VALU (of the previous step's output) in front of its MFMAs
    float v = vsink;
#pragma unroll
    for (int i = 0; i < NVALU; ++i) v = __builtin_fmaf(v, 1.0001f, sink[i & 3] * 1e-9f);
    vsink = v;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    const int ks = f % 3, t = (f / 3) % 5;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f16x8 bh = *reinterpret_cast<const f16x8*>(in + ((c * 3 + ks) * 2 + 0) * 1024 + lane * 16);
      const f16x8 bl = *reinterpret_cast<const f16x8*>(in + ((c * 3 + ks) * 2 + 1) * 1024 + lane * 16);
      acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][1], bh, acc[c][t], 0, 0, 0);
      acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][0], bl, acc[c][t], 0, 0, 0);
      acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][0], bh, acc[c][t], 0, 0, 0);
    }
  }
  if constexpr (!VALU_FIRST) {
    float v = vsink;
#pragma unroll
    for (int i = 0; i < NVALU; ++i) v = __builtin_fmaf(v, 1.0001f, acc[i & 1][i % 5][i & 3] * 1e-9f);
    vsink = v;
  }
  // hand the slice's output on: 3 k-steps x 2 planes per column tile (converted, not exactly split: the VALU count is NVALU's job)
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const f32x4 a = acc[c][ks], b = acc[c][ks + 2];
      const f16x8 h = {(_Float16)(a[0] * 1e-3f), (_Float16)(a[1] * 1e-3f), (_Float16)(a[2] * 1e-3f), (_Float16)(a[3] * 1e-3f),
                       (_Float16)(b[0] * 1e-3f), (_Float16)(b[1] * 1e-3f), (_Float16)(b[2] * 1e-3f), (_Float16)(b[3] * 1e-3f)};
      *reinterpret_cast<f16x8*>(out + ((c * 3 + ks) * 2 + 0) * 1024 + lane * 16) = h;
      *reinterpret_cast<f16x8*>(out + ((c * 3 + ks) * 2 + 1) * 1024 + lane * 16) = h;
    }
  sink += acc[0][0] + acc[1][4];
}

template <int NFRAG>
__device__ __forceinline__ void init_w(f16x8 (&w)[NFRAG][2], unsigned seed) {
#pragma unroll
  for (int f = 0; f < NFRAG; ++f)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        seed = seed * 1664525u + 1013904223u;
        v[i] = (_Float16)(((int)(seed >> 20) - 2048) * (p ? 1e-6f : 1e-3f));
      }
      w[f][p] = v;
    }
}

// slices: fragments / VALU per wave (two waves per SIMD: w and w + 4 share one)
template <bool STAGGER>
__global__ void __launch_bounds__(512, 1) ws_kernel(float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 hand-off buffers x 2 (double-buffered) x 12 KiB (the probe shares them pairwise: timing only)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4 * 2 * 12288 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x2e663266u + i * 2654435761u % 0x0fff0fffu;
  __syncthreads();
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  float vsink = 1.f;
  auto bufp = [&](int stage, int step) { return smem + ((stage & 3) * 2 + (step & 1)) * 12288; };
#define SLICE(W, NF, NV)                                                                   \
  if (wave == W) {                                                                         \
    f16x8 w[NF][2];                                                                        \
    init_w<NF>(w, 12345u + 977u * (threadIdx.x + 512u * blockIdx.x));                      \
    for (int s = 0; s < steps; ++s) {                                                      \
      slice<NF, NV, (STAGGER && W < 4)>(w, bufp(W, s), bufp(W + 1, s + 1), lane, sink, vsink);                 \
      __builtin_amdgcn_s_barrier();                                                        \
    }                                                                                      \
  }
  // q | k + scores | v + message | merge + LN1 | mlp0 0-3 | mlp0 4-7 | mlp0 8-9 + mlp2 0-1 | mlp2 2-4 + rw + LN2 + outputs
  // SIMD pairing by wave id is up to the hardware's placement; the table alternates light / heavy VALU
  SLICE(0, 15, 60) SLICE(1, 15, 320) SLICE(2, 15, 220) SLICE(3, 15, 200)
  SLICE(4, 20, 40) SLICE(5, 20, 40) SLICE(6, 20, 60) SLICE(7, 20, 260)
#undef SLICE
  out[blockIdx.x * 512 + threadIdx.x] = sink[0] + sink[1] + sink[2] + sink[3] + vsink;
}

extern "C" float ws_run(int blocks, int steps, float* out, int stagger) {
  const int lds = 4 * 2 * 12288;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    if (stagger) hipLaunchKernelGGL(ws_kernel<true>, dim3(blocks), dim3(512), lds, 0, out, steps);
    else hipLaunchKernelGGL(ws_kernel<false>, dim3(blocks), dim3(512), lds, 0, out, steps);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
