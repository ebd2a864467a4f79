// Per-frame re-layout of the encoder outputs to channel-last, so that one bilinear / trilinear
// tap of the gather kernel is one contiguous line (feat: 128 B, match: 128*(NV-1) B, volume
// texel: 48 B, rgb: 16 B).  HBM-bound streaming kernels, run once per frame
// (the reference keeps NCHW and lets F.grid_sample stride over channels, model.py:251,370).
#include "ufr_internal.h"
#include "ufr_layout_f16.h"

namespace ufr {

// in: (N,C,S) -> out: (N,S,Cpad).  Reads are coalesced over s for every channel; each thread
// then writes its Cpad-float row with 16-byte stores.
template <int CPAD>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int C, int S) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int n = blockIdx.y;
  if (s >= S) return;
  const float* src = in + (size_t)n * C * S + s;
  float v[CPAD];
#pragma unroll
  for (int c = 0; c < CPAD; ++c) v[c] = c < C ? src[(size_t)c * S] : 0.f;
  float4* dst = reinterpret_cast<float4*>(out + ((size_t)n * S + s) * CPAD);
#pragma unroll
  for (int c = 0; c < CPAD / 4; ++c) dst[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int S, int Cpad, hipStream_t s) {
  dim3 grid((S + 255) / 256, N), block(256);
  switch (Cpad) {
    case 4: hipLaunchKernelGGL(nchw_to_nhwc_kernel<4>, grid, block, 0, s, in, out, C, S); break;
    case 32: hipLaunchKernelGGL(nchw_to_nhwc_kernel<32>, grid, block, 0, s, in, out, C, S); break;
    case 64: hipLaunchKernelGGL(nchw_to_nhwc_kernel<64>, grid, block, 0, s, in, out, C, S); break;
    case 96: hipLaunchKernelGGL(nchw_to_nhwc_kernel<96>, grid, block, 0, s, in, out, C, S); break;
    case 128: hipLaunchKernelGGL(nchw_to_nhwc_kernel<128>, grid, block, 0, s, in, out, C, S); break;
    case 160: hipLaunchKernelGGL(nchw_to_nhwc_kernel<160>, grid, block, 0, s, in, out, C, S); break;
    case 192: hipLaunchKernelGGL(nchw_to_nhwc_kernel<192>, grid, block, 0, s, in, out, C, S); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// feature volume (N,8,S) + weight volume (N,1,S) -> (N,S,12) = [f0..f7, w, 0, 0, 0]
__global__ void __launch_bounds__(256) volume_pack_kernel(const float* __restrict__ feat,
                                                           const float* __restrict__ weight,
                                                           float* __restrict__ out, int S) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int n = blockIdx.y;
  if (s >= S) return;
  const float* f = feat + (size_t)n * 8 * S + s;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = f[(size_t)c * S];
  float w = weight[(size_t)n * S + s];
  float4* dst = reinterpret_cast<float4*>(out + ((size_t)n * S + s) * kVolCh);
  dst[0] = make_float4(v[0], v[1], v[2], v[3]);
  dst[1] = make_float4(v[4], v[5], v[6], v[7]);
  dst[2] = make_float4(w, 0.f, 0.f, 0.f);
}

hipError_t launch_volume_pack(const float* feat, const float* weight, float* out, int N, int S, hipStream_t s) {
  dim3 grid((S + 255) / 256, N), block(256);
  hipLaunchKernelGGL(volume_pack_kernel, grid, block, 0, s, feat, weight, out, S);
  return hipGetLastError();
}

// packed[i] = raw[param][elem] per ufr_layout.h:plan_entry (zero for padding), for i in [first, n).  Only the vector
// fragments (biases, LayerNorm, view token) of the fp32 region are read by the kernels: the fp32 A-fragment part in front
// of them (the layout of the first, fp32-MFMA version; still described by ufr_pack_plan for the CPU layout tests) is not
// written on the device any more.
__global__ void __launch_bounds__(256) pack_weights_kernel(RawPtrs raw, float* __restrict__ packed, int first, int n) {
  int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p, e;
  plan_entry(i, &p, &e);
  packed[i] = p >= 0 ? raw.p[p][e] : 0.f;
}

// fp16 plane region (ufr_layout_f16.h): halfword h = plane p of 2^kWScaleLog2 * raw[param][elem]; hi = fp16(w'),
// lo = fp16(w' - hi), both round-to-nearest-even (hi + lo carries 22+ significand bits of w').  A weight outside the
// fp16 range after scaling (or not finite) raises *flag.
__global__ void __launch_bounds__(256) pack_weights_f16_kernel(RawPtrs raw, unsigned short* __restrict__ packed, int n,
                                                                int* __restrict__ flag) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  int p, e, plane, bf;
  plan_entry_f16(h, &p, &e, &plane, &bf);
  unsigned short out = 0;
  if (p >= 0) {
    if (bf) {
      // backward streams: bf16 planes of the unscaled weight, hi = bf16(w), lo = bf16(w - hi) (round to nearest even)
      const float w = raw.p[p][e];
      const __bf16 hi = (__bf16)w;
      const __bf16 lo = (__bf16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, plane == 0 ? hi : lo);
    } else {
      const float w = raw.p[p][e] * kWScale;
      if (!(fabsf(w) <= 65504.f)) atomicOr(flag, 4);   // bit 2 of the sticky range status (include/ufr.h)
      const _Float16 hi = (_Float16)w;
      const _Float16 lo = (_Float16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, plane == 0 ? hi : lo);
    }
  }
  packed[h] = out;
}

hipError_t launch_pack_weights(const RawPtrs& raw, float* packed, int* range_flag, hipStream_t s) {
  const int n = blob_floats(), first = vec_region_offset();
  hipLaunchKernelGGL(pack_weights_kernel, dim3((n - first + 255) / 256), dim3(256), 0, s, raw, packed, first, n);
  unsigned short* planes = reinterpret_cast<unsigned short*>(packed + n);
  constexpr int n_half = kF16Halfwords + kBwdHalfwords;   // forward fp16 planes, then the backward kernels' bf16 planes
  hipLaunchKernelGGL(pack_weights_f16_kernel, dim3((n_half + 255) / 256), dim3(256), 0, s, raw, planes, n_half, range_flag);
  return hipGetLastError();
}

}  // namespace ufr
