// Per-frame re-layout of the encoder outputs to channel-last, so that one bilinear / trilinear
// tap of the gather kernel is one contiguous line (feat: 128 B, match: 128*(NV-1) B, volume
// texel: 48 B, rgb: 16 B).  HBM-bound streaming kernels, run once per frame
// (the reference keeps NCHW and lets F.grid_sample stride over channels, model.py:251,370).
#include <mutex>

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

// Plane regions (ufr_layout_f16.h).  Forward streams: halfword h = fp16 plane p of 2^kWScaleLog2 * raw[param][elem]; hi =
// fp16(w'), lo = fp16(w' - hi), both round-to-nearest-even (hi + lo carries 22+ significand bits of w'); a weight outside
// the fp16 range after scaling (or not finite) raises *flag.  Backward streams: bf16 planes of the unscaled weight.
// Where a halfword comes from is a pure function of the layout: it is evaluated ONCE per device into a table (param, elem,
// plane | bf16 flag), and a pack is a gather through that table -- training re-packs after every optimizer step, and
// walking the panel lists per halfword (plan_entry_f16) cost 0.41 ms per step once the backward streams had tripled the region.
struct PlaneSrc { int elem; short param; unsigned char plane, bf16; };
static_assert(sizeof(PlaneSrc) == 8, "one 8-byte entry per halfword");

__global__ void __launch_bounds__(256) plane_plan_kernel(PlaneSrc* __restrict__ plan, int n) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  int p, e, plane, bf;
  plan_entry_f16(h, &p, &e, &plane, &bf);
  plan[h] = PlaneSrc{e, (short)p, (unsigned char)plane, (unsigned char)bf};
}

__global__ void __launch_bounds__(256) pack_weights_f16_kernel(RawPtrs raw, const PlaneSrc* __restrict__ plan,
                                                                unsigned short* __restrict__ packed, int n,
                                                                int* __restrict__ flag) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  const PlaneSrc src = plan[h];
  unsigned short out = 0;
  if (src.param >= 0) {
    if (src.bf16) {
      const float w = raw.p[src.param][src.elem];
      const __bf16 hi = (__bf16)w;
      const __bf16 lo = (__bf16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, src.plane == 0 ? hi : lo);
    } else {
      const float w = raw.p[src.param][src.elem] * kWScale;
      if (!(fabsf(w) <= 65504.f)) atomicOr(flag, 4);   // bit 2 of the sticky range status (include/ufr.h)
      const _Float16 hi = (_Float16)w;
      const _Float16 lo = (_Float16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, src.plane == 0 ? hi : lo);
    }
  }
  packed[h] = out;
}

namespace {
struct PlanCache {
  std::once_flag once[16];
  hipError_t err[16];
  PlaneSrc* plan[16];
};
PlanCache g_plan;
}  // namespace

hipError_t launch_pack_weights(const RawPtrs& raw, float* packed, int* range_flag, hipStream_t s) {
  const int n = blob_floats(), first = vec_region_offset();
  constexpr int n_half = kF16Halfwords + kBwdHalfwords;   // forward fp16 planes, then the backward kernels' bf16 planes
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  std::call_once(g_plan.once[dev], [&] {      // process lifetime, 8 bytes per halfword (~13 MB)
    g_plan.err[dev] = hipMalloc(reinterpret_cast<void**>(&g_plan.plan[dev]), sizeof(PlaneSrc) * (size_t)n_half);
    if (g_plan.err[dev] != hipSuccess) return;
    hipLaunchKernelGGL(plane_plan_kernel, dim3((n_half + 255) / 256), dim3(256), 0, s, g_plan.plan[dev], n_half);
    g_plan.err[dev] = hipGetLastError();
    if (g_plan.err[dev] == hipSuccess) g_plan.err[dev] = hipStreamSynchronize(s);   // other streams may pack next
  });
  if (g_plan.err[dev] != hipSuccess) return g_plan.err[dev];
  hipLaunchKernelGGL(pack_weights_kernel, dim3((n - first + 255) / 256), dim3(256), 0, s, raw, packed, first, n);
  unsigned short* planes = reinterpret_cast<unsigned short*>(packed + n);
  hipLaunchKernelGGL(pack_weights_f16_kernel, dim3((n_half + 255) / 256), dim3(256), 0, s, raw, g_plan.plan[dev], planes, n_half,
                     range_flag);
  return hipGetLastError();
}

}  // namespace ufr
