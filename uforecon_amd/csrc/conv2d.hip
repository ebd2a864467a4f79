// The plain 2-D convolutions of the feature backbone (FeatureNet: code1/encoder_utils/fmt/module.py:26-62, 388-468 -- the
// Conv2d = conv + BatchNorm + ReLU blocks, the 1x1 lateral connections of the FPN, and DCN.conv_offset_mask of
// encoder_utils/fmt/dcn.py:41-80) as ONE kernel family on channel-last tensors, with everything that follows a
// convolution in the reference folded into its store:
//     out = [relu]( conv(in) * scale + shift ) [+ nearest-2x-upsampled skip]      scale / shift = eval-mode BatchNorm and bias
// and, for the offset / mask convolution of a deformable layer, the layout the deformable kernel reads: planar channels,
// sigmoid on the mask channels (dcn.py:66-70: chunk -> cat((o1, o2)) IS channels 0..17 as they are; mask = sigmoid(18..26)).
// Until round 5 these were library (MIOpen) convolutions on NCHW tensors with separate BatchNorm / ReLU / cat / sigmoid /
// interpolate + add passes and an NCHW -> NHWC re-layout in front of every deformable convolution: 7.8 ms of a 28 ms
// encode_frame for ~90 GFLOP.
//
// gfx950 mapping: an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 products and sums, like the library's fp32
// convolution up to summation order).  Output pixels are the 16 MFMA columns, output channels the rows (NT tiles of 16), the
// k axis is (tap, input channel): lane group g of the B operand holds channels CPG g .. CPG g + CPG - 1 of its pixel's tap
// (CPG = CIN / 4 contiguous floats: one or two 16-byte loads), so k-step (tap, r) contracts channels {CPG g + r}; the weight
// fragments sit in LDS in exactly that order ([row tile][tap][r][lane]) and are read once per four pixel tiles.  Padding is
// the buffer descriptor's: a tap outside the image is loaded from kBufOut and reads zeros.  The loads of tap k + 1 are
// issued before the MFMAs of tap k.  A workgroup keeps its weight fragments for several groups of 256 pixels.
#include "ufr_device.h"
#include "ufr_internal.h"

#ifndef UFR_C2_ROLLED
#define UFR_C2_ROLLED 1     // the tap loop stays a loop: unrolled, the 32-channel 3x3 kernel held 302 registers (one wave per SIMD), 9 % slower
#endif

namespace ufr {

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// epilogue + store of one accumulator tile: lane (g, j) holds channels o0 + 4g .. + 3 of output pixel p
__device__ __forceinline__ void conv2d_store(const Conv2dArgs& a, int b, int p, int ox, int oy, int o0g, const f32x4& acc) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = o0g + r;
    const float sc = (a.scale && o < a.cout) ? a.scale[o] : 1.f, sh = (a.shift && o < a.cout) ? a.shift[o] : 0.f;
    float x = fmaf(acc[r], sc, sh);
    if (a.relu) x = fmaxf(x, 0.f);
    if (a.sigmoid_from >= 0 && o >= a.sigmoid_from) x = sigmoidf_(x);
    v[r] = x;
  }
  if (a.skip) {   // nearest 2x upsampling of a half-resolution channel-last tensor (F.interpolate(scale_factor=2, 'nearest'))
    const float* sk = a.skip + (((size_t)b * (a.Ho / 2) + (oy >> 1)) * (a.Wo / 2) + (ox >> 1)) * a.cout + o0g;
    if (o0g + 3 < a.cout) {
      const f32x4 s = ld4(sk);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += s[r];
    }
  }
  const size_t HWo = (size_t)a.Ho * a.Wo;
  if (a.out_planar) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o0g + r < a.cout) a.out[((size_t)b * a.cout + o0g + r) * HWo + p] = v[r];
  } else if (o0g + 3 < a.cout) {
    st4(a.out + ((size_t)b * HWo + p) * a.cout + o0g, f32x4{v[0], v[1], v[2], v[3]});
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o0g + r < a.cout) a.out[((size_t)b * HWo + p) * a.cout + o0g + r] = v[r];
  }
}

template <int CPG>
struct TapVals { float v[CPG]; };

template <int CPG>
__device__ __forceinline__ TapVals<CPG> conv2d_tap_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  TapVals<CPG> t;
  if constexpr (CPG == 8) {
    const f32x4 a = buf_ld4(r, off), b = buf_ld4(r, off + 16u);
#pragma unroll
    for (int e = 0; e < 4; ++e) { t.v[e] = a[e]; t.v[4 + e] = b[e]; }
  } else if constexpr (CPG == 4) {
    const f32x4 a = buf_ld4(r, off);
#pragma unroll
    for (int e = 0; e < 4; ++e) t.v[e] = a[e];
  } else {
    static_assert(CPG == 2, "CIN in {8, 16, 32}");
    t.v[0] = buf_ld1(r, off);
    t.v[1] = buf_ld1(r, off + 4u);
  }
  return t;
}

template <int CIN, int KS, int S, int NT>
__global__ void __launch_bounds__(256) conv2d_mfma_kernel(Conv2dArgs a) {
  constexpr int CPG = CIN / 4, KK = KS * KS, PAD = KS / 2, T = 4;
  extern __shared__ __attribute__((aligned(16))) float a_lds[];   // [NT][KK][CPG][64]
  for (int i = threadIdx.x; i < NT * KK * CPG * 64; i += blockDim.x) {
    const int l = i & 63, r = (i >> 6) % CPG, k = (i / (64 * CPG)) % KK, to = i / (64 * CPG * KK);
    const int o = 16 * to + (l & 15), c = CPG * (l >> 4) + r;
    a_lds[i] = o < a.cout ? a.w[((size_t)o * CIN + c) * KK + k] : 0.f;
  }
  __syncthreads();
  const int b = blockIdx.y, HWo = a.Ho * a.Wo;
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.in + (size_t)b * a.H * a.W * CIN, (unsigned)(a.H * a.W) * (CIN * 4u));
  const int n_groups = (HWo + 255) / 256;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int wave_base = grp * 256 + wave * (16 * T);
    int oy[T], ox[T];
    bool okp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int p = wave_base + 16 * t + j;
      okp[t] = p < HWo;
      oy[t] = p / a.Wo;
      ox[t] = p - oy[t] * a.Wo;
    }
    f32x4 acc[T][NT];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int to = 0; to < NT; ++to) acc[t][to] = splat4(0.f);
    auto load = [&](int k, TapVals<CPG> (&val)[T]) {
      const int ky = k / KS, kx = k - ky * KS;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int iy = oy[t] * S + ky - PAD, ix = ox[t] * S + kx - PAD;
        const bool ok = okp[t] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        val[t] = conv2d_tap_load<CPG>(rin, ok ? ((unsigned)(iy * a.W + ix) * CIN + CPG * g) * 4u : kBufOut);
      }
    };
    auto contract = [&](int k, const TapVals<CPG> (&val)[T]) {
#pragma unroll
      for (int r = 0; r < CPG; ++r)
#pragma unroll
        for (int to = 0; to < NT; ++to) {
          const float w = a_lds[((to * KK + k) * CPG + r) * 64 + lane];
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t][to] = mfma16(w, val[t].v[r], acc[t][to]);
        }
    };
    TapVals<CPG> va[T], vb[T];
    load(0, va);
#if UFR_C2_ROLLED
#pragma unroll 1
#endif
    for (int k = 0; k < KK; k += 2) {
      if (k + 1 < KK) load(k + 1, vb);
      contract(k, va);
      if (k + 1 < KK) {
        if (k + 2 < KK) load(k + 2, va);
        contract(k + 1, vb);
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int p = wave_base + 16 * t + j;
      if (p >= HWo) continue;
#pragma unroll
      for (int to = 0; to < NT; ++to) conv2d_store(a, b, p, ox[t], oy[t], 16 * to + 4 * g, acc[t][to]);
    }
  }
}

// The stem: 3 input channels read from the planar image (B,3,H,W), 3x3, stride 1 -- 27 products per output channel on the
// vector ALU (bound by its 8-channel store); weights as LDS broadcasts.
template <int COUT>
__global__ void __launch_bounds__(256) conv2d_stem_kernel(Conv2dArgs a) {
  __shared__ float w_lds[27 * COUT];     // [c][ky][kx][o]
  for (int i = threadIdx.x; i < 27 * COUT; i += blockDim.x) {
    const int o = i % COUT, ck = i / COUT;
    w_lds[i] = a.w[(size_t)o * 27 + ck];
  }
  __syncthreads();
  const int HW = a.H * a.W, b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int y = p / a.W, x = p - y * a.W;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.in + (size_t)b * 3 * HW, (unsigned)HW * 12u);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = y + ky - 1, ix = x + kx - 1;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const float v = buf_ld1(rin, ok ? (unsigned)(c * HW + iy * a.W + ix) * 4u : kBufOut);
        const float* wk = w_lds + ((c * 3 + ky) * 3 + kx) * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(wk[o], v, acc[o]);
      }
  float* dst = a.out + ((size_t)b * HW + p) * COUT;
#pragma unroll
  for (int o0 = 0; o0 < COUT; o0 += 4) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = fmaf(acc[o0 + r], a.scale ? a.scale[o0 + r] : 1.f, a.shift ? a.shift[o0 + r] : 0.f);
      v[r] = a.relu ? fmaxf(t, 0.f) : t;
    }
    st4(dst + o0, f32x4{v[0], v[1], v[2], v[3]});
  }
}

// The top-down pathway of the FMT (FMT_with_pathway._push_down, code1/encoder_utils/fmt/FMT.py:226-235): the sum that the
// smoothing convolution reads, out = bilinear_2x(reduced) + fine, written channel-last.  reduced [B][h][w][C] channel-last
// (the 1x1 reduction of the coarser level), fine [B][C][2h][2w] planar (the backbone's map as the reference holds it).
// F.interpolate(size = 2x, mode = 'bilinear', align_corners = False): source coordinate max(0.5 (d + 0.5) - 0.5, 0), the
// second tap clamped at the border, value = h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11) -- torch's order.
template <int C>
__global__ void __launch_bounds__(256) upsample_add_kernel(const float* __restrict__ reduced, const float* __restrict__ fine,
                                                          float* __restrict__ out, int h, int w) {
  constexpr int Q = C / 4;
  const int H = 2 * h, W = 2 * w, b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= H * W * Q) return;
  const int q = t % Q, p = t / Q, y = p / W, x = p - y * W;
  const float sy = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float h1 = sy - (float)y0, h0 = 1.f - h1, w1 = sx - (float)x0, w0 = 1.f - w1;
  const float* r = reduced + (size_t)b * h * w * C + 4 * q;
  const f32x4 v00 = ld4(r + ((size_t)y0 * w + x0) * C), v01 = ld4(r + ((size_t)y0 * w + x1) * C),
              v10 = ld4(r + ((size_t)y1 * w + x0) * C), v11 = ld4(r + ((size_t)y1 * w + x1) * C);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = (h0 * (w0 * v00[e] + w1 * v01[e]) + h1 * (w0 * v10[e] + w1 * v11[e])) +
           fine[((size_t)b * C + 4 * q + e) * H * W + p];
  st4(out + ((size_t)b * H * W + p) * C + 4 * q, o);
}

template <int CIN, int KS, int S, int NT>
hipError_t launch_conv2d_t(const Conv2dArgs& a, hipStream_t s) {
  const int n_groups = (a.Ho * a.Wo + 255) / 256;
  int blocks = n_groups < 1024 ? n_groups : 1024;          // (a few groups per workgroup: the weight fragments are staged once)
  if (a.B > 1 && blocks * a.B > 2048) blocks = (2048 + a.B - 1) / a.B;
  const size_t lds = (size_t)NT * KS * KS * (CIN / 4) * 64 * sizeof(float);
  hipLaunchKernelGGL((conv2d_mfma_kernel<CIN, KS, S, NT>), dim3(blocks, a.B), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_conv2d(const Conv2dArgs& a, int cin, int ks, int stride, int in_planar, hipStream_t s) {
  if (in_planar) {       // the stem
    if (cin != 3 || ks != 3 || stride != 1 || a.cout != 8 || a.out_planar || a.skip || a.sigmoid_from >= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv2d_stem_kernel<8>, dim3((a.H * a.W + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const int nt = (a.cout + 15) / 16;
#define UFR_C2_CASE(CIN_, KS_, S_, NT_) \
  if (cin == CIN_ && ks == KS_ && stride == S_ && nt == NT_) return launch_conv2d_t<CIN_, KS_, S_, NT_>(a, s);
  UFR_C2_CASE(8, 3, 1, 1)      // conv0.1 (8 -> 8)
  UFR_C2_CASE(8, 5, 2, 1)      // conv1.0 (8 -> 16)
  UFR_C2_CASE(16, 3, 1, 1)     // conv1.1, conv1.2
  UFR_C2_CASE(16, 5, 2, 2)     // conv2.0 (16 -> 32)
  UFR_C2_CASE(32, 3, 1, 2)     // conv2.1, conv2.2, out2.0, out3.0, the offset / mask convolutions (32 -> 27)
  UFR_C2_CASE(32, 1, 1, 2)     // out1.0
  UFR_C2_CASE(16, 1, 1, 2)     // inner1 (16 -> 32)
  UFR_C2_CASE(8, 1, 1, 2)      // inner2 (8 -> 32)
  UFR_C2_CASE(32, 1, 1, 1)     // FMT pathway: dim_reduction_1 (32 -> 16)
  UFR_C2_CASE(16, 1, 1, 1)     // FMT pathway: dim_reduction_2 (16 -> 8)
#undef UFR_C2_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_upsample_add(const float* reduced, const float* fine, float* out, int B, int C, int h, int w, hipStream_t s) {
  const int n = 4 * h * w * (C / 4);
  const dim3 grid((n + 255) / 256, B), block(256);
  if (C == 16) hipLaunchKernelGGL(upsample_add_kernel<16>, grid, block, 0, s, reduced, fine, out, h, w);
  else if (C == 8) hipLaunchKernelGGL(upsample_add_kernel<8>, grid, block, 0, s, reduced, fine, out, h, w);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace ufr
