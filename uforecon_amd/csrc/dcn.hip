// Modulated deformable convolution (DCNv2, 3x3, stride 1, padding 1, dilation 1, one offset group) as the reference's
// FeatureNet uses it: DCN.forward (code1/encoder_utils/fmt/dcn.py:66-80) -> torchvision.ops.deform_conv2d.
// torchvision builds the (C*9, H*W) im2col matrix of bilinearly sampled, mask-modulated inputs and multiplies it by the
// flattened weight; here both steps are one kernel and the column matrix never exists: a thread owns one output pixel,
// walks the 9 taps, samples the C input channels of a tap from the channel-last image (4 corners x C/4 16-byte loads,
// torchvision's `bilinear_interpolate` rule: zero outside (-1,H)x(-1,W), out-of-image corners contribute zero) and
// accumulates all Cout outputs with the weights read as LDS broadcasts.  Arithmetic: fp32 fma, taps and channels in
// im2col order (c outer in torchvision's GEMM k-index c*9+k; here k outer, c inner: a re-association of the same sum).
// For C = 32 (every deformable layer of FeatureNet) the contraction runs on the matrix cores instead
// (deform_conv3x3_mfma_kernel): pixels are the 16 columns of v_mfma_f32_16x16x4_f32 (exact fp32), out channels the rows,
// the 288-long k axis is (tap, channel); lane group g samples channels 8g..8g+7 of its pixel, so k-step (tap, r)
// contracts channels {8g + r} and the weight fragments are stored in LDS in that order; a wave keeps 4 pixel tiles in
// flight per fragment read.
#include "ufr_device.h"
#include "ufr_internal.h"

namespace ufr {

constexpr int kDcnMaxC = 32;

template <int COUT>
__global__ void __launch_bounds__(256) deform_conv3x3_kernel(const float* __restrict__ in_cl,   // [B][H][W][C]
                                                              const float* __restrict__ offset,  // [B][18][H][W]
                                                              const float* __restrict__ mask,    // [B][9][H][W] or null
                                                              const float* __restrict__ weight,  // [COUT][C][3][3]
                                                              const float* __restrict__ bias,    // [COUT] or null
                                                              float* __restrict__ out,           // [B][COUT][H][W]
                                                              int C, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float w_lds[];   // [9][C][COUT]
  for (int i = threadIdx.x; i < 9 * C * COUT; i += blockDim.x) {
    const int o = i % COUT, c = (i / COUT) % C, k = i / (COUT * C);
    w_lds[i] = weight[((size_t)o * C + c) * 9 + k];
  }
  __syncthreads();
  const int HW = H * W;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const int py = pix / W, px = pix - py * W;
  const float* img = in_cl + (size_t)b * HW * C;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
  for (int k = 0; k < 9; ++k) {
    const int ki = k / 3, kj = k - 3 * ki;
    const float y = (float)(py - 1 + ki) + offset[((size_t)b * 18 + 2 * k) * HW + pix];
    const float x = (float)(px - 1 + kj) + offset[((size_t)b * 18 + 2 * k + 1) * HW + pix];
    const float m = mask ? mask[((size_t)b * 9 + k) * HW + pix] : 1.f;
    if (y <= -1.f || y >= (float)H || x <= -1.f || x >= (float)W) continue;     // whole tap is zero
    const float yl = floorf(y), xl = floorf(x);
    const int y0 = (int)yl, x0 = (int)xl, y1 = y0 + 1, x1 = x0 + 1;
    const float lh = y - yl, lw = x - xl, hh = 1.f - lh, hw = 1.f - lw;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    const bool v1 = y0 >= 0 && x0 >= 0, v2 = y0 >= 0 && x1 <= W - 1, v3 = y1 <= H - 1 && x0 >= 0, v4 = y1 <= H - 1 && x1 <= W - 1;
    const float* p1 = img + ((size_t)(v1 ? y0 : 0) * W + (v1 ? x0 : 0)) * C;
    const float* p2 = img + ((size_t)(v2 ? y0 : 0) * W + (v2 ? x1 : 0)) * C;
    const float* p3 = img + ((size_t)(v3 ? y1 : 0) * W + (v3 ? x0 : 0)) * C;
    const float* p4 = img + ((size_t)(v4 ? y1 : 0) * W + (v4 ? x1 : 0)) * C;
    const float a1 = v1 ? w1 : 0.f, a2 = v2 ? w2 : 0.f, a3 = v3 ? w3 : 0.f, a4 = v4 ? w4 : 0.f;
    const float* wk = w_lds + (size_t)k * C * COUT;
    for (int c = 0; c < C; c += 4) {
      const f32x4 s1 = ld4(p1 + c), s2 = ld4(p2 + c), s3 = ld4(p3 + c), s4 = ld4(p4 + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float val = (a1 * s1[e] + a2 * s2[e] + a3 * s3[e] + a4 * s4[e]) * m;   // w1*v1 + w2*v2 + w3*v3 + w4*v4, then mask
        const float* wc = wk + (size_t)(c + e) * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(wc[o], val, acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) out[((size_t)b * COUT + o) * HW + pix] = acc[o];
}

// ---- C = 32 on the fp32 MFMA.  NT = ceil(Cout / 16) row tiles.
template <int NT>
__global__ void __launch_bounds__(256) deform_conv3x3_mfma_kernel(const float* __restrict__ in_cl,   // [B][H][W][32]
                                                                   const float* __restrict__ offset,  // [B][18][H][W]
                                                                   const float* __restrict__ mask,    // [B][9][H][W] or null
                                                                   const float* __restrict__ weight,  // [Cout][32][3][3]
                                                                   const float* __restrict__ bias,    // [Cout] or null
                                                                   float* __restrict__ out,           // [B][Cout][H][W]
                                                                   int Cout, int H, int W, DcnEpilogue ep) {
  constexpr int C = 32, T = 4;
  extern __shared__ __attribute__((aligned(16))) float a_lds[];   // [NT][9][8][64]: A fragment of (row tile, tap, r)
  for (int i = threadIdx.x; i < NT * 72 * 64; i += blockDim.x) {
    const int l = i & 63, r = (i >> 6) & 7, k = (i >> 9) % 9, to = i / (72 * 64);
    const int o = 16 * to + (l & 15), c = 8 * (l >> 4) + r;
    a_lds[i] = o < Cout ? weight[((size_t)o * C + c) * 9 + k] : 0.f;
  }
  __syncthreads();
  const int HW = H * W, b = blockIdx.y;
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int wave_base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * T);
  const float* img = in_cl + (size_t)b * HW * C + 8 * g;
  f32x4 acc[T][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int to = 0; to < NT; ++to)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * to + 4 * g + r;
        acc[t][to][r] = (bias && o < Cout) ? bias[o] : 0.f;
      }
  for (int k = 0; k < 9; ++k) {
    const int ki = k / 3, kj = k - 3 * ki;
    float val[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int p_raw = wave_base + 16 * t + j, p = p_raw < HW ? p_raw : HW - 1;
      const int py = p / W, px = p - py * W;
      // (ep.om_planes: offsets and masks of an image are ONE block of that many planes -- the planar output of the offset |
      // mask convolution, conv2d.hip -- instead of two tensors of 18 and 9)
      const size_t ob = (size_t)b * (ep.om_planes ? ep.om_planes : 18) * HW, mb = (size_t)b * (ep.om_planes ? ep.om_planes : 9) * HW;
      const float y = (float)(py - 1 + ki) + offset[ob + (size_t)(2 * k) * HW + p];
      const float x = (float)(px - 1 + kj) + offset[ob + (size_t)(2 * k + 1) * HW + p];
      const float m = mask ? mask[mb + (size_t)k * HW + p] : 1.f;
      const bool in = !(y <= -1.f || y >= (float)H || x <= -1.f || x >= (float)W);
      const float yl = floorf(y), xl = floorf(x);
      const int y0 = in ? (int)yl : 0, x0 = in ? (int)xl : 0, y1 = y0 + 1, x1 = x0 + 1;
      const float lh = y - yl, lw = x - xl, hh = 1.f - lh, hw = 1.f - lw;
      const bool v1 = in && y0 >= 0 && x0 >= 0, v2 = in && y0 >= 0 && x1 <= W - 1, v3 = in && y1 <= H - 1 && x0 >= 0,
                 v4 = in && y1 <= H - 1 && x1 <= W - 1;
      const float a1 = v1 ? hh * hw : 0.f, a2 = v2 ? hh * lw : 0.f, a3 = v3 ? lh * hw : 0.f, a4 = v4 ? lh * lw : 0.f;
      const float* p1 = img + ((size_t)(v1 ? y0 : 0) * W + (v1 ? x0 : 0)) * C;
      const float* p2 = img + ((size_t)(v2 ? y0 : 0) * W + (v2 ? x1 : 0)) * C;
      const float* p3 = img + ((size_t)(v3 ? y1 : 0) * W + (v3 ? x0 : 0)) * C;
      const float* p4 = img + ((size_t)(v4 ? y1 : 0) * W + (v4 ? x1 : 0)) * C;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 s1 = ld4(p1 + 4 * h), s2 = ld4(p2 + 4 * h), s3 = ld4(p3 + 4 * h), s4 = ld4(p4 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[t][4 * h + e] = (a1 * s1[e] + a2 * s2[e] + a3 * s3[e] + a4 * s4[e]) * m;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int to = 0; to < NT; ++to) {
        const float a = a_lds[((to * 9 + k) * 8 + r) * 64 + lane];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t][to] = mfma16(a, val[t][r], acc[t][to]);
      }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int p = wave_base + 16 * t + j;
    if (p >= HW) continue;
#pragma unroll
    for (int to = 0; to < NT; ++to) {
      // epilogue of the channel-last pipeline (featurenet.py): the BatchNorm + ReLU that follow a deformable layer
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * to + 4 * g + r;
        float x = acc[t][to][r];
        if (ep.scale && o < Cout) x = fmaf(x, ep.scale[o], ep.shift[o]);
        v[r] = ep.relu ? fmaxf(x, 0.f) : x;
      }
      if (ep.out_cl) {
        if (16 * to + 4 * g + 3 < Cout) st4(out + ((size_t)b * HW + p) * Cout + 16 * to + 4 * g, f32x4{v[0], v[1], v[2], v[3]});
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * to + 4 * g + r;
          if (o < Cout) out[((size_t)b * Cout + o) * HW + p] = v[r];
        }
      }
    }
  }
}

hipError_t launch_deform_conv3x3(const float* in_cl, const float* offset, const float* mask, const float* weight,
                                 const float* bias, float* out, int B, int C, int Cout, int H, int W, hipStream_t s,
                                 DcnEpilogue ep) {
  if (C == 32) {   // every deformable layer of the reference's FeatureNet
    const dim3 grid((H * W + 255) / 256, B), block(256);
    if (Cout <= 16)
      hipLaunchKernelGGL(deform_conv3x3_mfma_kernel<1>, grid, block, 72 * 64 * sizeof(float), s, in_cl, offset, mask, weight, bias,
                         out, Cout, H, W, ep);
    else
      hipLaunchKernelGGL(deform_conv3x3_mfma_kernel<2>, grid, block, 2 * 72 * 64 * sizeof(float), s, in_cl, offset, mask, weight,
                         bias, out, Cout, H, W, ep);
    return hipGetLastError();
  }
  if (ep.scale || ep.relu || ep.out_cl) return hipErrorInvalidValue;     // the epilogue exists in the C = 32 kernel only
  const dim3 grid((H * W + 255) / 256, B), block(256);
  const size_t lds = (size_t)9 * C * Cout * sizeof(float);
  switch (Cout) {
    case 8: hipLaunchKernelGGL(deform_conv3x3_kernel<8>, grid, block, lds, s, in_cl, offset, mask, weight, bias, out, C, H, W); break;
    case 16: hipLaunchKernelGGL(deform_conv3x3_kernel<16>, grid, block, lds, s, in_cl, offset, mask, weight, bias, out, C, H, W); break;
    case 32: hipLaunchKernelGGL(deform_conv3x3_kernel<32>, grid, block, lds, s, in_cl, offset, mask, weight, bias, out, C, H, W); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ufr
