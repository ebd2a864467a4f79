// Correlation-volume construction, first step: similarity of every source view with the reference view over
// the depth hypotheses of a cascade stage, and its pixel-wise weighted aggregation over the views.
//   homo_warping_trans       code1/encoder_utils/fmt/module.py:329-367   (warp source features onto the hypotheses)
//   DepthNet.forward step 2  code1/encoder_utils/fmt/TransMVSNet.py:66-97 (similarity = mean_c(warped * ref), weighting)
//
// The reference materialises the warped volume (C x D x H x W floats per view: 126 MB at stage 1 of a 512x640
// frame) and reduces it over the channels afterwards.  Here one lane group walks the D hypotheses of one
// reference pixel: the reference feature stays in registers, every hypothesis is four bilinear taps into the
// channel-last source map, multiplied into the dot product on the spot -- only the (D,H,W) similarity is written.
// A group is LP = C/4 adjacent lanes, one float4 of channels each, so a tap is one 16*LP-byte line segment per
// load instruction (cache-line-wide for C = 32).
// The arithmetic follows torch's CPU kernels operation by operation (k-ordered fma chain of the 3x3 matmul,
// unfused scale / translate, exact divisions, grid_sample's align_corners=True un-normalisation and
// accumulation order: ufr_device.h); only the channel reduction is re-associated (4 sequential + log2(LP) butterfly).
#include "ufr_device.h"
#include "ufr_internal.h"

namespace ufr {

struct CorrViews {
  int NS;
  float m[UFR_MAX_VIEWS][12];  // per source view: rows of (src_proj_new @ inverse(ref_proj_new))[:3,:4]
};

// (N,C,S) -> (N,S,C)
__global__ void __launch_bounds__(256) chw_to_hwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (s >= S) return;
  const float* src = in + (size_t)n * C * S + s;
  float* dst = out + ((size_t)n * S + s) * C;
  for (int c = 0; c < C; c += 4) st4(dst + c, f32x4{src[(size_t)c * S], src[(size_t)(c + 1) * S], src[(size_t)(c + 2) * S], src[(size_t)(c + 3) * S]});
}

template <int LP>
__global__ void __launch_bounds__(256) correlate_kernel(const float* __restrict__ ref_cl, const float* __restrict__ src_cl,
                                                         CorrViews cv, const float* __restrict__ depth,
                                                         const float* __restrict__ vw, float* __restrict__ sim,
                                                         float* __restrict__ agg, int H, int W, int D) {
  constexpr int C = 4 * LP;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pix_raw = tid / LP, cl = tid % LP, HW = H * W;
  const bool active = pix_raw < HW;
  const int pix = active ? pix_raw : HW - 1;   // whole lane groups stay converged for the shuffles
  const int py = pix / W, px = pix - py * W;
  const f32x4 ref = ld4(ref_cl + (size_t)pix * C + 4 * cl);
  const float x = (float)px, y = (float)py;
  // pixel-wise weight sum (TransMVSNet.py:70, 90/94): starts from 1e-5, views in order
  float wsum = 1e-5f;
  if (vw)
    for (int i = 0; i < cv.NS; ++i) wsum += vw[(size_t)i * HW + pix];
  const float half_w = (float)(W - 1) / 2.f, half_h = (float)(H - 1) / 2.f;  // python float (W-1)/2 is exact in fp32 here
  // The hypotheses are independent: blockIdx.y takes a run of them (round 5).  A lane group walking all D x NS hypotheses
  // one dependent bilinear footprint after the other left the coarse stages with 2.5 waves per SIMD and nothing to overlap
  // (141 us for 32 MB of traffic); with the runs spread over workgroups and the four taps of a footprint as bounded buffer
  // loads (ufr_device.h: a masked corner reads zeros from past the descriptor's extent -- no branch, all four in flight)
  // the same work has 8+ waves per SIMD.
  const int d_per = (D + (int)gridDim.y - 1) / (int)gridDim.y, d0 = (int)blockIdx.y * d_per, d1 = min(D, d0 + d_per);
  const __amdgpu_buffer_rsrc_t rsrc_v = buf_rsrc(src_cl, (unsigned)((size_t)cv.NS * HW * C * 4));
  for (int d = d0; d < d1; ++d) {
    const float dep = depth[(size_t)d * HW + pix];
    float ssum = 0.f;
    for (int i = 0; i < cv.NS; ++i) {
      const float* M = cv.m[i];
      // rot @ [x, y, 1] as torch.matmul does it (k-ordered fma chain), then * depth, + trans, unfused (module.py:349-353)
      const float rx = fmaf(M[2], 1.f, fmaf(M[1], y, mul_rn(M[0], x)));
      const float ry = fmaf(M[6], 1.f, fmaf(M[5], y, mul_rn(M[4], x)));
      const float rz = fmaf(M[10], 1.f, fmaf(M[9], y, mul_rn(M[8], x)));
      const float qx = mul_add_unfused(rx, dep, M[3]), qy = mul_add_unfused(ry, dep, M[7]), qz = mul_add_unfused(rz, dep, M[11]);
      const bool invalid = qz < 1e-6f;
      float xn = (qx / qz) / half_w - 1.f, yn = (qy / qz) / half_h - 1.f;  // :355-358
      if (invalid) { xn = -99.f; yn = -99.f; }
      const Tap2 t = taps_zeros(unnorm2d_ac(xn, W), unnorm2d_ac(yn, H), W, H);
      const unsigned base = ((unsigned)i * (unsigned)HW * C + 4u * cl) * 4u;
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = buf_ld4(rsrc_v, t.o[k] >= 0 ? base + (unsigned)t.o[k] * (C * 4u) : kBufOut);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = mul_rn(v[0][e], t.w[0]);
        a = fmaf(v[1][e], t.w[1], a);
        a = fmaf(v[2][e], t.w[2], a);
        a = fmaf(v[3][e], t.w[3], a);           // warped feature, grid_sample's accumulation order
        part += mul_rn(a, ref[e]);              // (warped * ref), then summed (TransMVSNet.py:78)
      }
#pragma unroll
      for (int o = LP / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o);
      const float s = part / (float)C;          // .mean(1)
      if (sim && active && cl == 0) sim[((size_t)i * D + d) * HW + pix] = s;
      if (vw) ssum += mul_rn(s, vw[(size_t)i * HW + pix]);
    }
    if (agg && active && cl == 0) agg[(size_t)d * HW + pix] = ssum / wsum;      // :97
  }
}

// ---- pixel-wise view weights of the first cascade stage (PixelwiseNet, code1/encoder_utils/fmt/TransMVSNet.py:23-41) and
// the weighted aggregate that follows (:80-97), fused.  PixelwiseNet is three 1x1x1 convolutions (1 -> 16 -> 8 -> 1, BatchNorm +
// ReLU after the first two), a sigmoid and a max over the depth hypotheses: per voxel 152 multiply-adds on ONE input value.
// As library convolutions it wrote a 16-channel and an 8-channel copy of the similarity volume (BatchNorm and ReLU as further
// passes): 0.44 ms per source view at stage 1, 2.6 ms per frame.  Here a thread owns a pixel of a source view, walks its D
// hypotheses and keeps the maximum: the volume is read once, nothing but the (NS,H,W) weights is written.
// params: [a0 16 | b0 16 | W1 8 x 16 | a1 8 | b1 8 | w2 8 | b2 1] with the eval-mode BatchNorms folded (a0 = w0 * scale0,
// b0 = shift0; a1 = scale1, b1 = shift1): h0 = relu(a0 x + b0), h1 = relu(a1 (W1 h0) + b1), o = w2 . h1 + b2.
constexpr int kPixelwiseParams = 16 + 16 + 128 + 8 + 8 + 8 + 1;
__global__ void __launch_bounds__(256) pixelwise_weight_kernel(const float* __restrict__ sim, const float* __restrict__ params,
                                                              float* __restrict__ vw, int D, int HW) {
  __shared__ float P[kPixelwiseParams];
  for (int i = threadIdx.x; i < kPixelwiseParams; i += blockDim.x) P[i] = params[i];
  __syncthreads();
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (pix >= HW) return;
  const float* a0 = P, *b0 = P + 16, *W1 = P + 32, *a1 = P + 160, *b1 = P + 168, *w2 = P + 176;
  const float b2 = P[184];
  const float* col = sim + (size_t)i * D * HW + pix;
  float best = 0.f;      // sigmoid > 0
  for (int d = 0; d < D; ++d) {
    const float x = col[(size_t)d * HW];
    float h0[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) h0[c] = fmaxf(fmaf(a0[c], x, b0[c]), 0.f);
    float o = b2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) t = fmaf(W1[j * 16 + c], h0[c], t);
      o = fmaf(w2[j], fmaxf(fmaf(a1[j], t, b1[j]), 0.f), o);
    }
    best = fmaxf(best, 1.f / (1.f + __expf(-o)));
  }
  vw[(size_t)i * HW + pix] = best;
}

// aggregated[d] = (sum_i sim_i[d] vw_i) / (1e-5 + sum_i vw_i), in the reference's order (TransMVSNet.py:86-97)
__global__ void __launch_bounds__(256) weighted_aggregate_kernel(const float* __restrict__ sim, const float* __restrict__ vw,
                                                                float* __restrict__ agg, int NS, int D, int HW) {
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (size_t)D * HW) return;
  const int pix = (int)(v % HW);
  float s_sum = 0.f, w_sum = 1e-5f;
  for (int i = 0; i < NS; ++i) {
    const float w = vw[(size_t)i * HW + pix];
    s_sum = s_sum + sim[(size_t)i * D * HW + v] * w;
    w_sum = w_sum + w;
  }
  agg[v] = s_sum / w_sum;
}

hipError_t launch_pixelwise_weights(const float* sim, const float* params, float* vw, float* agg, int NS, int D, int H, int W,
                                    hipStream_t s) {
  const int HW = H * W;
  hipLaunchKernelGGL(pixelwise_weight_kernel, dim3((HW + 255) / 256, NS), dim3(256), 0, s, sim, params, vw, D, HW);
  if (agg) {
    const size_t n = (size_t)D * HW;
    hipLaunchKernelGGL(weighted_aggregate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sim, vw, agg, NS, D, HW);
  }
  return hipGetLastError();
}

hipError_t launch_chw_to_hwc(const float* in, float* out, int N, int C, int S, hipStream_t s) {
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((S + 255) / 256, N), dim3(256), 0, s, in, out, C, S);
  return hipGetLastError();
}

hipError_t launch_correlate(const float* ref_cl, const float* src_cl, const float* proj_host, int NS, const float* depth,
                            const float* vw, float* sim, float* agg, int C, int H, int W, int D, hipStream_t s) {
  CorrViews cv;
  cv.NS = NS;
  for (int i = 0; i < NS; ++i)
    for (int k = 0; k < 12; ++k) cv.m[i][k] = proj_host[i * 12 + k];
  const int LP = C / 4;
  const size_t threads = (size_t)H * W * LP;
  // runs of hypotheses per workgroup row: enough for ~8 waves per SIMD, at least 4 hypotheses per run
  int dsplit = (int)((8192 * 64 + threads - 1) / threads);
  if (dsplit > D / 4) dsplit = D / 4;
  if (dsplit < 1) dsplit = 1;
  if ((size_t)NS * H * W * C * 4 >= (1ull << 31)) return hipErrorInvalidValue;      // (bounded loads: 2^31 bytes of source maps)
  const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)dsplit), block(256);
  switch (LP) {
    case 1: hipLaunchKernelGGL(correlate_kernel<1>, grid, block, 0, s, ref_cl, src_cl, cv, depth, vw, sim, agg, H, W, D); break;
    case 2: hipLaunchKernelGGL(correlate_kernel<2>, grid, block, 0, s, ref_cl, src_cl, cv, depth, vw, sim, agg, H, W, D); break;
    case 4: hipLaunchKernelGGL(correlate_kernel<4>, grid, block, 0, s, ref_cl, src_cl, cv, depth, vw, sim, agg, H, W, D); break;
    case 8: hipLaunchKernelGGL(correlate_kernel<8>, grid, block, 0, s, ref_cl, src_cl, cv, depth, vw, sim, agg, H, W, D); break;
    case 16: hipLaunchKernelGGL(correlate_kernel<16>, grid, block, 0, s, ref_cl, src_cl, cv, depth, vw, sim, agg, H, W, D); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ufr
