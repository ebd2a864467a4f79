// The training loss of a ray batch and its cotangents in ONE launch (ufr_render_loss).
//
// Reference: code1/model.py:552-566 -- MSE of both passes' colours, L1 of both passes' ray depths over the rays whose GT
// depth is valid ((gt != 0) & (gt >= near) & (gt <= far), near / far of the batch element's reference view), zero depth
// terms when no ray is valid; loss = weight_rgb (rgb_c + rgb_f) + weight_depth (depth_c + depth_f).
//
// Why a kernel: as torch expressions the loss is ~25 launches forwards and ~30 autograd nodes backwards on tensors of 1 024
// rays -- 0.9 ms of a 4.7 ms step with the GPU idle between launches (tools/dev/step_timeline.py).  The cotangents
// d loss / d (rgb, depth) of both passes depend on nothing but these inputs, so the same launch writes them and the backward
// of the autograd node is a scale by the upstream gradient.
//
// One workgroup (the batch is a few thousand rays; the sums are reduced in a fixed order: same bits every run).
#include <hip/hip_runtime.h>

#include "ufr_internal.h"

namespace ufr {

namespace {
constexpr int kLossThreads = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                       // (red is reused between calls)
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kLossThreads / 64; ++w) t += red[w];
  return t;
}

__global__ void __launch_bounds__(kLossThreads) render_loss_kernel(
    const float* __restrict__ rgb_c, const float* __restrict__ depth_c, const float* __restrict__ rgb_f,
    const float* __restrict__ depth_f, const float* __restrict__ rgb_gt, const float* __restrict__ depth_gt,
    const float* __restrict__ near_far, int nf_stride, int B, int RN, float weight_rgb, float weight_depth,
    float* __restrict__ loss, float* __restrict__ d_rgb_c, float* __restrict__ d_depth_c, float* __restrict__ d_rgb_f,
    float* __restrict__ d_depth_f) {
  __shared__ float red[kLossThreads / 64];
  const int n = B * RN;
  float se_c = 0.f, se_f = 0.f, ad_c = 0.f, ad_f = 0.f, cnt = 0.f;
  for (int r = threadIdx.x; r < n; r += kLossThreads) {
    const int b = r / RN;
    const float gt = depth_gt[r], near = near_far[(size_t)b * nf_stride], far = near_far[(size_t)b * nf_stride + 1];
    const bool valid = gt != 0.f && gt >= near && gt <= far;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = rgb_gt[3 * r + c], ec = rgb_c[3 * r + c] - g, ef = rgb_f[3 * r + c] - g;
      se_c += ec * ec;
      se_f += ef * ef;
    }
    if (valid) {
      ad_c += fabsf(depth_c[r] - gt);
      ad_f += fabsf(depth_f[r] - gt);
      cnt += 1.f;
    }
  }
  se_c = block_sum(se_c, red);
  se_f = block_sum(se_f, red);
  ad_c = block_sum(ad_c, red);
  ad_f = block_sum(ad_f, red);
  cnt = block_sum(cnt, red);
  const float inv_rgb = 1.f / (3.f * (float)n), inv_cnt = cnt > 0.f ? 1.f / cnt : 0.f;
  if (threadIdx.x == 0) {
    const float lc = se_c * inv_rgb, lf = se_f * inv_rgb, dc = ad_c * inv_cnt, df = ad_f * inv_cnt;
    loss[0] = weight_rgb * (lc + lf) + weight_depth * (dc + df);
    loss[1] = lc;
    loss[2] = lf;
    loss[3] = dc;
    loss[4] = df;
  }
  const float k_rgb = 2.f * weight_rgb * inv_rgb, k_depth = weight_depth * inv_cnt;
  for (int r = threadIdx.x; r < n; r += kLossThreads) {
    const int b = r / RN;
    const float gt = depth_gt[r], near = near_far[(size_t)b * nf_stride], far = near_far[(size_t)b * nf_stride + 1];
    const bool valid = gt != 0.f && gt >= near && gt <= far;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = rgb_gt[3 * r + c];
      d_rgb_c[3 * r + c] = k_rgb * (rgb_c[3 * r + c] - g);
      d_rgb_f[3 * r + c] = k_rgb * (rgb_f[3 * r + c] - g);
    }
    const float ec = depth_c[r] - gt, ef = depth_f[r] - gt;      // sign(0) = 0, as torch's l1 backward
    d_depth_c[r] = valid ? k_depth * (float)((ec > 0.f) - (ec < 0.f)) : 0.f;
    d_depth_f[r] = valid ? k_depth * (float)((ef > 0.f) - (ef < 0.f)) : 0.f;
  }
}
}  // namespace

hipError_t launch_render_loss(const float* rgb_c, const float* depth_c, const float* rgb_f, const float* depth_f,
                              const float* rgb_gt, const float* depth_gt, const float* near_far, int nf_stride, int B, int RN,
                              float weight_rgb, float weight_depth, float* loss, float* d_rgb_c, float* d_depth_c,
                              float* d_rgb_f, float* d_depth_f, hipStream_t s) {
  hipLaunchKernelGGL(render_loss_kernel, dim3(1), dim3(kLossThreads), 0, s, rgb_c, depth_c, rgb_f, depth_f, rgb_gt, depth_gt,
                     near_far, nf_stride, B, RN, weight_rgb, weight_depth, loss, d_rgb_c, d_depth_c, d_rgb_f, d_depth_f);
  return hipGetLastError();
}

}  // namespace ufr
