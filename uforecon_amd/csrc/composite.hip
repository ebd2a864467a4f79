// SDF-to-alpha compositor: VolumeRenderer.render (code1/encoder_utils/renderer.py:7-48) with
// SingleVarianceNetwork (single_variance_network.py:10-11).  One wavefront per ray: each lane owns
// K = ceil(SN/64) consecutive samples, the transmittance is an exclusive product scan across the
// 64 lanes (DPP/shuffle Hillis-Steele) and the three weighted sums are wave reductions.
#include "ufr_internal.h"

#pragma clang fp contract(off)

namespace ufr {

constexpr int kMaxK = 4;  // SN <= 256

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__global__ void __launch_bounds__(256) composite_kernel(const float* __restrict__ z, const float* __restrict__ radiance,
                                                         const int* __restrict__ rad_row, const float* __restrict__ srdf,
                                                         const float* __restrict__ variance, int RN, int SN,
                                                         float* __restrict__ rgb, float* __restrict__ depth,
                                                         float* __restrict__ opacity, float* __restrict__ weight,
                                                         const float* __restrict__ camz, float* __restrict__ depth_z) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= RN) return;
  const float* zr = z + (size_t)ray * SN;
  const float* sr = srdf + (size_t)ray * SN;
  const float inv_s = fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);  // renderer.py:25
  const int K = (SN + 63) / 64;

  float alpha[kMaxK], zz[kMaxK];
  float prod = 1.f;  // product of (1 - alpha + 1e-7) over this lane's samples
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    alpha[k] = 0.f;
    zz[k] = 0.f;
    int i = lane * K + k;
    if (k < K && i < SN) {
      float zc = zr[i];
      // interval: mean of the two adjacent gaps with edge replication (renderer.py:19-21)
      float gl = (i == 0) ? zr[1] - zr[0] : zc - zr[i - 1];
      float gr = (i == SN - 1) ? zr[SN - 1] - zr[SN - 2] : zr[i + 1] - zc;
      float interval = (gl + gr) / 2.f;
      float s = sr[i];
      const float iter_cos = -1.5f;                       // renderer.py:28-29 with cos_anneal_ratio = 1
      float nxt = s + iter_cos * interval * 0.5f;         // :31
      float prv = s - iter_cos * interval * 0.5f;         // :32
      float pc = sigmoidf(prv * inv_s), nc = sigmoidf(nxt * inv_s);
      float a = ((pc - nc) + 1e-5f) / (pc + 1e-5f);       // :37-40
      a = fminf(fmaxf(a, 0.f), 1.f);
      alpha[k] = a;
      zz[k] = zc;
      prod *= (1.f - a) + 1e-7f;
    }
  }
  // exclusive product scan over lanes
  float incl = prod;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(incl, d);
    if (lane >= d) incl *= o;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.f;

  float acc_d = 0.f, acc_o = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    int i = lane * K + k;
    if (k < K && i < SN) {
      float w = alpha[k] * T;                             // :42
      T *= (1.f - alpha[k]) + 1e-7f;
      if (weight) weight[(size_t)ray * SN + i] = w;
      const float* c = radiance + (rad_row ? (size_t)rad_row[(size_t)ray * SN + i] : (size_t)ray * SN + i) * 3;
      acc_r += c[0] * w;
      acc_g += c[1] * w;
      acc_b += c[2] * w;
      acc_d += w * zz[k];
      acc_o += w;
    }
  }
  acc_d = wave_sum(acc_d);
  acc_o = wave_sum(acc_o);
  acc_r = wave_sum(acc_r);
  acc_g = wave_sum(acc_g);
  acc_b = wave_sum(acc_b);
  if (lane == 0) {
    depth[ray] = acc_d;
    if (opacity) opacity[ray] = acc_o;
    if (rgb) {
      rgb[3 * ray + 0] = acc_r;
      rgb[3 * ray + 1] = acc_g;
      rgb[3 * ray + 2] = acc_b;
    }
    if (depth_z) depth_z[ray] = acc_d * camz[ray];        // model.py:821
  }
}

// Adjoint of composite_kernel (autograd of renderer.py:19-46): gradients w.r.t. the radiance, the signed ray distance
// and the variance parameter given d rgb / d depth / d opacity / d weight (any may be NULL = zero).  Same lane <-> sample
// map as the forward; the transmittance product becomes an exclusive SUFFIX sum of d w_k w_k:
//   w_k = alpha_k prod_{j<k} (1 - alpha_j + 1e-7)   =>   d alpha_i = d w_i T_i - (sum_{k>i} d w_k w_k) / (1 - alpha_i + 1e-7)
// rad_row (nullable): row of `radiance` / `d_radiance` holding slot (ray, i) -- the sample pool of the two-pass training step;
// accumulate: d_radiance += instead of = (a pool row receives the coarse pass's cotangent on top of the fine pass's)
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ z, const float* __restrict__ radiance,
                                                             const int* __restrict__ rad_row, int accumulate,
                                                             const float* __restrict__ srdf,
                                                             const float* __restrict__ variance, int RN, int SN,
                                                             const float* __restrict__ d_rgb, const float* __restrict__ d_depth,
                                                             const float* __restrict__ d_opacity,
                                                             const float* __restrict__ d_weight,
                                                             float* __restrict__ d_radiance, float* __restrict__ d_srdf,
                                                             float* __restrict__ d_variance) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= RN) return;
  const float* zr = z + (size_t)ray * SN;
  const float* sr = srdf + (size_t)ray * SN;
  const float e10 = expf(variance[0] * 10.0f);
  const float inv_s = fminf(fmaxf(e10, 1e-6f), 1e6f);
  const bool s_live = e10 >= 1e-6f && e10 <= 1e6f;       // clip passes the gradient only inside the range
  const int K = (SN + 63) / 64;
  const float gr = d_rgb ? d_rgb[3 * ray + 0] : 0.f, gg = d_rgb ? d_rgb[3 * ray + 1] : 0.f, gb = d_rgb ? d_rgb[3 * ray + 2] : 0.f;
  const float gd = d_depth ? d_depth[ray] : 0.f, go = d_opacity ? d_opacity[ray] : 0.f;

  float alpha[kMaxK], araw[kMaxK], pcv[kMaxK], ncv[kMaxK], prvv[kMaxK], nxtv[kMaxK];
  float prod = 1.f;
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    alpha[k] = 0.f; araw[k] = 0.f; pcv[k] = 0.f; ncv[k] = 0.f; prvv[k] = 0.f; nxtv[k] = 0.f;
    const int i = lane * K + k;
    if (k < K && i < SN) {
      const float zc = zr[i];
      const float gl = (i == 0) ? zr[1] - zr[0] : zc - zr[i - 1];
      const float gr_ = (i == SN - 1) ? zr[SN - 1] - zr[SN - 2] : zr[i + 1] - zc;
      const float interval = (gl + gr_) / 2.f;
      const float s = sr[i];
      const float nxt = s + -1.5f * interval * 0.5f, prv = s - -1.5f * interval * 0.5f;
      const float pc = sigmoidf(prv * inv_s), nc = sigmoidf(nxt * inv_s);
      const float a = ((pc - nc) + 1e-5f) / (pc + 1e-5f);
      araw[k] = a; pcv[k] = pc; ncv[k] = nc; prvv[k] = prv; nxtv[k] = nxt;
      alpha[k] = fminf(fmaxf(a, 0.f), 1.f);
      prod *= (1.f - alpha[k]) + 1e-7f;
    }
  }
  float incl = prod;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(incl, d);
    if (lane >= d) incl *= o;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.f;

  // forward weights + d w; per-lane sum of d w_k w_k for the suffix scan
  float w[kMaxK], dw[kMaxK], Tk[kMaxK], lane_sum = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    w[k] = 0.f; dw[k] = 0.f; Tk[k] = 0.f;
    const int i = lane * K + k;
    if (k < K && i < SN) {
      Tk[k] = T;
      w[k] = alpha[k] * T;
      T *= (1.f - alpha[k]) + 1e-7f;
      const size_t slot = (size_t)ray * SN + i, rrow = rad_row ? (size_t)rad_row[slot] : slot;
      const float* c = radiance + rrow * 3;
      dw[k] = (d_weight ? d_weight[slot] : 0.f) + gr * c[0] + gg * c[1] + gb * c[2] + gd * zr[i] + go;
      lane_sum += dw[k] * w[k];
      float* dr = d_radiance + rrow * 3;
      if (accumulate) { dr[0] += w[k] * gr; dr[1] += w[k] * gg; dr[2] += w[k] * gb; }
      else { dr[0] = w[k] * gr; dr[1] = w[k] * gg; dr[2] = w[k] * gb; }
    }
  }
  // exclusive suffix sum over lanes
  float suf = lane_sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(suf, d);
    if (lane + d < 64) suf += o;
  }
  float after = __shfl_down(suf, 1);     // sum over lanes > lane
  if (lane == 63) after = 0.f;

  float dvar = 0.f;
#pragma unroll
  for (int k = kMaxK - 1; k >= 0; --k) {
    const int i = lane * K + k;
    if (k < K && i < SN) {
      const float dalpha = dw[k] * Tk[k] - after / ((1.f - alpha[k]) + 1e-7f);
      after += dw[k] * w[k];
      const float da = (araw[k] >= 0.f && araw[k] <= 1.f) ? dalpha : 0.f;
      const float den = pcv[k] + 1e-5f;
      const float dpc = da * (1.f - araw[k]) / den, dnc = -da / den;
      const float spc = pcv[k] * (1.f - pcv[k]), snc = ncv[k] * (1.f - ncv[k]);
      d_srdf[(size_t)ray * SN + i] = (dpc * spc + dnc * snc) * inv_s;
      dvar += dpc * spc * prvv[k] + dnc * snc * nxtv[k];
    }
  }
  dvar = wave_sum(dvar);
  if (lane == 0 && s_live) atomicAdd(d_variance, dvar * 10.f * inv_s);   // inv_s = exp(10 variance)
}

hipError_t launch_composite_bwd(const float* z, const float* radiance, const int* rad_row, bool accumulate, const float* srdf,
                                const float* variance, int RN, int SN, const float* d_rgb, const float* d_depth,
                                const float* d_opacity, const float* d_weight, float* d_radiance, float* d_srdf,
                                float* d_variance, hipStream_t s) {
  if (SN > 64 * kMaxK || SN < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((RN + 3) / 4), dim3(256), 0, s, z, radiance, rad_row, accumulate ? 1 : 0, srdf,
                     variance, RN, SN, d_rgb,
                     d_depth, d_opacity, d_weight, d_radiance, d_srdf, d_variance);
  return hipGetLastError();
}

hipError_t launch_composite(const float* z, const float* radiance, const int* rad_row, const float* srdf,
                            const float* variance, int RN, int SN, float* rgb, float* depth, float* opacity, float* weight,
                            const float* camz, float* depth_z, hipStream_t s) {
  if (SN > 64 * kMaxK || SN < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(composite_kernel, dim3((RN + 3) / 4), dim3(256), 0, s, z, radiance, rad_row, srdf, variance, RN, SN, rgb,
                     depth, opacity, weight, camz, depth_z);
  return hipGetLastError();
}

}  // namespace ufr
