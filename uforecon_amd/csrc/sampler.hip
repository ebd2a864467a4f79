// Ray set-up and the two samplers.
//   ray_setup          UFORecon.infer ray gather + near/far scaling   (code1/model.py:409-427)
//   sample_fixed       FixedSampler.sample_ray                         (code1/encoder_utils/sampler.py:15-50)
//   importance_merge   ImportanceSampler.sample_ray + coarse/fine merge (sampler.py:74-108, model.py:466-470)
// One wavefront per ray for the importance sampler: wave-wide scan for the CDF, per-lane binary
// search, rank-merge of the two sorted lists through LDS.
#include "ufr_internal.h"

// torch evaluates these expressions op by op (separate roundings); keep mul/add unfused so the
// sample positions come out bit-identical to the reference's.
#pragma clang fp contract(off)

namespace ufr {

__global__ void __launch_bounds__(256) ray_setup_kernel(const int64_t* __restrict__ ray_idx,
                                                         const float* __restrict__ ray_d,
                                                         const float* __restrict__ cam_ray_d, int HW, float near_z,
                                                         float far_z, int RN, float* __restrict__ rd_out,
                                                         float* __restrict__ near_out, float* __restrict__ far_out,
                                                         float* __restrict__ camz_out, float ox, float oy, float oz,
                                                         float* __restrict__ ray_o_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && ray_o_out) { ray_o_out[0] = ox; ray_o_out[1] = oy; ray_o_out[2] = oz; }
  if (i >= RN) return;
  int64_t p = ray_idx[i];
  // an index outside the H x W grid never becomes an out-of-bounds read: the ray is rendered from pixel 0 with NaN
  // near / far, so its depth and colour come out NaN (the caller's bug stays visible)
  const bool in_range = p >= 0 && p < (int64_t)HW;
  p = in_range ? p : 0;
  rd_out[3 * i + 0] = ray_d[p];
  rd_out[3 * i + 1] = ray_d[(size_t)HW + p];
  rd_out[3 * i + 2] = ray_d[2 * (size_t)HW + p];
  float cz = cam_ray_d ? cam_ray_d[2 * (size_t)HW + p] : 1.f;
  const float bad = __builtin_nanf("");
  near_out[i] = !in_range ? bad : cam_ray_d ? near_z / cz : near_z;  // model.py:426-427 (extract_geometry only)
  far_out[i] = !in_range ? bad : cam_ray_d ? far_z / cz : far_z;
  if (camz_out) camz_out[i] = cz;
}

hipError_t launch_ray_setup(const int64_t* ray_idx, const float* ray_d, const float* cam_ray_d, int HW, float near_z,
                            float far_z, int RN, float* rd_out, float* near_out, float* far_out, float* camz_out,
                            const float* ray_o_host, float* ray_o_out, hipStream_t s) {
  hipLaunchKernelGGL(ray_setup_kernel, dim3((RN + 255) / 256), dim3(256), 0, s, ray_idx, ray_d, cam_ray_d, HW, near_z,
                     far_z, RN, rd_out, near_out, far_out, camz_out, ray_o_host ? ray_o_host[0] : 0.f,
                     ray_o_host ? ray_o_host[1] : 0.f, ray_o_host ? ray_o_host[2] : 0.f, ray_o_out);
  return hipGetLastError();
}

// z[ray][s] = lin[s]*(far-near)+near + (U[s][ray]-0.5)*(1/(SN-1))*(far-near), lin = linspace(0,1,SN) as
// float32(float64 linspace) (sampler.py:33-43).  Threads run over (s, ray) with ray fastest so the
// U reads are coalesced; z writes are strided (RN*SN*4 bytes total, negligible).
__global__ void __launch_bounds__(256) sample_fixed_kernel(const float* __restrict__ near,
                                                            const float* __restrict__ far,
                                                            const float* __restrict__ U, int u_stride,
                                                            float* __restrict__ z, int RN, int SN) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)RN * SN) return;
  int s = (int)(i / RN), ray = (int)(i % RN);
  // np.linspace(0,1,SN): start + s*step in float64, last element forced to stop
  double step = 1.0 / (double)(SN - 1);
  float lin = (s == SN - 1) ? 1.0f : (float)((double)s * step);
  float n = near[ray], f = far[ray];
  float zz = lin * (f - n) + n;
  float interval = (float)(1.0 / (double)(SN - 1));  // python float (double) 1/(SN-1), multiplied into a float tensor
  zz = zz + (U[(size_t)s * u_stride + ray] - 0.5f) * interval * (f - n);
  z[(size_t)ray * SN + s] = zz;
}

hipError_t launch_sample_fixed(const float* near, const float* far, const float* U, int u_stride, float* z, int RN,
                               int SN, hipStream_t s) {
  size_t n = (size_t)RN * SN;
  hipLaunchKernelGGL(sample_fixed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, near, far, U, u_stride, z,
                     RN, SN);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) points_kernel(const float* __restrict__ ray_o, int o_stride,
                                                      const float* __restrict__ ray_d, const float* __restrict__ z,
                                                      float* __restrict__ pts, int RN, int SN) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)RN * SN) return;
  int ray = (int)(i / SN);
  float zz = z[i];
  const float* o = ray_o + (size_t)ray * o_stride;
  pts[3 * i + 0] = o[0] + zz * ray_d[3 * ray + 0];
  pts[3 * i + 1] = o[1] + zz * ray_d[3 * ray + 1];
  pts[3 * i + 2] = o[2] + zz * ray_d[3 * ray + 2];
}

hipError_t launch_points(const float* ray_o, int o_stride, const float* ray_d, const float* z, float* pts, int RN,
                         int SN, hipStream_t s) {
  size_t n = (size_t)RN * SN;
  hipLaunchKernelGGL(points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ray_o, o_stride, ray_d, z, pts,
                     RN, SN);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// One wave per ray, 4 rays per block.  LDS per wave: cdf[SN] | all[SN+PN] = coarse z then fine z.
constexpr int kMaxS = 256;  // SN, PN <= 256

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__global__ void __launch_bounds__(256) importance_merge_kernel(const float* __restrict__ weight,
                                                                const float* __restrict__ z,
                                                                const float* __restrict__ U2, int u_stride,
                                                                float* __restrict__ z_fine, float* __restrict__ z_all,
                                                                int RN, int SN, int PN, float* __restrict__ z_new,
                                                                int* __restrict__ src_row) {
  __shared__ float lds[4][3 * kMaxS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  const bool active = ray < RN;
  float* cdf = lds[wave];
  float* all = cdf + kMaxS;  // [0,SN) coarse, [SN,SN+PN) fine
  float* zf = all + SN;
  const float* w = weight + (size_t)(active ? ray : 0) * SN;

  // cdf = cumsum(w) / (sum(w) + 1e-6): torch's CPU cumsum accumulates in double and rounds each
  // prefix to float (sampler.py:84); lanes own contiguous chunks of K samples.
  const int K = (SN + 63) / 64;
  double run = 0.0;
  for (int k = 0; k < K; ++k) {
    int i = lane * K + k;
    if (i < SN) run += (double)w[i];
  }
  double incl = wave_incl_scan(run, lane);
  // sum(w): torch's float32 row reduction on the CPU (AVX2 kernel) keeps 4 accumulators of 8 lanes over
  // 32-element chunks, adds the accumulators in order and then the 8 lanes in order; reproducing that
  // order makes the CDF -- hence the fine sample positions -- bit-identical for SN % 32 == 0.
  float total;
  {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    const int l8 = lane & 7;
    for (int c = 0; c * 32 < SN; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int i = (c * 4 + k) * 8 + l8;
        if (i < SN) a4[k] += w[i];
      }
    const float t = ((a4[0] + a4[1]) + a4[2]) + a4[3];
    total = __shfl(t, 0);
#pragma unroll
    for (int l = 1; l < 8; ++l) total += __shfl(t, l);
  }
  float denom = total + 1e-6f;
  double acc = incl - run;
  for (int k = 0; k < K; ++k) {
    int i = lane * K + k;
    if (i < SN) {
      acc += (double)w[i];
      cdf[i] = (float)acc / denom;
      all[i] = z[(size_t)(active ? ray : 0) * SN + i];
    }
  }
  __syncthreads();

  const float c_first = cdf[0], c_last = cdf[SN - 1];
  for (int k = lane; k < PN; k += 64) {
    float u = U2[(size_t)k * u_stride + (active ? ray : 0)];   // transpose of torch.rand(PN,RN)  (sampler.py:86)
    u = fminf(fmaxf(u, c_first), c_last);                // clamp                            (:88)
    int lo = 0, hi = SN;                                 // searchsorted, right=False         (:90)
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    int ri = lo < 1 ? 1 : (lo > SN - 1 ? SN - 1 : lo);   // :92-93
    float cl = cdf[ri - 1], cr = cdf[ri], zl = all[ri - 1], zr = all[ri];
    zf[k] = (u - cl) / (cr - cl + 1e-6f) * (zr - zl) + zl;  // :101
  }
  __syncthreads();
  if (!active) return;

  // sorted fine samples (sampler.py:106) by rank; ties broken by index
  if (z_fine) {
    for (int k = lane; k < PN; k += 64) {
      float v = zf[k];
      int rank = 0;
      for (int j = 0; j < PN; ++j) {
        float o = zf[j];
        rank += (o < v || (o == v && j < k)) ? 1 : 0;
      }
      z_fine[(size_t)ray * PN + rank] = v;
    }
  }
  // coarse + fine sorted together (model.py:466-470), again by rank over the concatenation.
  // z_new / src_row (whole-path renderer): the PN new positions (sorted along the ray), and for every merged slot the
  // row of the pool [RN*SN coarse evaluations | RN*PN new evaluations] that holds its per-point results --
  // a point's view-transformer output does not depend on the other samples of the ray, so the fine pass
  // re-evaluates only the new points and reads the coarse ones back through this table.
  const int T = SN + PN;
  float* out = z_all + (size_t)ray * T;
  // The rank of every sample in the concatenation, ties broken by index (a stable sort, as torch.sort of model.py:466).
  // Fast path: when the coarse positions arrive sorted (FixedSampler's always do) the 128 x 128 comparisons become the
  // 64 x 64 that sort the new samples plus one binary search per sample -- a third of the kernel's time.
  bool sorted_coarse = true;
  for (int k = lane; k + 1 < SN; k += 64) sorted_coarse = sorted_coarse && all[k] <= all[k + 1];
  if (__builtin_amdgcn_ballot_w64(!sorted_coarse) == 0ull) {
    float* fs = cdf;                 // the cdf is dead: the new samples, sorted
    for (int k = lane; k < PN; k += 64) {
      const float v = zf[k];
      int r = 0;
      for (int j = 0; j < PN; ++j) {
        const float o = zf[j];
        r += (o < v || (o == v && j < k)) ? 1 : 0;
      }
      fs[r] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // coarse sample k: behind it come the new samples that are strictly smaller (an equal new sample has the larger index)
    for (int k = lane; k < SN; k += 64) {
      const float v = all[k];
      int lo = 0, hi = PN;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (fs[mid] < v) lo = mid + 1; else hi = mid; }
      // (equal coarse samples keep their order: k counts the coarse samples before this one)
      const int rank = k + lo;
      out[rank] = v;
      if (src_row) src_row[(size_t)ray * T + rank] = ray * SN + k;
    }
    // new sample of sorted rank r: in front of it the r new ones before it and the coarse ones that are <= it
    for (int r = lane; r < PN; r += 64) {
      const float v = fs[r];
      int lo = 0, hi = SN;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (all[mid] <= v) lo = mid + 1; else hi = mid; }
      const int rank = r + lo;
      out[rank] = v;
      if (src_row) src_row[(size_t)ray * T + rank] = RN * SN + ray * PN + r;
      if (z_new) z_new[(size_t)ray * PN + r] = v;
    }
    return;
  }
  for (int k = lane; k < T; k += 64) {
    float v = all[k];
    int rank = 0, rank_new = 0;   // among all merged samples / among the new ones only
    for (int j = 0; j < T; ++j) {
      float o = all[j];
      const int before = (o < v || (o == v && j < k)) ? 1 : 0;
      rank += before;
      rank_new += j >= SN ? before : 0;
    }
    out[rank] = v;
    // the new samples are emitted SORTED along the ray (their pool rows are only ever reached through src_row, so any order
    // is valid): neighbouring lanes of the gather kernels then hold neighbouring positions -- shared footprints in the
    // forward gathers, and the backward scatter folds runs of equal voxel corners before it issues atomics
    if (src_row) src_row[(size_t)ray * T + rank] = k < SN ? ray * SN + k : RN * SN + ray * PN + rank_new;
    if (z_new && k >= SN) z_new[(size_t)ray * PN + rank_new] = v;
  }
}

hipError_t launch_importance_merge(const float* weight, const float* z, const float* U2, int u_stride, float* z_fine,
                                   float* z_all, int RN, int SN, int PN, float* z_new, int* src_row, hipStream_t s) {
  if (SN > kMaxS || PN > kMaxS || SN < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(importance_merge_kernel, dim3((RN + 3) / 4), dim3(256), 0, s, weight, z, U2, u_stride, z_fine,
                     z_all, RN, SN, PN, z_new, src_row);
  return hipGetLastError();
}

}  // namespace ufr
