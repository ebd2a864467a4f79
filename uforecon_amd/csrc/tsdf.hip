// TSDF fusion: integrate one depth (+ optional colour) observation into the voxel volumes.
// Replaces the reference's only native kernel -- a CUDA source string compiled at run time through pycuda
// (tsdf_fusion.py:77-152), which cannot exist on ROCm -- with the same per-voxel algorithm: voxel -> world ->
// camera (R^T (p - t) of the camera-to-world pose) -> pixel (roundf) -> truncated signed distance to the observed
// depth -> running weighted average.  fp32, one operation per statement, unfused, so that the numpy restatement in
// oracle/tsdf_oracle.py reproduces it bit for bit.
// One thread per voxel, z fastest = the volumes' memory order: tsdf / weight / colour accesses are coalesced
// 4-byte streams; the kernel is HBM-bound on 16 B per updated voxel (8 B read, 8 B written; + 8 B with colour).
// Behaviour kept from the reference kernel: colour integration is OFF unless asked for (upstream's colour block is
// unreachable behind an early `return`, tsdf_fusion.py:139).  Not kept: its `voxel_idx > N` bound (off by one, an
// out-of-range access for the thread with voxel_idx == N).
#include "ufr_internal.h"

#pragma clang fp contract(off)

namespace ufr {

struct TsdfParams {
  int dim[3];
  float origin[3];
  float voxel_size, trunc_margin, obs_weight;
  float K[9], P[16];
  int im_h, im_w, integrate_color;
};

__global__ void __launch_bounds__(256) tsdf_integrate_kernel(float* __restrict__ tsdf, float* __restrict__ weight,
                                                              float* __restrict__ color, TsdfParams p,
                                                              const float* __restrict__ depth_im,
                                                              const float* __restrict__ color_im) {
  const size_t n = (size_t)p.dim[0] * p.dim[1] * p.dim[2];
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int yz = p.dim[1] * p.dim[2];
  const int vx = (int)(idx / yz), rem = (int)(idx - (size_t)vx * yz);
  const int vy = rem / p.dim[2], vz = rem - vy * p.dim[2];
  // voxel -> world -> camera (tsdf_fusion.py:104-113)
  const float pt_x = p.origin[0] + (float)vx * p.voxel_size;
  const float pt_y = p.origin[1] + (float)vy * p.voxel_size;
  const float pt_z = p.origin[2] + (float)vz * p.voxel_size;
  const float tx = pt_x - p.P[3], ty = pt_y - p.P[7], tz = pt_z - p.P[11];
  const float cx = (p.P[0] * tx + p.P[4] * ty) + p.P[8] * tz;
  const float cy = (p.P[1] * tx + p.P[5] * ty) + p.P[9] * tz;
  const float cz = (p.P[2] * tx + p.P[6] * ty) + p.P[10] * tz;
  // camera -> pixel (:115-121); the range test is done on the rounded float so no out-of-range float->int cast happens
  const float px = roundf(p.K[0] * (cx / cz) + p.K[2]);
  const float py = roundf(p.K[4] * (cy / cz) + p.K[5]);
  if (!(px >= 0.f && px < (float)p.im_w && py >= 0.f && py < (float)p.im_h) || cz < 0.f) return;
  const size_t pix = (size_t)(int)py * p.im_w + (int)px;
  const float depth = depth_im[pix];
  if (depth == 0.f) return;                                         // :123-125
  const float diff = depth - cz;
  if (diff < -p.trunc_margin) return;                               // :128-130
  const float dist = fminf(1.f, diff / p.trunc_margin);
  const float w_old = weight[idx], w_new = w_old + p.obs_weight;
  weight[idx] = w_new;
  tsdf[idx] = (tsdf[idx] * w_old + p.obs_weight * dist) / w_new;   // :132-136
  if (p.integrate_color && color && color_im) {                     // :140-151
    const float c256 = 65536.f;
    const float old = color[idx];
    const float old_b = floorf(old / c256);
    const float old_g = floorf((old - old_b * c256) / 256.f);
    const float old_r = (old - old_b * c256) - old_g * 256.f;
    const float nw = color_im[pix];
    const float new_b = floorf(nw / c256);
    const float new_g = floorf((nw - new_b * c256) / 256.f);
    const float new_r = (nw - new_b * c256) - new_g * 256.f;
    const float b = fminf(roundf((old_b * w_old + p.obs_weight * new_b) / w_new), 255.f);
    const float g = fminf(roundf((old_g * w_old + p.obs_weight * new_g) / w_new), 255.f);
    const float r = fminf(roundf((old_r * w_old + p.obs_weight * new_r) / w_new), 255.f);
    color[idx] = (b * c256 + g * 256.f) + r;
  }
}

hipError_t launch_tsdf_integrate(float* tsdf, float* weight, float* color, const int* dim, const float* origin,
                                 float voxel_size, float trunc_margin, const float* K, const float* P,
                                 const float* depth_im, const float* color_im, int im_h, int im_w, float obs_weight,
                                 int integrate_color, hipStream_t s) {
  TsdfParams p;
  for (int i = 0; i < 3; ++i) { p.dim[i] = dim[i]; p.origin[i] = origin[i]; }
  for (int i = 0; i < 9; ++i) p.K[i] = K[i];
  for (int i = 0; i < 16; ++i) p.P[i] = P[i];
  p.voxel_size = voxel_size; p.trunc_margin = trunc_margin; p.obs_weight = obs_weight;
  p.im_h = im_h; p.im_w = im_w; p.integrate_color = integrate_color;
  const size_t n = (size_t)dim[0] * dim[1] * dim[2];
  hipLaunchKernelGGL(tsdf_integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tsdf, weight, color, p,
                     depth_im, color_im);
  return hipGetLastError();
}

}  // namespace ufr
