// Trilinear lookup into one channel-last correlation frustum (shared by the forward gather and its backward).
#pragma once
#include "ufr_device.h"
#include "ufr_internal.h"

namespace ufr {

// trilinear sample of one channel-last volume (12 floats per texel), zeros padding, align_corners=True
__device__ __forceinline__ void sample_volume(const float* __restrict__ vol, int D, int H, int W, float x, float y,
                                              float zn, float (&f)[8], float& wgt) {
  float ix = unnorm3d_ac(x, W), iy = unnorm3d_ac(y, H), iz = unnorm3d_ac(zn, D);
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
  f32x4 a0 = splat4(0.f), a1 = splat4(0.f);
  float aw = 0.f;
  // torch accumulates the corners in the order tnw,tne,tsw,tse,bnw,bne,bsw,bse (x fastest)
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float cx = fx + dx, cy = fy + dy, cz = fz + dz;
        bool ok = cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1) && cz >= 0.f &&
                  cz <= (float)(D - 1);
        if (ok) {
          const float wt = mul_rn(mul_rn(wx[dx], wy[dy]), wz[dz]);
          // in range: 0 <= cz*H + cy < 2^24 and W * kVolCh < 2^24, so the texel offset is two 24-bit multiply-adds (full
          // rate) instead of 64-bit integer multiplies; ufr_frame_prepare refuses volumes beyond 2^31 floats per view
          const unsigned row = __umul24((unsigned)(int)cz, (unsigned)H) + (unsigned)(int)cy;
          const float* t = vol + (__umul24(row, (unsigned)(W * kVolCh)) + __umul24((unsigned)(int)cx, (unsigned)kVolCh));
          const f32x4 v0 = ld4(t), v1 = ld4(t + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a0[e] = mul_add_unfused(v0[e], wt, a0[e]);
            a1[e] = mul_add_unfused(v1[e], wt, a1[e]);
          }
          aw = mul_add_unfused(t[8], wt, aw);
        }
      }
  f[0] = a0[0]; f[1] = a0[1]; f[2] = a0[2]; f[3] = a0[3];
  f[4] = a1[0]; f[5] = a1[1]; f[6] = a1[2]; f[7] = a1[3];
  wgt = aw;
}

// The same lookup through a bounded buffer descriptor over ONE view's volume (forward gather): the 24 loads of the eight
// corners are issued back to back -- a corner outside the volume reads zeros from kBufOut and adds 0 * weight -- instead of
// eight conditional blocks with a memory round trip each.  Same products, same order.
__device__ __forceinline__ void sample_volume_buf(__amdgpu_buffer_rsrc_t r, int D, int H, int W, float x, float y, float zn,
                                                  float (&f)[8], float& wgt) {
  float ix = unnorm3d_ac(x, W), iy = unnorm3d_ac(y, H), iz = unnorm3d_ac(zn, D);
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
  unsigned off[8];
  float wt[8];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int k = 4 * dz + 2 * dy + dx;
        float cx = fx + dx, cy = fy + dy, cz = fz + dz;
        bool ok = cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1) && cz >= 0.f &&
                  cz <= (float)(D - 1);
        wt[k] = mul_rn(mul_rn(wx[dx], wy[dy]), wz[dz]);
        const unsigned row = __umul24((unsigned)(int)cz, (unsigned)H) + (unsigned)(int)cy;
        const unsigned texel = __umul24(row, (unsigned)W) + (unsigned)(int)cx;       // < 2^24 rows, < 2^31 bytes per view
        off[k] = ok ? texel * (unsigned)(kVolCh * 4) : kBufOut;
      }
  f32x4 v0[8], v1[8];
  float v2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v0[k] = buf_ld4(r, off[k]);
    v1[k] = buf_ld4(r, off[k] + 16u);
    v2[k] = buf_ld1(r, off[k] + 32u);
  }
  f32x4 a0 = splat4(0.f), a1 = splat4(0.f);
  float aw = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {       // torch's corner order: tnw,tne,tsw,tse,bnw,bne,bsw,bse (x fastest)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a0[e] = mul_add_unfused(v0[k][e], wt[k], a0[e]);
      a1[e] = mul_add_unfused(v1[k][e], wt[k], a1[e]);
    }
    aw = mul_add_unfused(v2[k], wt[k], aw);
  }
  f[0] = a0[0]; f[1] = a0[1]; f[2] = a0[2]; f[3] = a0[3];
  f[4] = a1[0]; f[5] = a1[1]; f[6] = a1[2]; f[7] = a1[3];
  wgt = aw;
}

}  // namespace ufr
