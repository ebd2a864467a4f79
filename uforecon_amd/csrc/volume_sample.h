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

}  // namespace ufr
