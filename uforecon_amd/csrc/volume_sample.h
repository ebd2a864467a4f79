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
#ifdef UFR_GABL_FIVE   // development ablation (timing only, wrong results): five instead of six 16-byte accesses per x pair
    if (k & 1) v2[k] = v1[k][3]; else
#endif
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

// ---- the lookup SHARED BY A LANE PAIR (forward gather, round 5).  The two x-neighbours of a cell are 96 contiguous bytes
// of the channel-last volume, but a lane that walks all eight corners touches them with six 16-byte loads of its own, and
// the L1 spends a line access on every lane of every load -- two thirds of the gather kernel's line accesses were these.
// Here lanes 2q and 2q + 1 work on the SAME lookup: each loads the four (dz, dy) corners of ONE x side (side 0 = x0, side
// 1 = x0 + 1), so the pair's loads of a corner are one contiguous 96-byte run, and forms its products value x weight; the
// owner then adds the eight products per channel in torch's corner order (x fastest), taking the partner's through a
// quad-swap DPP operand.  Same products, same order, same bits as sample_volume_buf.
__device__ __forceinline__ float pair_swap(float v) {      // the value of lane ^ 1
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
}
// side: which x corner this lane loads (0 / 1).  Returns in acc[0..8] the chain as seen by a lane whose side is 0 (the owner):
// ((((((p0[0] + p1[0]) + p0[1]) + p1[1]) + p0[2]) + p1[2]) + p0[3]) + p1[3], p0 = own products, p1 = the partner's.
__device__ __forceinline__ void sample_volume_pair(__amdgpu_buffer_rsrc_t r, int D, int H, int W, float x, float y, float zn,
                                                   int side, float (&acc)[9]) {
  float ix = unnorm3d_ac(x, W), iy = unnorm3d_ac(y, H), iz = unnorm3d_ac(zn, D);
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float wx0 = (fx + 1.f) - ix, wx1 = ix - fx;
  const float wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
  const float wxs = side ? wx1 : wx0, cx = fx + (float)side;
  const bool okx = cx >= 0.f && cx <= (float)(W - 1);
  unsigned off[4];
  float wt[4];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int q = 2 * dz + dy;
      const float cy = fy + dy, cz = fz + dz;
      const bool ok = okx && cy >= 0.f && cy <= (float)(H - 1) && cz >= 0.f && cz <= (float)(D - 1);
      wt[q] = mul_rn(mul_rn(wxs, wy[dy]), wz[dz]);
      const unsigned row = __umul24((unsigned)(int)cz, (unsigned)H) + (unsigned)(int)cy;
      const unsigned texel = __umul24(row, (unsigned)W) + (unsigned)(int)cx;
      off[q] = ok ? texel * (unsigned)(kVolCh * 4) : kBufOut;
    }
  f32x4 v0[4], v1[4];
  float v2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v0[q] = buf_ld4(r, off[q]);
    v1[q] = buf_ld4(r, off[q] + 16u);
    v2[q] = buf_ld1(r, off[q] + 32u);
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float pr[9];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pr[e] = mul_rn(v0[q][e], wt[q]);
      pr[4 + e] = mul_rn(v1[q][e], wt[q]);
    }
    pr[8] = mul_rn(v2[q], wt[q]);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      acc[c] = acc[c] + pr[c];                 // corner (dz, dy, x0): this lane's (when it is the owner)
      acc[c] = acc[c] + pair_swap(pr[c]);      // corner (dz, dy, x0 + 1): the partner's
    }
  }
}

}  // namespace ufr
