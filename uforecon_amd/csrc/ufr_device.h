// Device-side building blocks shared by the gfx950 kernels: MFMA helpers, lane exchange, the
// three grid_sample conventions of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "ufr_layout.h"

namespace ufr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15] and owns D[4*(l>>4)+r][l&15], r=0..3.  Exact fp32 (k-ordered fma chain).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

__host__ __device__ constexpr int in_steps(int cm, int t) {
  return (cm == COL_NAT88 && t == 5) ? 2
       : (cm == COL_CAT88 && (t == 5 || t == 11)) ? 2
       : (cm == COL_RW0 && t == 5) ? 1
       : 4;
}

// ---- lane exchange among the L tokens of one point (tokens of a point are adjacent columns)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// value of x held by the lane of token (tv + S) % L of the same point; src_lane precomputed for L != 4
template <int L, int S>
__device__ __forceinline__ float rot(float x, const int (&src)[8]) {
  if constexpr (S == 0) return x;
  if constexpr (L == 4) {
    if constexpr (S == 1) return dpp_quad<0x39>(x);
    if constexpr (S == 2) return dpp_quad<0x4E>(x);
    return dpp_quad<0x93>(x);
  } else {
    return __shfl(x, src[S]);
  }
}

// elu(x)+1 (linear_attention.py:10-11).  The negative branch is exp(x) in (0,1]: v_exp_f32 on
// x*log2(e) (rel. error ~|x| 2^-24, i.e. < 1e-6 for the |x| < 16 that matter) instead of the
// 15-instruction ocml expf -- 80 of them per view-transformer iteration.
#ifdef UFR_ACCURATE_EXP
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : expf(x); }
#else
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __expf(x); }
#endif

// sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float sum_groups(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// ------------------------------------------------------------------ grid_sample conventions
// The reference runs F.grid_sample on the CPU; its arithmetic (probed against torch 2.10, see
// DESIGN.md "numerics") is reproduced operation by operation so that the interpolation weights
// come out bit-identical -- white-noise test maps amplify a 1-ulp coordinate difference by W/2.
//   2-D (vectorised kernel): align_corners=True : u = (c+1) * ((size-1)/2)
//                            align_corners=False: u = fma(c+1, size/2, -0.5)
//        w = u-floor(u), e = 1-w (same in y: n, s); nw=s*e, ne=s*w, sw=n*e, se=n*w;
//        out = fma(v_se,se, fma(v_sw,sw, fma(v_ne,ne, v_nw*nw)))
//   3-D (scalar kernel):     u = ((c+1)/2)*(size-1); corner weight (dx*dy)*dz with dx = (fl+1)-u | u-fl;
//        out += v*w with separate roundings (no FMA), x fastest.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float mul_add_unfused(float a, float b, float c) {
#pragma clang fp contract(off)
  float p = a * b;
  return p + c;
}
__device__ __forceinline__ float unnorm2d_ac(float c, int size) { return mul_rn(c + 1.f, (float)(size - 1) / 2.f); }
__device__ __forceinline__ float unnorm2d_nac(float c, int size) { return fmaf(c + 1.f, (float)size / 2.f, -0.5f); }
__device__ __forceinline__ float unnorm3d_ac(float c, int size) { return mul_rn((c + 1.f) / 2.f, (float)(size - 1)); }

struct Tap2 {  // bilinear footprint in torch's order nw, ne, sw, se: texel offsets (-1 = zero) and weights
  int o[4];
  float w[4];
};

// zeros padding: out-of-range corners drop out
__device__ __forceinline__ Tap2 taps_zeros(float ix, float iy, int W, int H) {
  Tap2 t;
  // keep the float->int conversions defined for far-away / non-finite coordinates
  const bool sane = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
  float fx = floorf(ix), fy = floorf(iy);
  int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4, x1 = x0 + 1, y1 = y0 + 1;
  float w_ = ix - fx, e_ = 1.f - w_, n_ = iy - fy, s_ = 1.f - n_;
  bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  t.o[0] = (vx0 && vy0) ? y0 * W + x0 : -1; t.w[0] = mul_rn(s_, e_);
  t.o[1] = (vx1 && vy0) ? y0 * W + x1 : -1; t.w[1] = mul_rn(s_, w_);
  t.o[2] = (vx0 && vy1) ? y1 * W + x0 : -1; t.w[2] = mul_rn(n_, e_);
  t.o[3] = (vx1 && vy1) ? y1 * W + x1 : -1; t.w[3] = mul_rn(n_, w_);
  return t;
}
// border padding: clip the coordinate first (torch clip_coordinates), then the same footprint
__device__ __forceinline__ Tap2 taps_border(float ix, float iy, int W, int H) {
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  return taps_zeros(ix, iy, W, H);
}

}  // namespace ufr
