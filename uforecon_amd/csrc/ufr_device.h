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

// 1 / x and 1 / sqrt(x) as the hardware approximations (v_rcp_f32 / v_rsq_f32: 1 ulp) for the normalisers INSIDE the
// two transformer kernels (attention 1 / (Q.sum K + eps), LayerNorm 1 / sigma, the softmax denominator): an IEEE
// division is ten vector instructions, a correctly rounded sqrt a dozen more -- 5 % of those kernels' vector work for
// bits far below their own accumulation noise (4e-7, tests/accuracy_report.py).  -DUFR_FAST_DIV=0 restores the divisions.
#ifndef UFR_FAST_DIV
#define UFR_FAST_DIV 1
#endif
__device__ __forceinline__ float fast_rcp(float x) {
#if UFR_FAST_DIV
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.f / x;
#endif
}
__device__ __forceinline__ float fast_rsqrt(float x) {
#if UFR_FAST_DIV
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.f / sqrtf(x);
#endif
}
// ReLU of a matrix-core accumulator as ONE instruction (v_med3_f32 x, 0, FLT_MAX; an infinite accumulator is reported by the probe).  fmaxf(x, 0.f) compiles to two: hipcc
// first canonicalises an operand it cannot prove quiet (v_max_f32 x, x, x), then takes the maximum -- 96 extra vector
// instructions per view-transformer iteration.  NOT inline assembly: the compiler's hazard recogniser does not look
// inside an asm statement, and a VALU read of an MFMA result needs software wait states on gfx950 -- a v_max_f32 written
// as asm right behind a layer's last MFMA read stale accumulators (view_out off by 0.1).  NaN never reaches a ReLU
// unreported: the range probe of the dense layer runs on the accumulators first (weight_stream_f16.h: probe_gemm).
__device__ __forceinline__ float relu_acc(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 0x1.fffffep+127f); }   // FLT_MAX: with +inf hipcc folds the median back into canonicalise + max
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- bounded loads: a raw buffer descriptor over a tensor (< 2^31 bytes); an element that does not exist -- a texel outside
// the map, a masked corner -- is asked for at kBufOut, past the extent, and the hardware returns 0.  No branch and no select
// around the load: `ok ? p[i] : 0` compiles to a divergent block with its own s_waitcnt per load (the gather kernel had 110
// branches and 116 waits for 147 loads), so a thread's taps went to memory one round trip after the other.
constexpr unsigned kBufOut = 0x80000000u;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
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

// L = 4 (three source views): the exchange rides on the FMA itself as a quad-permute DPP operand.  hipcc leaves a
// v_mov_b32_dpp in front of every such FMA (270 extra VALU instructions per view-transformer iteration), hence the
// assembly.  One block = ten FMAs behind one s_nop: a DPP read needs two wait states after a VALU write of the same
// register, the hazard recogniser does not look inside inline assembly, and nothing inside a block writes a register
// the block permutes.  Same fused multiply-adds in the same order as the C++ form: bit-identical results.
// sum_d Q[d] * (K[d] held by the lane of token (tv + S) % 4 of the same point)
template <int S>
__device__ __forceinline__ float dot10_rot4(const float (&Q)[10], const float (&K)[10]) {
  static_assert(S >= 1 && S <= 3, "token offset");
  float a = 0.f;
#define UFR_DOT10_CASE(PERM)                                                                                          \
  asm("s_nop 1\n\t"                                                                                                   \
      "v_fmac_f32_dpp %0, %11, %1 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %12, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %13, %3 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %14, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %15, %5 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %16, %6 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %17, %7 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %18, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %19, %9 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
      "v_fmac_f32_dpp %0, %20, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1"                         \
      : "+v"(a)                                                                                                       \
      : "v"(Q[0]), "v"(Q[1]), "v"(Q[2]), "v"(Q[3]), "v"(Q[4]), "v"(Q[5]), "v"(Q[6]), "v"(Q[7]), "v"(Q[8]), "v"(Q[9]),   \
        "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "v"(K[8]), "v"(K[9]))
  if constexpr (S == 1) UFR_DOT10_CASE("[1,2,3,0]");
  if constexpr (S == 2) UFR_DOT10_CASE("[2,3,0,1]");
  if constexpr (S == 3) UFR_DOT10_CASE("[3,0,1,2]");
#undef UFR_DOT10_CASE
  return a;
}
// acc[d] += w * (V[d] held by the lane of token (tv + S) % 4 of the same point)
template <int S>
__device__ __forceinline__ void axpy10_rot4(float (&acc)[10], float w, const float (&V)[10]) {
  static_assert(S >= 1 && S <= 3, "token offset");
#define UFR_AXPY10_CASE(PERM)                                                                                         \
  asm("s_nop 1\n\t"                                                                                                   \
      "v_fmac_f32_dpp %0, %11, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %1, %12, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %2, %13, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %3, %14, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %4, %15, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %5, %16, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %6, %17, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %7, %18, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %8, %19, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
      "v_fmac_f32_dpp %9, %20, %10 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1"                         \
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), \
        "+v"(acc[8]), "+v"(acc[9])                                                                                    \
      : "v"(w), "v"(V[0]), "v"(V[1]), "v"(V[2]), "v"(V[3]), "v"(V[4]), "v"(V[5]), "v"(V[6]), "v"(V[7]), "v"(V[8]), "v"(V[9]))
  if constexpr (S == 1) UFR_AXPY10_CASE("[1,2,3,0]");
  if constexpr (S == 2) UFR_AXPY10_CASE("[2,3,0,1]");
  if constexpr (S == 3) UFR_AXPY10_CASE("[3,0,1,2]");
#undef UFR_AXPY10_CASE
}

// elu(x)+1 (linear_attention.py:10-11).  The negative branch is exp(x) in (0,1]: v_exp_f32 on
// x*log2(e) (rel. error ~|x| 2^-24, i.e. < 1e-6 for the |x| < 16 that matter) instead of the
// 15-instruction ocml expf -- 80 of them per view-transformer iteration.
#ifdef UFR_ACCURATE_EXP
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : expf(x); }
#else
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __expf(x); }
#endif
// elu(s a) + 1 for a power-of-two s (a raw accumulator of the split-precision GEMMs): bit-identical to elu1(s * a), the
// scale rides on the fma / on the exponent's log2(e) multiply
template <int LOG2S>
__device__ __forceinline__ float elu1_scaled(float a) {
  constexpr float s = LOG2S >= 0 ? (float)(1u << (LOG2S >= 0 ? LOG2S : 0)) : 1.f / (float)(1u << (LOG2S < 0 ? -LOG2S : 0));
#ifdef UFR_ACCURATE_EXP
  return a > 0.f ? __builtin_fmaf(a, s, 1.f) : expf(a * s);
#else
  return a > 0.f ? __builtin_fmaf(a, s, 1.f) : __builtin_amdgcn_exp2f((0x1.715476p+0f * s) * a);
#endif
}

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
