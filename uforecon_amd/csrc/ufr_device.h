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

// Hide a (wave-uniform) pointer's provenance from the optimiser.  The weight fragments are loop
// invariant, so LICM would otherwise hoist all ~1000 float4 loads out of the tile loop and spill.
// (An opaque zero OFFSET rather than an opaque pointer: the pointer keeps its kernel-argument
// provenance, so the loads stay global_load with counted vmcnt instead of flat_load + vmcnt(0).)
template <class T>
__device__ __forceinline__ const T* launder(const T* p) {
  int zero = 0;
  asm volatile("" : "+s"(zero));
  return p + zero;
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

// out[c][to] += W_M (tile to, all in tiles) x in[c][*].
//   C  : token column tiles sharing each A fragment (independent MFMA chains)
//   OT : output tiles interleaved (more independent chains when C == 1)
//   SWAP: put the activations in the A slot -> the result tile is [token][out feature]
// The A fragments stream from L2 through a small register ring: stage s (one in-tile of OT output
// tiles) issues the loads of stage s+PF and then runs its 4*OT*C MFMAs.  sched_barrier pins that
// order -- left alone, the scheduler hoists every load of the unrolled body to the top and spills.
constexpr int kPrefetch = 3;

template <int M, int OT>
struct GemmStages {
  static constexpr MatDesc d = mat_desc(M);
  static constexpr int n_groups = (d.n_out + OT - 1) / OT;
  static constexpr int n_stages = n_groups * d.n_in;
  __host__ __device__ static constexpr int to0(int s) { return (s / d.n_in) * OT; }
  __host__ __device__ static constexpr int ti(int s) { return s % d.n_in; }
  __host__ __device__ static constexpr int no(int s) { return (d.n_out - to0(s)) < OT ? (d.n_out - to0(s)) : OT; }
};

template <int OT>
struct WRing {  // PF+1 slots of OT fragments: the in-flight window of the weight stream
  f32x4 r[kPrefetch + 1][OT];
};

#ifdef UFR_ABL_NOWLOAD  // ablation build: no weight traffic (results are wrong, timing only)
struct FragSrc {
  f32x4 v;
  __device__ FragSrc(const f32x4*, int, int lane) : v(splat4((float)lane * 1e-3f)) {}
  __device__ f32x4 operator[](int) const { f32x4 r = v; asm volatile("" : "+v"(r)); return r; }
};
#else
struct FragSrc {
  const f32x4* p;
  __device__ FragSrc(const f32x4* w4, int m_off, int lane) : p(w4 + m_off / 4 + lane) {}
  __device__ f32x4 operator[](int i) const { return p[i]; }
};
#endif

// issue the loads of stage s of matrix M into ring slot (BASE + s) % (PF+1)
template <int M, int OT, int BASE>
__device__ __forceinline__ void ring_load(const FragSrc& A, WRing<OT>& ring, int s) {
  using G = GemmStages<M, OT>;
  constexpr MatDesc d = mat_desc(M);
#pragma unroll
  for (int o = 0; o < OT; ++o)
    if (o < G::no(s)) ring.r[(BASE + s) % (kPrefetch + 1)][o] = A[((G::to0(s) + o) * d.n_in + G::ti(s)) * 64];
}

// start the weight stream of matrix M (its first PF stages)
template <int M, int OT, int BASE>
__device__ __forceinline__ void prefetch_head(const f32x4* __restrict__ w4, int lane, WRing<OT>& ring) {
  const FragSrc A(w4, mat_offset(M), lane);
#pragma unroll
  for (int s = 0; s < kPrefetch; ++s)
    if (s < GemmStages<M, OT>::n_stages) ring_load<M, OT, BASE>(A, ring, s);
}

__host__ __device__ constexpr int ring_advance(int base, int n_stages) { return (base + n_stages) % (kPrefetch + 1); }

// Streamed GEMM.  Precondition: the first PF stages of M are already in flight in `ring` at phase
// BASE (prefetch_head, or the previous gemm's NEXT).  While it runs its last PF stages it starts
// the stream of matrix NEXT (phase ring_advance(BASE, n_stages)), so the L2 latency of the next
// layer's first fragments hides behind this layer's MFMAs and the VALU work in between.
template <int M, int C, int OT, bool SWAP, int BASE, int NEXT>
__device__ __forceinline__ void gemm_stream(const f32x4* __restrict__ w4, int lane,
                                            const f32x4 (&in)[C][mat_desc(M).n_in],
                                            f32x4 (&out)[C][mat_desc(M).n_out], WRing<OT>& ring) {
  using G = GemmStages<M, OT>;
  constexpr MatDesc d = mat_desc(M);
  constexpr int PF = kPrefetch;
  constexpr int NM = NEXT >= 0 ? NEXT : M;
  static_assert(G::n_stages >= PF, "streamed matrices must have at least PF stages");
  const FragSrc A(w4, mat_offset(M), lane);
  const FragSrc AN(w4, mat_offset(NM), lane);
#pragma unroll
  for (int s = 0; s < G::n_stages; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    const int sp = s + PF;
    if (sp < G::n_stages) {
      ring_load<M, OT, BASE>(A, ring, sp);
    } else if (NEXT >= 0 && sp - G::n_stages < GemmStages<NM, OT>::n_stages) {
      ring_load<NM, OT, ring_advance(BASE, G::n_stages)>(AN, ring, sp - G::n_stages);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int to = G::to0(s), ti = G::ti(s), slot = (BASE + s) % (PF + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r < in_steps(d.cm, ti)) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          if (o < G::no(s)) {
#pragma unroll
            for (int c = 0; c < C; ++c)
              out[c][to + o] = SWAP ? mfma16(in[c][ti][r], ring.r[slot][o][r], out[c][to + o])
                                    : mfma16(ring.r[slot][o][r], in[c][ti][r], out[c][to + o]);
          }
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// self-contained GEMM for the tiny heads (own ring, latency exposed once)
template <int M, int C, int OT, bool SWAP = false>
__device__ __forceinline__ void gemm(const f32x4* __restrict__ w4, int lane,
                                     const f32x4 (&in)[C][mat_desc(M).n_in],
                                     f32x4 (&out)[C][mat_desc(M).n_out]) {
  using G = GemmStages<M, OT>;
  constexpr MatDesc d = mat_desc(M);
  constexpr int PF = kPrefetch;
  const FragSrc A(w4, mat_offset(M), lane);
  WRing<OT> ring;
  prefetch_head<M, OT, 0>(w4, lane, ring);
#pragma unroll
  for (int s = 0; s < G::n_stages; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    if (s + PF < G::n_stages) ring_load<M, OT, 0>(A, ring, s + PF);
    __builtin_amdgcn_sched_barrier(0);
    const int to = G::to0(s), ti = G::ti(s), slot = s % (PF + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r < in_steps(d.cm, ti)) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          if (o < G::no(s)) {
#pragma unroll
            for (int c = 0; c < C; ++c)
              out[c][to + o] = SWAP ? mfma16(in[c][ti][r], ring.r[slot][o][r], out[c][to + o])
                                    : mfma16(ring.r[slot][o][r], in[c][ti][r], out[c][to + o]);
          }
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// per-lane vector fragment (bias / LayerNorm affine / view token): float4 for (tile t, lane group g)
template <int V>
__device__ __forceinline__ f32x4 vec_frag(const f32x4* __restrict__ w4, int t, int g) {
  return w4[vec_offset(V) / 4 + t * 4 + g];
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

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : expf(x); }

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
