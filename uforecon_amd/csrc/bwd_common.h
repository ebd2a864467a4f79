// Building blocks of the backward kernels (config C5: training step through the HIP path).
//
// Design (gfx950): a 256-thread workgroup (one wave per SIMD, one workgroup per CU: every wave owns the full 512-entry
// register file of its SIMD, which is what the register-resident weight-gradient tiles need) walks
// tiles of kTT = 16 tokens.  Every activation of the tile lives in LDS, feature-major [feature][kLD]
// (kLD = 17: odd stride -> the three MFMA operand access patterns below are bank-conflict free), in
// exact fp32.  All contractions run on v_mfma_f32_16x16x4_f32 (bitwise an fp32 fma chain):
//   data GEMM      Y[o][t]  = sum_i W[o][i]  X[i][t]    A = weights from global (L1/L2-resident, 0.6 MB),
//   transposed     dX[i][t] = sum_o W[o][i] dY[o][t]     B = activations from LDS, tokens are the 16 columns
//   weight grad    dW[o][i] = sum_t dY[o][t] X[i][t]     A and B from LDS, k = the tile's 16 tokens
// Weight gradients accumulate in REGISTERS across the workgroup's persistent tile loop (output tile
// (s*8 + wave) of the kernel's gradient-tile list lives in accumulator slot s of that wave) and are
// flushed once per workgroup with float atomics into the reference-layout gradient tensors.
#pragma once
#include "ufr_device.h"
#include "weight_stream.h"   // static_for

namespace ufr {

constexpr int kBwdThreads = 256;
constexpr int kBwdWaves = kBwdThreads / 64;
constexpr int kTT = 16;   // tokens per tile = MFMA columns
constexpr int kLD = 17;   // LDS row stride (floats)

struct GradPtrs { float* p[P_COUNT]; };

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// Every LDS / global address of these kernels is a function of the thread index and compile-time constants, i.e.
// invariant over the persistent tile loop: left alone, LICM precomputes all of them ahead of the loop (several hundred
// registers, spilled to scratch and reloaded one by one in front of every MFMA).  Passing the lane / thread index through
// an opaque asm at the top of a phase keeps the address arithmetic inside that phase (a base register + immediates).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---------------------------------------------------------------------------------------------------
// Y tile rows [16*rt, 16*rt+16) for every rt owned by this wave; epi(row, col, value) consumes the result.
//   TRANS = false: A[o][i] = W[o*ldw + i]          (forward layer)
//   TRANS = true : A[o][i] = W[i*ldw + o]          (input gradient: OUT = the layer's inputs, IN = its outputs)
// Row tile rt belongs to wave (rt + rt_shift) % kBwdWaves: consecutive GEMMs of one phase pass the running tile count
// so that their tiles are dealt round-robin over the waves.
// k-order inside a 16-chunk: MFMA step kk contracts k = 16*kc + 4*g + kk (lane group g), for both operands.
template <int OUT, int IN, bool TRANS, typename Epi>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ W, int ldw, const float* X, int wave, int lane,
                                         Epi epi, int rt_shift = 0) {
  constexpr int RT = (OUT + 15) / 16, KC = (IN + 15) / 16;
  lane = opaque(lane);
  const int g = lane >> 4, j = lane & 15;
  // the weights are loop-invariant over the kernel's tile loop: without an opaque offset LICM hoists every A-operand
  // load of every layer out of that loop (hundreds of spilled registers)
  int opaque_zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
  W += opaque_zero;
  for (int rt = (wave + kBwdWaves * 64 - rt_shift) % kBwdWaves; rt < RT; rt += kBwdWaves) {
    f32x4 acc0 = splat4(0.f), acc1 = splat4(0.f);
    const int row = rt * 16 + j;
    const bool row_ok = row < OUT;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int kb = kc * 16 + 4 * g;
      float a[4], b[4];
      if constexpr (!TRANS && (IN % 4 == 0)) {
        // ldw % 4 == 0 for every such matrix (rows are 16-byte aligned)
        f32x4 a4 = (row_ok && kb < IN) ? ld4(W + (size_t)row * ldw + kb) : splat4(0.f);
        a[0] = a4[0]; a[1] = a4[1]; a[2] = a4[2]; a[3] = a4[3];
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = kb + kk;
          const bool ok = row_ok && k < IN;
          a[kk] = ok ? (TRANS ? W[(size_t)k * ldw + row] : W[(size_t)row * ldw + k]) : 0.f;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b[kk] = (kb + kk < IN) ? X[(kb + kk) * kLD + j] : 0.f;
      acc0 = mfma16(a[0], b[0], acc0);
      acc1 = mfma16(a[1], b[1], acc1);
      acc0 = mfma16(a[2], b[2], acc0);
      acc1 = mfma16(a[3], b[3], acc1);
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rt * 16 + 4 * g + r;
      if (orow < OUT) epi(orow, j, acc[r]);
    }
  }
}

// one 16x16 tile of dW += dY X^T over the tile's 16 tokens
__device__ __forceinline__ f32x4 wgrad_tile(f32x4 acc, const float* dY, const float* X, int o0, int i0, int OUT, int IN,
                                            int lane) {
  const int g = lane >> 4, j = lane & 15;
  const bool ao = o0 + j < OUT, bo = i0 + j < IN;
  const float* pa = dY + (o0 + j) * kLD + g;
  const float* pb = X + (i0 + j) * kLD + g;
#pragma unroll
  for (int t0 = 0; t0 < kTT; t0 += 4) acc = mfma16(ao ? pa[t0] : 0.f, bo ? pb[t0] : 0.f, acc);
  return acc;
}

__device__ __forceinline__ void wgrad_flush(f32x4 acc, float* dW, int ldw, int o0, int i0, int OUT, int IN, int lane) {
  const int g = lane >> 4, j = lane & 15;
  if (i0 + j >= IN) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = o0 + 4 * g + r;
    if (o < OUT) atomic_add_f32(dW + (size_t)o * ldw + i0 + j, acc[r]);
  }
}

// One weight-gradient matrix of a kernel's list: dW[OUT][IN] += dY (LDS offset dy) x X (LDS offset x).
struct WgMat { int param, OUT, IN, dy, x; };

template <int N>
struct WgList {
  WgMat m[N];
  int first[N + 1];   // first tile id of each matrix
};
template <int N>
__host__ __device__ constexpr WgList<N> make_wglist(const WgMat (&mats)[N]) {
  WgList<N> l{};
  int acc = 0;
  for (int i = 0; i < N; ++i) {
    l.m[i] = mats[i];
    l.first[i] = acc;
    acc += ((mats[i].OUT + 15) / 16) * ((mats[i].IN + 15) / 16);
  }
  l.first[N] = acc;
  return l;
}

// tile id -> (matrix, row tile origin, col tile origin), evaluated at compile time: the slot index is a template
// constant and the wave id is turned into one by a switch in the kernel, so every operand address of a gradient tile
// is a base register + immediate (a run-time decode costs a table walk and an integer division per slot).
struct WgTile { int param, OUT, IN, dy, x, o0, i0; bool valid; };
template <int N>
__host__ __device__ constexpr WgTile wg_decode(const WgList<N>& l, int tile) {
  if (tile >= l.first[N]) return WgTile{0, 0, 0, 0, 0, 0, 0, false};
  int mi = 0;
  for (int i = 1; i < N; ++i) mi += tile >= l.first[i] ? 1 : 0;
  const int local = tile - l.first[mi];
  const int ct = (l.m[mi].IN + 15) / 16;
  return WgTile{l.m[mi].param, l.m[mi].OUT, l.m[mi].IN, l.m[mi].dy, l.m[mi].x, (local / ct) * 16, (local % ct) * 16, true};
}

// accumulate every tile of the list that wave WAVE owns (slot s <-> tile s*kBwdWaves + WAVE); SLOT0 = first slot
template <const auto& LIST, int N, int NSLOT, int SLOT0, int WAVE, int NACC>
__device__ __forceinline__ void wgrad_wave(f32x4 (&acc)[NACC], const float* lds, int lane) {
  static_assert(SLOT0 + NSLOT <= NACC, "accumulator slots");
  lane = opaque(lane);
  static_for<NSLOT>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    constexpr WgTile t = wg_decode(LIST, s * kBwdWaves + WAVE);
    if constexpr (t.valid)
      acc[SLOT0 + s] = wgrad_tile(acc[SLOT0 + s], lds + t.dy * kLD, lds + t.x * kLD, t.o0, t.i0, t.OUT, t.IN, lane);
  });
}
template <const auto& LIST, int N, int NSLOT, int SLOT0, int NACC>
__device__ __forceinline__ void wgrad_all(f32x4 (&acc)[NACC], const float* lds, int wave, int lane) {
  static_assert(kBwdWaves == 4, "one case per wave");
  switch (__builtin_amdgcn_readfirstlane(wave)) {
    case 0: wgrad_wave<LIST, N, NSLOT, SLOT0, 0>(acc, lds, lane); break;
    case 1: wgrad_wave<LIST, N, NSLOT, SLOT0, 1>(acc, lds, lane); break;
    case 2: wgrad_wave<LIST, N, NSLOT, SLOT0, 2>(acc, lds, lane); break;
    default: wgrad_wave<LIST, N, NSLOT, SLOT0, 3>(acc, lds, lane); break;
  }
}
template <const auto& LIST, int N, int NSLOT, int SLOT0, int WAVE, int NACC>
__device__ __forceinline__ void wgrad_flush_wave(const f32x4 (&acc)[NACC], const GradPtrs& gp, int lane) {
  static_for<NSLOT>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    constexpr WgTile t = wg_decode(LIST, s * kBwdWaves + WAVE);
    if constexpr (t.valid) wgrad_flush(acc[SLOT0 + s], gp.p[t.param], t.IN, t.o0, t.i0, t.OUT, t.IN, lane);
  });
}
template <const auto& LIST, int N, int NSLOT, int SLOT0, int NACC>
__device__ __forceinline__ void wgrad_flush_all(const f32x4 (&acc)[NACC], const GradPtrs& gp, int wave, int lane) {
  switch (__builtin_amdgcn_readfirstlane(wave)) {
    case 0: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 0>(acc, gp, lane); break;
    case 1: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 1>(acc, gp, lane); break;
    case 2: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 2>(acc, gp, lane); break;
    default: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 3>(acc, gp, lane); break;
  }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over D features of each of the 16 tokens: kTPT consecutive threads per token.
//   forward : buf[D][kLD] holds the input and receives xhat; out[f] = xhat*gamma + beta (+ res[f] if res); rstd[t]
//   backward: dout[D][kLD] -> din = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dout*gamma   (written to din)
constexpr int kTPT = kBwdThreads / kTT;   // threads per token in the per-token phases
template <int D>
__device__ __forceinline__ void ln_forward(float* buf, float* out, const float* res, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* rstd, int tid) {
  tid = opaque(tid);
  const int tok = tid / kTPT, sub = tid % kTPT;
  float s = 0.f;
  for (int f = sub; f < D; f += kTPT) s += buf[f * kLD + tok];
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  const float mean = s * (1.f / D);
  float q = 0.f;
  for (int f = sub; f < D; f += kTPT) {
    const float c = buf[f * kLD + tok] - mean;
    q = fmaf(c, c, q);
  }
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) q += __shfl_xor(q, d);
  const float rs = 1.f / sqrtf(q * (1.f / D) + 1e-5f);
  if (sub == 0) rstd[tok] = rs;
  for (int f = sub; f < D; f += kTPT) {
    const float xh = (buf[f * kLD + tok] - mean) * rs;
    buf[f * kLD + tok] = xh;
    float y = fmaf(xh, gamma[f], beta[f]);
    if (res) y += res[f * kLD + tok];
    out[f * kLD + tok] = y;
  }
}

template <int D>
__device__ __forceinline__ void ln_backward(const float* dout, const float* xhat, const float* __restrict__ gamma,
                                            const float* rstd, float* din, int tid) {
  tid = opaque(tid);
  const int tok = tid / kTPT, sub = tid % kTPT;
  float s1 = 0.f, s2 = 0.f;
  for (int f = sub; f < D; f += kTPT) {
    const float gg = dout[f * kLD + tok] * gamma[f];
    s1 += gg;
    s2 = fmaf(gg, xhat[f * kLD + tok], s2);
  }
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) {
    s1 += __shfl_xor(s1, d);
    s2 += __shfl_xor(s2, d);
  }
  const float m1 = s1 * (1.f / D), m2 = s2 * (1.f / D), rs = rstd[tok];
  for (int f = sub; f < D; f += kTPT) {
    const float gg = dout[f * kLD + tok] * gamma[f];
    din[f * kLD + tok] = rs * (gg - m1 - xhat[f * kLD + tok] * m2);
  }
}

// per-feature row sums over the tile's tokens, accumulated into a register of thread f (f < D):
//   sum_t a[f][t] (* b[f][t] if b)
__device__ __forceinline__ float row_dot(const float* a, const float* b, int f) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < kTT; ++t) s = b ? fmaf(a[f * kLD + t], b[f * kLD + t], s) : s + a[f * kLD + t];
  return s;
}

__device__ __forceinline__ float elu1_grad(float x) { return x > 0.f ? 1.f : __expf(x); }

}  // namespace ufr
