// Per-frame re-layout of the encoder outputs to channel-last, so that one bilinear / trilinear
// tap of the gather kernel is one contiguous line (feat: 128 B, match: 128*(NV-1) B, volume
// texel: 48 B, rgb: 16 B).  HBM-bound streaming kernels, run once per frame
// (the reference keeps NCHW and lets F.grid_sample stride over channels, model.py:251,370).
#include <cstdlib>
#include <mutex>

#include "ufr_internal.h"
#include "ufr_layout_f16.h"

namespace ufr {

// in: (N,C,S) -> out: (N,S,Cpad).  Reads are coalesced over s for every channel; each thread
// then writes its Cpad-float row with 16-byte stores.
// abs_max (optional): the largest |value| that passes through, as the bit pattern of a non-negative float (an unsigned
// maximum orders those like the floats; NaN patterns lie above +inf, so a NaN or an infinity in the maps surfaces as a
// non-finite bound, which the table derivation reports -- weight_scale_chain).  One atomic per wave.
__device__ __forceinline__ void wave_abs_max(float m, unsigned* __restrict__ abs_max) {
  unsigned b = __builtin_bit_cast(unsigned, m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
  // The word only grows, so a wave whose maximum does not exceed what it READS there has nothing to add -- and after the
  // first few waves that is every wave.  Without the check every wave's atomic went to the one address (123 000 of them per
  // full-resolution volume: same-address atomics serialise in the L2, and volume_pack_kernel took 1.1 ms instead of 0.1;
  // a stale read only costs a redundant atomic).
  if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(abs_max, __ATOMIC_RELAXED)) atomicMax(abs_max, b);
}

template <int CPAD>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int C, int S, unsigned* __restrict__ abs_max) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int n = blockIdx.y;
  float m = 0.f;
  if (s < S) {
    const float* src = in + (size_t)n * C * S + s;
    float v[CPAD];
#pragma unroll
    for (int c = 0; c < CPAD; ++c) v[c] = c < C ? src[(size_t)c * S] : 0.f;
    float4* dst = reinterpret_cast<float4*>(out + ((size_t)n * S + s) * CPAD);
#pragma unroll
    for (int c = 0; c < CPAD / 4; ++c) dst[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    if (abs_max) {
      unsigned mb = 0u;     // integer maximum of the magnitudes' bit patterns: a NaN is not dropped (fmaxf would)
#pragma unroll
      for (int c = 0; c < CPAD; ++c) mb = max(mb, __builtin_bit_cast(unsigned, v[c]) & 0x7fffffffu);
      m = __builtin_bit_cast(float, mb);
    }
  }
  if (abs_max) wave_abs_max(m, abs_max);
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int S, int Cpad, hipStream_t s, unsigned* abs_max) {
  dim3 grid((S + 255) / 256, N), block(256);
  switch (Cpad) {
    case 4: hipLaunchKernelGGL(nchw_to_nhwc_kernel<4>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 32: hipLaunchKernelGGL(nchw_to_nhwc_kernel<32>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 64: hipLaunchKernelGGL(nchw_to_nhwc_kernel<64>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 96: hipLaunchKernelGGL(nchw_to_nhwc_kernel<96>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 128: hipLaunchKernelGGL(nchw_to_nhwc_kernel<128>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 160: hipLaunchKernelGGL(nchw_to_nhwc_kernel<160>, grid, block, 0, s, in, out, C, S, abs_max); break;
    case 192: hipLaunchKernelGGL(nchw_to_nhwc_kernel<192>, grid, block, 0, s, in, out, C, S, abs_max); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// feature volume (N,8,S) + weight volume (N,1,S) -> (N,S,12) = [f0..f7, w, 0, 0, 0]
// (abs_max: the 8 feature channels only -- the weight channel blends them, model.py:376-386, it is not a token feature)
__global__ void __launch_bounds__(256) volume_pack_kernel(const float* __restrict__ feat,
                                                           const float* __restrict__ weight,
                                                           float* __restrict__ out, int S, unsigned* __restrict__ abs_max) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int n = blockIdx.y;
  float m = 0.f;
  if (s < S) {
    const float* f = feat + (size_t)n * 8 * S + s;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = f[(size_t)c * S];
    float w = weight[(size_t)n * S + s];
    float4* dst = reinterpret_cast<float4*>(out + ((size_t)n * S + s) * kVolCh);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    dst[2] = make_float4(w, 0.f, 0.f, 0.f);
    unsigned mb = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) mb = max(mb, __builtin_bit_cast(unsigned, v[c]) & 0x7fffffffu);
    m = __builtin_bit_cast(float, mb);
  }
  if (abs_max) wave_abs_max(m, abs_max);
}

hipError_t launch_volume_pack(const float* feat, const float* weight, float* out, int N, int S, hipStream_t s, unsigned* abs_max) {
  dim3 grid((S + 255) / 256, N), block(256);
  hipLaunchKernelGGL(volume_pack_kernel, grid, block, 0, s, feat, weight, out, S, abs_max);
  return hipGetLastError();
}

// packed[i] = raw[param][elem] per ufr_layout.h:plan_entry (zero for padding), for i in [first, n).  Only the vector
// fragments (biases, LayerNorm, view token) of the fp32 region are read by the kernels: the fp32 A-fragment part in front
// of them (the layout of the first, fp32-MFMA version; still described by ufr_pack_plan for the CPU layout tests) is not
// written on the device any more.
// (like the plane regions below, through a per-device table of (param, elem): walking plan_entry per float took 18 us)
__global__ void __launch_bounds__(256) vec_plan_kernel(int2* __restrict__ plan, int first, int n) {
  int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p, e;
  plan_entry(i, &p, &e);
  plan[i - first] = make_int2(p, e);
}
__global__ void __launch_bounds__(256) pack_weights_kernel(RawPtrs raw, const int2* __restrict__ plan, float* __restrict__ packed,
                                                            int first, int n) {
  int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int2 src = plan[i - first];
  packed[i] = src.x >= 0 ? raw.p[src.x][src.y] : 0.f;
}

// ---- the scale table (ufr_layout.h: scale_table_offset; ufr_layout_f16.h: what s_M and a_M mean).
// Per forward matrix: max |w| and the infinity norm (largest absolute row sum); per vector parameter: max |.|
// (weight_stats_kernel).  One thread then walks the two layer chains once, carrying an upper bound of every dense
// layer's input (weight_scale_kernel):
//   projections     |x| <= X = max(the caller's bound of the feature maps / volume features, the learned view token, the
//                   bound of pre_sim_mlp's output on cosine similarities in [-1, 1])
//   attention       the message is a (sub-)convex combination of the values (the scores Q'.K' are positive):
//                   |msg| <= |v| <= ||W_v||_inf X
//   LayerNorm       |(x - mean) / sigma| <= sqrt(D - 1), so |out| <= max|gamma| sqrt(D - 1) + max|beta|
//   ReLU MLPs       |W x + b| <= ||W||_inf |x| + max|b|
//   residual        |x + LN(.)| <= X + the LayerNorm bound
// (ray transformer: its tokens are [view-transformer output of token 0 | order encoding], |PE| <= 1).  The bounds are
// pessimistic by the usual gap between an infinity norm and a typical gain (3..7 bits here) -- which costs nothing: the
// planes keep 22 significand bits for everything above 2^-17 of the bound.
// A non-finite parameter raises bit 2 of the sticky status (include/ufr.h) and leaves default exponents.
namespace {
// largest exponent in [lo, hi] with 2^e bound <= 2^15 (bound > 0, finite; fp16 holds up to 65504)
__device__ int plane_exponent(float bound, int lo, int hi) {
  int e;
  const float f = frexpf(bound, &e);   // bound = f 2^e, f in [0.5, 1)
  return max(lo, min(hi, f == 0.5f ? 16 - e : 15 - e));
}
}  // namespace

// first kernel: one workgroup per matrix (its max |w| and infinity norm) and per vector parameter (max |.|); the numbers
// go to the table's own tail (the kernels' scalar lists, which the second kernel overwrites after reading them) --
// training re-packs after every optimizer step, a single workgroup walking all 33 parameters took 0.3 ms
constexpr int kStatVecs = 18;
__device__ constexpr int kStatVecParam[kStatVecs][2] = {{P_VT_N1W, 80}, {P_VT_N1B, 80}, {P_VT_N2W, 80}, {P_VT_N2B, 80}, {P_RT_N1W, 88},
                                                        {P_RT_N1B, 88}, {P_RT_N2W, 88}, {P_RT_N2B, 88}, {P_DM_B0, 32}, {P_DM_B2, 16},
                                                        {P_DM_B4, 1}, {P_RW_B0, 16}, {P_RW_B2, 8}, {P_RW_B4, 1}, {P_VIEW_TOKEN, 80},
                                                        {P_PS_B0, 32}, {P_PS_B2, 32}, {P_PS_B4, 16}};
// statistics "matrices": the M_COUNT dense matrices of the two transformers, then the three layers of pre_sim_mlp (8 -> 32
// -> 32 -> 16, evaluated in fp32 by the gather kernel: only their infinity norms matter -- they bound the 16 pre-similarity
// features among the token inputs)
constexpr int kStatMats = M_COUNT + 3;
__device__ inline void stat_mat(int b, int* param, int* k_raw, int* out_dim) {
  if (b < M_COUNT) {
    const MatDesc d = mat_desc(b);
    *param = d.param; *k_raw = d.k_raw; *out_dim = d.out_dim;
  } else {
    *param = b == M_COUNT ? P_PS_W0 : b == M_COUNT + 1 ? P_PS_W2 : P_PS_W4;
    *k_raw = b == M_COUNT ? 8 : 32;
    *out_dim = b == M_COUNT + 2 ? 16 : 32;
  }
}
// layout of the statistics: max |w| [kStatMats] | infinity norm [kStatMats] | vector maxima [kStatVecs]
static_assert(2 * kStatMats + kStatVecs <= kStatBoundSlot && kStatBoundSlot < kStatFloats, "the statistics region of the scale table (ufr_layout.h)");
constexpr int kStatThreads = 1024;
__global__ void __launch_bounds__(kStatThreads) weight_stats_kernel(RawPtrs raw, float* __restrict__ stats, int* __restrict__ flag) {
  __shared__ float red[3][kStatThreads / 64];
  __shared__ float rowsum[176];
  __shared__ float part[8 * kStatThreads];      // one slot per float4 of the largest matrix (176 x 176 = 7 744 of them)
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float m = 0.f, rs = 0.f;
  bool nf = false;
  if (b < kStatMats) {
    struct { int param, k_raw, out_dim; } d;
    stat_mat(b, &d.param, &d.k_raw, &d.out_dim);
    const float4* w4 = reinterpret_cast<const float4*>(raw.p[d.param]);
    for (int r = threadIdx.x; r < d.out_dim; r += kStatThreads) rowsum[r] = 0.f;
    __syncthreads();
    // The parameters were just written by the optimizer, on other XCDs: every read is a round trip past the L2 (~2 us).
    // So all of a thread's reads are issued before any is used: 8 independent 16-byte loads cover the largest matrix
    // (176 x 176) with 1024 threads -- a loop that waits per element took 56 us, per step.
    const int n4 = d.out_dim * d.k_raw / 4;     // every matrix's element count is a multiple of 4
    const bool quads = d.k_raw % 4 == 0;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i4 = threadIdx.x + u * kStatThreads;
      v[u] = w4[i4 < n4 ? i4 : n4 - 1];       // unconditional (a branch per load serialises them), masked below
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i4 = threadIdx.x + u * kStatThreads;
      const bool in = i4 < n4;
      const float e[4] = {in ? fabsf(v[u].x) : 0.f, in ? fabsf(v[u].y) : 0.f, in ? fabsf(v[u].z) : 0.f, in ? fabsf(v[u].w) : 0.f};
      const float s4 = (e[0] + e[1]) + (e[2] + e[3]);
      nf |= !(s4 <= 3.0e38f);
      m = fmaxf(fmaxf(m, fmaxf(e[0], e[1])), fmaxf(e[2], e[3]));
      // |w| into its row's sum.  Rows that are whole float4s (every matrix but the 83-column rw0): the float4's sum goes to
      // its own LDS slot and one thread per row adds the row's slots afterwards, in order -- no atomics (per element they were
      // 32 LDS atomics per thread onto <= 176 addresses: most of this kernel's 55 us, on the critical path of every training
      // step's re-pack).  Otherwise: a wave whose 256 elements lie in one row adds once, the others per element.
      if (quads) {
        if (i4 < n4) part[i4] = s4;
        continue;
      }
      const int r0 = (4 * i4) / d.k_raw, r3 = (4 * i4 + 3) / d.k_raw;
      const bool one_row = i4 < n4 && r0 == r3 && r0 == __shfl(r0, 0);
      if (__builtin_amdgcn_ballot_w64(one_row) == ~0ull) {
        float t = s4;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) atomicAdd(&rowsum[r0], t);
      } else if (i4 < n4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&rowsum[(4 * i4 + k) / d.k_raw], e[k]);
      }
    }
    __syncthreads();
    if (quads) {
      const int q = d.k_raw / 4;
      for (int r = threadIdx.x; r < d.out_dim; r += kStatThreads) {
        float t = 0.f;
        for (int k = 0; k < q; ++k) t += part[r * q + k];
        rs = fmaxf(rs, t);
      }
    } else {
      for (int r = threadIdx.x; r < d.out_dim; r += kStatThreads) rs = fmaxf(rs, rowsum[r]);
    }
  } else {
    const int v = b - kStatMats;
    const float* p = raw.p[kStatVecParam[v][0]];
    for (int i = threadIdx.x; i < kStatVecParam[v][1]; i += kStatThreads) {
      const float a = fabsf(p[i]);
      nf |= !(a <= 3.0e38f);
      m = fmaxf(m, a);
    }
  }
  // block maxima of (non-finite seen, max |w|, largest row sum): one round
  float t[3] = {nf ? 1.f : 0.f, m, rs};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int o = 32; o > 0; o >>= 1) t[q] = fmaxf(t[q], __shfl_xor(t[q], o));
    if (lane == 0) red[q][wave] = t[q];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float r = red[threadIdx.x][0];
#pragma unroll
    for (int i = 1; i < kStatThreads / 64; ++i) r = fmaxf(r, red[threadIdx.x][i]);
    if (threadIdx.x == 0 && r != 0.f) atomicOr(flag, 4);
    if (threadIdx.x == 1) stats[b < kStatMats ? b : 2 * kStatMats + (b - kStatMats)] = r;
    if (threadIdx.x == 2 && b < kStatMats) stats[kStatMats + b] = r;
  }
}

// second kernel: the chain of bounds, the exponents, the table (one thread computes, the wave loads and stores).
// frame_bound == nullptr: the pack -- the table for the caller's bound x_max.  Otherwise the REFIT for a frame
// (ufr_weights_fit_frame): *frame_bound is the frame's measured feature bound (ufr_frame_prepare); nothing happens unless it
// exceeds the bound the table serves, else the table is re-derived for the next power of two above it.  The bound only
// grows between two packs, so a forward and its backward see compatible tables whatever frames come in between.
__device__ void weight_scale_chain(const float* stats, float* table, float x_max, int* __restrict__ flag, int fixed);
__global__ void __launch_bounds__(64) weight_scale_kernel(float* __restrict__ table_out, float x_max, const unsigned* __restrict__ frame_bound,
                                                          int* __restrict__ flag, int fixed) {
  __shared__ float stats[kStatFloats];
  __shared__ float table[kScaleFloats];          // built by thread 0, written out by the wave
  constexpr int stats_at = stats_offset() - scale_table_offset();   // (constexpr: or the offset walk runs on the device)
  constexpr int n_table = stats_at;                                 // everything in front of the statistics is derived
  for (int i = threadIdx.x; i < kStatFloats; i += 64) stats[i] = table_out[stats_at + i];
  __syncthreads();
  if (frame_bound) {
    const float fb = __builtin_bit_cast(float, *frame_bound);       // NaN / inf pass: the chain reports them (bit 2)
    if (fb <= stats[kStatBoundSlot]) return;                        // (uniform) the table already covers this frame
    int e;
    const float f = frexpf(fb, &e);
    x_max = (fb <= 3.0e38f) ? (f == 0.5f ? fb : ldexpf(1.f, e)) : fb;
  }
  if (threadIdx.x == 0) weight_scale_chain(stats, table, x_max, flag, fixed);
  __syncthreads();
  for (int i = threadIdx.x; i < n_table; i += 64) table_out[i] = table[i];
  if (threadIdx.x == 0) table_out[stats_at + kStatBoundSlot] = x_max;
}
__device__ void weight_scale_chain(const float* stats, float* table, float x_max, int* __restrict__ flag, int fixed) {
  float wmax[kStatMats], ninf[kStatMats], vmax[P_COUNT];
  bool bad = false;
  // (every loop below is unrolled: the arrays are indexed with constants and stay in registers -- as private memory they
  // cost a 50 us kernel)
#pragma unroll
  for (int m = 0; m < kStatMats; ++m) {
    wmax[m] = stats[m];
    ninf[m] = stats[kStatMats + m];
    bad |= !(wmax[m] <= 3.0e38f) || !(ninf[m] <= 3.0e38f);
  }
#pragma unroll
  for (int v = 0; v < kStatVecs; ++v) {
    vmax[kStatVecParam[v][0]] = stats[2 * kStatMats + v];
    bad |= !(stats[2 * kStatMats + v] <= 3.0e38f);
  }
  bad |= !(x_max > 0.f && x_max <= 3.0e38f);
  float in[M_COUNT];
  // the token features: what the caller bounds (feature maps, volume features: x_max), the learned view token, and the 16
  // pre-similarity features = pre_sim_mlp of 8 mean cosine similarities in [-1, 1]
  const float ps1 = ninf[M_COUNT] + vmax[P_PS_B0], ps2 = ninf[M_COUNT + 1] * ps1 + vmax[P_PS_B2];
  const float ps3 = ninf[M_COUNT + 2] * ps2 + vmax[P_PS_B4];
  const float X = fmaxf(fmaxf(x_max, vmax[P_VIEW_TOKEN]), ps3);
  // q, k, v and mlp0 (whose input is [x | LayerNorm1 output]) all split the token x: ONE exponent for the four, so the
  // kernels split x once per phase with the same multiplier (and the compiler merges the repeats)
  const float vm = vmax[P_VT_N1W] * sqrtf(79.f) + vmax[P_VT_N1B];
  in[M_VT_Q] = in[M_VT_K] = in[M_VT_V] = in[M_VT_MLP0] = fmaxf(X, vm);
  in[M_VT_MERGE] = ninf[M_VT_V] * X;
  in[M_VT_MLP2] = ninf[M_VT_MLP0] * in[M_VT_MLP0];
  const float Y = X + vmax[P_VT_N2W] * sqrtf(79.f) + vmax[P_VT_N2B];
  in[M_RW0] = fmaxf(Y, 1.f);                                   // [y | unit direction]
  in[M_RW2] = ninf[M_RW0] * in[M_RW0] + vmax[P_RW_B0];
  in[M_RW4] = ninf[M_RW2] * in[M_RW2] + vmax[P_RW_B2];
  const float XR = fmaxf(Y, 1.f);                              // [y of token 0 | order encoding]
  const float rm = vmax[P_RT_N1W] * sqrtf(87.f) + vmax[P_RT_N1B];
  in[M_RT_Q] = in[M_RT_K] = in[M_RT_V] = in[M_RT_MLP0] = fmaxf(XR, rm);
  in[M_RT_MERGE] = ninf[M_RT_V] * XR;
  in[M_RT_MLP2] = ninf[M_RT_MLP0] * in[M_RT_MLP0];
  const float O = XR + vmax[P_RT_N2W] * sqrtf(87.f) + vmax[P_RT_N2B];
  in[M_DM0] = O;
  in[M_DM2] = ninf[M_DM0] * O + vmax[P_DM_B0];
  in[M_DM4] = ninf[M_DM2] * in[M_DM2] + vmax[P_DM_B2];
#pragma unroll
  for (int m = 0; m < M_COUNT; ++m) bad |= !(in[m] <= 3.0e38f);
  if (bad) atomicOr(flag, 4);
#pragma unroll
  for (int m = 0; m < M_COUNT; ++m) {
    // (matrices that share one split of their input -- q / k / v / mlp0 of either transformer -- get the same a_M: it
    // depends on the input's bound only)
    // fixed (UFR_DEBUG_FIXED_SCALES=1, ablation): round 3's constants for every matrix
    const int sw = (!fixed && !bad && wmax[m] > 0.f) ? plane_exponent(wmax[m], kScaleExpWMin, kScaleExpWMax) : 8;
    const int ax = (!fixed && !bad && in[m] > 0.f) ? plane_exponent(in[m], kScaleExpXMin, kScaleExpXMax) : 4;
    table[4 * m + 0] = ldexpf(1.f, ax);
    table[4 * m + 1] = ldexpf(1.f, -(sw + ax));
    table[4 * m + 2] = ldexpf(1.f, sw + ax);
    table[4 * m + 3] = ldexpf(1.f, sw);
  }
  // the kernels' scalar lists (ufr_layout.h: ViewScalar / RayScalar); every product below is a product of powers of two
  // (exact), except the two constants 1e-5 and log2(e)
  auto xs = [&](int m) { return table[4 * m + 0]; };
  auto dsc = [&](int m) { return table[4 * m + 1]; };
  auto asc = [&](int m) { return table[4 * m + 2]; };
  constexpr float l2e = 0x1.715476p+0f;
  constexpr int vs_at = view_scalars_offset() - scale_table_offset(), rs_at = ray_scalars_offset() - scale_table_offset();
  float* vs = table + vs_at;
  float* rs = table + rs_at;
  for (int i = 0; i < 2 * kKernelScalars; ++i) vs[i] = 0.f;     // (the two lists are adjacent)
  vs[VS_XS_X] = xs(M_VT_Q);
  vs[VS_Q_DSC] = dsc(M_VT_Q);     vs[VS_Q_L2E] = dsc(M_VT_Q) * l2e;
  vs[VS_K_DSC] = dsc(M_VT_K);     vs[VS_K_L2E] = dsc(M_VT_K) * l2e;
  vs[VS_V_DSC] = dsc(M_VT_V);
  vs[VS_M_XS] = xs(M_VT_MERGE);   vs[VS_EPS1] = 1e-5f * asc(M_VT_MERGE) * asc(M_VT_MERGE);   vs[VS_M_ASC] = asc(M_VT_MERGE);
  vs[VS_MLP0_DSC] = dsc(M_VT_MLP0);
  vs[VS_M_MLP2] = xs(M_VT_MLP2) * dsc(M_VT_MLP0);
  vs[VS_EPS2] = 1e-5f * asc(M_VT_MLP2) * asc(M_VT_MLP2);   vs[VS_MLP2_ASC] = asc(M_VT_MLP2);
  vs[VS_RW0_XS] = xs(M_RW0);      vs[VS_RW0_ASC] = asc(M_RW0);   vs[VS_RW0_DSC] = dsc(M_RW0);
  vs[VS_M_RW2] = xs(M_RW2) * dsc(M_RW0);   vs[VS_RW2_ASC] = asc(M_RW2);   vs[VS_RW2_DSC] = dsc(M_RW2);
  vs[VS_M_RW4] = xs(M_RW4) * dsc(M_RW2);   vs[VS_RW4_ASC] = asc(M_RW4);   vs[VS_RW4_DSC] = dsc(M_RW4);
  rs[RS_XS_X] = xs(M_RT_K);
  rs[RS_K_DSC] = dsc(M_RT_K);     rs[RS_K_L2E] = dsc(M_RT_K) * l2e;
  rs[RS_V_DSC] = dsc(M_RT_V);     rs[RS_V_ASC] = asc(M_RT_V);
  rs[RS_Q_DSC] = dsc(M_RT_Q);     rs[RS_Q_L2E] = dsc(M_RT_Q) * l2e;
  rs[RS_M_XS] = xs(M_RT_MERGE);   rs[RS_EPS1] = 1e-5f * asc(M_RT_MERGE) * asc(M_RT_MERGE);   rs[RS_M_ASC] = asc(M_RT_MERGE);
  rs[RS_MLP0_DSC] = dsc(M_RT_MLP0);
  rs[RS_M_MLP2] = xs(M_RT_MLP2) * dsc(M_RT_MLP0);
  rs[RS_EPS2] = 1e-5f * asc(M_RT_MLP2) * asc(M_RT_MLP2);   rs[RS_MLP2_ASC] = asc(M_RT_MLP2);
  rs[RS_DM0_XS] = xs(M_DM0);      rs[RS_DM0_ASC] = asc(M_DM0);   rs[RS_DM0_DSC] = dsc(M_DM0);
  rs[RS_M_DM2] = xs(M_DM2) * dsc(M_DM0);   rs[RS_DM2_ASC] = asc(M_DM2);   rs[RS_DM2_DSC] = dsc(M_DM2);
  rs[RS_M_DM4] = xs(M_DM4) * dsc(M_DM2);   rs[RS_DM4_ASC] = asc(M_DM4);   rs[RS_DM4_DSC] = dsc(M_DM4);
}

// Plane regions (ufr_layout_f16.h).  Forward streams: halfword h = fp16 plane p of 2^s_M * raw[param][elem] (2^s_M from the
// scale table); hi = fp16(w'), lo = fp16(w' - hi), both round-to-nearest-even (hi + lo carries 22+ significand bits of w').
// Backward streams: bf16 planes of the unscaled weight.
// Where a halfword comes from is a pure function of the layout: it is evaluated ONCE per device into a table (param, elem,
// plane, matrix), and a pack is a gather through that table -- training re-packs after every optimizer step, and
// walking the panel lists per halfword (plan_entry_f16) cost 0.41 ms per step once the backward streams had tripled the region.
struct PlaneSrc { int elem; short param; unsigned char plane, mat; };
static_assert(sizeof(PlaneSrc) == 8, "one 8-byte entry per halfword");

__global__ void __launch_bounds__(256) plane_plan_kernel(PlaneSrc* __restrict__ plan, int n) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  int p, e, plane, bf, mat;
  plan_entry_f16(h, &p, &e, &plane, &bf, &mat);
  plan[h] = PlaneSrc{e, (short)p, (unsigned char)plane, (unsigned char)(mat < 0 ? 0 : mat)};
}

__global__ void __launch_bounds__(256) pack_weights_f16_kernel(RawPtrs raw, const PlaneSrc* __restrict__ plan,
                                                                const float* __restrict__ table,
                                                                unsigned short* __restrict__ packed, int n) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  const PlaneSrc src = plan[h];
  unsigned short out = 0;
  if (src.param >= 0) {
    if (f16_mat_is_bf16(src.mat)) {
      const float w = raw.p[src.param][src.elem];
      const __bf16 hi = (__bf16)w;
      const __bf16 lo = (__bf16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, src.plane == 0 ? hi : lo);
    } else {
      const float w = raw.p[src.param][src.elem] * table[4 * src.mat + 3];
      const _Float16 hi = (_Float16)w;
      const _Float16 lo = (_Float16)(w - (float)hi);
      out = __builtin_bit_cast(unsigned short, src.plane == 0 ? hi : lo);
    }
  }
  packed[h] = out;
}

namespace {
struct PlanCache {
  std::once_flag once[16];
  hipError_t err[16];
  PlaneSrc* plan[16];
  int2* vec_plan[16];
};
PlanCache g_plan;
}  // namespace

hipError_t launch_pack_weights(const RawPtrs& raw, float* packed, float input_abs_max, int* range_flag, hipStream_t s) {
  const int n = blob_floats(), first = vec_region_offset(), n_vec = scale_table_offset();
  constexpr int n_half = kF16Halfwords + kBwdHalfwords;   // forward fp16 planes, then the backward kernels' bf16 planes
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  std::call_once(g_plan.once[dev], [&] {      // process lifetime, 8 bytes per halfword (~13 MB)
    g_plan.err[dev] = hipMalloc(reinterpret_cast<void**>(&g_plan.plan[dev]), sizeof(PlaneSrc) * (size_t)n_half);
    if (g_plan.err[dev] != hipSuccess) return;
    g_plan.err[dev] = hipMalloc(reinterpret_cast<void**>(&g_plan.vec_plan[dev]), sizeof(int2) * (size_t)(n_vec - first));
    if (g_plan.err[dev] != hipSuccess) return;
    hipLaunchKernelGGL(vec_plan_kernel, dim3((n_vec - first + 255) / 256), dim3(256), 0, s, g_plan.vec_plan[dev], first, n_vec);
    hipLaunchKernelGGL(plane_plan_kernel, dim3((n_half + 255) / 256), dim3(256), 0, s, g_plan.plan[dev], n_half);
    g_plan.err[dev] = hipGetLastError();
    if (g_plan.err[dev] == hipSuccess) g_plan.err[dev] = hipStreamSynchronize(s);   // other streams may pack next
  });
  if (g_plan.err[dev] != hipSuccess) return g_plan.err[dev];
  hipLaunchKernelGGL(pack_weights_kernel, dim3((n_vec - first + 255) / 256), dim3(256), 0, s, raw, g_plan.vec_plan[dev], packed, first,
                     n_vec);
  constexpr int table_at = scale_table_offset(), stats_at = stats_offset();
  float* table = packed + table_at;
  static const int fixed = [] { const char* e = getenv("UFR_DEBUG_FIXED_SCALES"); return (e && e[0] == '1') ? 1 : 0; }();
  hipLaunchKernelGGL(weight_stats_kernel, dim3(kStatMats + kStatVecs), dim3(kStatThreads), 0, s, raw,
                     packed + stats_at, range_flag);
  hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(64), 0, s, table, input_abs_max, (const unsigned*)nullptr, range_flag, fixed);
  unsigned short* planes = reinterpret_cast<unsigned short*>(packed + n);
  hipLaunchKernelGGL(pack_weights_f16_kernel, dim3((n_half + 255) / 256), dim3(256), 0, s, raw, g_plan.plan[dev], table, planes,
                     n_half);
  return hipGetLastError();
}


// ufr_weights_fit_frame: the table of an already packed blob follows a frame's measured feature bound (weight_scale_kernel)
hipError_t launch_refit_weights(float* packed, const unsigned* frame_bound, int* range_flag, hipStream_t s) {
  constexpr int table_at = scale_table_offset();
  static const int fixed = [] { const char* e = getenv("UFR_DEBUG_FIXED_SCALES"); return (e && e[0] == '1') ? 1 : 0; }();
  hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(64), 0, s, packed + table_at, 0.f, frame_bound, range_flag, fixed);
  return hipGetLastError();
}

}  // namespace ufr
