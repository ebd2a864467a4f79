// Backward of the cross-view aggregation (view_transformer.hip): recomputes the layer for a tile of 16 token
// columns (PPT = 16/L points x L = NV+1 tokens) from the saved token inputs, then walks it backwards.
//   RayTransformer.forward      code1/ray_transformer.py:283-294, 309-320   (autograd of these lines)
//   LoFTREncoderLayer.forward   code1/attention/transformer.py:35-58
//   LinearAttention.forward     code1/attention/linear_attention.py:20-47
// Inputs of the backward: d token0 (two partial buffers from ray_bwd.hip) and d radiance (compositor).
// Outputs: gradients of every view-transformer / radiance-MLP parameter and of the view token (float atomics into the
// reference-layout gradient tensors, once per workgroup), and d_pv (P,40) = the gradient w.r.t. the 24 frustum
// features and the 16 pre_sim_mlp outputs of each point, summed over its NV view tokens (gather_bwd.hip consumes it).
// Machinery: bwd_common.h (LDS-resident fp32 tile, fp32 MFMA, register-resident weight-gradient tiles).
#include "bwd_common.h"
#include "ufr_internal.h"

namespace ufr {

namespace vb {
// LDS rows (each kLD floats).  Buffers whose live ranges do not overlap share rows; the walk through a tile, with the
// phase that writes (W) / last reads (R) every buffer:
//   CAT   x | m                W P0 / P4     R B10 (weight gradients of q, k, v), B5 (of mlp0)
//   Q K V                      W P1          R B9
//   MSG                        W P2          R B7 (weight gradient of merge)          then  DV   W B9  R B10
//   XH1   merge out -> xhat1   W P3 / P4     R B6                                      then  DQ   W B8  R B10
//   HID   relu(mlp0)           W P5          R W1, B4 -> DHID in place (W B4, R B5)    then  DMPRE (rows 0..79, W B6, R B7) | DMSG (80..159, W B7, R B9)
//   XH2   mlp2 out -> xhat2    W P6 / P7     R B3                                      then  DK   W B9  R B10
//   RIN   y | dir              W P7 / P0     R B2 (weight gradient of rw0)             then  DOPRE (W B3, R B4)  then  DCATm (W B5, R B6)
//   H1 H2 DH1 DH2 DY, scalars  not shared
enum : int {
  O_CAT = 0,        // 160: x (0..79) | m = LN1(merge(msg)) (80..159)        transformer.py:55
  O_Q = 160, O_K = 240, O_V = 320,
  O_MSG = 400, O_DV = O_MSG,
  O_XH1 = 480, O_DQ = O_XH1,
  O_HID = 560, O_DHID = O_HID, O_DMPRE = O_HID, O_DMSG = O_HID + 80,
  O_XH2 = 720, O_DK = O_XH2,
  O_RIN = 800,      // 83 (+1 pad): layer output y (0..79) | dir (80..82)     ray_transformer.py:311-313
  O_DOPRE = O_RIN, O_DCATM = O_RIN,
  O_H1 = 884,       // 16
  O_H2 = 900,       // 8
  O_DH1 = 908,      // 16
  O_DH2 = 924,      // 8
  O_DY = 932,       // 80: d y, later the d x accumulator
  O_RSTD1 = 1012, O_RSTD2 = 1013, O_DLOGIT = 1014,
  O_U = 1015,       // 8 heads: L / (Q'.sum K' + eps) of (token, head)
  O_DDEN = 1023,    // 8 heads
  O_END = 1031
};
// flat regions behind the tile rows (floats): small parameters copied once per workgroup, and the per-tile colours / masks /
// d radiance of the tile's points (prefetched with the tokens)
enum : int {
  F_N1W = 0, F_N1B = 80, F_N2W = 160, F_N2B = 240,     // LayerNorm gamma / beta
  F_RW_B0 = 320, F_RW_W2 = 336, F_RW_B2 = 464, F_RW_W4 = 472, F_RW_B4 = 480,
  F_RGBM = 484,               // [PPT][NV][4]  (PPT * NV < kTT)
  F_DRAD = 484 + 4 * kTT,     // [PPT][3]
  F_END = 484 + 4 * kTT + kTT
};
constexpr int kFlatBase = (O_END * kLD + 3) / 4 * 4;   // 16-byte aligned (float4 reads of the colours)
constexpr int kLdsBytes = (kFlatBase + F_END) * 4;
static_assert(kLdsBytes <= 160 * 1024, "tile does not fit the CU's LDS");

// gradient tiles in the order their dY becomes final: rw0 | mlp2 | mlp0 | merge | q k v
constexpr WgMat kMats[] = {
    {P_RW_W0, 16, 83, O_DH1, O_RIN},      {P_VT_MLP2, 80, 160, O_DOPRE, O_HID},  {P_VT_MLP0, 160, 160, O_DHID, O_CAT},
    {P_VT_MERGE, 80, 80, O_DMPRE, O_MSG}, {P_VT_Q, 80, 80, O_DQ, O_CAT},         {P_VT_K, 80, 80, O_DK, O_CAT},
    {P_VT_V, 80, 80, O_DV, O_CAT}};
constexpr auto kList = make_wglist(kMats);
constexpr int kSlots = (kList.first[7] + kBwdWaves - 1) / kBwdWaves;   // 256 tiles -> 64 slots
constexpr int T_RW0 = kList.first[0], T_MLP2 = kList.first[1], T_MLP0 = kList.first[2], T_MERGE = kList.first[3],
              T_QKV = kList.first[4], T_END = kList.first[7];
constexpr int kDbgCols = 881;
}  // namespace vb

#ifdef UFR_BWD_TIMING
__device__ unsigned long long g_vb_phase[64];
#endif

template <bool LOWP>
__global__ void __launch_bounds__(kBwdThreads) view_bwd_kernel(RawPtrs wp, GradPtrs gp, const float* __restrict__ x_tokens,
                                                               const float* __restrict__ rgbm,
                                                               const float* __restrict__ dirs,
                                                               const float* __restrict__ d_tok_a,
                                                               const float* __restrict__ d_tok_b,
                                                               const float* __restrict__ d_radiance, int P, int NV,
                                                               float* __restrict__ d_pv, float* __restrict__ dbg) {
  using namespace vb;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid0 = threadIdx.x, wave = tid0 >> 6, lane = tid0 & 63;
  int tid = tid0;   // re-laundered after every barrier (bwd_common.h: opaque)
  const int L = NV + 1, PPT = kTT / L;
  const int n_tiles = (P + PPT - 1) / PPT;
  const float invL = 1.f / (float)L;

  f32x4 acc[kSlots];
  const auto wg_tab = wgrad_table<vb::kList, 7, kSlots>(wave, lane);
#pragma unroll
  for (int s = 0; s < kSlots; ++s) acc[s] = splat4(0.f);
  float accA = 0.f, accB = 0.f, accC = 0.f, accN1 = 0.f, accN2 = 0.f;   // small gradients, a few scalars per thread (see the flush)

  auto R = [&](int row) -> float* { return lds + row * kLD; };
  float* flat = lds + kFlatBase;
  {  // small parameters: read by every tile's VALU phases, so they live in LDS (a global load there is an exposed L2 trip)
    const int i = tid;
    if (i < 80) {
      flat[F_N1W + i] = wp.p[P_VT_N1W][i]; flat[F_N1B + i] = wp.p[P_VT_N1B][i];
      flat[F_N2W + i] = wp.p[P_VT_N2W][i]; flat[F_N2B + i] = wp.p[P_VT_N2B][i];
    }
    if (i < 128) flat[F_RW_W2 + i] = wp.p[P_RW_W2][i];
    if (i < 16) flat[F_RW_B0 + i] = wp.p[P_RW_B0][i];
    if (i < 8) { flat[F_RW_B2 + i] = wp.p[P_RW_B2][i]; flat[F_RW_W4 + i] = wp.p[P_RW_W4][i]; }
    if (i == 0) flat[F_RW_B4] = wp.p[P_RW_B4][0];
  }

  // A tile's global inputs (tokens, staged d token0, dir, colours / masks, d radiance) are fetched into registers one
  // tile AHEAD -- the loads are issued inside the last GEMM phase of the previous tile -- and committed to LDS at
  // the top of the tile: the HBM latency hides behind that phase's MFMAs instead of opening every tile.
  // The loads are branch-free (clamped / redirected addresses, selects at commit time) into plain register arrays with
  // compile-time indices: as a struct filled under divergent branches the compiler kept the buffer in scratch memory, and
  // every load was followed by s_waitcnt vmcnt(0) + scratch_store -- a dozen serial HBM round trips, 15 k cycles per tile.
  constexpr int kTokLoads = (kTT * 20 + kBwdThreads - 1) / kBwdThreads;   // float4 token loads per thread
  f32x4 in_v[kTokLoads], in_a[kTokLoads], in_b[kTokLoads];
  float in_misc = 0.f;
  // what token-load slot q of this thread holds: column, float4 index, point, token, validity
  // (`t` is the thread index laundered by the caller: everything derived from the raw index is invariant over the tile
  // loop, gets hoisted out of it, spilled, and reloaded from scratch one value at a time -- 12 k cycles per tile)
  auto tok_slot = [&](int t, int q, int p0, int& col, int& f4, int& tv, int& p, bool& ok) __attribute__((always_inline)) {
    const int idx = t + q * kBwdThreads;
    col = idx / 20;
    f4 = idx - col * 20;
    const int pt = col / L;
    tv = col - pt * L;
    p = p0 + pt;
    ok = idx < kTT * 20 && pt < PPT && p < P;
  };
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int p0 = tile * PPT;
    const int t_l = opaque(tid0);
    static_for<kTokLoads>([&](auto qi) __attribute__((always_inline)) {
      constexpr int q = decltype(qi)::value;
      int col, f4, tv, p;
      bool ok;
      tok_slot(t_l, q, p0, col, f4, tv, p, ok);
      const size_t pc = ok ? (size_t)p : 0;
      const float* vt = wp.p[P_VIEW_TOKEN] + 4 * f4;
      const float* src_v = tv == 0 ? vt : x_tokens + (pc * NV + (tv > 0 ? tv - 1 : 0)) * UFR_TOKEN_DIM + 4 * f4;
      const float* src_a = tv == 0 ? d_tok_a + pc * UFR_TOKEN_DIM + 4 * f4 : vt;      // d token0 exists for token 0 only
      const float* src_b = (tv == 0 && d_tok_b) ? d_tok_b + pc * UFR_TOKEN_DIM + 4 * f4 : vt;
      in_v[q] = ld4(src_v);
      in_a[q] = ld4(src_a);
      in_b[q] = ld4(src_b);
    });
    // one float per thread: [0, 3 kTT) dir of (col, e); then PPT*NV*4 colours / masks; then PPT*3 d radiance
    const int i = t_l;
    const float* src = dirs;   // any valid address for the threads without a value
    if (i < kTT * 3) {
      const int col = i / 3, e = i - col * 3, pt = col / L, tv = col - pt * L, p = p0 + pt;
      if (pt < PPT && p < P && tv > 0) src = dirs + ((size_t)p * NV + tv - 1) * 4 + e;
    } else if (i < kTT * 3 + PPT * NV * 4) {
      const int k = i - kTT * 3, pt = k / (NV * 4), p = p0 + pt;
      if (p < P) src = rgbm + (size_t)p * NV * 4 + (k - pt * NV * 4);
    } else if (i < kTT * 3 + PPT * NV * 4 + PPT * 3) {
      const int k = i - kTT * 3 - PPT * NV * 4, pt = k / 3, p = p0 + pt;
      if (p < P) src = d_radiance + (size_t)p * 3 + (k - pt * 3);
    }
    in_misc = *src;
  };
  static_assert(kTT * 3 + 4 * kTT + kTT <= kBwdThreads, "one misc value per thread");   // PPT*NV < kTT, PPT*3 <= kTT
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int p0 = tile * PPT;
    const int t_l = opaque(tid0);
    static_for<kTokLoads>([&](auto qi) __attribute__((always_inline)) {
      constexpr int q = decltype(qi)::value;
      int col, f4, tv, p;
      bool ok;
      tok_slot(t_l, q, p0, col, f4, tv, p, ok);
      if (t_l + q * kBwdThreads < kTT * 20) {
        const bool tok0 = ok && tv == 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          R(O_CAT + 4 * f4 + e)[col] = ok ? in_v[q][e] : 0.f;
          R(O_DY + 4 * f4 + e)[col] = tok0 ? in_a[q][e] + (d_tok_b ? in_b[q][e] : 0.f) : 0.f;
        }
      }
    });
    const int i = t_l;
    float misc = 0.f;   // same validity conditions as in fetch
    if (i < kTT * 3) {
      const int col = i / 3, pt = col / L, tv = col - pt * L;
      if (pt < PPT && p0 + pt < P && tv > 0) misc = in_misc;
      R(O_RIN + 80 + i % 3)[i / 3] = misc;
    } else if (i < kTT * 3 + PPT * NV * 4) {
      const int k = i - kTT * 3, pt = k / (NV * 4);
      flat[F_RGBM + k] = p0 + pt < P ? in_misc : 0.f;
    } else if (i < kTT * 3 + PPT * NV * 4 + PPT * 3) {
      const int k = i - kTT * 3 - PPT * NV * 4, pt = k / 3;
      flat[F_DRAD + k] = p0 + pt < P ? in_misc : 0.f;
    }
  };
  // development dump of intermediate gradients (tools/dev): rows [row0, row0 + n) -> columns [k0, k0 + n) of the tile's tokens
  auto dump = [&](int p0, int row0, int n, int k0) {
    for (int idx = tid; idx < kTT * n; idx += kBwdThreads) {
      const int col = idx / n, k = idx - col * n;
      const int pt = col / L, tv = col - pt * L, p = p0 + pt;
      if (pt < PPT && p < P) dbg[((size_t)p * L + tv) * kDbgCols + k0 + k] = R(row0 + k)[col];
    }
  };

#ifdef UFR_BWD_TIMING
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int p0 = tile * PPT;
    // ---------------- P0: token inputs (ray_transformer.py:284-286), dir, staged d token0, colours, d radiance
    commit(tile);
    auto pf0 = gemm_prefetch<80, 80, false>(wp.p[P_VT_Q], 80, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 0)

    // ---------------- P1: Q' = elu(q)+1, K' = elu(k)+1, v (15 row tiles dealt over the waves).  Only the feature-mapped
    // values are kept: elu'(q) = q > 0 ? 1 : exp(q) = (Q' > 1 ? 1 : Q')
    // (the first weight burst of the next matrix is requested before the current one's MFMAs: its L2 trip rides on them)
    auto pf0k = gemm_prefetch<80, 80, false>(wp.p[P_VT_K], 80, wave, lane, 5);
    gemm_compute<80, 80, false, LOWP>(pf0, wp.p[P_VT_Q], 80, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_Q + r)[c] = elu1(v); });
    auto pf0v = gemm_prefetch<80, 80, false>(wp.p[P_VT_V], 80, wave, lane, 10);
    gemm_compute<80, 80, false, LOWP>(pf0k, wp.p[P_VT_K], 80, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_K + r)[c] = elu1(v); }, 5);
    gemm_compute<80, 80, false, LOWP>(pf0v, wp.p[P_VT_V], 80, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_V + r)[c] = v; }, 10);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 1)

    // ---------------- P2: linear attention over the L tokens of each point (linear_attention.py:31-45) in score form:
    // A[s'] = Q'.K'_s', msg = u * sum_s' A[s'] V_s'  with V = v/L and u = L / (sum_s' A[s'] + eps)
    for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
      const int col = idx >> 3, h = idx & 7, pt = col / L;
      float msg[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) msg[e] = 0.f;
      float u = 0.f;
      if (pt < PPT) {
        float Qp[10], den = 0.f;
#pragma unroll
        for (int d = 0; d < 10; ++d) Qp[d] = R(O_Q + 10 * h + d)[col];
#pragma unroll 4
        for (int s = 0; s < L; ++s) {
          const int c2 = pt * L + s;
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < 10; ++d) a = fmaf(Qp[d], R(O_K + 10 * h + d)[c2], a);
          den += a;
#pragma unroll
          for (int e = 0; e < 10; ++e) msg[e] = fmaf(a, R(O_V + 10 * h + e)[c2] * invL, msg[e]);
        }
        u = (float)L / (den + 1e-6f);
      }
#pragma unroll
      for (int e = 0; e < 10; ++e) R(O_MSG + 10 * h + e)[col] = msg[e] * u;
      R(O_U + h)[col] = u;
    }
    auto pf1 = gemm_prefetch<80, 80, false>(wp.p[P_VT_MERGE], 80, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 2)

    // ---------------- P3/P4: merge + LayerNorm1 (transformer.py:51-52)
    gemm_compute<80, 80, false, LOWP>(pf1, wp.p[P_VT_MERGE], 80, R(O_MSG), wave, lane, [&](int r, int c, float v) { R(O_XH1 + r)[c] = v; });
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 3)
    ln_forward<80>(R(O_XH1), R(O_CAT + 80), nullptr, flat + F_N1W, flat + F_N1B, R(O_RSTD1), tid);
    auto pf2 = gemm_prefetch<160, 160, false>(wp.p[P_VT_MLP0], 160, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 4)
    // ---------------- P5-P7: MLP on [x | m], LayerNorm2, residual (transformer.py:55-58)
    gemm_compute<160, 160, false, LOWP>(pf2, wp.p[P_VT_MLP0], 160, R(O_CAT), wave, lane,
                              [&](int r, int c, float v) { R(O_HID + r)[c] = fmaxf(v, 0.f); });
    auto pf3 = gemm_prefetch<80, 160, false>(wp.p[P_VT_MLP2], 160, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 5)
    gemm_compute<80, 160, false, LOWP>(pf3, wp.p[P_VT_MLP2], 160, R(O_HID), wave, lane, [&](int r, int c, float v) { R(O_XH2 + r)[c] = v; });
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 6)
    ln_forward<80>(R(O_XH2), R(O_RIN), R(O_CAT), flat + F_N2W, flat + F_N2B, R(O_RSTD2), tid);
    auto pf4 = gemm_prefetch<16, 83, false>(wp.p[P_RW_W0], 83, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 7)
    // ---------------- P8-P10: radiance-weight MLP 83 -> 16 -> 8 -> 1 (ray_transformer.py:159-163, 313-314)
    gemm_compute<16, 83, false, LOWP>(pf4, wp.p[P_RW_W0], 83, R(O_RIN), wave, lane,
                            [&](int r, int c, float v) { R(O_H1 + r)[c] = fmaxf(v + flat[F_RW_B0 + r], 0.f); });
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 8)
    for (int idx = tid; idx < 8 * kTT; idx += kBwdThreads) {
      const int o = idx / kTT, c = idx - o * kTT;
      float s = flat[F_RW_B2 + o];
#pragma unroll
      for (int i = 0; i < 16; ++i) s = fmaf(flat[F_RW_W2 + o * 16 + i], R(O_H1 + i)[c], s);
      R(O_H2 + o)[c] = fmaxf(s, 0.f);
    }
    if (tid < kTT) R(O_DLOGIT)[tid] = 0.f;     // view tokens / empty columns carry no logit gradient
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 9)
    // masked softmax over the views of a point and its adjoint (ray_transformer.py:315-319): one thread per (point, view) in
    // wave 0, the views of a point meet through the logit row (same sums in the same order as one thread per point, which
    // walked its views serially with run-time indexed arrays: 4.6 k cycles per tile for 8 active threads)
    if (tid0 < 64) {
      auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };
      const int pt = tid / NV, v = tid - pt * NV;
      const bool on = tid < PPT * NV && p0 + pt < P;
      const int c = on ? pt * L + 1 + v : 0, c0 = on ? pt * L + 1 : 0;
      float* row = R(O_DLOGIT);
      f32x4 col = splat4(0.f);
      float lg = 0.f;
      if (on) {
        float s = flat[F_RW_B4];
#pragma unroll
        for (int i = 0; i < 8; ++i) s = fmaf(flat[F_RW_W4 + i], R(O_H2 + i)[c], s);
        col = ld4(flat + F_RGBM + (pt * NV + v) * 4);
        lg = col[3] == 0.f ? -1e9f : s;
        row[c] = lg;
      }
      wave_sync();
      float mx = -INFINITY;
      if (on)
        for (int u = 0; u < NV; ++u) mx = fmaxf(mx, row[c0 + u]);
      const float e = expf(lg - mx);
      wave_sync();
      if (on) row[c] = e;
      wave_sync();
      float dl = 0.f;
      if (on) {
        float den = 0.f;
        for (int u = 0; u < NV; ++u) den += row[c0 + u];
        float rr = 0.f, rg = 0.f, rb = 0.f;
        for (int u = 0; u < NV; ++u) {
          const float pu = row[c0 + u] / den;
          const f32x4 cu = ld4(flat + F_RGBM + (pt * NV + u) * 4);
          rr = fmaf(pu, cu[0], rr); rg = fmaf(pu, cu[1], rg); rb = fmaf(pu, cu[2], rb);
        }
        const float dr = flat[F_DRAD + pt * 3 + 0], dg = flat[F_DRAD + pt * 3 + 1], db = flat[F_DRAD + pt * 3 + 2];
        const float dot_r = rr * dr + rg * dg + rb * db;
        dl = (e / den) * ((col[0] * dr + col[1] * dg + col[2] * db) - dot_r);
      }
      wave_sync();
      if (on) row[c] = col[3] == 0.f ? 0.f : dl;     // torch.where: masked logits are constants
    }
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 11)
    // ---------------- B1: radiance MLP backwards
    for (int idx = tid; idx < 8 * kTT; idx += kBwdThreads) {
      const int o = idx / kTT, c = idx - o * kTT;
      R(O_DH2 + o)[c] = R(O_H2 + o)[c] > 0.f ? flat[F_RW_W4 + o] * R(O_DLOGIT)[c] : 0.f;
    }
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 12)
    for (int idx = tid; idx < 16 * kTT; idx += kBwdThreads) {
      const int i = idx / kTT, c = idx - i * kTT;
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) s = fmaf(flat[F_RW_W2 + o * 16 + i], R(O_DH2 + o)[c], s);
      R(O_DH1 + i)[c] = R(O_H1 + i)[c] > 0.f ? s : 0.f;
    }
    auto pf5 = gemm_prefetch<80, 16, true>(wp.p[P_RW_W0], 83, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 13)
    // ---------------- B2: d y = W0^T d h1 (the 80 feature columns) + d token0 (staged in P0); weight gradient of rw0
    gemm_compute<80, 16, true, LOWP>(pf5, wp.p[P_RW_W0], 83, R(O_DH1), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
    wgrad_range<kSlots, T_RW0, T_MLP2, LOWP>(acc, lds, wg_tab, wave, lane);
    // small gradients on the VALU: W2 (8x16), W4 (8), biases (operands final since B1)
    {  // one instruction stream for all threads: thread -> (row a, row b or none)
      const float* ra = R(O_DH2 + (tid >> 4));
      const float* rb = R(O_H1 + (tid & 15));
      if (tid >= 128) { ra = R(O_DLOGIT); rb = R(O_H2 + ((tid - 128) & 7)); }
      if (tid >= 136) { ra = R(O_DH1 + ((tid - 136) & 15)); rb = nullptr; }
      if (tid >= 152) ra = R(O_DH2 + ((tid - 152) & 7));
      if (tid >= 160) ra = R(O_DLOGIT);
      const float v = row_dot(ra, rb, 0);
      if (tid < 128) accA += v;
      else if (tid <= 160) accB += v;
    }

    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 14)
    // ---------------- B3: LayerNorm2 backwards (d opre takes the rows of [y | dir], dead since B2); y = x + LN2(.) so d x
    // starts as d y
    ln_backward<80>(R(O_DY), R(O_XH2), flat + F_N2W, R(O_RSTD2), R(O_DOPRE), tid);
    {
      const float v = row_dot(R(O_DY), tid < 80 ? R(O_XH2) : nullptr, tid < 80 ? tid : (tid < 160 ? tid - 80 : 0));
      if (tid < 160) accN2 += v;
    }
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 15)
    if (dbg) dump(p0, O_DOPRE, 80, 80);
    // ---------------- W1: weight gradient of mlp2 (needs the hidden layer, which B4 overwrites in place)
    auto pf6 = gemm_prefetch<160, 80, true>(wp.p[P_VT_MLP2], 160, wave, lane, 0);
    wgrad_range<kSlots, T_MLP2, T_MLP0, LOWP>(acc, lds, wg_tab, wave, lane);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 16)
    // ---------------- B4/B5: MLP backwards; d hid replaces hid element by element
    gemm_compute<160, 80, true, LOWP>(pf6, wp.p[P_VT_MLP2], 160, R(O_DOPRE), wave, lane,
                            [&](int r, int c, float v) { R(O_DHID + r)[c] = R(O_HID + r)[c] > 0.f ? v : 0.f; });
    auto pf7 = gemm_prefetch<160, 160, true>(wp.p[P_VT_MLP0], 160, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 17)
    if (dbg) dump(p0, O_DHID, 160, 160);
    // d cat: the x half joins the d x accumulator at once (an element has one owner lane), the message half takes d opre's
    // rows (dead since B4); weight gradient of mlp0
    gemm_compute<160, 160, true, LOWP>(pf7, wp.p[P_VT_MLP0], 160, R(O_DHID), wave, lane, [&](int r, int c, float v) {
      if (r < 80) R(O_DY + r)[c] += v;
      else R(O_DCATM + r - 80)[c] = v;
    });
    UFR_BWD_PHASE(g_vb_phase, 34)
    wgrad_range<kSlots, T_MLP0, T_MERGE, LOWP>(acc, lds, wg_tab, wave, lane);
    UFR_BWD_PHASE(g_vb_phase, 35)
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 18)
    if (dbg) dump(p0, O_DCATM, 80, 400);
    // ---------------- B6: LayerNorm1 backwards on the message half (d mpre takes d hid's rows)
    ln_backward<80>(R(O_DCATM), R(O_XH1), flat + F_N1W, R(O_RSTD1), R(O_DMPRE), tid);
    {
      const float v = row_dot(R(O_DCATM), tid < 80 ? R(O_XH1) : nullptr, tid < 80 ? tid : (tid < 160 ? tid - 80 : 0));
      if (tid < 160) accN1 += v;
    }
    auto pf8 = gemm_prefetch<80, 80, true>(wp.p[P_VT_MERGE], 80, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 19)
    // ---------------- B7: merge backwards; weight gradient of merge
    gemm_compute<80, 80, true, LOWP>(pf8, wp.p[P_VT_MERGE], 80, R(O_DMPRE), wave, lane, [&](int r, int c, float v) { R(O_DMSG + r)[c] = v; });
    wgrad_range<kSlots, T_MERGE, T_QKV, LOWP>(acc, lds, wg_tab, wave, lane);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 20)
    // ---------------- B8: attention backwards, query side (thread = (token s, head)):
    //   msg = u r, r = sum_s' A[s'] V_s';  d r = u d msg;  d u = d msg . r;  d den = -d u u^2 / L;
    //   d A[s'] = d r . V_s' + d den;  d Q' = sum_s' d A[s'] K'_s'        (d q takes xhat1's rows, dead since B6)
    for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
      const int col = idx >> 3, h = idx & 7, pt = col / L;
      float dq[10];
#pragma unroll
      for (int d = 0; d < 10; ++d) dq[d] = 0.f;
      float dden = 0.f;
      if (pt < PPT) {
        float Qp[10], dm[10], r[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          Qp[d] = R(O_Q + 10 * h + d)[col];
          dm[d] = R(O_DMSG + 10 * h + d)[col];
          r[d] = 0.f;
        }
#pragma unroll 4
        for (int s = 0; s < L; ++s) {
          const int c2 = pt * L + s;
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < 10; ++d) a = fmaf(Qp[d], R(O_K + 10 * h + d)[c2], a);
#pragma unroll
          for (int e = 0; e < 10; ++e) r[e] = fmaf(a, R(O_V + 10 * h + e)[c2] * invL, r[e]);
        }
        const float u = R(O_U + h)[col];
        float du = 0.f;
#pragma unroll
        for (int e = 0; e < 10; ++e) du = fmaf(dm[e], r[e], du);
        dden = -du * u * u * invL;
#pragma unroll 4
        for (int s = 0; s < L; ++s) {
          const int c2 = pt * L + s;
          float dA = dden;
#pragma unroll
          for (int e = 0; e < 10; ++e) dA = fmaf(u * dm[e], R(O_V + 10 * h + e)[c2] * invL, dA);
#pragma unroll
          for (int d = 0; d < 10; ++d) dq[d] = fmaf(dA, R(O_K + 10 * h + d)[c2], dq[d]);
        }
      }
#pragma unroll
      for (int d = 0; d < 10; ++d) {
        const float qp = R(O_Q + 10 * h + d)[col];
        R(O_DQ + 10 * h + d)[col] = dq[d] * (qp > 1.f ? 1.f : qp);
      }
      R(O_DDEN + h)[col] = dden;
    }
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 21)
    // ---------------- B9: key / value side (thread = (token s', head)): d K'_s' = sum_s d A[s][s'] Q'_s,
    //   d V_s' = sum_s A[s][s'] d r_s        (d k takes xhat2's rows, d v the message's: dead since B3 / B7)
    for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
      const int col = idx >> 3, h = idx & 7, pt = col / L;
      float dk[10], dv[10], kp_own[10];
#pragma unroll
      for (int d = 0; d < 10; ++d) {
        dk[d] = 0.f;
        dv[d] = 0.f;
        kp_own[d] = R(O_K + 10 * h + d)[col];
      }
      if (pt < PPT) {
        float V[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) V[d] = R(O_V + 10 * h + d)[col] * invL;
#pragma unroll 4
        for (int s = 0; s < L; ++s) {
          const int c2 = pt * L + s;
          const float u = R(O_U + h)[c2];
          float a = 0.f, dA = R(O_DDEN + h)[c2];
          float Qs[10], dr[10];
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            Qs[d] = R(O_Q + 10 * h + d)[c2];
            dr[d] = u * R(O_DMSG + 10 * h + d)[c2];
            a = fmaf(Qs[d], kp_own[d], a);
            dA = fmaf(dr[d], V[d], dA);
          }
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            dk[d] = fmaf(dA, Qs[d], dk[d]);
            dv[d] = fmaf(a, dr[d], dv[d]);
          }
        }
      }
#pragma unroll
      for (int d = 0; d < 10; ++d) {
        R(O_DK + 10 * h + d)[col] = dk[d] * (kp_own[d] > 1.f ? 1.f : kp_own[d]);
        R(O_DV + 10 * h + d)[col] = dv[d] * invL;
      }
    }
    auto pf9 = gemm_prefetch<80, 80, true>(wp.p[P_VT_Q], 80, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 22)
    // ---------------- B10: projections backwards into the d x accumulator (same lane owns an element in all three); weight
    // gradients of q, k, v
    auto pf9k = gemm_prefetch<80, 80, true>(wp.p[P_VT_K], 80, wave, lane, 0);
    gemm_compute<80, 80, true, LOWP>(pf9, wp.p[P_VT_Q], 80, R(O_DQ), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
    UFR_BWD_PHASE(g_vb_phase, 31)
    auto pf9v = gemm_prefetch<80, 80, true>(wp.p[P_VT_V], 80, wave, lane, 0);
    gemm_compute<80, 80, true, LOWP>(pf9k, wp.p[P_VT_K], 80, R(O_DK), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
    gemm_compute<80, 80, true, LOWP>(pf9v, wp.p[P_VT_V], 80, R(O_DV), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
    UFR_BWD_PHASE(g_vb_phase, 32)
    wgrad_range<kSlots, T_QKV, T_END, LOWP>(acc, lds, wg_tab, wave, lane);
    UFR_BWD_PHASE(g_vb_phase, 33)
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 23)
    // the next tile's global inputs are requested here and land while B11 runs.  (Anywhere inside the GEMM / weight-gradient
    // phases a reload of a spilled register -- these kernels keep 256 accumulators per lane and spill ~100 others -- waits
    // with vmcnt(0) for every vector-memory operation in flight, these HBM loads included: 7..15 k cycles per tile.)
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    UFR_BWD_PHASE(g_vb_phase, 30)
    // ---------------- B11: outputs.  Token columns 32..55 (frustum features) and 56..71 (pre_sim_mlp) are the same for
    // all NV view tokens of a point (ray_transformer.py:258-281): their gradients add up.  Token 0 is the view token.
    for (int idx = tid; idx < PPT * 40; idx += kBwdThreads) {
      const int pt = idx / 40, c = idx - pt * 40, p = p0 + pt;
      if (p < P) {
        float s = 0.f;
        for (int tv = 1; tv < L; ++tv) s += R(O_DY + 32 + c)[pt * L + tv];
        d_pv[(size_t)p * 40 + c] = s;
      }
    }
    if (tid < 80) {
      for (int pt = 0; pt < PPT; ++pt)
        if (p0 + pt < P) accC += R(O_DY + tid)[pt * L];
    }
    if (dbg) {   // what is still live at the end of the tile (d opre / d hid / d cat were dumped where they were final)
      dump(p0, O_DY, 80, 0); dump(p0, O_DMPRE, 80, 480); dump(p0, O_DMSG, 80, 560);
      dump(p0, O_DQ, 80, 640); dump(p0, O_DK, 80, 720); dump(p0, O_DV, 80, 800); dump(p0, O_DLOGIT, 1, 880);
    }
    __syncthreads();   // the next tile's commit overwrites CAT / DY / RIN rows that this phase reads
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_vb_phase, 24)
  }

  // ---------------- flush (once per workgroup)
  wgrad_flush_all<vb::kList, 7, kSlots, 0>(acc, gp, wave, lane);
  if (tid < 128) atomic_add_f32(gp.p[P_RW_W2] + tid, accA);
  if (tid < 80) atomic_add_f32(gp.p[P_VIEW_TOKEN] + tid, accC);
  if (tid >= 128 && tid < 136) atomic_add_f32(gp.p[P_RW_W4] + (tid - 128), accB);
  if (tid >= 136 && tid < 152) atomic_add_f32(gp.p[P_RW_B0] + (tid - 136), accB);
  if (tid >= 152 && tid < 160) atomic_add_f32(gp.p[P_RW_B2] + (tid - 152), accB);
  if (tid == 160) atomic_add_f32(gp.p[P_RW_B4], accB);
  if (tid < 80) {
    atomic_add_f32(gp.p[P_VT_N2W] + tid, accN2);
    atomic_add_f32(gp.p[P_VT_N1W] + tid, accN1);
  } else if (tid < 160) {
    atomic_add_f32(gp.p[P_VT_N2B] + (tid - 80), accN2);
    atomic_add_f32(gp.p[P_VT_N1B] + (tid - 80), accN1);
  }
}

template <bool LOWP>
static hipError_t launch_view_bwd_t(const RawPtrs& wp, const GradPtrs& gp, const float* x_tokens, const float* rgbm,
                                    const float* dirs, const float* d_tok_a, const float* d_tok_b, const float* d_radiance,
                                    int P, int NV, float* d_pv, float* dbg, hipStream_t s) {
  static bool attr_set[16] = {};   // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&view_bwd_kernel<LOWP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, vb::kLdsBytes);
    if (attr != hipSuccess) return attr;
    attr_set[dev] = true;
  }
  const int PPT = kTT / (NV + 1);
  const int n_tiles = (P + PPT - 1) / PPT;
  const int blocks = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(view_bwd_kernel<LOWP>, dim3(blocks), dim3(kBwdThreads), vb::kLdsBytes, s, wp, gp, x_tokens, rgbm, dirs,
                     d_tok_a, d_tok_b, d_radiance, P, NV, d_pv, dbg);
  return hipGetLastError();
}

hipError_t launch_view_bwd(const RawPtrs& wp, const GradPtrs& gp, const float* x_tokens, const float* rgbm,
                           const float* dirs, const float* d_tok_a, const float* d_tok_b, const float* d_radiance, int P,
                           int NV, float* d_pv, float* dbg, bool lowp, hipStream_t s) {
  return lowp
             ? launch_view_bwd_t<true>(wp, gp, x_tokens, rgbm, dirs, d_tok_a, d_tok_b, d_radiance, P, NV, d_pv, dbg, s)
             : launch_view_bwd_t<false>(wp, gp, x_tokens, rgbm, dirs, d_tok_a, d_tok_b, d_radiance, P, NV, d_pv, dbg, s);
}

}  // namespace ufr

#ifdef UFR_BWD_TIMING
extern "C" int ufr_debug_vb_phases(unsigned long long* out, int n, int reset) {
  unsigned long long h[64] = {};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ufr::g_vb_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 64; ++i) out[i] = h[i];
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ufr::g_vb_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
