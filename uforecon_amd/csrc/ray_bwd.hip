// Backward of the along-ray aggregation (ray_transformer.hip) and of pre_sim_mlp.
//   RayTransformer.forward     code1/ray_transformer.py:296-307 (+ order_posenc :165-173), pre_sim_mlp :128-132, 268
//   LoFTREncoderLayer.forward  code1/attention/transformer.py:35-58
//   LinearAttention.forward    code1/attention/linear_attention.py:20-47
// One workgroup per ray (persistent over rays); the SN tokens of the ray are walked in tiles of 16 in three sweeps,
// because linear attention couples all tokens of a ray through the per-head KV state:
//   sweep 1: K', V -> KV_h = sum_s K'_s^T V_s (+ the K' sum as column 11)                    (recompute)
//   sweep 2: q .. DensityMLP recomputed per tile, then backwards down to d msg; d Q' needs KV; d KV_h accumulates;
//            d x (residual, MLP and q paths) -> d_tok_a; weight gradients of everything but k, v
//   sweep 3: k, v recomputed, d K', d V from d KV -> d k, d v -> d x contribution -> d_tok_b; dWk, dWv
// Machinery: bwd_common.h.
#include "bwd_common.h"
#include "ufr_internal.h"

namespace ufr {

namespace rb {
// LDS rows (each kLD floats); buffers whose live ranges do not overlap share rows (sweep 2 unless noted):
//   CAT   x | m                      W load / P4    R B6 (weight gradient of mlp0), B10 (of q); sweeps 1, 3: x only
//   Q                                W P1           R B10
//   MSG                              W P2           R B8 (weight gradient of merge)           sweep 3: DV
//   XH1   merge out -> xhat1         W P3 / P4      R B7     then DQ (W B9, R B10)             sweep 3: DK
//   HID   relu(mlp0)                 W P5           R W1, B5 -> DHID in place (W B5, R B6)  then DMPRE (0..87, W B7, R B8) |
//                                                   DMSG (88..175, W B8, R B10)              sweeps 1, 3: K | V
//   XH2   mlp2 out -> xhat2          W P6 / P7      R B4     then DCATM (W B6, R B7)
//   Y     layer output               W P7           R B3 (weight gradient of dm0)  then DOPRE (W B4, R B5)
//   D1 D2 DD1 DD2 DY, scalars        not shared
enum : int {
  O_CAT = 0,        // 176: x = [token0 80 | order PE 8] (0..87) | m (88..175)
  O_Q = 176,
  O_MSG = 264, O_DV = O_MSG,
  O_XH1 = 352, O_DQ = O_XH1, O_DK = O_XH1,
  O_HID = 440, O_DHID = O_HID, O_DMPRE = O_HID, O_DMSG = O_HID + 88, O_K = O_HID, O_V = O_HID + 88,
  O_XH2 = 616, O_DCATM = O_XH2,
  O_Y = 704, O_DOPRE = O_Y,
  O_D1 = 792,       // 32
  O_D2 = 824,       // 16
  O_DD1 = 840,      // 32
  O_DD2 = 872,      // 16
  O_DY = 888,       // 88
  O_RSTD1 = 976, O_RSTD2 = 977, O_DSRDF = 978,
  O_Z = 979,        // 8 heads
  O_DDEN = 987,     // 8 heads
  O_END = 995
};
constexpr int kKV = 8 * 11 * 12;   // per head [d][e], e = 11: sum of K' (linear_attention.py:43)
constexpr int kKVPer = (kKV + kBwdThreads - 1) / kBwdThreads;
// flat region behind the KV state: small parameters copied once per workgroup (read by every tile's VALU phases)
enum : int { F_N1W = 0, F_N1B = 88, F_N2W = 176, F_N2B = 264, F_DM_B0 = 352, F_DM_B2 = 384, F_DM_W4 = 400, F_END = 416 };
constexpr int kLdsBytes = (O_END * kLD + 2 * kKV + F_END) * 4;
static_assert(kLdsBytes <= 160 * 1024, "tile does not fit the CU's LDS");

// sweep 2's gradient tiles in the order their dY becomes final: dm2 | dm0 | mlp2 | mlp0 | merge | q
constexpr WgMat kMats2[] = {{P_DM_W2, 16, 32, O_DD2, O_D1},       {P_DM_W0, 32, 88, O_DD1, O_Y},
                            {P_RT_MLP2, 88, 176, O_DOPRE, O_HID}, {P_RT_MLP0, 176, 176, O_DHID, O_CAT},
                            {P_RT_MERGE, 88, 88, O_DMPRE, O_MSG}, {P_RT_Q, 88, 88, O_DQ, O_CAT}};
constexpr WgMat kMats3[] = {{P_RT_K, 88, 88, O_DK, O_CAT}, {P_RT_V, 88, 88, O_DV, O_CAT}};
constexpr auto kList2 = make_wglist(kMats2);
constexpr auto kList3 = make_wglist(kMats3);
constexpr int kSlots2 = (kList2.first[6] + kBwdWaves - 1) / kBwdWaves;   // 273 tiles -> 69
constexpr int kSlots3 = (kList3.first[2] + kBwdWaves - 1) / kBwdWaves;   // 72 tiles -> 18
constexpr int T_DM0 = kList2.first[1], T_MLP2 = kList2.first[2], T_MLP0 = kList2.first[3], T_MERGE = kList2.first[4],
              T_Q = kList2.first[5], T_END2 = kList2.first[6];
}  // namespace rb

#ifdef UFR_BWD_TIMING
__device__ unsigned long long g_rb_phase[64];   // cycles between consecutive barriers of workgroup 0 (tools/dev/bwd_phases.py)
#endif

template <bool LOWP>
__global__ void __launch_bounds__(kBwdThreads) ray_bwd_kernel(RawPtrs wp, GradPtrs gp, const float* __restrict__ token0,
                                                              const int* __restrict__ tok_row, int accumulate,
                                                              const float* __restrict__ order_pe,
                                                              const float* __restrict__ d_srdf, int RN, int SN,
                                                              float* __restrict__ d_tok_a, float* __restrict__ d_tok_b,
                                                              float* __restrict__ dbg) {
  using namespace rb;
#ifdef UFR_BWD_TIMING
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* KV = lds + O_END * kLD;
  float* dKV = KV + kKV;
  float* flat = dKV + kKV;
  const int tid0 = threadIdx.x, wave = tid0 >> 6, lane = tid0 & 63;
  int tid = tid0;   // re-laundered after every barrier (bwd_common.h: opaque)
  const int n_sub = (SN + kTT - 1) / kTT;   // SN is a multiple of 16: the last tile of a ray may hold 16 tokens only
  const float fS = (float)SN;

  f32x4 acc[kSlots2], acc3[kSlots3];
  const auto wg_tab2 = wgrad_table<rb::kList2, 6, kSlots2>(wave, lane);
  const auto wg_tab3 = wgrad_table<rb::kList3, 2, kSlots3>(wave, lane);
#pragma unroll
  for (int s = 0; s < kSlots2; ++s) acc[s] = splat4(0.f);
#pragma unroll
  for (int s = 0; s < kSlots3; ++s) acc3[s] = splat4(0.f);
  float accB = 0.f, accN1 = 0.f, accN2 = 0.f;

  auto R = [&](int row) -> float* { return lds + row * kLD; };
  if (tid0 < 88) {
    flat[F_N1W + tid0] = wp.p[P_RT_N1W][tid0]; flat[F_N1B + tid0] = wp.p[P_RT_N1B][tid0];
    flat[F_N2W + tid0] = wp.p[P_RT_N2W][tid0]; flat[F_N2B + tid0] = wp.p[P_RT_N2B][tid0];
  }
  if (tid0 < 32) flat[F_DM_B0 + tid0] = wp.p[P_DM_B0][tid0];
  if (tid0 < 16) { flat[F_DM_B2 + tid0] = wp.p[P_DM_B2][tid0]; flat[F_DM_W4 + tid0] = wp.p[P_DM_W4][tid0]; }
  // (the first barrier of the ray loop publishes them)
  // x tile: token-0 feature of sample (ray, s0 + col) | order PE (ray_transformer.py:301-303); columns past the ray's end
  // (a 16-token last tile) are zero: every gradient of such a column is zero, only the K' sums have to skip it
  auto load_x = [&](int ray, int s0) {
    for (int idx = tid; idx < kTT * 22; idx += kBwdThreads) {
      const int col = idx / 22, f4 = idx - col * 22;
      f32x4 v = splat4(0.f);
      if (s0 + col < SN) {
        const size_t slot = (size_t)ray * SN + s0 + col;
        v = f4 < 20 ? ld4(token0 + (tok_row ? (size_t)tok_row[slot] : slot) * UFR_TOKEN_DIM + 4 * f4)
                    : ld4(order_pe + (size_t)(s0 + col) * 8 + 4 * (f4 - 20));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) R(O_CAT + 4 * f4 + e)[col] = v[e];
    }
  };
  auto store_dx = [&](int ray, int s0, float* __restrict__ dst_base) {   // d x of a sweep (order-PE rows carry no gradient)
    for (int idx = tid; idx < kTT * 20; idx += kBwdThreads) {
      const int col = idx / 20, f4 = idx - col * 20;
      if (s0 + col >= SN) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = R(O_DY + 4 * f4 + e)[col];
      const size_t slot = (size_t)ray * SN + s0 + col;
      float* dst = dst_base + (tok_row ? (size_t)tok_row[slot] : slot) * UFR_TOKEN_DIM + 4 * f4;
      if (accumulate) v += ld4(dst);
      st4(dst, v);
    }
  };
  // the 1056 = 8 x 11 x 12 per-head state entries are dealt kKVPer per thread
  auto kv_entry = [&](int o, int& h, int& d, int& e) { h = o / 132; d = (o - h * 132) / 12; e = o % 12; };
  auto dump = [&](int ray, int s0, int row0, int k0) {
    for (int idx = tid; idx < kTT * 88; idx += kBwdThreads) {
      const int col = idx / 88, k = idx - col * 88;
      if (s0 + col < SN) dbg[((size_t)ray * SN + s0 + col) * 440 + k0 + k] = R(row0 + k)[col];
    }
  };

  for (int ray = blockIdx.x; ray < RN; ray += gridDim.x) {
    // ================= sweep 1: KV state
    float kv[kKVPer] = {};
    for (int sub = 0; sub < n_sub; ++sub) {
      const int nt = min(kTT, SN - sub * kTT);
      load_x(ray, sub * kTT);
      auto pf0 = gemm_prefetch<88, 88, false>(wp.p[P_RT_K], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 0)
      auto pf0v = gemm_prefetch<88, 88, false>(wp.p[P_RT_V], 88, wave, lane, 6);
      // (columns past the ray's end hold x = 0: V = 0 there, but K' = elu(0) + 1 = 1 -- zeroed here so that the sums below can
      // run over whole tiles)
      gemm_compute<88, 88, false, LOWP>(pf0, wp.p[P_RT_K], 88, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_K + r)[c] = c < nt ? elu1(v) : 0.f; });
      gemm_compute<88, 88, false, LOWP>(pf0v, wp.p[P_RT_V], 88, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_V + r)[c] = v / fS; }, 6);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 1)
#pragma unroll
      for (int i = 0; i < kKVPer; ++i) {
        const int o = tid + i * kBwdThreads;
        if (o < kKV) {
          int h, d, e;
          kv_entry(o, h, d, e);
          // batched operand reads (row_dot): as a running sum over a run-time token count every read waited for its own
          // LDS round trip -- 31 k cycles per tile, an eighth of this kernel
          kv[i] += row_dot(R(O_K + 11 * h + d), e < 11 ? R(O_V + 11 * h + e) : nullptr, 0);
        }
      }
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 2)
    }
#pragma unroll
    for (int i = 0; i < kKVPer; ++i)
      if (tid + i * kBwdThreads < kKV) KV[tid + i * kBwdThreads] = kv[i];
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_rb_phase, 3)

    // ================= sweep 2
    float dkv[kKVPer] = {};
    for (int sub = 0; sub < n_sub; ++sub) {
      const int s0 = sub * kTT;
      load_x(ray, s0);
      if (tid < kTT) R(O_DSRDF)[tid] = s0 + tid < SN ? d_srdf[(size_t)ray * SN + s0 + tid] : 0.f;
      auto pf1 = gemm_prefetch<88, 88, false>(wp.p[P_RT_Q], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 4)
      gemm_compute<88, 88, false, LOWP>(pf1, wp.p[P_RT_Q], 88, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_Q + r)[c] = elu1(v); });   // Q' kept: elu'(q) = (Q' > 1 ? 1 : Q')
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 5)
      // message of (token, head): t = Q' KV_h, den = Q'.sum K', msg = t * Z * SN (linear_attention.py:43-44)
      for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
        const int col = idx >> 3, h = idx & 7;
        float Qp[11], t[11], den = 0.f;
#pragma unroll
        for (int d = 0; d < 11; ++d) Qp[d] = R(O_Q + 11 * h + d)[col];
#pragma unroll
        for (int e = 0; e < 11; ++e) t[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          const float* row = KV + (h * 11 + d) * 12;
#pragma unroll
          for (int e = 0; e < 11; ++e) t[e] = fmaf(Qp[d], row[e], t[e]);
          den = fmaf(Qp[d], row[11], den);
        }
        const float Z = 1.f / (den + 1e-6f);
#pragma unroll
        for (int e = 0; e < 11; ++e) R(O_MSG + 11 * h + e)[col] = t[e] * (Z * fS);
        R(O_Z + h)[col] = Z;
      }
      auto pf2 = gemm_prefetch<88, 88, false>(wp.p[P_RT_MERGE], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 6)
      gemm_compute<88, 88, false, LOWP>(pf2, wp.p[P_RT_MERGE], 88, R(O_MSG), wave, lane, [&](int r, int c, float v) { R(O_XH1 + r)[c] = v; });
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 7)
      ln_forward<88>(R(O_XH1), R(O_CAT + 88), nullptr, flat + F_N1W, flat + F_N1B, R(O_RSTD1), tid);
      auto pf3 = gemm_prefetch<176, 176, false>(wp.p[P_RT_MLP0], 176, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 8)
      gemm_compute<176, 176, false, LOWP>(pf3, wp.p[P_RT_MLP0], 176, R(O_CAT), wave, lane,
                                [&](int r, int c, float v) { R(O_HID + r)[c] = fmaxf(v, 0.f); });
      auto pf4 = gemm_prefetch<88, 176, false>(wp.p[P_RT_MLP2], 176, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 9)
      gemm_compute<88, 176, false, LOWP>(pf4, wp.p[P_RT_MLP2], 176, R(O_HID), wave, lane, [&](int r, int c, float v) { R(O_XH2 + r)[c] = v; });
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 10)
      ln_forward<88>(R(O_XH2), R(O_Y), R(O_CAT), flat + F_N2W, flat + F_N2B, R(O_RSTD2), tid);
      auto pf5 = gemm_prefetch<32, 88, false>(wp.p[P_DM_W0], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 11)
      // DensityMLP 88 -> 32 -> 16 (-> 1) (ray_transformer.py:147-150, 307)
      gemm_compute<32, 88, false, LOWP>(pf5, wp.p[P_DM_W0], 88, R(O_Y), wave, lane,
                              [&](int r, int c, float v) { R(O_D1 + r)[c] = fmaxf(v + flat[F_DM_B0 + r], 0.f); });
      auto pf6 = gemm_prefetch<16, 32, false>(wp.p[P_DM_W2], 32, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 12)
      gemm_compute<16, 32, false, LOWP>(pf6, wp.p[P_DM_W2], 32, R(O_D1), wave, lane,
                              [&](int r, int c, float v) { R(O_D2 + r)[c] = fmaxf(v + flat[F_DM_B2 + r], 0.f); });
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 13)
      // ---- backwards: srdf = W4 d2 + b4
      for (int idx = tid; idx < 16 * kTT; idx += kBwdThreads) {
        const int o = idx / kTT, c = idx - o * kTT;
        R(O_DD2 + o)[c] = R(O_D2 + o)[c] > 0.f ? flat[F_DM_W4 + o] * R(O_DSRDF)[c] : 0.f;
      }
      auto pf7 = gemm_prefetch<32, 16, true>(wp.p[P_DM_W2], 32, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 14)
      gemm_compute<32, 16, true, LOWP>(pf7, wp.p[P_DM_W2], 32, R(O_DD2), wave, lane,
                             [&](int r, int c, float v) { R(O_DD1 + r)[c] = R(O_D1 + r)[c] > 0.f ? v : 0.f; });
      auto pf8 = gemm_prefetch<88, 32, true>(wp.p[P_DM_W0], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 15)
      // d y; weight gradients of the DensityMLP (dm2: d d2 x d1, dm0: d d1 x y); its small gradients on the VALU
      gemm_compute<88, 32, true, LOWP>(pf8, wp.p[P_DM_W0], 88, R(O_DD1), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] = v; });
      wgrad_range<kSlots2, 0, T_MLP2, LOWP>(acc, lds, wg_tab2, wave, lane);
      {   // biases and last layer: one instruction stream, thread -> (row a, row b or none)
        const float* ra = R(O_DD1 + ((tid - 176) & 31));
        const float* rb = nullptr;
        if (tid >= 208) ra = R(O_DD2 + ((tid - 208) & 15));
        if (tid >= 224) { ra = R(O_DSRDF); rb = R(O_D2 + ((tid - 224) & 15)); }
        if (tid >= 240) rb = nullptr;
        const float v = row_dot(ra, rb, 0);
        if (tid >= 176 && tid <= 240) accB += v;
      }
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 16)
      // LayerNorm2 backwards (d opre takes y's rows, dead since the weight gradient of dm0)
      ln_backward<88>(R(O_DY), R(O_XH2), flat + F_N2W, R(O_RSTD2), R(O_DOPRE), tid);
      {
        const float v = row_dot(R(O_DY), tid < 88 ? R(O_XH2) : nullptr, tid < 88 ? tid : (tid < 176 ? tid - 88 : 0));
        if (tid < 176) accN2 += v;
      }
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 17)
      if (dbg) dump(ray, s0, O_DOPRE, 88);
      // weight gradient of mlp2 (needs the hidden layer, which the next phase overwrites in place)
      auto pf9 = gemm_prefetch<176, 88, true>(wp.p[P_RT_MLP2], 176, wave, lane, 0);
      wgrad_range<kSlots2, T_MLP2, T_MLP0, LOWP>(acc, lds, wg_tab2, wave, lane);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 18)
      gemm_compute<176, 88, true, LOWP>(pf9, wp.p[P_RT_MLP2], 176, R(O_DOPRE), wave, lane,
                              [&](int r, int c, float v) { R(O_DHID + r)[c] = R(O_HID + r)[c] > 0.f ? v : 0.f; });
      auto pf10 = gemm_prefetch<176, 176, true>(wp.p[P_RT_MLP0], 176, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 19)
      // d cat: the x half joins d x at once, the message half takes xhat2's rows (dead since LN2 backwards); weight gradient of mlp0
      gemm_compute<176, 176, true, LOWP>(pf10, wp.p[P_RT_MLP0], 176, R(O_DHID), wave, lane, [&](int r, int c, float v) {
        if (r < 88) R(O_DY + r)[c] += v;
        else R(O_DCATM + r - 88)[c] = v;
      });
      wgrad_range<kSlots2, T_MLP0, T_MERGE, LOWP>(acc, lds, wg_tab2, wave, lane);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 20)
      ln_backward<88>(R(O_DCATM), R(O_XH1), flat + F_N1W, R(O_RSTD1), R(O_DMPRE), tid);
      {
        const float v = row_dot(R(O_DCATM), tid < 88 ? R(O_XH1) : nullptr, tid < 88 ? tid : (tid < 176 ? tid - 88 : 0));
        if (tid < 176) accN1 += v;
      }
      auto pf11 = gemm_prefetch<88, 88, true>(wp.p[P_RT_MERGE], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 21)
      if (dbg) dump(ray, s0, O_DMPRE, 176);
      gemm_compute<88, 88, true, LOWP>(pf11, wp.p[P_RT_MERGE], 88, R(O_DMPRE), wave, lane, [&](int r, int c, float v) { R(O_DMSG + r)[c] = v; });
      wgrad_range<kSlots2, T_MERGE, T_Q, LOWP>(acc, lds, wg_tab2, wave, lane);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 22)
      // attention backwards, query side: d t = d msg Z SN; d den = -SN Z^2 (d msg . t); d Q' = KV d t + d den sum K'
      // (d q takes xhat1's rows, dead since LN1 backwards)
      for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
        const int col = idx >> 3, h = idx & 7;
        float Qp[11], t[11], dt[11], dq[11];
#pragma unroll
        for (int d = 0; d < 11; ++d) { Qp[d] = R(O_Q + 11 * h + d)[col]; t[d] = 0.f; }
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          const float* row = KV + (h * 11 + d) * 12;
#pragma unroll
          for (int e = 0; e < 11; ++e) t[e] = fmaf(Qp[d], row[e], t[e]);
        }
        const float Z = R(O_Z + h)[col];
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 11; ++e) {
          const float dm = R(O_DMSG + 11 * h + e)[col];
          dot = fmaf(dm, t[e], dot);
          dt[e] = dm * (Z * fS);
        }
        const float dden = -fS * Z * Z * dot;
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          const float* row = KV + (h * 11 + d) * 12;
          float s = dden * row[11];
#pragma unroll
          for (int e = 0; e < 11; ++e) s = fmaf(row[e], dt[e], s);
          dq[d] = s;
        }
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          R(O_DQ + 11 * h + d)[col] = dq[d] * (Qp[d] > 1.f ? 1.f : Qp[d]);
          R(O_DMSG + 11 * h + d)[col] = dt[d];
        }
        R(O_DDEN + h)[col] = dden;
      }
      auto pf12 = gemm_prefetch<88, 88, true>(wp.p[P_RT_Q], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 23)
      // d KV_h[d][e] += sum_t Q'_t[d] d t_t[e];   d (sum K')[d] += sum_t d den_t Q'_t[d]   (columns past the ray's end: d t = 0)
#pragma unroll
      for (int i = 0; i < kKVPer; ++i) {
        const int o = tid + i * kBwdThreads;
        if (o < kKV) {
          int h, d, e;
          kv_entry(o, h, d, e);
          dkv[i] += row_dot(R(O_Q + 11 * h + d), e < 11 ? R(O_DMSG + 11 * h + e) : R(O_DDEN + h), 0);
        }
      }
      gemm_compute<88, 88, true, LOWP>(pf12, wp.p[P_RT_Q], 88, R(O_DQ), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
      wgrad_range<kSlots2, T_Q, T_END2, LOWP>(acc, lds, wg_tab2, wave, lane);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 24)
      store_dx(ray, s0, d_tok_a);
      if (dbg) { dump(ray, s0, O_DY, 0); dump(ray, s0, O_DQ, 264); }
      __syncthreads();   // the next tile's load_x / d y overwrite rows this phase reads
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 25)
    }
#pragma unroll
    for (int i = 0; i < kKVPer; ++i)
      if (tid + i * kBwdThreads < kKV) dKV[tid + i * kBwdThreads] = dkv[i];
    __syncthreads();
    tid = opaque(tid0);
    UFR_BWD_PHASE(g_rb_phase, 26)

    // ================= sweep 3: key / value side
    for (int sub = 0; sub < n_sub; ++sub) {
      const int s0 = sub * kTT;
      load_x(ray, s0);
      auto pf13 = gemm_prefetch<88, 88, false>(wp.p[P_RT_K], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 27)
      auto pf13v = gemm_prefetch<88, 88, false>(wp.p[P_RT_V], 88, wave, lane, 6);
      gemm_compute<88, 88, false, LOWP>(pf13, wp.p[P_RT_K], 88, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_K + r)[c] = elu1(v); });   // K'
      gemm_compute<88, 88, false, LOWP>(pf13v, wp.p[P_RT_V], 88, R(O_CAT), wave, lane, [&](int r, int c, float v) { R(O_V + r)[c] = v; }, 6);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 28)
      // d K'_s[d] = sum_e dKV[d][e] V_s[e] + d(sum K')[d];   d V_s[e] = sum_d K'_s[d] dKV[d][e];  V = v / SN
      // (a column past the ray's end took no part in the forward: its d k, d v are zero)
      for (int idx = tid; idx < kTT * 8; idx += kBwdThreads) {
        const int col = idx >> 3, h = idx & 7;
        const bool in_ray = s0 + col < SN;
        float Kp[11], V[11], dv[11];
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          Kp[d] = R(O_K + 11 * h + d)[col];
          V[d] = R(O_V + 11 * h + d)[col] / fS;
          dv[d] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          const float* row = dKV + (h * 11 + d) * 12;
          float s = row[11];
#pragma unroll
          for (int e = 0; e < 11; ++e) {
            s = fmaf(row[e], V[e], s);
            dv[e] = fmaf(Kp[d], row[e], dv[e]);
          }
          R(O_DK + 11 * h + d)[col] = in_ray ? s * (Kp[d] > 1.f ? 1.f : Kp[d]) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 11; ++e) R(O_DV + 11 * h + e)[col] = in_ray ? dv[e] / fS : 0.f;
      }
      auto pf14 = gemm_prefetch<88, 88, true>(wp.p[P_RT_K], 88, wave, lane, 0);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 29)
      auto pf14v = gemm_prefetch<88, 88, true>(wp.p[P_RT_V], 88, wave, lane, 0);
      gemm_compute<88, 88, true, LOWP>(pf14, wp.p[P_RT_K], 88, R(O_DK), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] = v; });
      gemm_compute<88, 88, true, LOWP>(pf14v, wp.p[P_RT_V], 88, R(O_DV), wave, lane, [&](int r, int c, float v) { R(O_DY + r)[c] += v; });
      wgrad_all<kSlots3, 0, LOWP>(acc3, lds, wg_tab3, lane);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 30)
      store_dx(ray, s0, d_tok_b);
      if (dbg) dump(ray, s0, O_DY, 352);
      __syncthreads();
      tid = opaque(tid0);
      UFR_BWD_PHASE(g_rb_phase, 31)
    }
  }

  wgrad_flush_all<rb::kList2, 6, kSlots2, 0>(acc, gp, wave, lane);
  wgrad_flush_all<rb::kList3, 2, kSlots3, 0>(acc3, gp, wave, lane);
  if (tid < 88) {
    atomic_add_f32(gp.p[P_RT_N2W] + tid, accN2);
    atomic_add_f32(gp.p[P_RT_N1W] + tid, accN1);
  } else if (tid < 176) {
    atomic_add_f32(gp.p[P_RT_N2B] + (tid - 88), accN2);
    atomic_add_f32(gp.p[P_RT_N1B] + (tid - 88), accN1);
  } else if (tid < 208) atomic_add_f32(gp.p[P_DM_B0] + (tid - 176), accB);
  else if (tid < 224) atomic_add_f32(gp.p[P_DM_B2] + (tid - 208), accB);
  else if (tid < 240) atomic_add_f32(gp.p[P_DM_W4] + (tid - 224), accB);
  else if (tid == 240) atomic_add_f32(gp.p[P_DM_B4], accB);
}

// ---------------------------------------------------------------------------------------------------
// pre_sim_mlp backwards (weights only: its input, the pair similarity, comes from the frozen matching features).
// Columns = 16 points; sim8 (P,8) saved by the forward gather; d out = columns 24..39 of d_pv.
namespace pb {
enum : int { O_S8 = 0, O_A1 = 8, O_A2 = 40, O_DO = 72, O_DA2 = 88, O_DA1 = 120, O_END = 152 };
constexpr WgMat kMats[] = {{P_PS_W4, 16, 32, O_DO, O_A2}, {P_PS_W2, 32, 32, O_DA2, O_A1}, {P_PS_W0, 32, 8, O_DA1, O_S8}};
constexpr auto kList = make_wglist(kMats);   // 2 + 4 + 2 = 8 tiles: two slots per wave
}  // namespace pb

template <bool LOWP>
__global__ void __launch_bounds__(kBwdThreads) presim_bwd_kernel(RawPtrs wp, GradPtrs gp, const float* __restrict__ sim8,
                                                                 const float* __restrict__ d_pv, int P) {
  using namespace pb;
  __shared__ float lds[O_END * kLD];
  const int tid0 = threadIdx.x, wave = tid0 >> 6, lane = tid0 & 63;
  int tid = tid0;   // re-laundered after every barrier (bwd_common.h: opaque)
  f32x4 acc[2] = {splat4(0.f), splat4(0.f)};   // 8 tiles over the waves: at most 2 per wave
  const auto wg_tab = wgrad_table<pb::kList, 3, 2>(wave, lane);
  float accB = 0.f;
  auto R = [&](int row) -> float* { return lds + row * kLD; };
  const int n_tiles = (P + kTT - 1) / kTT;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int p0 = tile * kTT;
    for (int idx = tid; idx < kTT * 24; idx += kBwdThreads) {
      if (idx < kTT * 8) {
        const int c = idx >> 3, i = idx & 7, p = p0 + c;
        R(O_S8 + i)[c] = p < P ? sim8[(size_t)p * 8 + i] : 0.f;
      } else {
        const int c = (idx - kTT * 8) >> 4, i = (idx - kTT * 8) & 15, p = p0 + c;
        R(O_DO + i)[c] = p < P ? d_pv[(size_t)p * 40 + 24 + i] : 0.f;
      }
    }
    auto pf14 = gemm_prefetch<32, 8, false>(wp.p[P_PS_W0], 8, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 8, false, LOWP>(pf14, wp.p[P_PS_W0], 8, R(O_S8), wave, lane,
                           [&](int r, int c, float v) { R(O_A1 + r)[c] = fmaxf(v + wp.p[P_PS_B0][r], 0.f); });
    auto pf15 = gemm_prefetch<32, 32, false>(wp.p[P_PS_W2], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 32, false, LOWP>(pf15, wp.p[P_PS_W2], 32, R(O_A1), wave, lane,
                            [&](int r, int c, float v) { R(O_A2 + r)[c] = fmaxf(v + wp.p[P_PS_B2][r], 0.f); });
    auto pf16 = gemm_prefetch<32, 16, true>(wp.p[P_PS_W4], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 16, true, LOWP>(pf16, wp.p[P_PS_W4], 32, R(O_DO), wave, lane,
                           [&](int r, int c, float v) { R(O_DA2 + r)[c] = R(O_A2 + r)[c] > 0.f ? v : 0.f; });
    auto pf17 = gemm_prefetch<32, 32, true>(wp.p[P_PS_W2], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 32, true, LOWP>(pf17, wp.p[P_PS_W2], 32, R(O_DA2), wave, lane,
                           [&](int r, int c, float v) { R(O_DA1 + r)[c] = R(O_A1 + r)[c] > 0.f ? v : 0.f; });
    __syncthreads();
    tid = opaque(tid0);
    if (tid < 16) accB += row_dot(R(O_DO + tid), nullptr, 0);
    else if (tid < 48) accB += row_dot(R(O_DA2 + (tid - 16)), nullptr, 0);
    else if (tid < 80) accB += row_dot(R(O_DA1 + (tid - 48)), nullptr, 0);
    wgrad_all<2, 0, LOWP>(acc, lds, wg_tab, lane);
    __syncthreads();
    tid = opaque(tid0);
  }
  wgrad_flush_all<pb::kList, 3, 2, 0>(acc, gp, wave, lane);
  if (tid < 16) atomic_add_f32(gp.p[P_PS_B4] + tid, accB);
  else if (tid < 48) atomic_add_f32(gp.p[P_PS_B2] + (tid - 16), accB);
  else if (tid < 80) atomic_add_f32(gp.p[P_PS_B0] + (tid - 48), accB);
}

template <bool LOWP>
static hipError_t launch_ray_bwd_t(const RawPtrs& wp, const GradPtrs& gp, const float* token0, const int* tok_row,
                                   bool accumulate, const float* order_pe, const float* d_srdf, int RN, int SN,
                                   float* d_tok_a, float* d_tok_b, float* dbg, hipStream_t s) {
  static LdsAttrOnce lds_attr;   // per instantiation; thread-safe, once per device
  if (const hipError_t attr = lds_attr.set(reinterpret_cast<const void*>(&ray_bwd_kernel<LOWP>), rb::kLdsBytes); attr != hipSuccess) return attr;
  const int blocks = RN < 256 ? RN : 256;
  hipLaunchKernelGGL(ray_bwd_kernel<LOWP>, dim3(blocks), dim3(kBwdThreads), rb::kLdsBytes, s, wp, gp, token0, tok_row,
                     accumulate ? 1 : 0, order_pe, d_srdf, RN, SN, d_tok_a, d_tok_b, dbg);
  return hipGetLastError();
}

hipError_t launch_ray_bwd(const RawPtrs& wp, const GradPtrs& gp, const float* token0, const int* tok_row, bool accumulate,
                          const float* order_pe, const float* d_srdf, int RN, int SN, float* d_tok_a, float* d_tok_b,
                          float* dbg, bool lowp, hipStream_t s) {
  if (SN % 16 != 0 || SN < 16) return hipErrorInvalidValue;   // a ray's last tile may hold 16 of the kTT = 32 tokens
  return lowp ? launch_ray_bwd_t<true>(wp, gp, token0, tok_row, accumulate, order_pe, d_srdf, RN, SN, d_tok_a, d_tok_b, dbg, s)
              : launch_ray_bwd_t<false>(wp, gp, token0, tok_row, accumulate, order_pe, d_srdf, RN, SN, d_tok_a, d_tok_b, dbg, s);
}

hipError_t launch_presim_bwd(const RawPtrs& wp, const GradPtrs& gp, const float* sim8, const float* d_pv, int P, bool lowp,
                             hipStream_t s) {
  const int n_tiles = (P + kTT - 1) / kTT;
  const dim3 grid(n_tiles < 256 ? n_tiles : 256), block(kBwdThreads);
  if (lowp) hipLaunchKernelGGL(presim_bwd_kernel<true>, grid, block, 0, s, wp, gp, sim8, d_pv, P);
  else hipLaunchKernelGGL(presim_bwd_kernel<false>, grid, block, 0, s, wp, gp, sim8, d_pv, P);
  return hipGetLastError();
}

}  // namespace ufr

#ifdef UFR_BWD_TIMING
extern "C" int ufr_debug_rb_phases(unsigned long long* out, int n, int reset) {
  unsigned long long h[64] = {};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ufr::g_rb_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 64; ++i) out[i] = h[i];
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ufr::g_rb_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
