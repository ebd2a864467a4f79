// Along-ray aggregation: order positional encoding, LoFTR linear-attention layer over the SN samples
// of a ray (d = 88, 8 heads of 11), DensityMLP -> signed ray distance.
//   RayTransformer.forward     code1/ray_transformer.py:296-307 (+ order_posenc :165-173)
//   LoFTREncoderLayer.forward  code1/attention/transformer.py:35-58
//   LinearAttention.forward    code1/attention/linear_attention.py:20-47
//
// One wavefront per ray, two sweeps over its SN/16 column tiles, everything in registers; the dense
// layers run split-precision on the bf16 matrix cores (ufr_layout_bf.h), the tiny per-head KV / message
// products (K = 16 tokens / 16 head dims) stay on the fp32 MFMA:
//  sweep 1: K^T, V^T tiles ([token][head dim], obtained by swapping the MFMA operands), then
//           KV_h += K'_h^T V_h as 4 MFMAs per head; a ones column appended to V makes
//           column 11 of KV_h the K' sum needed for the normaliser.
//  sweep 2: Q tiles, message_h = KV_h^T-chained MFMA with Q'_h (row 11 of the result is Q'.sum K'),
//           merge, LayerNorm, MLP, LayerNorm, residual, DensityMLP.
// Each head occupies its own 16-row tile (11 real rows) so that head boundaries coincide with MFMA
// tiles; the 88-wide activations use the "nat88" layout of ufr_layout.h.
#include "ufr_internal.h"
#include "weight_stream_bf.h"

namespace ufr {

// load the ray-transformer input tile: [token-0 feature (80) | order PE (8)] in nat88 layout
__device__ __forceinline__ void load_ray_tile(const float* __restrict__ token0, const int* __restrict__ tok_row,
                                              const float* __restrict__ order_pe, size_t tok_base, int s_base, int g,
                                              int j, f32x4 (&x)[1][6]) {
  const float* row = token0 + (tok_row ? (size_t)tok_row[tok_base + j] : tok_base + j) * UFR_TOKEN_DIM;
#pragma unroll
  for (int t = 0; t < 5; ++t) x[0][t] = ld4(row + 16 * t + 4 * g);
  const float* pe = order_pe + (size_t)(s_base + j) * 8 + 2 * g;  // features 80+2g, 81+2g in registers 0,1
  x[0][5] = f32x4{pe[0], pe[1], 0.f, 0.f};
}

// LayerNorm over 88 features in nat88 layout (tile 5: registers 0,1 real)
template <int VW, int VB>
__device__ __forceinline__ void layer_norm88(f32x4 (&t)[1][6], const WStreamBf& ws, int g) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) s += (t[0][i][0] + t[0][i][1]) + (t[0][i][2] + t[0][i][3]);
  s += t[0][5][0] + t[0][5][1];
  const float mean = sum_groups(s) * (1.f / 88.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (i < 5 || r < 2) {
        float d = t[0][i][r] - mean;
        q = fmaf(d, d, q);
      }
    }
  const float rstd = 1.f / sqrtf(sum_groups(q) * (1.f / 88.f) + 1e-5f);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const f32x4 gw = vec_frag<VW>(ws, i, g), gb = vec_frag<VB>(ws, i, g);  // zero in the padding slots
    t[0][i] = (t[0][i] - mean) * rstd * gw + gb;
  }
}

// 256-thread workgroups = 4 rays (one wave each, one per SIMD), two workgroups per CU; the four waves
// walk the weight streams B_RT1 / B_RT2 together through LDS (weight_stream_bf.h).
#ifndef UFR_RT_BLOCK
#define UFR_RT_BLOCK 256
#define UFR_RT_MINW 2
#endif
constexpr int kRtBlock = UFR_RT_BLOCK;
constexpr int kRtWaves = kRtBlock / 64;

__global__ void __launch_bounds__(kRtBlock, UFR_RT_MINW) ray_transformer_kernel(const float* __restrict__ packed,
                                                                  const float* __restrict__ token0,
                                                                  const int* __restrict__ tok_row,
                                                                  const float* __restrict__ order_pe, int RN, int SN,
                                                                  float* __restrict__ srdf,
                                                                  float* __restrict__ ray_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WStreamBf ws = wstream_bf_begin<kRtWaves>(packed, smem);
  wstream_bf_prime<B_RT1, kRtWaves>(ws);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int ray_raw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool valid = ray_raw < RN;           // no early exit: every wave meets every chunk barrier
  const int ray = valid ? ray_raw : RN - 1;
  const int n_tiles = SN / 16;
  // values / v_length (linear_attention.py:41): a multiply by 1/SN is exact only for power-of-two sample counts;
  // any other total (64 + 32, 48, ...) takes the true division the reference performs
  const float inv_len = 1.f / (float)SN, f_len = (float)SN;
  const bool pow2_len = (SN & (SN - 1)) == 0;

  // ---------------- sweep 1: KV_h[d][v] = sum_s K'_h[s][d] * V_h[s][v] / SN   (linear_attention.py:41-42)
  f32x4 KV[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) KV[h] = splat4(0.f);
  for (int tile = 0; tile < n_tiles; ++tile) {
    const bool wrap = tile + 1 < n_tiles;
    f32x4 x[1][6], kt[1][8], vt[1][8];
    load_ray_tile(token0, tok_row, order_pe, (size_t)ray * SN + tile * 16, tile * 16, g, j, x);
#pragma unroll
    for (int h = 0; h < 8; ++h) { kt[0][h] = splat4(0.f); vt[0][h] = splat4(0.f); }
    {  // swapped operands: kt[h], vt[h] rows = tokens 4g+r, column j = head dim; x is split once per k-step
      BWords<1> cur;
      split_units<0, 0, 4>(x, cur);
      static_for<3>([&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        BStep b[1];
        bwords_to_bstep(cur, b);
        if constexpr (s < 2) {
          BWords<1> nxt;
          gemm_bf_panel<M_RT_K, s, 1, kRtWaves, true>(ws, b, kt, wrap, [&](auto ti) __attribute__((always_inline)) {
            constexpr int to = decltype(ti)::value;
            split_units<s + 1, to * 4 / 8, (to + 1) * 4 / 8>(x, nxt);
          });
          gemm_bf_panel<M_RT_V, s, 1, kRtWaves, true>(ws, b, vt, wrap);
          cur = nxt;
        } else {
          gemm_bf_panel<M_RT_K, s, 1, kRtWaves, true>(ws, b, kt, wrap);
          gemm_bf_panel<M_RT_V, s, 1, kRtWaves, true>(ws, b, vt, wrap);
        }
      });
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float kk = j < 11 ? elu1(kt[0][h][r]) : 0.f;                      // padded dims contribute nothing
        const float vs = pow2_len ? vt[0][h][r] * inv_len : vt[0][h][r] / f_len;
        const float vv = j < 11 ? vs : (j == 11 ? 1.f : 0.f);                     // ones column -> sum of K'
        KV[h] = mfma16(kk, vv, KV[h]);
      }
    }
    wstream_bf_finish<B_RT1, kRtWaves>(ws, wrap);
  }

  // ---------------- sweep 2 (slot 0 is free: every wave passed the barrier that opened sweep 1's last chunk)
  wstream_bf_prime<B_RT2, kRtWaves>(ws);
  for (int tile = 0; tile < n_tiles; ++tile) {
    const bool wrap = tile + 1 < n_tiles;
    f32x4 x[1][6], q[1][8], msg[1][8];
    load_ray_tile(token0, tok_row, order_pe, (size_t)ray * SN + tile * 16, tile * 16, g, j, x);
#pragma unroll
    for (int h = 0; h < 8; ++h) q[0][h] = splat4(0.f);
    gemm_bf<M_RT_Q, 1, kRtWaves>(ws, x, q, wrap);  // q[h]: rows = head dims 4g+r, column j = token
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      f32x4 acc = splat4(0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float qq = (4 * g + r < 11) ? elu1(q[0][h][r]) : 0.f;
        acc = mfma16(KV[h][r], qq, acc);           // rows v = 4g+r: sum_d KV[d][v] Q'[d]; row 11 = Q'.sum(K')
      }
      const float den = __shfl(acc[3], 32 + j);    // row 11 lives in lane group 2, register 3
      const float Z = 1.f / (den + 1e-6f);         // linear_attention.py:43
      msg[0][h] = acc * (Z * (float)SN);           // :44  (rows >= 11 meet zero merge columns)
    }
    f32x4 m[1][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) m[0][t] = splat4(0.f);
    gemm_bf<M_RT_MERGE, 1, kRtWaves>(ws, msg, m, wrap);
    layer_norm88<V_RT_N1W, V_RT_N1B>(m, ws, g);

    f32x4 cat[1][12], hid[1][11], o[1][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { cat[0][t] = x[0][t]; cat[0][6 + t] = m[0][t]; }
#pragma unroll
    for (int t = 0; t < 11; ++t) hid[0][t] = splat4(0.f);
    gemm_bf<M_RT_MLP0, 1, kRtWaves>(ws, cat, hid, wrap);
#pragma unroll
    for (int t = 0; t < 11; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[0][t][r] = fmaxf(hid[0][t][r], 0.f);
#pragma unroll
    for (int t = 0; t < 6; ++t) o[0][t] = splat4(0.f);
    gemm_bf<M_RT_MLP2, 1, kRtWaves>(ws, hid, o, wrap);
    layer_norm88<V_RT_N2W, V_RT_N2B>(o, ws, g);
#pragma unroll
    for (int t = 0; t < 6; ++t) o[0][t] += x[0][t];

    if (ray_out && valid) {
      float* row = ray_out + ((size_t)ray * SN + tile * 16 + j) * UFR_RAY_DIM;
#pragma unroll
      for (int t = 0; t < 5; ++t) st4(row + 16 * t + 4 * g, o[0][t]);
      row[80 + 2 * g] = o[0][5][0];
      row[81 + 2 * g] = o[0][5][1];
    }

    // ---------------- DensityMLP 88 -> 32 -> 16 -> 1 (ray_transformer.py:147-150, 307)
    f32x4 d1[1][2], d2[1][1], d3[1][1];
    d1[0][0] = vec_frag<V_DM_B0>(ws, 0, g);
    d1[0][1] = vec_frag<V_DM_B0>(ws, 1, g);
    d2[0][0] = vec_frag<V_DM_B2>(ws, 0, g);
    d3[0][0] = vec_frag<V_DM_B4>(ws, 0, g);
    gemm_bf<M_DM0, 1, kRtWaves>(ws, o, d1, wrap);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) d1[0][t][r] = fmaxf(d1[0][t][r], 0.f);
    gemm_bf<M_DM2, 1, kRtWaves>(ws, d1, d2, wrap);
#pragma unroll
    for (int r = 0; r < 4; ++r) d2[0][0][r] = fmaxf(d2[0][0][r], 0.f);
    gemm_bf<M_DM4, 1, kRtWaves>(ws, d2, d3, wrap);
    if (g == 0 && valid) srdf[(size_t)ray * SN + tile * 16 + j] = d3[0][0][0];
    wstream_bf_finish<B_RT2, kRtWaves>(ws, wrap);
  }
}

hipError_t launch_ray_transformer(const float* packed, const float* token0, const int* tok_row, const float* order_pe,
                                  int RN, int SN, float* srdf, float* ray_out, hipStream_t s) {
  if (SN % 16 != 0 || SN < 16) return hipErrorInvalidValue;
  // the attribute is per device: set it once on every device this process launches on
  static bool attr_set[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&ray_transformer_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kBfLdsBytes);
    if (attr != hipSuccess) return attr;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(ray_transformer_kernel, dim3((RN + kRtWaves - 1) / kRtWaves), dim3(kRtBlock), kBfLdsBytes, s,
                     packed, token0, tok_row, order_pe, RN, SN, srdf, ray_out);
  return hipGetLastError();
}

}  // namespace ufr
