// Data-gradient chain of the along-ray aggregation (round 4; replaces the barrier-phased ray_bwd kernel).
//   autograd of RayTransformer.forward     code1/ray_transformer.py:296-307
//               LoFTREncoderLayer.forward   code1/attention/transformer.py:35-58
//               LinearAttention.forward     code1/attention/linear_attention.py:20-47
// Built like the forward kernel (ray_transformer.hip): one wavefront per ray, the ray's tokens in tiles of 16 MFMA columns,
// cotangents chained through registers into the TRANSPOSED weights (bf16 planes streamed through the LDS ring,
// ufr_layout_f16.h: B_RTB1 / B_RTB2), everything the chain needs from the forward read from the tape
// (bwd_tape.h, written by ray_transformer_kernel<.., TAPE>), every layer-output cotangent written as a tile for the
// weight-gradient contraction (wgrad_stream.hip).  Linear attention couples the tokens of a ray through the per-head
// state KV_h = sum_t K'_t (x) V_t, so the walk has two sweeps:
//   sweep 1, per tile: DensityMLP .. merge backwards to d msg;  d acc = Zs d msg, d den from d msg . msg;
//            d Q' = KV_h d acc (four fp32 MFMAs per head against the taped KV_h^T; the ones column of V carries d den);
//            d KV_h += Q' (x) d acc over the tile's tokens -- both factors transposed to [token][slot] ON THE MATRIX CORE (four
//            fp32 MFMAs against the identity, exact) and contracted like the forward's KV accumulation;  d q -> q^T -> d x
//   sweep 2, per tile: k, v recomputed in the plain orientation (the forward's fp16 planes again);
//            d V = d KV_h^T-contract K', d K' = d KV_h V (fp32 MFMAs);  d k, d v -> k^T, v^T -> d x;  d token0 rows out
// LayerNorm gamma / beta are reduced here (DPP row all-reduce into per-wave LDS accumulators, flushed once); all other
// parameter gradients are contractions of the tiles.
#include "bwd_common.h"   // GradPtrs
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"

namespace ufr {

constexpr int kRdBlock = 256, kRdWaves = 4;
constexpr int kRdAccFloats = 4 * 3 * 64;   // per wave: 4 vectors (norm1 / norm2 gamma, beta) x 3 blocks of ten values x 64 lanes

// v[i] <- sum of v[i] over the 16 lanes of the DPP row (view_dgrad.hip: row_allreduce10)
__device__ __forceinline__ void rd_row_allreduce10(float (&v)[10]) {
#define UFR_RR_STEP(CTRL)                                                    \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %8, %8, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"        \
  "v_add_f32_dpp %9, %9, %9 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" UFR_RR_STEP("quad_perm:[1,0,3,2]") UFR_RR_STEP("quad_perm:[2,3,0,1]") UFR_RR_STEP("row_half_mirror")
               UFR_RR_STEP("row_mirror")
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
#undef UFR_RR_STEP
}
// acc[64 b] += sum over the 16 token lanes of value 10 b + j of a nat88 vector (6 tiles x 4 registers = 24 values per lane
// group, the padding ones zero): lane j < 10 keeps the j-th sum of block b
__device__ __forceinline__ void rd_reduce_acc88(const f32x4 (&t)[6], float* acc, int j) {
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    float v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = 10 * b + i < 24 ? t[(10 * b + i) >> 2][(10 * b + i) & 3] : 0.f;
    rd_row_allreduce10(v);
    float mine = v[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) mine = j == i ? v[i] : mine;
    acc[64 * b] += mine;
  }
}
__device__ __forceinline__ void rd_flush88(const float* acc, float* __restrict__ dst, int g, int j) {
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int idx = 10 * b + j;
    const int f = idx < 24 ? nat88(idx >> 2, g, idx & 3) : -1;
    if (j < 10 && f >= 0) unsafeAtomicAdd(dst + f, acc[64 * b]);
  }
}

// X^T of a 16 x 16 fp32 tile in accumulator layout (lane (g, j): rows 4g + r, column j), exact: four fp32 MFMAs against
// the identity -- MFMA r takes A[m = j][k = g] = X[4g + r][j] and B[k = g][n] = (n == 4g + r): D[m][n] = X[n][m]
__device__ __forceinline__ f32x4 transpose_tile(const f32x4& x, const f32x4& ident) {
  f32x4 t = splat4(0.f);
#pragma unroll
  for (int r = 0; r < 4; ++r) t = mfma16(x[r], ident[r], t);
  return t;
}

template <bool LOWP>
__global__ void __launch_bounds__(kRdBlock, 2) ray_dgrad_kernel(const float* __restrict__ packed, const float* __restrict__ tape,
                                                               const float* __restrict__ ray_state,
                                                               const float* __restrict__ d_srdf,
                                                               const int* __restrict__ tok_row, int accumulate, int RN, int SN,
                                                               float* __restrict__ dbuf, float* __restrict__ d_tok_a,
                                                               float* __restrict__ d_tok_b, float* __restrict__ g_n1w,
                                                               float* __restrict__ g_n1b, float* __restrict__ g_n2w,
                                                               float* __restrict__ g_n2b) {
  typedef RayTapeLayout<LOWP> TapeL;
  typedef RayGradLayout<LOWP> GradL;
  constexpr int C = 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto ws = wstream_f16_begin<kRdWaves, LOWP>(packed, smem);
  wstream_f16_prime<B_RTB1, kRdWaves>(ws);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int ray_raw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool valid = ray_raw < RN;           // no early exit: every wave meets every chunk barrier
  const int ray = valid ? ray_raw : RN - 1;
  const int n_tiles = SN / 16, n2 = n_tiles + (n_tiles & 1), nb = n2 / 2;
  const float fSN = (float)SN;
  const auto sc = scalar_file<false, ray_scalars_offset()>(ws);   // sweep 2 recomputes k, v from the forward planes
  const float xs_x = sc[RS_XS_X], k_dsc = sc[RS_K_DSC], k_l2e = sc[RS_K_L2E];
  const float inv_len = uniform_f32(sc[RS_V_DSC] / fSN), f_len = uniform_f32(fSN * sc[RS_V_ASC]);   // values / v_length on raw accumulators (ray_transformer.hip)
  const bool pow2_len = (SN & (SN - 1)) == 0;

  float* const a_base = reinterpret_cast<float*>(smem + kF16LdsBytes) + (threadIdx.x >> 6) * kRdAccFloats + lane;
  float* const a_n1w = a_base;
  float* const a_n1b = a_base + 192;
  float* const a_n2w = a_base + 384;
  float* const a_n2b = a_base + 576;
#pragma unroll
  for (int i = 0; i < 12; ++i) a_base[64 * i] = 0.f;

  const f32x4 ident = {j == 4 * g ? 1.f : 0.f, j == 4 * g + 1 ? 1.f : 0.f, j == 4 * g + 2 ? 1.f : 0.f, j == 4 * g + 3 ? 1.f : 0.f};
  f32x4 KVT[8], dKV[8];
  {
    const float* st = ray_state + (size_t)ray * (kRayStateTiles * kTileFloats) + lane * 4;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      KVT[h] = ld4(st + (8 + h) * kTileFloats);
      dKV[h] = splat4(0.f);
    }
  }
  const char* const tape_ray = reinterpret_cast<const char*>(tape) + (size_t)ray * nb * (TapeL::block_units * 512);
  char* const dy_ray = reinterpret_cast<char*>(dbuf) + (size_t)ray * nb * (GradL::block_units * 512);

  // ================================================================ sweep 1
  for (int it = 0; it < n2; ++it) {
    const bool wrap = it + 1 < n2;
    const bool live = valid && it < n_tiles;
    const char* const tape_blk = tape_ray + (size_t)(it >> 1) * (TapeL::block_units * 512);
    char* const dy_blk = dy_ray + (size_t)(it >> 1) * (GradL::block_units * 512);
    const int c = it & 1;
    auto tape_ld = [&](int tile) __attribute__((always_inline)) -> f32x4 { return tile_load<TapeL>(tape_blk, tile, c, lane); };
    auto dy_st = [&](int tile, f32x4 v) __attribute__((always_inline)) {
      if (valid) tile_store<GradL>(dy_blk, tile, c, lane, v);       // a padding tile's cotangents are written too: zeros
    };

    const f32x4 misc = tape_ld(RT_MISC);
    const float rstd1 = misc[0], rstd2 = misc[1];
    // (through scalars: __builtin_bit_cast applied to an ext_vector ELEMENT reads element 0 with this hipcc)
    const float misc2 = misc[2], misc3 = misc[3];
    const unsigned bits0 = __builtin_bit_cast(unsigned, misc2), bits1 = __builtin_bit_cast(unsigned, misc3);
    auto relu_on = [&](int bit) __attribute__((always_inline)) -> bool { return ((bit < 32 ? bits0 >> bit : bits1 >> (bit - 32)) & 1u) != 0u; };
    const float dsr = live ? d_srdf[(size_t)ray * SN + 16 * it + j] : 0.f;

    // ---------------- DensityMLP backwards (ray_transformer.py:147-150, 307)
    f32x4 dd2[C][1], dd1[C][2], dout[C][6];
    {
      const f32x4 w4 = vec_frag<V_DM_W4>(ws, 0, g);            // rows 4g + r of the 16-vector
#pragma unroll
      for (int r = 0; r < 4; ++r) dd2[0][0][r] = relu_on(52 + r) ? w4[r] * dsr : 0.f;
      dy_st(DR_SR, f32x4{g == 0 ? dsr : 0.f, 0.f, 0.f, 0.f});
      dy_st(DR_D2, dd2[0][0]);
      dd1[0][0] = dd1[0][1] = splat4(0.f);
    }
    gemm_f16<M_DM2T, C, kRdWaves>(ws, dd2, dd1, wrap);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dd1[0][t][r] = relu_on(44 + 4 * t + r) ? dd1[0][t][r] : 0.f;
      dy_st(DR_D1 + t, dd1[0][t]);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) dout[0][t] = splat4(0.f);
    gemm_f16<M_DM0T, C, kRdWaves>(ws, dd1, dout, wrap);

    // ---------------- LayerNorm2 backwards (transformer.py:56-58); o = x + LN2(opre): d x starts as d o (parked in DR_SCR)
    f32x4 dopre[C][6];
    {
      // (loads before stores, and no read-modify-write of the scratch tiles: view_dgrad.hip on the order of memory operations)
      f32x4 xh[6], gy[6], dgam[6];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 6; ++t) xh[t] = tape_ld(RT_XH2 + t);
#pragma unroll
      for (int t = 0; t < 6; ++t) dy_st(DR_SCR + t, dout[0][t]);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        dgam[t] = dout[0][t] * xh[t];
        gy[t] = dout[0][t] * vec_frag<V_RT_N2W>(ws, t, g);        // gamma is zero in the padding slots
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += gy[t][r];
          s2 = fmaf(gy[t][r], xh[t][r], s2);
        }
      }
      const float m1 = sum_groups(s1) * (1.f / 88.f), m2 = sum_groups(s2) * (1.f / 88.f);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        dopre[0][t] = (gy[t] - m1 - xh[t] * m2) * rstd2;
        if (t == 5) { dopre[0][t][2] = 0.f; dopre[0][t][3] = 0.f; }   // nat88 padding: no such feature
        dy_st(DR_OPRE + t, dopre[0][t]);
      }
      rd_reduce_acc88(dgam, a_n2w, j);
      rd_reduce_acc88(dout[0], a_n2b, j);
    }

    // ---------------- MLP backwards (transformer.py:55-56)
    f32x4 dhid[C][11];
#pragma unroll
    for (int t = 0; t < 11; ++t) dhid[0][t] = splat4(0.f);
    gemm_f16<M_RT_MLP2T, C, kRdWaves>(ws, dopre, dhid, wrap);
#pragma unroll
    for (int t = 0; t < 11; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dhid[0][t][r] = relu_on(4 * t + r) ? dhid[0][t][r] : 0.f;
      dy_st(DR_HID + t, dhid[0][t]);
    }
    f32x4 dcat[C][12];
#pragma unroll
    for (int t = 0; t < 12; ++t) dcat[0][t] = splat4(0.f);
    gemm_f16<M_RT_MLP0T, C, kRdWaves>(ws, dhid, dcat, wrap);
    // ---------------- LayerNorm1 backwards on the message half (transformer.py:52); the x half joins the scratch
    f32x4 dmpre[C][6];
    {
      f32x4 xh[6], gm[6], dgam[6], dbet[6];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 6; ++t) xh[t] = tape_ld(RT_XH1 + t);
#pragma unroll
      for (int t = 0; t < 6; ++t) dy_st(DR_SCR2 + t, dcat[0][t]);   // a second scratch set: sweep 2 adds the three up
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        dbet[t] = dcat[0][6 + t];
        dgam[t] = dbet[t] * xh[t];
        gm[t] = dbet[t] * vec_frag<V_RT_N1W>(ws, t, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += gm[t][r];
          s2 = fmaf(gm[t][r], xh[t][r], s2);
        }
      }
      const float m1 = sum_groups(s1) * (1.f / 88.f), m2 = sum_groups(s2) * (1.f / 88.f);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        dmpre[0][t] = (gm[t] - m1 - xh[t] * m2) * rstd1;
        if (t == 5) { dmpre[0][t][2] = 0.f; dmpre[0][t][3] = 0.f; }
        dy_st(DR_MPRE + t, dmpre[0][t]);
      }
      rd_reduce_acc88(dgam, a_n1w, j);
      rd_reduce_acc88(dbet, a_n1b, j);
    }
    // ---------------- merge backwards: d msg, quad-packed like the message (quad 3h + q: head h, dims 3g + q)
    f32x4 dmsg[C][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) dmsg[0][t] = splat4(0.f);
    gemm_f16<M_RT_MERGET, C, kRdWaves>(ws, dmpre, dmsg, wrap);

    // ---------------- linear attention backwards, query side (linear_attention.py:41-44):
    //   acc[v] = sum_d KV[d][v] Q'[d] (slot 3 of a head tile = the ones column: acc = Q'.sum K' = den),  Zs = SN / (den + eps),
    //   msg = Zs acc   =>   d acc = Zs d msg,  d den = -(d msg . acc) Zs^2 / SN = -(d msg . msg) Zs / SN
    f32x4 dq[C][6];
    {
      f32x4 qt[6], mt[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) { qt[t] = tape_ld(RT_Q + t); mt[t] = tape_ld(RT_MSG + t); }
      const f32x4 zs0 = tape_ld(RT_ZS), zs1 = tape_ld(RT_ZS + 1);
      static_for<8>([&](auto hi) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value;
        const float zs = h < 4 ? zs0[h & 3] : zs1[h & 3];
        float part = 0.f;
        f32x4 dacc, qx;
        static_for<3>([&](auto qi) __attribute__((always_inline)) {
          constexpr int qd = decltype(qi)::value, quad = 3 * h + qd;
          const float dm = dmsg[0][quad >> 2][quad & 3];
          part = fmaf(dm, mt[quad >> 2][quad & 3], part);
          dacc[qd] = zs * dm;
          qx[qd] = qt[quad >> 2][quad & 3];
        });
        const float dden = -sum_groups(part) * zs / fSN;
        dacc[3] = g == 0 ? dden : 0.f;
        qx[3] = 0.f;
        f32x4 dQ = splat4(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) dQ = mfma16(KVT[h][r], dacc[r], dQ);     // rows = K slots 4g + r, column = token
        static_for<3>([&](auto qi) __attribute__((always_inline)) {
          constexpr int qd = decltype(qi)::value, quad = 3 * h + qd;
          dq[0][quad >> 2][quad & 3] = dQ[qd] * (qx[qd] > 1.f ? 1.f : qx[qd]);   // elu'(q) = (Q' > 1 ? 1 : Q'); padding: Q' = 0
        });
        // d KV_h += sum over the tile's tokens of Q' (x) d acc: both to [token][slot], then the forward's KV contraction
        const f32x4 qT = transpose_tile(qx, ident), aT = transpose_tile(dacc, ident);
#pragma unroll
        for (int r = 0; r < 4; ++r) dKV[h] = mfma16(qT[r], aT[r], dKV[h]);
      });
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) dy_st(DR_Q + t, dq[0][t]);
    // ---------------- q projection backwards; the scratch now holds d o + d cat[0..87] + this
    f32x4 dx[C][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) dx[0][t] = splat4(0.f);
    gemm_f16<M_RT_QT, C, kRdWaves>(ws, dq, dx, wrap);
#pragma unroll
    for (int t = 0; t < 6; ++t) dy_st(DR_SCR3 + t, dx[0][t]);
    wstream_f16_finish<B_RTB1, kRdWaves>(ws, wrap);
  }

  // ================================================================ sweep 2 (the ring is free: every wave passed sweep 1's last barrier)
  f32x4 dKVT[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) dKVT[h] = transpose_tile(dKV[h], ident);
  wstream_f16_prime<B_RTB2, kRdWaves>(ws);
  for (int it = 0; it < n2; ++it) {
    const bool wrap = it + 1 < n2;
    const bool live = valid && it < n_tiles;
    const char* const tape_blk = tape_ray + (size_t)(it >> 1) * (TapeL::block_units * 512);
    char* const dy_blk = dy_ray + (size_t)(it >> 1) * (GradL::block_units * 512);
    const int c = it & 1;
    f32x4 x[C][6], kt[C][8], vt[C][8];
#pragma unroll
    for (int t = 0; t < 6; ++t) x[0][t] = tile_load<TapeL>(tape_blk, RT_X + t, c, lane);
#pragma unroll
    for (int h = 0; h < 8; ++h) { kt[0][h] = splat4(0.f); vt[0][h] = splat4(0.f); }
    {  // k, v in the plain orientation: rows = the head tile's slots 4g + r, column = token; x is split once per k-step
      BWords<C> cur;
      split_units<0, 0, 4 * C>(x, cur, xs_x);
      static_for<3>([&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        BStep b[C];
        bwords_to_bstep(cur, b);
        if constexpr (s < 2) {
          BWords<C> nxt;
          gemm_f16_panel<M_RT_K, s, C, kRdWaves, false, B_RTB2>(ws, b, kt, wrap, [&](auto ti) __attribute__((always_inline)) {
            constexpr int to = decltype(ti)::value;
            split_units<s + 1, to * 4 * C / 8, (to + 1) * 4 * C / 8>(x, nxt, xs_x);
          });
          gemm_f16_panel<M_RT_V, s, C, kRdWaves, false, B_RTB2>(ws, b, vt, wrap);
          cur = nxt;
        } else {
          gemm_f16_panel<M_RT_K, s, C, kRdWaves, false, B_RTB2>(ws, b, kt, wrap);
          gemm_f16_panel<M_RT_V, s, C, kRdWaves, false, B_RTB2>(ws, b, vt, wrap);
        }
      });
    }
    // ---------------- key / value side: d V[v][t] = sum_d d KV[d][v] K'[d][t],  d K'[d][t] = sum_v d KV[d][v] V[v][t]
    f32x4 dk[C][8], dv[C][8];
    static_for<8>([&](auto hi) __attribute__((always_inline)) {
      constexpr int h = decltype(hi)::value;
      f32x4 Kp, Vx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool slot_ok = head11_slot(4 * g + r) >= 0;
        Kp[r] = slot_ok ? elu1_acc(kt[0][h][r], k_dsc, k_l2e) : 0.f;
        const float vs = pow2_len ? vt[0][h][r] * inv_len : vt[0][h][r] / f_len;
        Vx[r] = slot_ok ? vs : ((g == 0 && r == 3) ? 1.f : 0.f);              // ones column (slot 3) <-> the K' sum
      }
      f32x4 dV = splat4(0.f), dK = splat4(0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) dV = mfma16(dKV[h][r], Kp[r], dV);
#pragma unroll
      for (int r = 0; r < 4; ++r) dK = mfma16(dKVT[h][r], Vx[r], dK);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = live && head11_slot(4 * g + r) >= 0;
        dk[0][h][r] = on ? dK[r] * (Kp[r] > 1.f ? 1.f : Kp[r]) : 0.f;
        dv[0][h][r] = on ? dV[r] / fSN : 0.f;                                  // values = v / SN
      }
      if (valid) {
        tile_store<GradL>(dy_blk, DR_K + h, c, lane, dk[0][h]);
        tile_store<GradL>(dy_blk, DR_V + h, c, lane, dv[0][h]);
      }
    });
    f32x4 dx[C][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) dx[0][t] = splat4(0.f);
    gemm_f16<M_RT_KT, C, kRdWaves>(ws, dk, dx, wrap);
    gemm_f16<M_RT_VT, C, kRdWaves>(ws, dv, dx, wrap);
    // ---------------- d token0 rows: features 0..79 of d x (the order code has no consumer)
    if (live) {
      const size_t slot = (size_t)ray * SN + 16 * it + j;
      const size_t row = tok_row ? (size_t)tok_row[slot] : slot;
      float* pa = d_tok_a + row * UFR_TOKEN_DIM + 4 * g;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        f32x4 v = dx[0][t] + (tile_load<GradL>(dy_blk, DR_SCR + t, c, lane) + tile_load<GradL>(dy_blk, DR_SCR2 + t, c, lane)) +
                  tile_load<GradL>(dy_blk, DR_SCR3 + t, c, lane);
        if (accumulate) v += ld4(pa + 16 * t);
        st4(pa + 16 * t, v);
      }
      if (!accumulate && d_tok_b) {
        float* pb = d_tok_b + row * UFR_TOKEN_DIM + 4 * g;
#pragma unroll
        for (int t = 0; t < 5; ++t) st4(pb + 16 * t, splat4(0.f));
      }
    }
    wstream_f16_finish<B_RTB2, kRdWaves>(ws, wrap);
  }
  if (valid) {
    rd_flush88(a_n1w, g_n1w, g, j);
    rd_flush88(a_n1b, g_n1b, g, j);
    rd_flush88(a_n2w, g_n2w, g, j);
    rd_flush88(a_n2b, g_n2b, g, j);
  }
}

template <bool LOWP>
static hipError_t launch_rd(const float* packed, const float* tape, const float* ray_state, const float* d_srdf, const int* tok_row,
                            bool accumulate, int RN, int SN, float* dbuf, float* d_tok_a, float* d_tok_b, const GradPtrs& gp,
                            hipStream_t s) {
  static LdsAttrOnce lds_attr;
  constexpr int lds = kF16LdsBytes + kRdWaves * kRdAccFloats * 4;
  if (const hipError_t attr = lds_attr.set(reinterpret_cast<const void*>(&ray_dgrad_kernel<LOWP>), lds); attr != hipSuccess) return attr;
  hipLaunchKernelGGL(ray_dgrad_kernel<LOWP>, dim3((RN + kRdWaves - 1) / kRdWaves), dim3(kRdBlock), lds, s, packed, tape, ray_state,
                     d_srdf, tok_row, accumulate ? 1 : 0, RN, SN, dbuf, d_tok_a, d_tok_b, gp.p[P_RT_N1W], gp.p[P_RT_N1B],
                     gp.p[P_RT_N2W], gp.p[P_RT_N2B]);
  return hipGetLastError();
}

hipError_t launch_ray_dgrad(const float* packed, const float* tape, const float* ray_state, const float* d_srdf, const int* tok_row,
                            bool accumulate, int RN, int SN, float* dbuf, float* d_tok_a, float* d_tok_b, const GradPtrs& gp, bool lowp,
                            hipStream_t s) {
  if (SN % 16 != 0 || SN < 16) return hipErrorInvalidValue;
  return lowp ? launch_rd<true>(packed, tape, ray_state, d_srdf, tok_row, accumulate, RN, SN, dbuf, d_tok_a, d_tok_b, gp, s)
              : launch_rd<false>(packed, tape, ray_state, d_srdf, tok_row, accumulate, RN, SN, dbuf, d_tok_a, d_tok_b, gp, s);
}

}  // namespace ufr
