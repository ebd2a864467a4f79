// Along-ray aggregation: order positional encoding, LoFTR linear-attention layer over the SN samples
// of a ray (d = 88, 8 heads of 11), DensityMLP -> signed ray distance.
//   RayTransformer.forward     code1/ray_transformer.py:296-307 (+ order_posenc :165-173)
//   LoFTREncoderLayer.forward  code1/attention/transformer.py:35-58
//   LinearAttention.forward    code1/attention/linear_attention.py:20-47
//
// One wavefront per ray, two sweeps over its SN/16 column tiles, everything in registers; the dense
// layers run split-precision on the fp16 matrix cores (ufr_layout_f16.h), the tiny per-head KV / message
// products (K = 16 tokens / 16 head dims) stay on the fp32 MFMA:
//  sweep 1: K^T, V^T tiles ([token][head slot], obtained by swapping the MFMA operands), then
//           KV_h += K'_h^T V_h as 4 MFMAs per head (k = the tile's 16 tokens); a ones column in a padding
//           slot of V makes that column of KV_h the K' sum needed for the normaliser.
//  sweep 2: Q, message_h = KV_h^T-chained MFMA with Q'_h (k = head dims: THREE MFMAs per head, the padding
//           row 3 of the result is Q'.sum K'), merge, LayerNorm, MLP, LayerNorm, residual, DensityMLP.
// K and V give each head its own 16-column tile with the 11 dims in the slots 4g + r, r < 3 (ufr_layout.h:
// head11_slot), which makes register 3 of every KV / message tile padding: Q is computed for the live registers
// only (24 "quads" = 6 tiles instead of 8) and merge contracts 6 input tiles instead of 8 -- weight rows and columns
// are permuted at pack time, the kernel only renames registers.  The 88-wide activations use the "nat88" layout.
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"

namespace ufr {

// load the ray-transformer input tile: [token-0 feature (80) | order PE (8)] in nat88 layout
// Column tiles per wave and iteration.  Sweep 1 (K, V and the per-head state) always walks ONE 16-token tile: its two sets of
// eight head tiles fill the registers.  Sweep 2 (Q, message, merge, MLP, DensityMLP: three quarters of the weight stream)
// walks UFR_RT_C2 tiles per pass over the weights.  C2 = 2 (a weight fragment read from LDS feeds 6 MFMAs and the stream
// is fetched half as often per token; the registers come from the per-head state KV_h waiting in LDS between the sweeps,
// 8 KiB per wave) was built and measured in round 5: bit-identical output, static instructions per two tiles 8 166 ->
// ~6 900, but 256 registers with 36 spilled and 0.391 vs 0.377 ms per 4096 x 128 launch on the same box -- like the
// round-2 attempt over both sweeps (0.706 vs 0.641), the ray kernel is not bound by its weight stream.  Kept as a switch.
#ifndef UFR_RT_C2
#define UFR_RT_C2 1
#endif
constexpr int kRtC2 = UFR_RT_C2;
// The NEXT tile's token rows are requested at the top of an iteration (24 registers): per-phase cycle counters put the
// K / V and Q GEMMs -- the phases that open with the row-table lookup and the dependent row loads -- at 57 and 72 cycles
// per MFMA against 33 for MLP0.
#ifndef UFR_RT_LOCAL_DEN
#define UFR_RT_LOCAL_DEN 1   // forward-only build: the ones column in every lane group's padding slot (see the KV accumulation)
#endif
#ifndef UFR_RT_PREFETCH
#define UFR_RT_PREFETCH 0   // measured round 5: 0.378 vs 0.379 ms per 4096 x 128 launch -- the partner wave already covers those loads (as in round 3)
#endif

// UFR_RT_NT (development): bit 0 / bit 1 = sweep 1 / sweep 2 read the token rows with the non-temporal hint.  The rows are
// read twice (2.05 x the algorithmic bytes reach the fabric: profiles/r5_pmc.json); measured round 6, see DESIGN.md.
#ifndef UFR_RT_NT
#define UFR_RT_NT 0
#endif
template <bool NT = false>
__device__ __forceinline__ void load_ray_tile(const float* __restrict__ token0, const int* __restrict__ tok_row,
                                              const float* __restrict__ order_pe, size_t tok_base, int s_base, int g,
                                              int j, f32x4 (&x)[6]) {
  const float* row = token0 + (tok_row ? (size_t)tok_row[tok_base + j] : tok_base + j) * UFR_TOKEN_DIM;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if constexpr (NT) x[t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + 16 * t + 4 * g));
    else x[t] = ld4(row + 16 * t + 4 * g);
  }
  const float* pe = order_pe + (size_t)(s_base + j) * 8 + 2 * g;  // features 80+2g, 81+2g in registers 0,1
  x[5] = f32x4{pe[0], pe[1], 0.f, 0.f};
}

// LayerNorm over 88 features in nat88 layout (tile 5: registers 0,1 real)
// XH / RS (TAPE builds): the normalised input and 1 / sigma of the TRUE values, which the backward needs
template <int VW, int VB, int C, class WS>
__device__ __forceinline__ void layer_norm88(f32x4 (&tt)[C][6], const WS& ws, int g, float eps, float asc, f32x4 (*XH)[6] = nullptr, float* RS = nullptr) {
  // eps = 1e-5 asc^2: raw accumulators (asc times the values), view_transformer.hip layer_norm80
#pragma unroll
  for (int c = 0; c < C; ++c) {
    f32x4 (&t)[6] = tt[c];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) s += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
    s += t[5][0] + t[5][1];
    const float mean = sum_groups(s) * (1.f / 88.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (i < 5 || r < 2) {
          float d = t[i][r] - mean;
          q = fmaf(d, d, q);
        }
      }
    const float rstd = fast_rsqrt(sum_groups(q) * (1.f / 88.f) + eps);
    if (RS) RS[c] = rstd * asc;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const f32x4 gw = vec_frag<VW>(ws, i, g), gb = vec_frag<VB>(ws, i, g);  // zero in the padding slots
      // element by element: f32x4 expressions become packed-f32 VALU, an anti-lever beside MFMAs (view_transformer.hip)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (t[i][r] - mean) * rstd;
        if (XH) XH[c][i][r] = (i == 5 && r >= 2) ? 0.f : xh;                  // nat88: registers 2, 3 of tile 5 are padding
        t[i][r] = xh * gw[r] + gb[r];
      }
    }
  }
}

// 256-thread workgroups = 4 rays (one wave each, one per SIMD), two workgroups per CU; the four waves
// walk the weight streams B_RT1 / B_RT2 together through LDS (weight_stream_f16.h).
#ifndef UFR_RT_BLOCK
#define UFR_RT_BLOCK 256
#define UFR_RT_MINW 2
#endif
constexpr int kRtBlock = UFR_RT_BLOCK;
constexpr int kRtWaves = kRtBlock / 64;

#ifdef UFR_PHASE_TIMING  // development build: cycle counts per phase of wave 5 (tools/bench_kernels.py prints them)
__device__ unsigned long long g_rt_phase[32];
#define UFR_RT_PHASE(i)                                                                \
  {                                                                                    \
    const unsigned long long t_now = __builtin_readcyclecounter();                     \
    rt_acc[i] += t_now - rt_prev;                                                      \
    rt_prev = t_now;                                                                   \
  }
#else
#define UFR_RT_PHASE(i)
#endif

// TAPE: the instantiation the backward launches (bwd_tape.h): the same arithmetic, plus one store per activation tile, the
// transposed per-head state KV_h^T beside KV_h, and an even number of sweep-2 tiles (blocks of two).
constexpr int kRtKvLdsBytes = kRtWaves * 8 * 1024;     // the waves' per-head states between the sweeps (C2 > 1)
template <bool LOWP, bool TAPE = false, int C2 = 1>
__global__ void __launch_bounds__(kRtBlock, TAPE ? 2 : UFR_RT_MINW) ray_transformer_kernel(const float* __restrict__ packed,
                                                                  const float* __restrict__ token0,
                                                                  const int* __restrict__ tok_row,
                                                                  const float* __restrict__ order_pe, int RN, int SN,
                                                                  float* __restrict__ srdf,
                                                                  float* __restrict__ ray_out,
                                                                  int* __restrict__ status, float* __restrict__ tape = nullptr,
                                                                  float* __restrict__ ray_state = nullptr) {
  static_assert(!TAPE || C2 == 1, "the tape build walks one tile per iteration");
  constexpr bool kKvLds = C2 > 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto ws = wstream_f16_begin<kRtWaves, LOWP>(packed, smem);
  wstream_f16_prime<B_RT1, kRtWaves>(ws);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int ray_raw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool valid = ray_raw < RN;           // no early exit: every wave meets every chunk barrier
  const int ray = valid ? ray_raw : RN - 1;
  const int n_tiles = SN / 16;
  const int n_iter = n_tiles;                 // sweep 1: one tile per iteration
  // values / v_length (linear_attention.py:41): a multiply by 1/SN is exact only for power-of-two sample counts;
  // any other total (64 + 32, 48, ...) takes the true division the reference performs
  // the layers' plane / accumulator scales (ufr_layout.h: RayScalar; weight_stream_f16.h: ScalarFile -- this kernel has the
  // scalar registers to keep them all resident).  q, k, v and mlp0 all split x: their a_M agree by construction
  // (prep.hip), one multiplier (RS_XS_X) serves the four.
  const auto sc = scalar_file<false, ray_scalars_offset()>(ws);
  const float xs_x = sc[RS_XS_X];
  const float k_dsc = sc[RS_K_DSC], k_l2e = sc[RS_K_L2E], q_dsc = sc[RS_Q_DSC], q_l2e = sc[RS_Q_L2E];
  const float inv_len = uniform_f32(sc[RS_V_DSC] / (float)SN), f_len = uniform_f32((float)SN * sc[RS_V_ASC]);   // applied to raw accumulators
  const bool pow2_len = (SN & (SN - 1)) == 0;

#ifdef UFR_PHASE_TIMING
  unsigned long long rt_acc[16] = {};
  unsigned long long rt_prev = __builtin_readcyclecounter();
#endif
  // ---------------- sweep 1: KV_h[d][v] = sum_s K'_h[s][d] * V_h[s][v] / SN   (linear_attention.py:41-42)
  f32x4 KV[8], KVT[TAPE ? 8 : 1];
#pragma unroll
  for (int h = 0; h < 8; ++h) KV[h] = splat4(0.f);
#pragma unroll
  for (int h = 0; h < (TAPE ? 8 : 1); ++h) KVT[h] = splat4(0.f);
  constexpr bool kPrefetch = UFR_RT_PREFETCH && C2 == 1 && !TAPE;   // (the tape build has no registers to spare)
  f32x4 xn[6];     // kPrefetch: the next tile's rows, in flight
  if constexpr (kPrefetch) load_ray_tile(token0, tok_row, order_pe, (size_t)ray * SN, 0, g, j, xn);
  for (int it = 0; it < n_iter; ++it) {
    constexpr int C = 1;
    const bool wrap = it + 1 < n_iter;
    const bool slot_ok = head11_slot(j) >= 0;   // column j of a head tile carries a head dim
    f32x4 x[C][6], kt[C][8], vt[C][8];
    bool live[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int tile = it * C + c;
      live[c] = tile < n_tiles;
      const int tl = live[c] ? tile : n_tiles - 1;
      if constexpr (kPrefetch) {
#pragma unroll
        for (int t = 0; t < 6; ++t) x[c][t] = xn[t];
        const int nx = wrap ? tl + 1 : 0;        // after the last tile: sweep 2's first one
        load_ray_tile(token0, tok_row, order_pe, (size_t)ray * SN + nx * 16, nx * 16, g, j, xn);
      } else {
        load_ray_tile<(UFR_RT_NT & 1) != 0>(token0, tok_row, order_pe, (size_t)ray * SN + tl * 16, tl * 16, g, j, x[c]);
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) { kt[c][h] = splat4(0.f); vt[c][h] = splat4(0.f); }
    }
    {  // swapped operands: kt[h], vt[h] rows = tokens 4g+r, column j = head dim; x is split once per k-step
      BWords<C> cur;
      split_units<0, 0, 4 * C>(x, cur, xs_x);
      static_for<3>([&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        BStep b[C];
        bwords_to_bstep(cur, b);
        if constexpr (s < 2) {
          BWords<C> nxt;
          gemm_f16_panel<M_RT_K, s, C, kRtWaves, true>(ws, b, kt, wrap, [&](auto ti) __attribute__((always_inline)) {
            constexpr int to = decltype(ti)::value;
            split_units<s + 1, to * 4 * C / 8, (to + 1) * 4 * C / 8>(x, nxt, xs_x);
          });
          gemm_f16_panel<M_RT_V, s, C, kRtWaves, true>(ws, b, vt, wrap);
          cur = nxt;
        } else {
          gemm_f16_panel<M_RT_K, s, C, kRtWaves, true>(ws, b, kt, wrap);
          gemm_f16_panel<M_RT_V, s, C, kRtWaves, true>(ws, b, vt, wrap);
        }
      });
      probe_gemm(kt, ws);   // raw accumulators: the scale joins elu1 / the division by the sample count
      probe_gemm(vt, ws);
    }
    UFR_RT_PHASE(0)  // sweep 1: token load + K, V GEMMs
    // KV_h += K'_h^T V_h: the operands of all 32 products first (branch-free: the division by a non-power-of-two sample
    // count is chosen once per tile, padding slots by selects), then the MFMAs register-major over the heads, so that
    // consecutive matrix instructions add into different accumulators (a head's own four still run r = 0..3: same bits).
    // Written per (head, register) this phase compiled into 32 x {exec-masked elu, a uniform branch around an IEEE
    // division, one MFMA}: 128 branches per tile and nothing for the scheduler to overlap.
    // ones column -> sum of K'.  The forward-only build puts it into EVERY lane group's padding slot (3, 7, 11, 15): the
    // normaliser Q'.sum K' then comes out of the message MFMAs in register 3 of every lane group -- no cross-lane exchange
    // (8 ds_bpermute round trips per tile); the tape build keeps slot 3 alone (the backward reads the state's layout)
    const float pad_v = (!(UFR_RT_LOCAL_DEN && !TAPE) ? j == 3 : (j & 3) == 3) ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float kk[8][4], vv[8][4];
      if (pow2_len) {
#pragma unroll
        for (int h = 0; h < 8; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[h][r] = vt[c][h][r] * inv_len;
      } else {
#pragma unroll
        for (int h = 0; h < 8; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[h][r] = vt[c][h][r] / f_len;
      }
      const bool k_ok = slot_ok && live[c];                                               // padding slots / empty tile contribute nothing
#pragma unroll
      for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = elu1_acc(kt[c][h][r], k_dsc, k_l2e);
          kk[h][r] = k_ok ? e : 0.f;
          vv[h][r] = slot_ok ? vv[h][r] : pad_v;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {
#ifdef UFR_ABL_NOKV   // ablation (timing only): no fp32 MFMAs for the per-head KV state
          KV[h][r] += kk[h][r] * vv[h][r];
#else
          KV[h] = mfma16(kk[h][r], vv[h][r], KV[h]);
          if constexpr (TAPE) KVT[h] = mfma16(vv[h][r], kk[h][r], KVT[h]);   // [V slot][K slot]: the A operand of d Q' = KV d acc
#endif
        }
      }
    }
    wstream_f16_finish<B_RT1, kRtWaves>(ws, wrap);
    UFR_RT_PHASE(1)  // sweep 1: KV accumulation (fp32 MFMA)
  }

  if constexpr (TAPE) {
    if (valid) {
      float* st = ray_state + (size_t)ray * (kRayStateTiles * kTileFloats) + lane * 4;
#pragma unroll
      for (int h = 0; h < 8; ++h) { st4(st + h * kTileFloats, KV[h]); st4(st + (8 + h) * kTileFloats, KVT[h]); }
    }
  }
  // ---------------- sweep 2 (slot 0 is free: every wave passed the barrier that opened sweep 1's last chunk)
  wstream_f16_prime<B_RT2, kRtWaves>(ws);
  // C2 > 1: the per-head state leaves the registers (the wave's own 8 KiB behind the ring and the vector fragments; only
  // this wave reads it back: program order within the wave is all the ordering it needs)
  f32x4* const kv_lds = reinterpret_cast<f32x4*>(smem + kF16LdsBytes) + (threadIdx.x >> 6) * 512 + lane;
  if constexpr (kKvLds) {
#pragma unroll
    for (int h = 0; h < 8; ++h) kv_lds[h * 64] = KV[h];
  }
  constexpr int C = C2;
  // TAPE: whole blocks of two tiles (a padding tile is not live); C2 > 1: an odd tile count leaves the last iteration's
  // second tile empty (masked)
  const int n_iter2 = TAPE ? n_iter + (n_iter & 1) : (n_tiles + C - 1) / C;
  typedef RayTapeLayout<LOWP> TapeL;
  for (int it = 0; it < n_iter2; ++it) {
    const bool wrap = it + 1 < n_iter2;
    char* const tape_blk = reinterpret_cast<char*>(tape) + ((size_t)ray * (n_iter2 / 2) + it / 2) * (TapeL::block_units * 512);
    auto tape_st = [&](int tile, f32x4 v) __attribute__((always_inline)) {
      if (valid) tile_store<TapeL>(tape_blk, tile, it & 1, lane, v);
    };
    f32x4 x[C][6], q[C][6], msg[C][6];   // q, msg: quad-packed (ROW_QUAD11 / COL_QUAD11)
    bool live[C];
    int tbase[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int tile = it * C + c;
      live[c] = tile < n_tiles;
      tbase[c] = (live[c] ? tile : n_tiles - 1) * 16;
      if constexpr (kPrefetch) {
#pragma unroll
        for (int t = 0; t < 6; ++t) x[c][t] = xn[t];
        const int nx = (tile + 1 < n_tiles ? tile + 1 : n_tiles - 1) * 16;
        if (wrap) load_ray_tile(token0, tok_row, order_pe, (size_t)ray * SN + nx, nx, g, j, xn);
      } else {
        load_ray_tile<(UFR_RT_NT & 2) != 0>(token0, tok_row, order_pe, (size_t)ray * SN + tbase[c], tbase[c], g, j, x[c]);
      }
#pragma unroll
      for (int t = 0; t < 6; ++t) q[c][t] = splat4(0.f);
    }
    track_external(x, ws);
    gemm_f16<M_RT_Q, C, kRtWaves>(ws, x, q, wrap, xs_x);  // raw accumulators, quad-packed rows, column j = token
    float zs_all[8];
    if constexpr (TAPE) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        tape_st(RT_X + t, live[0] ? x[0][t] : splat4(0.f));
        f32x4 qq;
#pragma unroll
        for (int r = 0; r < 4; ++r) qq[r] = quad11(t, g, r) >= 0 ? elu1_acc(q[0][t][r], q_dsc, q_l2e) : 0.f;
        tape_st(RT_Q + t, qq);
      }
    }
    UFR_RT_PHASE(2)  // sweep 2: token load + Q GEMM
#pragma unroll
    for (int c = 0; c < C; ++c) {
      static_for<8>([&](auto hi) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value;
        f32x4 acc = splat4(0.f);
        f32x4 kvh;
        if constexpr (kKvLds) kvh = kv_lds[h * 64];
        else kvh = KV[h];
        static_for<3>([&](auto qi) __attribute__((always_inline)) {
          constexpr int qd = decltype(qi)::value, quad = 3 * h + qd;   // lane group g: head dim 3g + qd
          const float qq = (3 * g + qd < 11) ? elu1_acc(q[c][quad >> 2][quad & 3], q_dsc, q_l2e) : 0.f;
#ifdef UFR_ABL_NOKV
          acc[qd] += kvh[qd] * qq;
#else
          acc = mfma16(kvh[qd], qq, acc);            // rows = V slots: sum_d KV[d][v] Q'[d]; slot 3 = Q'.sum(K')
#endif
        });
        const float den = (UFR_RT_LOCAL_DEN && !TAPE) ? acc[3] : __shfl(acc[3], j);   // every group's own / slot 3 of lane group 0
        const float Z = fast_rcp(den + 1e-6f);         // linear_attention.py:43
        const float zs = Z * (float)SN;              // :44
        if constexpr (TAPE) zs_all[h] = zs;
        static_for<3>([&](auto ri) __attribute__((always_inline)) {
          constexpr int rr = decltype(ri)::value, quad = 3 * h + rr;
          msg[c][quad >> 2][quad & 3] = acc[rr] * zs;   // the live registers, in merge's input order
        });
      });
    }
    UFR_RT_PHASE(3)  // message (fp32 MFMA)
    if constexpr (TAPE) {
#pragma unroll
      for (int t = 0; t < 6; ++t) tape_st(RT_MSG + t, msg[0][t]);
      tape_st(RT_ZS, f32x4{zs_all[0], zs_all[1], zs_all[2], zs_all[3]});
      tape_st(RT_ZS + 1, f32x4{zs_all[4], zs_all[5], zs_all[6], zs_all[7]});
    }
    f32x4 m[C][6];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 6; ++t) m[c][t] = splat4(0.f);
    gemm_f16<M_RT_MERGE, C, kRtWaves>(ws, msg, m, wrap, sc[RS_M_XS]);
    UFR_RT_PHASE(4)  // merge GEMM
    float rstd1[C] = {}, rstd2[C] = {};
    if constexpr (TAPE) {
      f32x4 xh[C][6];
      layer_norm88<V_RT_N1W, V_RT_N1B, C>(m, ws, g, sc[RS_EPS1], sc[RS_M_ASC], xh, rstd1);
#pragma unroll
      for (int t = 0; t < 6; ++t) { tape_st(RT_XH1 + t, xh[0][t]); tape_st(RT_M + t, m[0][t]); }
    } else {
      layer_norm88<V_RT_N1W, V_RT_N1B, C>(m, ws, g, sc[RS_EPS1], 1.f);
    }
    UFR_RT_PHASE(5)  // LayerNorm 1

    f32x4 cat[C][12], hid[C][11], o[C][6];
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int t = 0; t < 6; ++t) { cat[c][t] = x[c][t]; cat[c][6 + t] = m[c][t]; }
#pragma unroll
      for (int t = 0; t < 11; ++t) hid[c][t] = splat4(0.f);
    }
    gemm_f16<M_RT_MLP0, C, kRtWaves>(ws, cat, hid, wrap, xs_x);   // hid: raw accumulators through the ReLU
    UFR_RT_PHASE(6)  // MLP0
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int t = 0; t < 11; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[c][t][r] = relu_acc(hid[c][t][r]);
#pragma unroll
      for (int t = 0; t < 6; ++t) o[c][t] = splat4(0.f);
    }
    unsigned relu_bits[2] = {0u, 0u};      // TAPE: bit 4 t + r <-> unit (t, r) of this lane is active (hid, then d1, d2)
    if constexpr (TAPE) {
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        tape_st(RT_HID + t, hid[0][t] * sc[RS_MLP0_DSC]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (hid[0][t][r] > 0.f) relu_bits[(4 * t + r) >> 5] |= 1u << ((4 * t + r) & 31);
      }
    }
    gemm_f16<M_RT_MLP2, C, kRtWaves>(ws, hid, o, wrap, sc[RS_M_MLP2]);
    UFR_RT_PHASE(7)  // ReLU + MLP2
    if constexpr (TAPE) {
      f32x4 xh[C][6];
      layer_norm88<V_RT_N2W, V_RT_N2B, C>(o, ws, g, sc[RS_EPS2], sc[RS_MLP2_ASC], xh, rstd2);
#pragma unroll
      for (int t = 0; t < 6; ++t) tape_st(RT_XH2 + t, xh[0][t]);
    } else {
      layer_norm88<V_RT_N2W, V_RT_N2B, C>(o, ws, g, sc[RS_EPS2], 1.f);
    }
    UFR_RT_PHASE(8)  // LayerNorm 2
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[c][t][r] += x[c][t][r];   // scalar adds: no v_pk_add_f32 (layer_norm88)
      if constexpr (TAPE) {
#pragma unroll
        for (int t = 0; t < 6; ++t) tape_st(RT_O + t, o[c][t]);
      }
      if (ray_out && valid && live[c]) {
        float* row = ray_out + ((size_t)ray * SN + tbase[c] + j) * UFR_RAY_DIM;
#pragma unroll
        for (int t = 0; t < 5; ++t) st4(row + 16 * t + 4 * g, o[c][t]);
        row[80 + 2 * g] = o[c][5][0];
        row[81 + 2 * g] = o[c][5][1];
      }
    }

    // ---------------- DensityMLP 88 -> 32 -> 16 -> 1 (ray_transformer.py:147-150, 307)
    f32x4 d1[C][2], d2[C][1], d3[C][1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      d1[c][0] = vec_frag<V_DM_B0>(ws, 0, g) * sc[RS_DM0_ASC];   // biases enter the scaled accumulators (weight_stream_f16.h)
      d1[c][1] = vec_frag<V_DM_B0>(ws, 1, g) * sc[RS_DM0_ASC];
      d2[c][0] = vec_frag<V_DM_B2>(ws, 0, g) * sc[RS_DM2_ASC];
      d3[c][0] = vec_frag<V_DM_B4>(ws, 0, g) * sc[RS_DM4_ASC];
    }
    gemm_f16<M_DM0, C, kRtWaves>(ws, o, d1, wrap, sc[RS_DM0_XS]);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[c][t][r] = relu_acc(d1[c][t][r]);
    gemm_f16<M_DM2, C, kRtWaves>(ws, d1, d2, wrap, sc[RS_M_DM2]);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) d2[c][0][r] = relu_acc(d2[c][0][r]);
    if constexpr (TAPE) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        tape_st(RT_D1 + t, d1[0][t] * sc[RS_DM0_DSC]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (d1[0][t][r] > 0.f) relu_bits[(44 + 4 * t + r) >> 5] |= 1u << ((44 + 4 * t + r) & 31);
      }
      tape_st(RT_D2, d2[0][0] * sc[RS_DM2_DSC]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (d2[0][0][r] > 0.f) relu_bits[1] |= 1u << (52 + r - 32);
      tape_st(RT_MISC, f32x4{rstd1[0], rstd2[0], __builtin_bit_cast(float, relu_bits[0]), __builtin_bit_cast(float, relu_bits[1])});
    }
    gemm_f16<M_DM4, C, kRtWaves>(ws, d2, d3, wrap, sc[RS_M_DM4]);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (g == 0 && valid && live[c]) srdf[(size_t)ray * SN + tbase[c] + j] = d3[c][0][0] * sc[RS_DM4_DSC];
    wstream_f16_finish<B_RT2, kRtWaves>(ws, wrap);
    UFR_RT_PHASE(9)  // residual, stores, DensityMLP
  }
  wstream_report_range(ws, status);
#ifdef UFR_PHASE_TIMING
  if (ray_raw == 5 && lane == 0)
    for (int i = 0; i < 16; ++i) g_rt_phase[i] += rt_acc[i];
#endif
}

template <bool LOWP, bool TAPE = false>
static hipError_t launch_rt(const float* packed, const float* token0, const int* tok_row, const float* order_pe, int RN,
                            int SN, float* srdf, float* ray_out, int* status, hipStream_t s, float* tape = nullptr,
                            float* ray_state = nullptr) {
  constexpr int C2 = TAPE ? 1 : kRtC2;
  constexpr int lds = kF16LdsBytes + (C2 > 1 ? kRtKvLdsBytes : 0);
  static LdsAttrOnce lds_attr;   // per instantiation; thread-safe, once per device
  if (const hipError_t attr = lds_attr.set(reinterpret_cast<const void*>(&ray_transformer_kernel<LOWP, TAPE, C2>), lds); attr != hipSuccess) return attr;
  hipLaunchKernelGGL((ray_transformer_kernel<LOWP, TAPE, C2>), dim3((RN + kRtWaves - 1) / kRtWaves), dim3(kRtBlock), lds, s,
                     packed, token0, tok_row, order_pe, RN, SN, srdf, ray_out, status, tape, ray_state);
  return hipGetLastError();
}

// The forward again for the backward (bwd_tape.h): tape = RN x ceil(SN / 32) blocks of RT_COUNT tiles, ray_state = RN x
// kRayStateTiles tiles; srdf is written as usual (the caller passes scratch).
hipError_t launch_ray_tape(const float* packed, const float* token0, const int* tok_row, const float* order_pe, int RN, int SN,
                           float* srdf, float* tape, float* ray_state, bool lowp, int* status, hipStream_t s) {
  if (SN % 16 != 0 || SN < 16) return hipErrorInvalidValue;
  return lowp ? launch_rt<true, true>(packed, token0, tok_row, order_pe, RN, SN, srdf, nullptr, status, s, tape, ray_state)
              : launch_rt<false, true>(packed, token0, tok_row, order_pe, RN, SN, srdf, nullptr, status, s, tape, ray_state);
}

hipError_t launch_ray_transformer(const float* packed, const float* token0, const int* tok_row, const float* order_pe,
                                  int RN, int SN, float* srdf, float* ray_out, bool lowp, int* status, hipStream_t s) {
  if (SN % 16 != 0 || SN < 16) return hipErrorInvalidValue;
  return lowp ? launch_rt<true>(packed, token0, tok_row, order_pe, RN, SN, srdf, ray_out, status, s)
              : launch_rt<false>(packed, token0, tok_row, order_pe, RN, SN, srdf, ray_out, status, s);
}

}  // namespace ufr

#ifdef UFR_PHASE_TIMING
extern "C" int ufr_debug_rt_phases(unsigned long long* out, int n, int reset) {
  unsigned long long h[32] = {};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ufr::g_rt_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 32; ++i) out[i] = h[i];
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ufr::g_rt_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
