// Data-gradient chain of the cross-view aggregation (round 4; replaces the barrier-phased view_bwd kernel).
//   autograd of RayTransformer.forward     code1/ray_transformer.py:283-294, 309-320
//               LoFTREncoderLayer.forward   code1/attention/transformer.py:35-58
//               LinearAttention.forward     code1/attention/linear_attention.py:20-47
// Built like the forward kernel (view_transformer.hip): tokens are the 16 MFMA columns, a wave owns two column tiles
// (the same point <-> column map as the forward, so block b of the tape is block b here), cotangents never leave registers
// between layers -- the accumulator tile of  d in = W^T d out  is the B operand of the next transposed layer -- and the
// TRANSPOSED weights stream through the same LDS ring as bf16 planes (ufr_layout_f16.h: B_VTB; three plane products per
// fp32 product, one in the 16-bit mode).  Everything the chain needs from the forward comes from the tape
// (bwd_tape.h, written by view_transformer_kernel<.., TAPE>); every layer-output cotangent goes to the dY tile buffer, from
// which wgrad_stream.hip contracts the weight gradients over the tokens.  The only token reductions here are the
// elementwise ones -- LayerNorm gamma / beta and the view token: 100 values per lane group and iteration, all-reduced over
// the 16 token lanes of a DPP row (four v_add_f32_dpp each) and added to the gradient tensors with one atomic instruction
// per ten values; as tiles through HBM they were a sixth of the three kernels' traffic.
// Outputs of this kernel itself: the dY tiles and d_pv (P,40) = the gradient w.r.t. the 24 frustum features and the 16
// pre_sim_mlp outputs of each point, summed over its NV view tokens (gather_bwd.hip consumes it).
#include "bwd_common.h"   // GradPtrs
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"

namespace ufr {

constexpr int kVdBlock = 256, kVdWaves = 4;

// v[i] <- sum of v[i] over the 16 lanes of the DPP row (= the 16 token columns of a lane group), for ten values.  One block
// of assembly: a DPP read needs two wait states after a VALU write of the same register and the hazard recogniser does not
// look inside inline assembly -- within the block a register's next read is ten instructions after its write.
__device__ __forceinline__ void row_allreduce10(float (&v)[10]) {
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
// acc[b] += sum over the 16 token lanes of value 10 b + j of the 80-feature natural-layout vector t (20 values per lane
// group: two blocks of ten): after the all-reduce every lane of a row holds all ten sums and lane j < 10 keeps the j-th.
// Accumulated over the wave's whole persistent loop -- in a private LDS slot per (vector, block, lane): ten more live
// registers cost this kernel 50..70 spills -- and flushed once (flush80): per iteration the atomics of all waves would
// queue up on the same 400 addresses (measured: +0.1 ms, more than the tiles had cost).
__device__ __forceinline__ void reduce_acc80(const f32x4 (&t)[5], float* acc /* LDS: [2][64], this lane's column */, int j) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = t[(10 * b + i) >> 2][(10 * b + i) & 3];
    row_allreduce10(v);
    float mine = v[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) mine = j == i ? v[i] : mine;
    acc[64 * b] += mine;
  }
}
__device__ __forceinline__ void flush80(const float* acc, float* __restrict__ dst, int g, int j) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int idx = 10 * b + j;
    if (j < 10) unsafeAtomicAdd(dst + 16 * (idx >> 2) + 4 * g + (idx & 3), acc[64 * b]);
  }
}
constexpr int kVdAccFloats = 5 * 2 * 64;   // per wave: 5 vectors x 2 blocks x 64 lanes

// sum_d Q[d] * (K[d] of the lane holding token (tv + S) % L of the same point); acc[d] += w * (V[d] of that lane)
template <int L, int S>
__device__ __forceinline__ float dot10r(const float (&Q)[10], const float (&K)[10], const int (&src)[8]) {
  if constexpr (L == 4 && S >= 1 && S <= 3) {
    return dot10_rot4<(S >= 1 && S <= 3) ? S : 1>(Q, K);
  } else {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 10; ++d) a = fmaf(Q[d], rot<L, S>(K[d], src), a);
    return a;
  }
}
template <int L, int S>
__device__ __forceinline__ void axpy10r(float (&acc)[10], float w, const float (&V)[10], const int (&src)[8]) {
  if constexpr (L == 4 && S >= 1 && S <= 3) {
    axpy10_rot4<(S >= 1 && S <= 3) ? S : 1>(acc, w, V);
  } else {
#pragma unroll
    for (int d = 0; d < 10; ++d) acc[d] = fmaf(w, rot<L, S>(V[d], src), acc[d]);
  }
}

template <int L, bool LOWP>
__global__ void __launch_bounds__(kVdBlock, 2) view_dgrad_kernel(const float* __restrict__ packed, const float* __restrict__ tape,
                                                                const float* __restrict__ rgbm,
                                                                const float* __restrict__ d_tok_a,
                                                                const float* __restrict__ d_tok_b,
                                                                const float* __restrict__ d_radiance, int P,
                                                                float* __restrict__ dbuf, float* __restrict__ d_pv,
                                                                float* __restrict__ g_n1w, float* __restrict__ g_n1b,
                                                                float* __restrict__ g_n2w, float* __restrict__ g_n2b,
                                                                float* __restrict__ g_vtok) {
  typedef ViewTapeLayout<LOWP> TapeL;
  typedef ViewGradLayout<LOWP> GradL;
  constexpr int NV = L - 1, C = kBlockCols;
  constexpr int PPT = 16 / L, PPW = PPT * C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto ws = wstream_f16_begin<kVdWaves, LOWP, true>(packed, smem);
  wstream_f16_prime<B_VTB, kVdWaves>(ws);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  int ptw[C], tv[C];
  bool okc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    ptw[c] = c * PPT + j / L;
    tv[c] = j % L;
    okc[c] = j < PPT * L;
  }
  int src[8];    // L != 4: lane of token (tv + S) % L of the same point
#pragma unroll
  for (int s = 0; s < 8; ++s) src[s] = okc[0] ? (lane - tv[0] + (tv[0] + s) % L) : lane;

  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  const int n_groups = (P + PPW - 1) / PPW;
  const int n_iter = (n_groups + n_waves - 1) / n_waves;   // uniform over the grid: every wave meets every chunk barrier

  // the small-gradient accumulators behind the weight ring and the vector fragments
  float* const a_base = reinterpret_cast<float*>(smem + kF16LdsBytes) + (threadIdx.x >> 6) * kVdAccFloats + lane;
  float* const a_n1w = a_base;
  float* const a_n1b = a_base + 128;
  float* const a_n2w = a_base + 256;
  float* const a_n2b = a_base + 384;
  float* const a_vtok = a_base + 512;
#pragma unroll
  for (int i = 0; i < 10; ++i) a_base[64 * i] = 0.f;
  for (int it = 0; it < n_iter; ++it) {
    const int grp_raw = it * n_waves + wave_global;
    const bool wrap = it + 1 < n_iter;
    const bool live = grp_raw < n_groups;
    const size_t grp = live ? grp_raw : 0;                 // an idle wave re-reads block 0 and stores nothing
    const char* const tape_blk = reinterpret_cast<const char*>(tape) + grp * (TapeL::block_units * 512);
    char* const dy_blk = reinterpret_cast<char*>(dbuf) + grp * (GradL::block_units * 512);
    auto tape_ld = [&](int tile, int c) __attribute__((always_inline)) -> f32x4 { return tile_load<TapeL>(tape_blk, tile, c, lane); };
    auto dy_st = [&](int tile, int c, f32x4 v) __attribute__((always_inline)) {
      if (live) tile_store<GradL>(dy_blk, tile, c, lane, v);
    };
    auto dy_ld = [&](int tile, int c) __attribute__((always_inline)) -> f32x4 { return tile_load<GradL>(dy_blk, tile, c, lane); };
    int pidx[C];
    bool valid[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pidx[c] = (int)grp * PPW + ptw[c];
      valid[c] = okc[c] && live && pidx[c] < P;
    }

    // ---------------- masked softmax over the views and its adjoint (ray_transformer.py:315-319): d logit of this column
    // (memory-operation order, here and below: vmcnt counts loads AND stores in issue order on gfx950, so a load issued
    // after a tile store cannot be waited for without waiting for that store's round trip to HBM.  Every phase therefore
    // issues its tape loads first and its dY stores last, and the ReLU derivatives come from the bit masks of TV_MISC
    // instead of 22 more tile loads: as "load, wait, mask, store" per tile the kernel spent 3/4 of its time in s_waitcnt.)
    float rstd1[C], rstd2[C], dl[C];
    unsigned bits0[C], bits1[C];
    f32x4 miscv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) miscv[c] = tape_ld(TV_MISC, c);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float m0 = miscv[c][0], m1 = miscv[c][1], m2 = miscv[c][2], m3 = miscv[c][3];   // (scalars first: hipcc mis-reads
      bits0[c] = __builtin_bit_cast(unsigned, m0);                                          //  a vector ELEMENT in bit_cast)
      bits1[c] = __builtin_bit_cast(unsigned, m1);
      rstd1[c] = __shfl(m2, j);            // lane group 0 holds (rstd1, rstd2) of column j, lane group 1 (logit, 0)
      rstd2[c] = __shfl(m3, j);
      const float logit_raw = __shfl(m2, 16 + j);
      f32x4 col = splat4(0.f);
      float cd = 0.f;                                  // colour . d radiance of this (point, view)
      if (valid[c] && tv[c] > 0) {
        col = ld4(rgbm + ((size_t)pidx[c] * NV + (tv[c] - 1)) * 4);
        const float* dr = d_radiance + (size_t)pidx[c] * 3;
        cd = col[0] * dr[0] + col[1] * dr[1] + col[2] * dr[2];
      }
      float logit = logit_raw;
      if (col[3] == 0.f) logit = -1e9f;
      if (tv[c] == 0) logit = -INFINITY;               // the view token is not a colour source
      float mx = logit;
#define UFR_MAX_STEP(S) if (S < L) mx = fmaxf(mx, rot<L, S>(logit, src));
      UFR_MAX_STEP(1) UFR_MAX_STEP(2) UFR_MAX_STEP(3) UFR_MAX_STEP(4) UFR_MAX_STEP(5) UFR_MAX_STEP(6) UFR_MAX_STEP(7)
#undef UFR_MAX_STEP
      const float e = tv[c] == 0 ? 0.f : expf(logit - mx);
      float den = e, ecd = e * cd;
#define UFR_SUM_STEP(S)                 \
      if (S < L) {                      \
        den += rot<L, S>(e, src);       \
        ecd += rot<L, S>(e * cd, src);  \
      }
      UFR_SUM_STEP(1) UFR_SUM_STEP(2) UFR_SUM_STEP(3) UFR_SUM_STEP(4) UFR_SUM_STEP(5) UFR_SUM_STEP(6) UFR_SUM_STEP(7)
#undef UFR_SUM_STEP
      // d logit_v = p_v (c_v . dR - sum_u p_u c_u . dR); masked logits are constants (torch.where)
      const float p = e / den;
      dl[c] = (valid[c] && tv[c] > 0 && col[3] != 0.f) ? p * (cd - ecd / den) : 0.f;
    }

    // ---------------- radiance-weight MLP backwards (ray_transformer.py:159-163, 313-314)
    f32x4 dh2[C][1], dh1[C][1], dy[C][5];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 w4 = vec_frag<V_RW_W4>(ws, 0, g);   // rows 4g + r; rows >= 8 are zero
#pragma unroll
      for (int r = 0; r < 4; ++r) dh2[c][0][r] = ((bits1[c] >> (12 + r)) & 1u) ? w4[r] * dl[c] : 0.f;
      dy_st(DV_LG, c, f32x4{g == 0 ? dl[c] : 0.f, 0.f, 0.f, 0.f});
      dy_st(DV_H2, c, dh2[c][0]);
      dh1[c][0] = splat4(0.f);
    }
    gemm_f16<M_RW2T, C, kVdWaves>(ws, dh2, dh1, wrap);
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh1[c][0][r] = ((bits1[c] >> (8 + r)) & 1u) ? dh1[c][0][r] : 0.f;
      dy_st(DV_H1, c, dh1[c][0]);
#pragma unroll
      for (int t = 0; t < 5; ++t) dy[c][t] = splat4(0.f);
    }
    gemm_f16<M_RW0T, C, kVdWaves>(ws, dh1, dy, wrap);
    // + d token0 from the ray transformer's backward (view-token columns only)
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (valid[c] && tv[c] == 0) {
        const float* ra = d_tok_a + (size_t)pidx[c] * UFR_TOKEN_DIM + 4 * g;
#pragma unroll
        for (int t = 0; t < 5; ++t) dy[c][t] += ld4(ra + 16 * t);
        if (d_tok_b) {
          const float* rb = d_tok_b + (size_t)pidx[c] * UFR_TOKEN_DIM + 4 * g;
#pragma unroll
          for (int t = 0; t < 5; ++t) dy[c][t] += ld4(rb + 16 * t);
        }
      }

    // ---------------- LayerNorm2 backwards (transformer.py:56-58): y = x + LN2(opre), so d x starts as d y (parked in DV_SCR)
    f32x4 dopre[C][5];
    f32x4 dgam[5], dbet[5];     // d gamma2 = sum_t d y xhat2, d beta2 = sum_t d y: summed over the column tiles, then the lanes
    {
      f32x4 xh[C][5];
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) xh[c][t] = tape_ld(TV_XH2 + t, c);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        f32x4 gy[5];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          dgam[t] = c == 0 ? dy[c][t] * xh[c][t] : dgam[t] + dy[c][t] * xh[c][t];
          dbet[t] = c == 0 ? dy[c][t] : dbet[t] + dy[c][t];
          gy[t] = dy[c][t] * vec_frag<V_VT_N2W>(ws, t, g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1 += gy[t][r];
            s2 = fmaf(gy[t][r], xh[c][t][r], s2);
          }
        }
        const float m1 = sum_groups(s1) * (1.f / 80.f), m2 = sum_groups(s2) * (1.f / 80.f);
#pragma unroll
        for (int t = 0; t < 5; ++t) dopre[c][t] = (gy[t] - m1 - xh[c][t] * m2) * rstd2[c];
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) { dy_st(DV_SCR + t, c, dy[c][t]); dy_st(DV_OPRE + t, c, dopre[c][t]); }
    }
    reduce_acc80(dgam, a_n2w, j);       // (idle / padding columns carry zeros)
    reduce_acc80(dbet, a_n2b, j);

    // ---------------- MLP backwards (transformer.py:55-56)
    f32x4 dhid[C][10];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t) dhid[c][t] = splat4(0.f);
    gemm_f16<M_VT_MLP2T, C, kVdWaves>(ws, dopre, dhid, wrap);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned on = t < 8 ? (bits0[c] >> (4 * t + r)) & 1u : (bits1[c] >> (4 * (t - 8) + r)) & 1u;
          dhid[c][t][r] = on ? dhid[c][t][r] : 0.f;
        }
        dy_st(DV_HID + t, c, dhid[c][t]);
      }
    f32x4 dcat[C][10];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t) dcat[c][t] = splat4(0.f);
    gemm_f16<M_VT_MLP0T, C, kVdWaves>(ws, dhid, dcat, wrap);
    // the x half of d cat is parked in a second set of scratch tiles (the attention phase below needs the registers; the
    // output phase adds both sets to the projections' share), the message half goes on
    f32x4 dmpre[C][5];
    {
      f32x4 xh[C][5];
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) xh[c][t] = tape_ld(TV_XH1 + t, c);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        f32x4 gm[5];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          dgam[t] = c == 0 ? dcat[c][5 + t] * xh[c][t] : dgam[t] + dcat[c][5 + t] * xh[c][t];
          dbet[t] = c == 0 ? dcat[c][5 + t] : dbet[t] + dcat[c][5 + t];
          gm[t] = dcat[c][5 + t] * vec_frag<V_VT_N1W>(ws, t, g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1 += gm[t][r];
            s2 = fmaf(gm[t][r], xh[c][t][r], s2);
          }
        }
        // ---------------- LayerNorm1 backwards (transformer.py:52)
        const float m1 = sum_groups(s1) * (1.f / 80.f), m2 = sum_groups(s2) * (1.f / 80.f);
#pragma unroll
        for (int t = 0; t < 5; ++t) dmpre[c][t] = (gm[t] - m1 - xh[c][t] * m2) * rstd1[c];
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) { dy_st(DV_SCR2 + t, c, dcat[c][t]); dy_st(DV_MPRE + t, c, dmpre[c][t]); }
    }
    reduce_acc80(dgam, a_n1w, j);
    reduce_acc80(dbet, a_n1b, j);
    // ---------------- merge backwards: d msg in the slot layout (lane group g <- heads 2g, 2g+1 of its token)
    f32x4 dmsg[C][5];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 5; ++t) dmsg[c][t] = splat4(0.f);
    gemm_f16<M_VT_MERGET, C, kVdWaves>(ws, dmpre, dmsg, wrap);

    // ---------------- linear attention backwards (linear_attention.py:31-45), score form, lane-local per (token, head):
    //   A[s][s'] = Q'_s . K'_s';  den_s = sum_s' A;  Zs = L / (den + eps);  msg_s = Zs sum_s' A[s][s'] V_s'
    //   d r_s = Zs d msg_s;  d den_s = -(d msg_s . r_s) Zs^2 / L;  d A[s][s'] = d r_s . V_s' + d den_s
    //   d Q'_s = sum_s' d A[s][s'] K'_s';  d K'_s' = sum_s d A[s][s'] Q'_s;  d V_s' = sum_s A[s][s'] d r_s
    f32x4 dq[C][5], dk[C][5], dv[C][5];
    f32x4 qt[5], kt[5], vt[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { qt[t] = tape_ld(TV_Q + t, 0); kt[t] = tape_ld(TV_K + t, 0); vt[t] = tape_ld(TV_V + t, 0); }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float Q[10], K[10], V[10], dm[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const int s = 10 * hh + d;
          Q[d] = qt[s >> 2][s & 3]; K[d] = kt[s >> 2][s & 3]; V[d] = vt[s >> 2][s & 3]; dm[d] = dmsg[c][s >> 2][s & 3];
        }
        float r[10], den = 0.f;
#pragma unroll
        for (int d = 0; d < 10; ++d) r[d] = 0.f;
#define UFR_FWD_STEP(S)                                   \
        if (S < L) {                                      \
          const float a = dot10r<L, S>(Q, K, src);        \
          den += a;                                       \
          axpy10r<L, S>(r, a, V, src);                    \
        }
        UFR_FWD_STEP(0) UFR_FWD_STEP(1) UFR_FWD_STEP(2) UFR_FWD_STEP(3) UFR_FWD_STEP(4) UFR_FWD_STEP(5) UFR_FWD_STEP(6) UFR_FWD_STEP(7)
#undef UFR_FWD_STEP
        const float Zs = (float)L / (den + 1e-6f);
        float du = 0.f, dr[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          du = fmaf(dm[d], r[d], du);
          dr[d] = Zs * dm[d];
        }
        const float dden = -du * Zs * Zs * (1.f / (float)L);
        float dQ[10], dK[10], dV[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) dQ[d] = dK[d] = dV[d] = 0.f;
        // query side: this lane is token s, its partner of step S is s' = (s + S) % L
#define UFR_Q_STEP(S)                                                    \
        if (S < L) {                                                     \
          const float dA = dden + dot10r<L, S>(dr, V, src);             \
          axpy10r<L, S>(dQ, dA, K, src);                                 \
        }
        UFR_Q_STEP(0) UFR_Q_STEP(1) UFR_Q_STEP(2) UFR_Q_STEP(3) UFR_Q_STEP(4) UFR_Q_STEP(5) UFR_Q_STEP(6) UFR_Q_STEP(7)
#undef UFR_Q_STEP
        // key / value side: this lane is token s', the query of step S is s = (s' + S) % L:
        //   d A[s][s'] = d r_s . V_s' + d den_s,  A[s][s'] = Q'_s . K'_s'
#define UFR_K_STEP(S)                                                    \
        if (S < L) {                                                     \
          const float dA = rot<L, S>(dden, src) + dot10r<L, S>(V, dr, src); \
          const float a = dot10r<L, S>(K, Q, src);                       \
          axpy10r<L, S>(dK, dA, Q, src);                                 \
          axpy10r<L, S>(dV, a, dr, src);                                 \
        }
        UFR_K_STEP(0) UFR_K_STEP(1) UFR_K_STEP(2) UFR_K_STEP(3) UFR_K_STEP(4) UFR_K_STEP(5) UFR_K_STEP(6) UFR_K_STEP(7)
#undef UFR_K_STEP
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const int s = 10 * hh + d;
          dq[c][s >> 2][s & 3] = dQ[d] * (Q[d] > 1.f ? 1.f : Q[d]);     // elu'(q) = q > 0 ? 1 : exp(q) = (Q' > 1 ? 1 : Q')
          dk[c][s >> 2][s & 3] = dK[d] * (K[d] > 1.f ? 1.f : K[d]);
          dv[c][s >> 2][s & 3] = dV[d] * (1.f / (float)L);              // values = v / L (linear_attention.py:41)
        }
      }
      if (c + 1 < C) {            // the next column tile's Q', K', V: issued before this one's stores
#pragma unroll
        for (int t = 0; t < 5; ++t) { qt[t] = tape_ld(TV_Q + t, c + 1); kt[t] = tape_ld(TV_K + t, c + 1); vt[t] = tape_ld(TV_V + t, c + 1); }
      }
#pragma unroll
      for (int t = 0; t < 5; ++t) { dy_st(DV_Q + t, c, dq[c][t]); dy_st(DV_K + t, c, dk[c][t]); dy_st(DV_V + t, c, dv[c][t]); }
    }

    // ---------------- projections backwards, all three into one accumulator
    f32x4 dx[C][5];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 5; ++t) dx[c][t] = splat4(0.f);
    gemm_f16<M_VT_QT, C, kVdWaves>(ws, dq, dx, wrap);
    gemm_f16<M_VT_KT, C, kVdWaves>(ws, dk, dx, wrap);
    gemm_f16<M_VT_VT, C, kVdWaves>(ws, dv, dx, wrap);

    // ---------------- outputs: d x = d y (residual) + d cat[0..79] + the projections.  Token columns 32..55 (frustum
    // features) and 56..71 (pre_sim_mlp) are the same for all NV view tokens of a point (ray_transformer.py:258-281):
    // their gradients add up into d_pv; column 0 of a point is the view token: its gradient is d x summed over those columns
    f32x4 dtok[5];
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        dx[c][t] += dy_ld(DV_SCR + t, c) + dy_ld(DV_SCR2 + t, c);
        const f32x4 v0 = (valid[c] && tv[c] == 0) ? dx[c][t] : splat4(0.f);
        dtok[t] = c == 0 ? v0 : dtok[t] + v0;
      }
      // features 32..71 = tiles 2, 3 and lane groups 0, 1 of tile 4
#pragma unroll
      for (int t = 2; t < 5; ++t) {
        f32x4 sum = splat4(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = tv[c] == 0 ? 0.f : dx[c][t][r];
          float acc = v0;
#define UFR_PV_STEP(S) if (S < L) acc += rot<L, S>(v0, src);
          UFR_PV_STEP(1) UFR_PV_STEP(2) UFR_PV_STEP(3) UFR_PV_STEP(4) UFR_PV_STEP(5) UFR_PV_STEP(6) UFR_PV_STEP(7)
#undef UFR_PV_STEP
          sum[r] = acc;
        }
        if (valid[c] && tv[c] == 0 && (t < 4 || g < 2)) st4(d_pv + (size_t)pidx[c] * 40 + 16 * (t - 2) + 4 * g, sum);
      }
    }
    reduce_acc80(dtok, a_vtok, j);
    wstream_f16_finish<B_VTB, kVdWaves>(ws, wrap);
  }
  flush80(a_n1w, g_n1w, g, j);
  flush80(a_n1b, g_n1b, g, j);
  flush80(a_n2w, g_n2w, g, j);
  flush80(a_n2b, g_n2b, g, j);
  flush80(a_vtok, g_vtok, g, j);
}

template <int L, bool LOWP>
static hipError_t launch_vd(const float* packed, const float* tape, const float* rgbm, const float* d_tok_a, const float* d_tok_b,
                            const float* d_radiance, int P, float* dbuf, float* d_pv, const GradPtrs& gp, hipStream_t s) {
  constexpr int PPW = (16 / L) * kBlockCols;
  const int n_groups = (P + PPW - 1) / PPW;
  int blocks = (n_groups + kVdWaves - 1) / kVdWaves;
  constexpr int resident = 256 * 2;              // two workgroups per CU
  constexpr int max_blocks = resident * 4;       // a few times more, shorter workgroups (view_transformer.hip: launch_vt)
  if (blocks > max_blocks) {
    const int base = (n_groups + max_blocks * kVdWaves - 1) / (max_blocks * kVdWaves);
    long best_cost = -1;
    for (int n_iter = base; n_iter <= 4 * base; ++n_iter) {
      const int b = (n_groups + kVdWaves * n_iter - 1) / (kVdWaves * n_iter);
      const long cost = (long)((b + resident - 1) / resident) * n_iter;
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; blocks = b; }
    }
  }
  static LdsAttrOnce lds_attr;   // per instantiation; thread-safe, once per device
  if (const hipError_t attr = lds_attr.set(reinterpret_cast<const void*>(&view_dgrad_kernel<L, LOWP>), kF16LdsBytes + kVdWaves * kVdAccFloats * 4); attr != hipSuccess) return attr;
  hipLaunchKernelGGL((view_dgrad_kernel<L, LOWP>), dim3(blocks), dim3(kVdBlock), kF16LdsBytes + kVdWaves * kVdAccFloats * 4, s, packed, tape, rgbm, d_tok_a,
                     d_tok_b, d_radiance, P, dbuf, d_pv, gp.p[P_VT_N1W], gp.p[P_VT_N1B], gp.p[P_VT_N2W], gp.p[P_VT_N2B],
                     gp.p[P_VIEW_TOKEN]);
  return hipGetLastError();
}

hipError_t launch_view_dgrad(const float* packed, const float* tape, const float* rgbm, const float* d_tok_a,
                             const float* d_tok_b, const float* d_radiance, int P, int NV, float* dbuf, float* d_pv,
                             const GradPtrs& gp, bool lowp, hipStream_t s) {
  switch (NV) {
#define UFR_VD_CASE(N)                                                                                              \
    case N:                                                                                                         \
      return lowp ? launch_vd<N + 1, true>(packed, tape, rgbm, d_tok_a, d_tok_b, d_radiance, P, dbuf, d_pv, gp, s)  \
                  : launch_vd<N + 1, false>(packed, tape, rgbm, d_tok_a, d_tok_b, d_radiance, P, dbuf, d_pv, gp, s);
    UFR_VD_CASE(2) UFR_VD_CASE(3) UFR_VD_CASE(4) UFR_VD_CASE(5) UFR_VD_CASE(6) UFR_VD_CASE(7)
#undef UFR_VD_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ufr
