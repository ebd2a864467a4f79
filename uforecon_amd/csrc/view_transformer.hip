// Cross-view aggregation of one pass: view token + LoFTR linear-attention layer over the NV+1
// tokens of every point, then the radiance-weight MLP, masked softmax over views and colour blend.
//   RayTransformer.forward      code1/ray_transformer.py:283-294, 309-320
//   LoFTREncoderLayer.forward   code1/attention/transformer.py:35-58   (bias-free Linear, post-norm message)
//   LinearAttention.forward     code1/attention/linear_attention.py:20-47
//
// MI355X mapping: tokens are the 16 MFMA columns; a wave owns C column tiles (PPT = 16/L points each,
// L = NV+1 tokens per point).  Activations never leave registers: the fp32 accumulator tiles of a
// layer, split into two fp16 planes, are the B operands of the next layer's
// v_mfma_f32_16x16x32_f16 ("fp16x3": three plane pairs per product, fp32-grade accuracy on the 16-bit
// matrix cores -- ufr_layout_f16.h); the weight planes stream through LDS (weight_stream_f16.h).  Q/K/V
// rows are permuted so lane group g holds heads 2g,2g+1 of its token, and the L-token attention is
// lane-local with quad DPP exchanges (L = 4) or ds_bpermute (other L).  LayerNorm, attention, elu and
// the softmax run on the VALU and overlap with the other resident wave's MFMAs.
#include <cstdlib>
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"

namespace ufr {

// base + 32-bit element offset, the byte offset computed in 32 bits: the access becomes "scalar base + 32-bit lane
// offset" (global_load ... v_off, s[base]) with no 64-bit per-lane address to keep alive (the launcher bounds the sizes)
template <class T>
__device__ __forceinline__ T* at32(T* base, unsigned elem) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + (elem * (unsigned)sizeof(T)));
}

template <int C, int N>
__device__ __forceinline__ void zero_tiles(f32x4 (&t)[C][N]) {
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) t[c][i] = splat4(0.f);
}

// LayerNorm over the 80 features of each token: 5 tiles x 4 regs in each of the 4 lane groups.  t holds raw accumulators
// (asc = 2^(s_M + a_M) times the values; asc = 1: plain values): the normalised value is scale-free once the epsilon carries
// the square of the scale (eps = 1e-5 asc^2, formed once per launch), and with a power-of-two scale every intermediate is the exact multiple -- bit-identical to
// descaling first.
// XH / RS (TAPE builds): the normalised input and 1 / sigma of the TRUE values (asc times the raw one), which the
// backward needs.
template <int C, int VW, int VB, class WS>
__device__ __forceinline__ void layer_norm80(f32x4 (&t)[C][5], const WS& ws, int g, float eps, float asc, f32x4 (*XH)[5] = nullptr, float* RS = nullptr) {
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) s += (t[c][i][0] + t[c][i][1]) + (t[c][i][2] + t[c][i][3]);
    const float mean = sum_groups(s) * (1.f / 80.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float d = t[c][i][r] - mean;
        q = fmaf(d, d, q);
      }
    const float rstd = fast_rsqrt(sum_groups(q) * (1.f / 80.f) + eps);
    if (RS) RS[c] = rstd * asc;
    // element by element on purpose: f32x4 expressions become v_pk_mul / v_pk_fma_f32, which cost more beside the
    // partner wave's MFMAs than the two scalar instructions they replace (MI355X_MICROARCH.md, price of a filler)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const f32x4 gw = vec_frag<VW>(ws, i, g), gb = vec_frag<VB>(ws, i, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (t[c][i][r] - mean) * rstd;
        if (XH) XH[c][i][r] = xh;
        t[c][i][r] = xh * gw[r] + gb[r];
      }
    }
  }
}

// 256-thread workgroups (one wave per SIMD), two per CU; each streams the layer chain's weight planes
// through its own ring of three 12 KiB LDS slots (weight_stream_f16.h).
#ifndef UFR_VT_BLOCK
#define UFR_VT_BLOCK 256   // threads per workgroup
#define UFR_VT_C 2         // token column tiles per wave
#define UFR_VT_MINW 2      // waves per SIMD the register budget is sized for
#endif
#ifndef UFR_VT_OVERSUB
#define UFR_VT_OVERSUB 4   // workgroups launched per resident slot
#endif
#ifdef UFR_PHASE_TIMING  // development build: cycle counts per phase of wave 0 (tools/bench_kernels.py prints them)
__device__ unsigned long long g_vt_phase[32];
__device__ unsigned long long g_vt_wave[4096 * 2];  // start / end tick of every wave of the last launch
#define UFR_PHASE(i)                                                                   \
  {                                                                                    \
    const unsigned long long t_now = __builtin_readcyclecounter();                     \
    ph_acc[i] += t_now - t_prev;                                                       \
    t_prev = t_now;                                                                    \
  }
#else
// production builds: optionally a scheduling fence at the phase boundaries (kPhaseFence).  Tried for the straddling L = 6
// kernel while it ran 15 % slower than it should (the cause was scalar-register pressure: weight_stream_f16.h, ScalarFile):
// no effect before or after that fix, so both knobs default to off.
#define UFR_PHASE(i) \
  { if constexpr (kPhaseFence) __builtin_amdgcn_sched_barrier(0); }
#endif
#ifndef UFR_VT_FENCE_ALL
#define UFR_VT_FENCE_ALL 0
#endif
#ifndef UFR_VT_FENCE_STRADDLE
#define UFR_VT_FENCE_STRADDLE 0
#endif
#ifndef UFR_VT_SCORES_HOOK
#define UFR_VT_SCORES_HOOK 1   // the attention scores ride on the v GEMM's MFMAs (L = 4, two column tiles).  Round 6, same box, two runs each: alone 0.3635-0.3641 vs 0.3648-0.3662 ms per launch (nothing); with UFR_HOOK_VALU 6 frame 122.1-122.3 vs 123.4-123.6 ms
#endif
#ifndef UFR_VT_RELOAD_X
#define UFR_VT_RELOAD_X 0   // 1: the L = 6 kernel re-reads the token rows for the residual (see there); measured 395 -> 444 ms per
#endif                      // 600x800 / 5-view frame once the scalar pressure was fixed (it had helped before: 434 -> 428)
constexpr int kVtBlock = UFR_VT_BLOCK;
constexpr int kVtWaves = kVtBlock / 64;

// TAPE: the instantiation the backward launches (bwd_tape.h): the same arithmetic, plus one store per activation tile.
template <int L, int C, bool LOWP, bool TAPE = false>
__global__ void __launch_bounds__(kVtBlock, UFR_VT_MINW) view_transformer_kernel(const float* __restrict__ packed,
                                                                             const float* __restrict__ x_tokens,
                                                                             const float* __restrict__ x_point,
                                                                             const float* __restrict__ rgbm,
                                                                             const float* __restrict__ dirs, int P,
                                                                             float* __restrict__ token0,
                                                                             float* __restrict__ radiance,
                                                                             float* __restrict__ view_out,
                                                                             int* __restrict__ status,
                                                                             float* __restrict__ tape = nullptr) {
  constexpr int NV = L - 1;
  constexpr int PPT = 16 / L;          // points per column tile
  // L = 6 (five source views) fills only 12 of a tile's 16 columns with whole points: with two column tiles per wave a
  // fifth point STRADDLES them -- tile 0 = points 0, 1 and tokens 0..3 of point 2; tile 1 = points 3, 4, then tokens 4, 5 of
  // point 2 in columns 12, 13 (30 of 32 columns).  The dense layers do not care where a token sits; the attention and the
  // softmax exchange tokens of a point through the schedule below.
  constexpr bool STRADDLE = (L == 6 && C == 2 && !TAPE);   // the tape's consumers use the plain slot map
  static_assert(!TAPE || kBlockCols % C == 0, "tape blocks");
  constexpr int PPW = STRADDLE ? 5 : PPT * C;         // points per wave iteration
  constexpr bool kReloadX = STRADDLE && UFR_VT_RELOAD_X;                 // the residual re-reads the token rows (see there)
  constexpr bool kPhaseFence = (STRADDLE && UFR_VT_FENCE_STRADDLE) || UFR_VT_FENCE_ALL;
  constexpr bool kScoresHook = UFR_VT_SCORES_HOOK && L == 4 && C == 2 && !TAPE;
  (void)kPhaseFence;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto ws = wstream_f16_begin<kVtWaves, LOWP>(packed, smem);
  wstream_f16_prime<B_VT, kVtWaves>(ws);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  // the layers' plane / accumulator scales (ufr_layout.h: ViewScalar; weight_stream_f16.h: ScalarFile).  q, k, v and mlp0
  // all split x: their a_M agree by construction (prep.hip), and the splits use ONE multiplier (VS_XS_X) so that the
  // compiler can merge them (41 of 171 value pairs per iteration).
  constexpr bool kLocalScalars = (L == 6 && C == 2 && !TAPE);   // = STRADDLE, defined below
  const auto sc = scalar_file<kLocalScalars, view_scalars_offset()>(ws);
  // column (c, j) holds token tvv[c] of the wave's point ptw[c] (tvv == 0: view token)
  int ptw[C], tvv[C];
  bool okc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if constexpr (STRADDLE) {
      ptw[c] = j < 12 ? 3 * c + j / 6 : 2;
      tvv[c] = j < 12 ? j % 6 : (c == 0 ? j - 12 : j - 8);
      okc[c] = c == 0 || j < 14;
    } else {
      ptw[c] = c * PPT + j / L;
      tvv[c] = j % L;
      okc[c] = j < PPT * L;
    }
  }
  int src[8];    // L != 4, no straddle: lane of token (tv + S) % L of the same point (the same for every tile)
#pragma unroll
  for (int s = 0; s < 8; ++s) src[s] = okc[0] ? (lane - tvv[0] + (tvv[0] + s) % L) : lane;
  // STRADDLE: a token meets the other five of its point in five STEPS, token t reading token kStep[t][step] -- the same
  // schedule for every point, so a point's arithmetic does not depend on the slot it lands in.  The schedule is chosen so
  // that ONE exchange per value, tile and step suffices: the two halves of the straddling point sit in the same lanes
  // (12..15) of the two tiles, and in no step does a tile's consumer set want both tiles' value of one lane -- the source
  // register of tile c at a step is the tile's own value with lanes 12..15 taken from the other tile where `ovr` says so
  // (tile 0: lane 12 at step 0 = token 4, lane 13 at step 1 = token 5; tile 1: lanes 12..15 = tokens 0..3 from step 1 on).
  int srcx[C][5];
  if constexpr (STRADDLE) {
    constexpr int kStep[6][5] = {{4, 5, 1, 2, 3}, {4, 5, 0, 2, 3}, {4, 5, 0, 1, 3}, {4, 5, 0, 1, 2}, {5, 0, 1, 2, 3}, {4, 0, 1, 2, 3}};
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        int u = 0;
#pragma unroll
        for (int t = 0; t < 6; ++t) u = tvv[c] == t ? kStep[t][st] : u;
        const int base = j < 12 ? (j / 6) * 6 : 12;            // first lane of the point's tokens in the source register
        const int off = j < 12 ? u : (u & 3);                  // straddling point: tokens 0..3 and 4, 5 both at lanes 12..
        srcx[c][st] = 4 * (okc[c] ? 16 * g + base + off : lane);   // byte address of ds_bpermute_b32
      }
  }
  // value that token kStep[tv][st] of this lane's point holds of (xc: this tile's register, xo: the other tile's)
  auto exch = [&](auto ci, auto sti, float xc, float xo) __attribute__((always_inline)) -> float {
    constexpr int c = decltype(ci)::value, st = decltype(sti)::value;
    float m = xc;
    if constexpr (c == 0 && st == 0) m = j == 12 ? xo : xc;
    if constexpr (c == 0 && st == 1) m = j == 13 ? xo : xc;
    if constexpr (c == 1 && st >= 1) m = j >= 12 ? xo : xc;
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcx[c][st], __builtin_bit_cast(int, m)));
  };

  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  // TAPE: whole blocks of kBlockCols column tiles (a trailing group without points writes finite values, never garbage)
  const int n_groups = TAPE ? ((P + PPT * kBlockCols - 1) / (PPT * kBlockCols)) * (kBlockCols / C) : (P + PPW - 1) / PPW;
  const int n_iter = (n_groups + n_waves - 1) / n_waves;  // uniform over the grid: every wave meets every barrier

#ifdef UFR_PHASE_TIMING
  unsigned long long ph_acc[12] = {};
  unsigned long long t_prev = __builtin_readcyclecounter();
  const unsigned long long t_start = t_prev;
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  for (int it = 0; it < n_iter; ++it) {
    const int grp = it * n_waves + wave_global;
    const bool wrap = it + 1 < n_iter;
    // tape tile (block, column tile) of this wave iteration (bwd_tape.h: per-tile layout, bf16 storage of some tiles in the
    // 16-bit mode)
    typedef ViewTapeLayout<LOWP> TapeL;
    const int col0 = __builtin_amdgcn_readfirstlane(grp < n_groups ? grp : 0) * C;      // first column tile, counted over all blocks
    char* const tape_blk = reinterpret_cast<char*>(tape) + (size_t)(col0 / kBlockCols) * (TapeL::block_units * 512);
    auto tape_st = [&](int tile, int c, f32x4 v) __attribute__((always_inline)) {
      if (grp < n_groups) tile_store<TapeL>(tape_blk, tile, col0 % kBlockCols + c, lane, v);
    };
#ifdef UFR_FUSION_PROBE
    // Feasibility probe for fusing the gather into this kernel: a synthetic producer phase with
    // the gather's per-iteration footprint -- 3 dependent rounds (projection -> footprint -> taps) of UFR_FUSION_PROBE
    // independent 16-byte loads inside a 256 KiB window (cache-resident like the feature maps), ~12 VALU per load --
    // whose result is kept alive but unused.  The real gather needs ~8 points x 3 views x 176 lane-loads / 64 = 66 loads
    // per wave iteration, i.e. UFR_FUSION_PROBE = 22.
    {
      f32x4 acc = splat4(0.f);
      unsigned cursor = (unsigned)(grp * 64 + lane) * 2654435761u;
      const float* window = x_tokens + (size_t)((unsigned)grp % 64u) * 65536u;
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int k = 0; k < UFR_FUSION_PROBE; ++k) {
          const unsigned off = ((cursor >> 8) + k * 977u) & 16383u;
          const f32x4 v = ld4(window + (size_t)off * 4);
          acc = acc * 1.0001f + v * v - acc * v * 0.5f + v * 0.25f;
        }
        cursor = cursor * 1664525u + 1013904223u + (unsigned)(int)(acc[0] * 1e-30f);
      }
      asm volatile("" ::"v"(acc));
    }
#endif
    // ---------------- load tokens: x[c][t] = features 16t+4g..+3 of token j
    f32x4 x[C][5];
    int pidx[C];
    bool valid[C];
    // token columns 0..31 and 72..79 are per (point, view), 32..71 per point: one 80-column row in the public layout,
    // a 40-column view row plus a 40-column point row in the compact one (ufr_internal.h)
    // (32-bit element offsets from the scalar bases -- the launcher bounds P: 64-bit per-lane addresses that the
    // optimiser hoists out of the loop cost this kernel spills it cannot afford)
    auto load_x = [&](int c, f32x4 (&dst)[5]) __attribute__((always_inline)) {
      const int pp = valid[c] ? pidx[c] : 0;
      const int tv = tvv[c];
      const unsigned vrow = ((unsigned)pp * NV + (tv > 0 ? tv - 1 : 0)) * (x_point ? kViewCols : UFR_TOKEN_DIM);
      const unsigned prow_e = x_point ? (unsigned)pp * kPointCols : vrow + 32;
      const float* pbase = x_point ? x_point : x_tokens;
      const float* row = at32(x_tokens, vrow + 4 * g);
      const float* prow = at32(pbase, prow_e + 4 * g);
      const float* last = g < 2 ? prow + 32 : row + (x_point ? 32 : 72) - 8;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        f32x4 tok = vec_frag<V_VIEW_TOKEN>(ws, t, g);
#ifdef UFR_ABL_NOTOKLOAD   // ablation (timing only): what the token loads' latency costs
        f32x4 val = splat4(0.001f * (float)(lane + t));
        asm volatile("" : "+v"(val));
#else
        f32x4 val = ld4(t < 2 ? row + 16 * t : t < 4 ? prow + 16 * (t - 2) : last);
#endif
        dst[t] = tv == 0 ? tok : val;                     // ray_transformer.py:284-286
        if (!valid[c]) dst[t] = splat4(0.f);
      }
    };
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pidx[c] = grp * PPW + ptw[c];
      valid[c] = okc[c] && grp < n_groups && pidx[c] < P;
      load_x(c, x[c]);
    }
    track_external(x, ws);
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) tape_st(TV_X + t, c, x[c][t]);
    }
    UFR_PHASE(0)  // token loads issued
    // ---------------- q,k projections (slot layout: lane group g <- heads 2g, 2g+1); x is split once per k-step
    // and feeds both matrices (stream order q0 k0 q1 k1 q2 k2)
    f32x4 q[C][5], k[C][5];
    zero_tiles(q); zero_tiles(k);
    {
      BWords<C> cur;
      const float xs_x = sc[VS_XS_X];
      split_units<0, 0, 4 * C>(x, cur, xs_x);
      static_for<3>([&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        BStep b[C];
        bwords_to_bstep(cur, b);
        if constexpr (s < 2) {       // the next step's split rides on the q panel's MFMAs
          BWords<C> nxt;
          gemm_f16_panel<M_VT_Q, s, C, kVtWaves, false>(ws, b, q, wrap, [&](auto ti) __attribute__((always_inline)) {
            constexpr int to = decltype(ti)::value;
            split_units<s + 1, to * 4 * C / 5, (to + 1) * 4 * C / 5>(x, nxt, xs_x);
          });
          gemm_f16_panel<M_VT_K, s, C, kVtWaves>(ws, b, k, wrap);
          cur = nxt;
        } else {
          gemm_f16_panel<M_VT_Q, s, C, kVtWaves>(ws, b, q, wrap);
          gemm_f16_panel<M_VT_K, s, C, kVtWaves>(ws, b, k, wrap);
        }
      });
      probe_gemm(q, ws);   // q, k stay raw accumulators: elu1_acc descales
      probe_gemm(k, ws);
    }
    const float q_dsc = sc[VS_Q_DSC], q_l2e = sc[VS_Q_L2E], k_dsc = sc[VS_K_DSC], k_l2e = sc[VS_K_L2E];
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          tape_st(TV_Q + t, c, f32x4{elu1_acc(q[c][t][0], q_dsc, q_l2e), elu1_acc(q[c][t][1], q_dsc, q_l2e), elu1_acc(q[c][t][2], q_dsc, q_l2e), elu1_acc(q[c][t][3], q_dsc, q_l2e)});
          tape_st(TV_K + t, c, f32x4{elu1_acc(k[c][t][0], k_dsc, k_l2e), elu1_acc(k[c][t][1], k_dsc, k_l2e), elu1_acc(k[c][t][2], k_dsc, k_l2e), elu1_acc(k[c][t][3], k_dsc, k_l2e)});
        }
    }
    UFR_PHASE(1)  // q,k GEMMs
    // ---------------- linear attention over the L tokens of each point (linear_attention.py:31-45), written as
    // msg = sum_S A_S V_S / sum_S A_S with A_S = Q'.K'_S for the token (tv+S)%L of the same point: the
    // scores are reduced to L numbers per head before v is even computed, so q and k die early
    float A[C][2][L], Zs[C][2];
    if constexpr (STRADDLE) {
      // index 0 = the token itself, 1 + step = its partner of that step (the schedule above)
      static_for<2>([&](auto hi) __attribute__((always_inline)) {
        constexpr int hh = decltype(hi)::value;
        float Q[C][10], K[C][10], den[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            const int s = 10 * hh + d;
            Q[c][d] = elu1_acc(q[c][s >> 2][s & 3], q_dsc, q_l2e);
            K[c][d] = elu1_acc(k[c][s >> 2][s & 3], k_dsc, k_l2e);
          }
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < 10; ++d) a = fmaf(Q[c][d], K[c][d], a);
          A[c][hh][0] = a;
          den[c] = a;
        }
        static_for<5>([&](auto sti) __attribute__((always_inline)) {
          constexpr int st = decltype(sti)::value;
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            a0 = fmaf(Q[0][d], exch(std::integral_constant<int, 0>{}, sti, K[0][d], K[1][d]), a0);
            a1 = fmaf(Q[1][d], exch(std::integral_constant<int, 1>{}, sti, K[1][d], K[0][d]), a1);
          }
          A[0][hh][1 + st] = a0; den[0] += a0;
          A[1][hh][1 + st] = a1; den[1] += a1;
        });
#pragma unroll
        for (int c = 0; c < C; ++c) Zs[c][hh] = (float)L * fast_rcp(den[c] + 1e-6f);   // Z * v_length (linear_attention.py:43-44)
      });
    } else if constexpr (kScoresHook) {
      // round-6 experiment (UFR_VT_SCORES_HOOK): the scores of (column tile, head) pair u = 2 c + hh ride on the v GEMM's
      // stages 3 u .. 3 u + 2 (elu of Q | elu of K | the L dot products, the normaliser) instead of running in front of it
    } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float Q[10], K[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const int s = 10 * hh + d;
          Q[d] = elu1_acc(q[c][s >> 2][s & 3], q_dsc, q_l2e);
          K[d] = elu1_acc(k[c][s >> 2][s & 3], k_dsc, k_l2e);
        }
        float den = 0.f;
#define UFR_ATT_STEP(S)                                                  \
        if (S < L) {                                                     \
          float a = 0.f;                                                 \
          if constexpr (L == 4 && S >= 1 && S <= 3) {                    \
            a = dot10_rot4<(S >= 1 && S <= 3) ? S : 1>(Q, K);           \
          } else {                                                       \
            _Pragma("unroll") for (int d = 0; d < 10; ++d) a = fmaf(Q[d], rot<L, S>(K[d], src), a); \
          }                                                              \
          den += a;                                                      \
          A[c][hh][S < L ? S : 0] = a;                                   \
        }
        UFR_ATT_STEP(0) UFR_ATT_STEP(1) UFR_ATT_STEP(2) UFR_ATT_STEP(3)
        UFR_ATT_STEP(4) UFR_ATT_STEP(5) UFR_ATT_STEP(6) UFR_ATT_STEP(7)
#undef UFR_ATT_STEP
        Zs[c][hh] = (float)L * fast_rcp(den + 1e-6f);             // Z * v_length (linear_attention.py:43-44)
      }
    }
    }
    UFR_PHASE(2)  // scores
    f32x4 v[C][5];
    zero_tiles(v);
    if constexpr (kScoresHook) {
      float Qs[10], Ks[10];
      gemm_f16<M_VT_V, C, kVtWaves>(ws, x, v, wrap, sc[VS_XS_X], [&](auto ui) __attribute__((always_inline)) {
        constexpr int slot = decltype(ui)::value, u = slot / 3, part = slot % 3;
        if constexpr (u < 2 * C) {
          constexpr int c = u / 2, hh = u % 2;
          if constexpr (part == 0) {
#pragma unroll
            for (int d = 0; d < 10; ++d) Qs[d] = elu1_acc(q[c][(10 * hh + d) >> 2][(10 * hh + d) & 3], q_dsc, q_l2e);
          } else if constexpr (part == 1) {
#pragma unroll
            for (int d = 0; d < 10; ++d) Ks[d] = elu1_acc(k[c][(10 * hh + d) >> 2][(10 * hh + d) & 3], k_dsc, k_l2e);
          } else {
            float a0 = 0.f;
#pragma unroll
            for (int d = 0; d < 10; ++d) a0 = fmaf(Qs[d], Ks[d], a0);
            const float a1 = dot10_rot4<1>(Qs, Ks), a2 = dot10_rot4<2>(Qs, Ks), a3 = dot10_rot4<3>(Qs, Ks);
            A[c][hh][0] = a0; A[c][hh][1] = a1; A[c][hh][2] = a2; A[c][hh][3] = a3;
            Zs[c][hh] = (float)L * fast_rcp((((0.f + a0) + a1) + a2) + a3 + 1e-6f);
          }
        }
      });
    } else {
      gemm_f16<M_VT_V, C, kVtWaves>(ws, x, v, wrap, sc[VS_XS_X]);   // raw accumulators: the descale joins the 1 / v_length
    }
    UFR_PHASE(3)  // v GEMM
    // values / v_length on raw accumulators: one exact multiply when L is a power of two; otherwise (L = 3, 5, 6, 7) a true
    // division of the descaled value (the power-of-two descale is exact, so this is v / (L 2^(s+a)) bit for bit)
    const float v_dsc = sc[VS_V_DSC];
    const float v_mul = v_dsc / (float)L;
    if constexpr (TAPE) {   // values / v_length, exactly as the message phase below forms them
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          f32x4 vv;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            vv[r] = (L & (L - 1)) == 0 ? v[c][t][r] * v_mul : (v[c][t][r] * v_dsc) / (float)L;
          tape_st(TV_V + t, c, vv);
        }
    }
    f32x4 msg[C][5];
    if constexpr (STRADDLE) {
      static_for<2>([&](auto hi) __attribute__((always_inline)) {
        constexpr int hh = decltype(hi)::value;
        float V[C][10], acc[C][10];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            const int s = 10 * hh + d;
            V[c][d] = (v[c][s >> 2][s & 3] * v_dsc) / (float)L;   // values / v_length (L = 6: a true division)
            acc[c][d] = A[c][hh][0] * V[c][d];
          }
        static_for<5>([&](auto sti) __attribute__((always_inline)) {
          constexpr int st = decltype(sti)::value;
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            acc[0][d] = fmaf(A[0][hh][1 + st], exch(std::integral_constant<int, 0>{}, sti, V[0][d], V[1][d]), acc[0][d]);
            acc[1][d] = fmaf(A[1][hh][1 + st], exch(std::integral_constant<int, 1>{}, sti, V[1][d], V[0][d]), acc[1][d]);
          }
        });
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int d = 0; d < 10; ++d) {
            const int s = 10 * hh + d;
            msg[c][s >> 2][s & 3] = acc[c][d] * Zs[c][hh];
          }
      });
    } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float V[10], acc[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const int s = 10 * hh + d;
          // values / v_length: exact as a multiply when L is a power of two (NV = 3, 7)
          V[d] = (L & (L - 1)) == 0 ? v[c][s >> 2][s & 3] * v_mul : (v[c][s >> 2][s & 3] * v_dsc) / (float)L;
          acc[d] = 0.f;
        }
#define UFR_ATT_STEP(S)                                                  \
        if (S < L) {                                                     \
          if constexpr (L == 4 && S >= 1 && S <= 3) {                    \
            axpy10_rot4<(S >= 1 && S <= 3) ? S : 1>(acc, A[c][hh][S < L ? S : 0], V); \
          } else {                                                       \
            _Pragma("unroll") for (int d = 0; d < 10; ++d) acc[d] = fmaf(A[c][hh][S < L ? S : 0], rot<L, S>(V[d], src), acc[d]); \
          }                                                              \
        }
        UFR_ATT_STEP(0) UFR_ATT_STEP(1) UFR_ATT_STEP(2) UFR_ATT_STEP(3)
        UFR_ATT_STEP(4) UFR_ATT_STEP(5) UFR_ATT_STEP(6) UFR_ATT_STEP(7)
#undef UFR_ATT_STEP
#pragma unroll
        for (int d = 0; d < 10; ++d) {
          const int s = 10 * hh + d;
          msg[c][s >> 2][s & 3] = acc[d] * Zs[c][hh];
        }
      }
    }
    }

    UFR_PHASE(4)  // message
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) tape_st(TV_MSG + t, c, msg[c][t]);
    }
    // ---------------- merge + LayerNorm1 (transformer.py:51-52)
    f32x4 m[C][5];
    zero_tiles(m);
    gemm_f16<M_VT_MERGE, C, kVtWaves>(ws, msg, m, wrap, sc[VS_M_XS]);
    UFR_PHASE(5)  // merge GEMM
    float rstd1[C] = {}, rstd2[C] = {};
    if constexpr (TAPE) {
      f32x4 xh[C][5];
      layer_norm80<C, V_VT_N1W, V_VT_N1B>(m, ws, g, sc[VS_EPS1], sc[VS_M_ASC], xh, rstd1);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) { tape_st(TV_XH1 + t, c, xh[c][t]); tape_st(TV_M + t, c, m[c][t]); }
    } else {
      layer_norm80<C, V_VT_N1W, V_VT_N1B>(m, ws, g, sc[VS_EPS1], 1.f);
    }

    UFR_PHASE(6)  // LN1
    // ---------------- MLP on [x | message] + LayerNorm2 + residual (transformer.py:55-58)
    f32x4 cat[C][10], hid[C][10], o[C][5];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 5; ++t) { cat[c][t] = x[c][t]; cat[c][5 + t] = m[c][t]; }
    zero_tiles(hid);
    gemm_f16<M_VT_MLP0, C, kVtWaves>(ws, cat, hid, wrap, sc[VS_XS_X]);   // hid: raw accumulators through the ReLU
    UFR_PHASE(7)  // MLP0
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[c][t][r] = relu_acc(hid[c][t][r]);
    unsigned relu_bits[C][2] = {};   // TAPE: bit 4 t + r <-> hidden unit (t, r) of this lane is active; then h1 (40..43), h2 (44..47)
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
          tape_st(TV_HID + t, c, hid[c][t] * sc[VS_MLP0_DSC]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (hid[c][t][r] > 0.f) relu_bits[c][(4 * t + r) >> 5] |= 1u << ((4 * t + r) & 31);
        }
    }
    zero_tiles(o);
    gemm_f16<M_VT_MLP2, C, kVtWaves>(ws, hid, o, wrap, sc[VS_M_MLP2]);
    // colour / mask / direction of this lane's (point, view): (issued here: hid is dead, so the 10 registers are free, and LayerNorm2 + the token stores cover the latency)
    f32x4 col[C];
    float dcomp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      col[c] = splat4(0.f);
      dcomp[c] = 0.f;
      const int tv = tvv[c];
      if (valid[c] && tv > 0) {
        col[c] = ld4(at32(rgbm, ((unsigned)pidx[c] * NV + (tv - 1)) * 4u));           // r,g,b,mask
        dcomp[c] = *at32(dirs, ((unsigned)pidx[c] * NV + (tv - 1)) * 4u + g);         // lane group g <- dir[g], 0 for g=3
      }
    }

    UFR_PHASE(8)  // relu + MLP2
    if constexpr (TAPE) {
      f32x4 xh[C][5];
      layer_norm80<C, V_VT_N2W, V_VT_N2B>(o, ws, g, sc[VS_EPS2], sc[VS_MLP2_ASC], xh, rstd2);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) tape_st(TV_XH2 + t, c, xh[c][t]);
    } else {
      layer_norm80<C, V_VT_N2W, V_VT_N2B>(o, ws, g, sc[VS_EPS2], 1.f);
    }
    // the residual.  UFR_VT_RELOAD_X (off): the straddling L = 6 kernel reads the token rows AGAIN here (they are in L2)
    // instead of keeping 40 registers alive from the top of the iteration -- no spills then, but slower (see the macro)
    if constexpr (kReloadX) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        f32x4 xr[5];
        load_x(c, xr);
#pragma unroll
        for (int t = 0; t < 5; ++t) o[c][t] += xr[t];
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[c][t][r] += x[c][t][r];   // scalar adds: no v_pk_add_f32 (layer_norm80)
    }
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int t = 0; t < 5; ++t) tape_st(TV_Y + t, c, o[c][t]);
        tape_st(TV_Y + 5, c, f32x4{dcomp[c], 0.f, 0.f, 0.f});
      }
    }

    // ---------------- outputs: token 0 -> ray transformer input; optional full dump
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int tv = tvv[c];
      if (valid[c] && tv == 0) {
#pragma unroll
        for (int t = 0; t < 5; ++t) st4(at32(token0, (unsigned)pidx[c] * UFR_TOKEN_DIM + 4 * g) + 16 * t, o[c][t]);
      }
      if (valid[c] && view_out) {
#pragma unroll
        for (int t = 0; t < 5; ++t)
          st4(at32(view_out, ((unsigned)pidx[c] * L + tv) * UFR_TOKEN_DIM + 4 * g) + 16 * t, o[c][t]);
      }
    }

    UFR_PHASE(9)  // LN2 + residual + stores
    // ---------------- radiance weight MLP on [view feature | dir] (ray_transformer.py:309-314)
    f32x4 rin[C][6], h1[C][1], h2[C][1], lg[C][1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int t = 0; t < 5; ++t) rin[c][t] = o[c][t];
      rin[c][5] = f32x4{dcomp[c], 0.f, 0.f, 0.f};
      ws.bad_out |= __builtin_amdgcn_ballot_w64(dcomp[c] != dcomp[c]);
      h1[c][0] = vec_frag<V_RW_B0>(ws, 0, g) * sc[VS_RW0_ASC];   // biases enter the scaled accumulators (weight_stream_f16.h)
      h2[c][0] = vec_frag<V_RW_B2>(ws, 0, g) * sc[VS_RW2_ASC];
      lg[c][0] = vec_frag<V_RW_B4>(ws, 0, g) * sc[VS_RW4_ASC];
    }
    gemm_f16<M_RW0, C, kVtWaves>(ws, rin, h1, wrap, sc[VS_RW0_XS]);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[c][0][r] = relu_acc(h1[c][0][r]);
    gemm_f16<M_RW2, C, kVtWaves>(ws, h1, h2, wrap, sc[VS_M_RW2]);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) h2[c][0][r] = relu_acc(h2[c][0][r]);
    gemm_f16<M_RW4, C, kVtWaves>(ws, h2, lg, wrap, sc[VS_M_RW4]);
    if constexpr (TAPE) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        tape_st(TV_H1, c, h1[c][0] * sc[VS_RW0_DSC]);
        tape_st(TV_H2, c, h2[c][0] * sc[VS_RW2_DSC]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (h1[c][0][r] > 0.f) relu_bits[c][1] |= 1u << (8 + r);
          if (h2[c][0][r] > 0.f) relu_bits[c][1] |= 1u << (12 + r);
        }
        // bwd_tape.h: TV_MISC = {this lane's ReLU bits (2 words), per-column scalars}: lane group 0 carries (rstd1, rstd2),
        // lane group 1 (logit, 0) -- the logit of column j sits in lane group 0, register 0 of lg
        const float lgt = __shfl(lg[c][0][0], j) * sc[VS_RW4_DSC];
        tape_st(TV_MISC, c, f32x4{__builtin_bit_cast(float, relu_bits[c][0]), __builtin_bit_cast(float, relu_bits[c][1]),
                                  g == 0 ? rstd1[c] : g == 1 ? lgt : 0.f, g == 0 ? rstd2[c] : 0.f});
      }
    }

    UFR_PHASE(10)  // radiance MLP
    const float rw4_dsc = sc[VS_RW4_DSC];
    // ---------------- masked softmax over the NV view tokens + colour blend (ray_transformer.py:315-319)
    // logit of token j sits in lane group 0, register 0; lanes of group 0 do the point-local reduction
    if constexpr (STRADDLE) {
      float logit[C], mx[C], e[C], den[C], cr[C], cg[C], cb[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        logit[c] = lg[c][0][0] * rw4_dsc;
        if (col[c][3] == 0.f) logit[c] = -1e9f;
        if (tvv[c] == 0) logit[c] = -INFINITY;  // the view token is not a colour source
        mx[c] = logit[c];
      }
      static_for<5>([&](auto sti) __attribute__((always_inline)) {
        mx[0] = fmaxf(mx[0], exch(std::integral_constant<int, 0>{}, sti, logit[0], logit[1]));
        mx[1] = fmaxf(mx[1], exch(std::integral_constant<int, 1>{}, sti, logit[1], logit[0]));
      });
#pragma unroll
      for (int c = 0; c < C; ++c) {
        e[c] = tvv[c] == 0 ? 0.f : expf(logit[c] - mx[c]);
        den[c] = e[c]; cr[c] = e[c] * col[c][0]; cg[c] = e[c] * col[c][1]; cb[c] = e[c] * col[c][2];
      }
      static_for<5>([&](auto sti) __attribute__((always_inline)) {
        constexpr std::integral_constant<int, 0> c0{};
        constexpr std::integral_constant<int, 1> c1{};
        den[0] += exch(c0, sti, e[0], e[1]);
        den[1] += exch(c1, sti, e[1], e[0]);
        cr[0] += exch(c0, sti, e[0] * col[0][0], e[1] * col[1][0]);
        cr[1] += exch(c1, sti, e[1] * col[1][0], e[0] * col[0][0]);
        cg[0] += exch(c0, sti, e[0] * col[0][1], e[1] * col[1][1]);
        cg[1] += exch(c1, sti, e[1] * col[1][1], e[0] * col[0][1]);
        cb[0] += exch(c0, sti, e[0] * col[0][2], e[1] * col[1][2]);
        cb[1] += exch(c1, sti, e[1] * col[1][2], e[0] * col[0][2]);
      });
#pragma unroll
      for (int c = 0; c < C; ++c)
        if (valid[c] && tvv[c] == 0 && g == 0) {
          float* dst = at32(radiance, (unsigned)pidx[c] * 3u);
          const float rden = fast_rcp(den[c]);
          dst[0] = cr[c] * rden;
          dst[1] = cg[c] * rden;
          dst[2] = cb[c] * rden;
        }
    } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int tv = tvv[c];
      float logit = lg[c][0][0] * rw4_dsc;
      if (col[c][3] == 0.f) logit = -1e9f;
      if (tv == 0) logit = -INFINITY;  // the view token is not a colour source
      float mx = logit;
#define UFR_MAX_STEP(S) if (S < L) mx = fmaxf(mx, rot<L, S>(logit, src));
      UFR_MAX_STEP(1) UFR_MAX_STEP(2) UFR_MAX_STEP(3) UFR_MAX_STEP(4) UFR_MAX_STEP(5) UFR_MAX_STEP(6) UFR_MAX_STEP(7)
#undef UFR_MAX_STEP
      const float e = tv == 0 ? 0.f : expf(logit - mx);
      float den = e, cr = e * col[c][0], cg = e * col[c][1], cb = e * col[c][2];
#define UFR_SUM_STEP(S)                                                       \
      if (S < L) {                                                            \
        den += rot<L, S>(e, src);                                             \
        cr += rot<L, S>(e * col[c][0], src);                                  \
        cg += rot<L, S>(e * col[c][1], src);                                  \
        cb += rot<L, S>(e * col[c][2], src);                                  \
      }
      UFR_SUM_STEP(1) UFR_SUM_STEP(2) UFR_SUM_STEP(3) UFR_SUM_STEP(4) UFR_SUM_STEP(5) UFR_SUM_STEP(6) UFR_SUM_STEP(7)
#undef UFR_SUM_STEP
      if (valid[c] && tv == 0 && g == 0) {
        float* dst = at32(radiance, (unsigned)pidx[c] * 3u);
        const float rden = fast_rcp(den);
        dst[0] = cr * rden;
        dst[1] = cg * rden;
        dst[2] = cb * rden;
      }
    }
    }
    wstream_f16_finish<B_VT, kVtWaves>(ws, wrap);
    UFR_PHASE(11)  // softmax blend
  }
  wstream_report_range(ws, status);
#ifdef UFR_PHASE_TIMING
  if (wave_global == 5 && lane == 0)
    for (int i = 0; i < 12; ++i) g_vt_phase[i] += ph_acc[i];
  if (wave_global == 5 && lane == 0) {
    g_vt_phase[20] += __builtin_amdgcn_s_memrealtime() - rt_start;   // constant 100 MHz
    g_vt_phase[21] += __builtin_readcyclecounter() - t_start;
  }
  if (lane == 0 && wave_global < 4096) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    g_vt_wave[2 * wave_global] = t_start;
    g_vt_wave[2 * wave_global + 1] = (__builtin_readcyclecounter() - t_start) | ((unsigned long long)(xcc & 0xf) << 60) |
                                     ((unsigned long long)(hwid & 0xffff) << 40);
  }
#endif
}

template <int L, bool LOWP, bool TAPE = false>
static hipError_t launch_vt(const float* packed, const float* x_tokens, const float* x_point, const float* rgb, const float* dir, int P,
                            float* token0, float* radiance, float* view_out, int* status, hipStream_t s, float* tape = nullptr) {
  // TAPE: one column tile per wave -- the tape stores keep a tile's activations alive longer, and with two column tiles
  // the kernel spilled 250..330 registers (1.2 ms per 131 072 points for 0.2 ms of arithmetic)
  constexpr int C = TAPE ? 1 : UFR_VT_C;
  constexpr int PPW = (L == 6 && C == 2 && !TAPE) ? 5 : (16 / L) * C;   // L = 6: a fifth point straddles the wave's two column tiles
  const int n_groups = TAPE ? ((P + (16 / L) * kBlockCols - 1) / ((16 / L) * kBlockCols)) * (kBlockCols / C) : (P + PPW - 1) / PPW;
  int blocks = (n_groups + kVtWaves - 1) / kVtWaves;
  // Two workgroups are resident per CU, and the older one wins the SIMD's issue arbitration: with exactly
  // 512 persistent workgroups the favoured half finishes ~25 % early and the rest runs alone, without a
  // partner wave to overlap its VALU phases with (measured per-wave lifetimes 1.27 .. 1.64 ms).  Launching a
  // few times more, shorter workgroups lets the dispatcher refill a CU as soon as one retires.
  constexpr int resident = 256 * (UFR_VT_MINW * 4 / kVtWaves);   // workgroups the chip holds at once
  const int max_blocks = resident * UFR_VT_OVERSUB;
  if (blocks > max_blocks) {
    // every wave runs the same number of iterations (the chunk barriers are workgroup-wide): size the grid so that
    // the groups divide evenly over them instead of leaving most waves idle in a last, partial iteration
    // (18 432 groups over 8 192 waves would be 3 iterations with 25 % of the slots empty; 6 144 waves x 3 is exact)
    // ... and so that the workgroups fill whole rounds of the resident slots: cost ~ rounds x iterations
    const int base = (n_groups + max_blocks * kVtWaves - 1) / (max_blocks * kVtWaves);
    long best_cost = -1;
    for (int n_iter = base; n_iter <= 4 * base; ++n_iter) {
      const int b = (n_groups + kVtWaves * n_iter - 1) / (kVtWaves * n_iter);
      const long cost = (long)((b + resident - 1) / resident) * n_iter;
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; blocks = b; }
    }
  }
  static LdsAttrOnce lds_attr;   // per instantiation; thread-safe, once per device
  if (const hipError_t attr = lds_attr.set(reinterpret_cast<const void*>(&view_transformer_kernel<L, C, LOWP, TAPE>), kF16LdsBytes); attr != hipSuccess) return attr;
#ifdef UFR_VT_OCC_PROBE   // development: UFR_VT_PAD_LDS=<bytes> inflates the LDS request to limit the workgroups per CU
  static const int pad_lds = getenv("UFR_VT_PAD_LDS") ? atoi(getenv("UFR_VT_PAD_LDS")) : 0;
  if (pad_lds > 0) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&view_transformer_kernel<L, C, LOWP>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, kF16LdsBytes + pad_lds);
    hipLaunchKernelGGL((view_transformer_kernel<L, C, LOWP>), dim3(blocks), dim3(kVtBlock), kF16LdsBytes + pad_lds, s, packed,
                       x_tokens, x_point, rgb, dir, P, token0, radiance, view_out, status);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((view_transformer_kernel<L, C, LOWP, TAPE>), dim3(blocks), dim3(kVtBlock), kF16LdsBytes, s, packed, x_tokens,
                     x_point, rgb, dir, P, token0, radiance, view_out, status, tape);
  return hipGetLastError();
}

// The forward again for the backward: every activation goes to `tape` (bwd_tape.h; view_tape_blocks(P, NV) blocks of
// TV_COUNT tiles); token0 / radiance are written as usual (the caller passes scratch rows).
int view_tape_blocks(int P, int NV) {
  const int PPW = (16 / (NV + 1)) * kBlockCols;
  return (P + PPW - 1) / PPW;
}
hipError_t launch_view_tape(const float* packed, const float* x_tokens, const float* rgb, const float* dir, int P, int NV,
                            float* token0, float* radiance, float* tape, bool lowp, int* status, hipStream_t s) {
  if (P <= 0 || (unsigned long long)P * (NV + 1) * UFR_TOKEN_DIM >= (1ull << 30)) return hipErrorInvalidValue;
  switch (NV) {
#define UFR_VT_CASE(N)                                                                                                   \
    case N:                                                                                                              \
      return lowp ? launch_vt<N + 1, true, true>(packed, x_tokens, nullptr, rgb, dir, P, token0, radiance, nullptr, status, s, tape) \
                  : launch_vt<N + 1, false, true>(packed, x_tokens, nullptr, rgb, dir, P, token0, radiance, nullptr, status, s, tape);
    UFR_VT_CASE(2) UFR_VT_CASE(3) UFR_VT_CASE(4) UFR_VT_CASE(5) UFR_VT_CASE(6) UFR_VT_CASE(7)
#undef UFR_VT_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_view_transformer(const float* packed, const float* x_tokens, const float* x_point, const float* rgb,
                                   const float* dir, int P, int NV, float* token0, float* radiance, float* view_out, bool lowp, int* status,
                                   hipStream_t s) {
  // the kernel addresses its buffers with 32-bit element offsets: the largest one, view_out (P, NV + 1, 80), must stay
  // below 2^32 floats' worth of bytes / 4 (6.7 M points at NV = 3; a chunk of the whole-path call has 0.5 M)
  if (P <= 0 || (unsigned long long)P * (NV + 1) * UFR_TOKEN_DIM >= (1ull << 30)) return hipErrorInvalidValue;
  switch (NV) {
#define UFR_VT_CASE(N)                                                                                       \
    case N:                                                                                                  \
      return lowp ? launch_vt<N + 1, true>(packed, x_tokens, x_point, rgb, dir, P, token0, radiance, view_out, status, s)     \
                  : launch_vt<N + 1, false>(packed, x_tokens, x_point, rgb, dir, P, token0, radiance, view_out, status, s);
    UFR_VT_CASE(2) UFR_VT_CASE(3) UFR_VT_CASE(4) UFR_VT_CASE(5) UFR_VT_CASE(6) UFR_VT_CASE(7)
#undef UFR_VT_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ufr

#ifdef UFR_PHASE_TIMING
extern "C" int ufr_debug_vt_phases(unsigned long long* out, int n, int reset) {
  unsigned long long h[32] = {};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ufr::g_vt_phase), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 32; ++i) out[i] = h[i];
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ufr::g_vt_phase), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int ufr_debug_vt_waves(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ufr::g_vt_wave), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
