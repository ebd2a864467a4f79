// Building blocks of the backward kernels (config C5: training step through the HIP path).
//
// Design (gfx950): a 256-thread workgroup (one wave per SIMD, one workgroup per CU: every wave owns the full 512-entry
// register file of its SIMD, which is what the register-resident weight-gradient tiles need) walks
// tiles of kTT = 32 tokens = kCT = 2 MFMA column tiles (round 2: 16.  The phases of a tile are separated by workgroup
// barriers and are latency-bound -- LDS round trips, weight loads from L2, shuffles -- so two column tiles per phase
// nearly halve the per-token cost: one weight fragment feeds two MFMAs, every per-token phase has all 256 threads busy).
// Every activation of the tile lives in LDS, feature-major [feature][kLD]
// (kLD = kTT + 1: odd stride -> the three MFMA operand access patterns below are bank-conflict free), in
// exact fp32; buffers whose live ranges do not overlap share rows (the kernels' row maps).
// All contractions run on v_mfma_f32_16x16x4_f32 (bitwise an fp32 fma chain):
//   data GEMM      Y[o][t]  = sum_i W[o][i]  X[i][t]    A = weights from global (L1/L2-resident, 0.6 MB),
//   transposed     dX[i][t] = sum_o W[o][i] dY[o][t]     B = activations from LDS, tokens are the 16 columns
//   weight grad    dW[o][i] = sum_t dY[o][t] X[i][t]     A and B from LDS, k = the tile's 16 tokens
// Weight gradients accumulate in REGISTERS across the workgroup's persistent tile loop (output tile
// (s*4 + wave) of the kernel's gradient-tile list lives in accumulator slot s of that wave) and are
// flushed once per workgroup with float atomics into the reference-layout gradient tensors.  A matrix's tiles are
// accumulated right after its dY is final (wgrad_range), beside the GEMM that consumes the same dY -- not in one phase at
// the end of the tile: that is what lets dY / X buffers die early and share LDS rows.
#pragma once
#include "ufr_device.h"
#include "weight_stream.h"   // static_for

namespace ufr {

#ifndef UFR_BWD_THREADS
#define UFR_BWD_THREADS 256
#endif
constexpr int kBwdThreads = UFR_BWD_THREADS;
constexpr int kBwdWaves = kBwdThreads / 64;
#ifndef UFR_BWD_TT
#define UFR_BWD_TT 32
#endif
constexpr int kTT = UFR_BWD_TT;   // tokens per tile
constexpr int kCT = kTT / 16;     // MFMA column tiles per tile
constexpr int kLD = kTT + 1;      // LDS row stride (floats)
static_assert(kTT % 16 == 0 && kBwdThreads % kTT == 0, "tile shape");

struct GradPtrs { float* p[P_COUNT]; };

// development build (-DUFR_BWD_TIMING): cycles between consecutive barriers of workgroup 0, summed over its tiles
#ifdef UFR_BWD_TIMING
#define UFR_BWD_PHASE(arr, i)                                          \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                           \
    const unsigned long long t_now = __builtin_readcyclecounter();     \
    arr[i] += t_now - t_prev;                                          \
    t_prev = t_now;                                                    \
  }
#else
#define UFR_BWD_PHASE(arr, i)
#endif

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
// ---- reduced-precision mode (ufr_set_matrix_precision(UFR_PRECISION_16BIT)): the GEMMs and weight-gradient products of the
// backward kernels take bf16 operands (v_mfma_f32_16x16x16_bf16: lane (g, j) supplies k = 4g..4g+3, exactly the four
// values it feeds to four consecutive v_mfma_f32_16x16x4_f32 in the fp32 mode), fp32 accumulation as before.
typedef short bf16x4_bits __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4_bits pack_bf16x4(float a, float b, float c, float d) {   // round to nearest even
  const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{c, d}, bf16x2_t));
  return __builtin_bit_cast(bf16x4_bits, u32x2_t{lo, hi});
}
__device__ __forceinline__ f32x4 mfma16_bf16(bf16x4_bits a, bf16x4_bits b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// A operands of one row tile of a GEMM: [k chunk][MFMA step]
template <int IN>
struct AFrag { float a[(IN + 15) / 16][4]; };

// All A operands of a row tile are fetched in one burst (branch-free: out-of-range rows read a valid row whose
// results are never stored, out-of-range k reads element 0 and is zeroed by a select).
template <int OUT, int IN, bool TRANS>
__device__ __forceinline__ void gemm_load_a(const float* __restrict__ W, int ldw, int rt, int lane, AFrag<IN>& f) {
  constexpr int KC = (IN + 15) / 16, FULL = IN / 16;
  const int g = lane >> 4, j = lane & 15;
  int row = rt * 16 + j;
  row = row < OUT ? row : OUT - 1;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int kb = kc * 16 + 4 * g;
    if constexpr (!TRANS && (IN % 4 == 0)) {
      const bool ok = kc < FULL || kb < IN;          // IN % 4 == 0: a float4 is all in or all out
      f32x4 a4 = ld4(W + (size_t)row * ldw + (ok ? kb : 0));
      if (!ok) a4 = splat4(0.f);
      f.a[kc][0] = a4[0]; f.a[kc][1] = a4[1]; f.a[kc][2] = a4[2]; f.a[kc][3] = a4[3];
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool ok = kc < FULL || kb + kk < IN;
        const int k = ok ? kb + kk : 0;
        const float v = TRANS ? W[(size_t)k * ldw + row] : W[(size_t)row * ldw + k];
        f.a[kc][kk] = ok ? v : 0.f;
      }
    }
  }
}

// first row tile of this wave in a GEMM whose tiles start at position rt_shift of the phase's round-robin
__device__ __forceinline__ int gemm_first_rt(int wave, int rt_shift) {
  return (wave + kBwdWaves * 64 - rt_shift) % kBwdWaves;
}

// Issue the A burst of this wave's first row tile.  Called BEFORE the barrier that publishes the GEMM's B operand, so
// the L2 latency of the weights overlaps the tail of the previous phase and the barrier wait.
template <int OUT, int IN, bool TRANS>
__device__ __forceinline__ AFrag<IN> gemm_prefetch(const float* __restrict__ W, int ldw, int wave, int lane, int rt_shift = 0) {
  constexpr int RT = (OUT + 15) / 16;
  AFrag<IN> f;
  int opaque_zero;   // weights are invariant over the tile loop: keep LICM from hoisting the loads out of it
  asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
  const int rt = gemm_first_rt(wave, rt_shift);
  if (rt < RT) gemm_load_a<OUT, IN, TRANS>(W + opaque_zero, ldw, rt, opaque(lane), f);
  return f;
}

// The B operand (this tile's activations, kCT column tiles) streams from LDS one 16-deep k-chunk AHEAD of the MFMAs that
// consume it (two 8-register buffers, instead of the whole K x kCT operand in registers: with two column tiles that was
// 80 registers for K = 160 and the kernels spilled 260); a chunk's 4 kCT MFMAs (>= 256 cycles) cover the LDS latency.
template <int IN>
__device__ __forceinline__ void gemm_load_b(const float* X, int kc, int g, int j, float (&b)[kCT][4]) {
  constexpr int FULL = IN / 16;
  const int kb = kc * 16 + 4 * g;
#pragma unroll
  for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bool ok = kc < FULL || kb + kk < IN;      // A is zero there, but the LDS row may hold anything (NaN)
      const float v = X[(ok ? kb + kk : 0) * kLD + 16 * ct + j];
      b[ct][kk] = ok ? v : 0.f;
    }
}

template <int OUT, int IN, bool TRANS, bool LOWP, typename Epi>
__device__ __forceinline__ void gemm_compute(AFrag<IN>& cur, const float* __restrict__ W, int ldw, const float* X, int wave,
                                             int lane, Epi epi, int rt_shift = 0) {
  constexpr int RT = (OUT + 15) / 16, KC = (IN + 15) / 16;
  lane = opaque(lane);
  const int g = lane >> 4, j = lane & 15;
  int opaque_zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
  W += opaque_zero;
  for (int rt = gemm_first_rt(wave, rt_shift); rt < RT; rt += kBwdWaves) {
    AFrag<IN> nxt;
    const bool more = rt + kBwdWaves < RT;
    if (more) gemm_load_a<OUT, IN, TRANS>(W, ldw, rt + kBwdWaves, lane, nxt);   // next row tile's burst rides on these MFMAs
    // ONE accumulator per column tile, its MFMAs of a k-chunk back to back: the accumulate chain stays in the matrix pipe
    // (two interleaved partial sums per column tile, alternating between the tiles, were 2 % slower: view_bwd 4.26 -> 4.16)
    f32x4 acc0[kCT];
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct) acc0[ct] = splat4(0.f);
    float bb[2][kCT][4];
    gemm_load_b<IN>(X, 0, g, j, bb[0]);
    static_for<KC>([&](auto kci) __attribute__((always_inline)) {
      constexpr int kc = decltype(kci)::value;
      if constexpr (kc + 1 < KC) gemm_load_b<IN>(X, kc + 1, g, j, bb[(kc + 1) & 1]);
      float (&b)[kCT][4] = bb[kc & 1];
      if constexpr (LOWP) {
        const bf16x4_bits a4 = pack_bf16x4(cur.a[kc][0], cur.a[kc][1], cur.a[kc][2], cur.a[kc][3]);
#pragma unroll
        for (int ct = 0; ct < kCT; ++ct) {
          const bf16x4_bits b4 = pack_bf16x4(b[ct][0], b[ct][1], b[ct][2], b[ct][3]);
          acc0[ct] = mfma16_bf16(a4, b4, acc0[ct]);
        }
      } else {
#pragma unroll
        for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc0[ct] = mfma16(cur.a[kc][kk], b[ct][kk], acc0[ct]);
      }
      // keep the chunks in order: without this the scheduler hoists every LDS read of the row tile to its top again
      __builtin_amdgcn_sched_barrier(0);
    });
    // Stores without per-element guards.  At run time a guard is a compare, an exec-mask save / restore and a branch per
    // ELEMENT (view_bwd 4.46 -> 4.27 ms, ray_bwd 2.80 -> 2.63 ms when they went): whole row tiles need none, and when
    // the row count is a multiple of 4 a lane's four rows are all in or all out -- one guard per lane and row tile.
    const int orow0 = rt * 16 + 4 * g;
    if (OUT % 16 == 0 || OUT % 4 != 0 || orow0 < OUT) {
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct) {
        const f32x4 acc = acc0[ct];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (OUT % 4 == 0 || orow0 + r < OUT) epi(orow0 + r, 16 * ct + j, acc[r]);
      }
    }
    if (more) cur = nxt;
  }
}

// prefetch + compute in one call (phases whose B operand is already published)
template <int OUT, int IN, bool TRANS, bool LOWP, typename Epi>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ W, int ldw, const float* X, int wave, int lane,
                                         Epi epi, int rt_shift = 0) {
  AFrag<IN> f = gemm_prefetch<OUT, IN, TRANS>(W, ldw, wave, lane, rt_shift);
  gemm_compute<OUT, IN, TRANS, LOWP>(f, W, ldw, X, wave, lane, epi, rt_shift);
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

// Operand rows of the tile (slot s, wave w) = tile s*kBwdWaves + w of a list, for the weight-gradient phase: a table in
// device memory that the waves read with scalar loads, so that ALL WAVES EXECUTE THE SAME INSTRUCTIONS.  (The first
// version decoded the tile at compile time per wave behind a switch over the wave id: four copies of a 15 KB unrolled
// phase, each fetched by one wave only -- with 140..190 KB of code per kernel against a 64 KB instruction cache shared by
// two CUs the phase was instruction-fetch-bound: 38 k cycles per tile, 20 k with one shared copy.)
// dY rows a_row + min(j, a_last), X rows b_row + min(j, b_last), stored as float offsets (row * kLD): a lane past the
// matrix edge re-reads the last valid row and fills accumulator rows / columns that wgrad_flush never writes out -- no
// masking of the operands needed.  One 16-byte entry = one s_load_dwordx4.
struct alignas(16) WgEntry { int a_off, a_last, b_off, b_last; };
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int NSLOT>
struct WgTab { WgEntry e[NSLOT * kBwdWaves]; };
template <int N, int NSLOT>
__host__ __device__ constexpr WgTab<NSLOT> make_wgtab(const WgList<N>& l) {
  WgTab<NSLOT> t{};
  for (int s = 0; s < NSLOT; ++s)
    for (int w = 0; w < kBwdWaves; ++w) {
      const WgTile d = wg_decode(l, s * kBwdWaves + w);
      t.e[s * kBwdWaves + w] = d.valid ? WgEntry{(d.dy + d.o0) * kLD, ((d.OUT - d.o0 < 16 ? d.OUT - d.o0 : 16) - 1) * kLD,
                                                 (d.x + d.i0) * kLD, ((d.IN - d.i0 < 16 ? d.IN - d.i0 : 16) - 1) * kLD}
                                       : WgEntry{0, 0, 0, 0};   // a slot past the list: any valid rows, never flushed
    }
  return t;
}
template <const auto& LIST, int N, int NSLOT>
__device__ const WgTab<NSLOT> g_wgtab = make_wgtab<N, NSLOT>(LIST);

// The wave's rows of that table live in VGPRs for the whole kernel, slot s in lane s % 64 (field f of slots 64k .. 64k+63
// in register r[f][k]); the phase pulls an entry out with four v_readlane.  No memory operation in the phase: scalar
// loads share lgkmcnt with the LDS operand reads and return out of order, so waiting for one drains the others.
template <int NSLOT>
struct WgRegs { int r[4][(NSLOT + 63) / 64]; };
template <const auto& LIST, int N, int NSLOT>
__device__ __forceinline__ WgRegs<NSLOT> wgrad_table(int wave, int lane) {
  WgRegs<NSLOT> t;
  const i32x4_t* tab = reinterpret_cast<const i32x4_t*>(g_wgtab<LIST, N, NSLOT>.e);
#pragma unroll
  for (int k = 0; k < (NSLOT + 63) / 64; ++k) {
    const int slot = min(64 * k + lane, NSLOT - 1);
    const i32x4_t e = tab[slot * kBwdWaves + wave];
    t.r[0][k] = e[0]; t.r[1][k] = e[1]; t.r[2][k] = e[2]; t.r[3][k] = e[3];
  }
  return t;
}

// Accumulate the tiles [TILE0, TILE1) of the list (one matrix, or several whose operands are live at the same time) that
// this wave owns (slot s <-> tile s*kBwdWaves + wave).  In a slot that straddles the range only some waves take part: a
// wave-uniform branch.  Slots are processed in groups of kWgGroup: all LDS operand reads of a group are issued first, then
// its MFMAs with the group's independent accumulators interleaved (four dependent chains keep the MFMA issue rate).
constexpr int kWgGroup = 4;
template <int NSLOT, int TILE0, int TILE1, bool LOWP, int NACC>
__device__ __forceinline__ void wgrad_range(f32x4 (&acc)[NACC], const float* lds, const WgRegs<NSLOT>& tab, int wave, int lane) {
  static_assert(NSLOT <= NACC && TILE0 < TILE1 && TILE1 <= NSLOT * kBwdWaves, "accumulator slots");
  constexpr int S0 = TILE0 / kBwdWaves, S1 = (TILE1 + kBwdWaves - 1) / kBwdWaves;
  constexpr int NQ = LOWP ? kCT : kTT / 4;      // MFMAs per tile: fp32 contracts 4 tokens each, bf16 16 tokens
  lane = opaque(lane);
  const int g = lane >> 4, j = lane & 15;
  const int jrow = j * kLD;
  const float* lane_base = lds + (LOWP ? 4 * g : g);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  static_for<(S1 - S0 + kWgGroup - 1) / kWgGroup>([&](auto gi) __attribute__((always_inline)) {
    constexpr int s0 = S0 + decltype(gi)::value * kWgGroup;
    float a[kWgGroup][LOWP ? 4 * kCT : kTT / 4], b[kWgGroup][LOWP ? 4 * kCT : kTT / 4];
    bool on[kWgGroup];
    static_for<kWgGroup>([&](auto ui) __attribute__((always_inline)) {
      constexpr int u = decltype(ui)::value;
      if constexpr (s0 + u < S1) {
        constexpr int sl = s0 + u;
        constexpr bool interior = sl * kBwdWaves >= TILE0 && sl * kBwdWaves + kBwdWaves - 1 < TILE1;
        on[u] = interior || (sl * kBwdWaves + wave_u >= TILE0 && sl * kBwdWaves + wave_u < TILE1);
        if (on[u]) {
          const int e0 = __builtin_amdgcn_readlane(tab.r[0][sl / 64], sl % 64), e1 = __builtin_amdgcn_readlane(tab.r[1][sl / 64], sl % 64);
          const int e2 = __builtin_amdgcn_readlane(tab.r[2][sl / 64], sl % 64), e3 = __builtin_amdgcn_readlane(tab.r[3][sl / 64], sl % 64);
          const float* pa = lane_base + (e0 + min(jrow, e1));
          const float* pb = lane_base + (e2 + min(jrow, e3));
          if constexpr (LOWP) {   // MFMA h takes tokens 16h + 4g .. 4g+3 from this lane
#pragma unroll
            for (int q = 0; q < 4 * kCT; ++q) {
              a[u][q] = pa[16 * (q >> 2) + (q & 3)];
              b[u][q] = pb[16 * (q >> 2) + (q & 3)];
            }
          } else {                // MFMA q contracts tokens 4q + g
#pragma unroll
            for (int q = 0; q < kTT / 4; ++q) {
              a[u][q] = pa[4 * q];
              b[u][q] = pb[4 * q];
            }
          }
        }
      }
    });
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      static_for<kWgGroup>([&](auto ui) __attribute__((always_inline)) {
        constexpr int u = decltype(ui)::value;
        if constexpr (s0 + u < S1) {
          if (on[u]) {
            if constexpr (LOWP)
              acc[s0 + u] = mfma16_bf16(pack_bf16x4(a[u][4 * q], a[u][4 * q + 1], a[u][4 * q + 2], a[u][4 * q + 3]),
                                        pack_bf16x4(b[u][4 * q], b[u][4 * q + 1], b[u][4 * q + 2], b[u][4 * q + 3]), acc[s0 + u]);
            else
              acc[s0 + u] = mfma16(a[u][q], b[u][q], acc[s0 + u]);
          }
        }
      });
  });
}
// all tiles of a list (kernels with one weight-gradient phase)
template <int NSLOT, int SLOT0, bool LOWP, int NACC>
__device__ __forceinline__ void wgrad_all(f32x4 (&acc)[NACC], const float* lds, const WgRegs<NSLOT>& tab, int lane) {
  static_assert(SLOT0 == 0, "lists that do not start at accumulator 0 use wgrad_range on a sub-array");
  wgrad_range<NSLOT, 0, NSLOT * kBwdWaves, LOWP>(acc, lds, tab, 0, lane);
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
    case 3: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 3>(acc, gp, lane); break;
    case 4: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 4 % kBwdWaves>(acc, gp, lane); break;
    case 5: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 5 % kBwdWaves>(acc, gp, lane); break;
    case 6: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 6 % kBwdWaves>(acc, gp, lane); break;
    default: wgrad_flush_wave<LIST, N, NSLOT, SLOT0, 7 % kBwdWaves>(acc, gp, lane); break;
  }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over D features of each of the 16 tokens: kTPT consecutive threads per token.
//   forward : buf[D][kLD] holds the input and receives xhat; out[f] = xhat*gamma + beta (+ res[f] if res); rstd[t]
//   backward: dout[D][kLD] -> din = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dout*gamma   (written to din)
constexpr int kTPT = kBwdThreads / kTT;   // threads per token in the per-token phases
// thread -> (token, sub): the kTPT threads of a token are adjacent lanes (shuffle reductions).  With 32-token tiles
// (kTPT = 8) the tokens of a half-wave are 8 apart, so that the 32 lanes of an LDS access [feature sub + kTPT i][token]
// hit 32 different banks (bank = sub + 8 (token / 8) + const); the plain tid / kTPT map put four tokens of a half-wave on
// overlapping banks (4-way conflicts in every LayerNorm phase).
__device__ __forceinline__ void ln_thread_map(int tid, int& tok, int& sub) {
  if constexpr (kTT == 32 && kBwdThreads == 256) {
    const int lane = tid & 63, wave = tid >> 6;
    sub = lane & 7;
    tok = 8 * ((lane >> 3) & 3) + 2 * wave + (lane >> 5);
  } else {
    tok = tid / kTPT;
    sub = tid % kTPT;
  }
}
template <int D>
__device__ __forceinline__ void ln_forward(float* buf, float* out, const float* res, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* rstd, int tid) {
  tid = opaque(tid);
  int tok, sub;
  ln_thread_map(tid, tok, sub);
  // a thread's features stay in registers over the three passes (one LDS read each instead of three)
  constexpr int NPT = (D + kTPT - 1) / kTPT;
  float v[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int f = sub + i * kTPT;
    v[i] = f < D ? buf[f * kLD + tok] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPT; ++i)
    if (sub + i * kTPT < D) s += v[i];
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  const float mean = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NPT; ++i)
    if (sub + i * kTPT < D) {
      const float c = v[i] - mean;
      q = fmaf(c, c, q);
    }
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) q += __shfl_xor(q, d);
  const float rs = 1.f / sqrtf(q * (1.f / D) + 1e-5f);
  if (sub == 0) rstd[tok] = rs;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int f = sub + i * kTPT;
    if (f < D) {
      const float xh = (v[i] - mean) * rs;
      buf[f * kLD + tok] = xh;
      float y = fmaf(xh, gamma[f], beta[f]);
      if (res) y += res[f * kLD + tok];
      out[f * kLD + tok] = y;
    }
  }
}

template <int D>
__device__ __forceinline__ void ln_backward(const float* dout, const float* xhat, const float* __restrict__ gamma,
                                            const float* rstd, float* din, int tid) {
  tid = opaque(tid);
  int tok, sub;
  ln_thread_map(tid, tok, sub);
  constexpr int NPT = (D + kTPT - 1) / kTPT;
  float gg[NPT], xh[NPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int f = sub + i * kTPT;
    gg[i] = 0.f;
    xh[i] = 0.f;
    if (f < D) {
      gg[i] = dout[f * kLD + tok] * gamma[f];
      xh[i] = xhat[f * kLD + tok];
      s1 += gg[i];
      s2 = fmaf(gg[i], xh[i], s2);
    }
  }
#pragma unroll
  for (int d = kTPT / 2; d >= 1; d >>= 1) {
    s1 += __shfl_xor(s1, d);
    s2 += __shfl_xor(s2, d);
  }
  const float m1 = s1 * (1.f / D), m2 = s2 * (1.f / D), rs = rstd[tok];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int f = sub + i * kTPT;
    if (f < D) din[f * kLD + tok] = rs * (gg[i] - m1 - xh[i] * m2);
  }
}

// row sums over the tile's tokens: sum_t a[t] * (b ? b[t] : 1).  All 2 kTT operands are requested before the first use
// (one LDS round trip): written as a running sum inside the callers' thread-range branches, every read was followed by
// its own s_waitcnt and the "small gradient" sums cost 7 k cycles per tile.  `b == nullptr` lanes read `a` twice and
// multiply by one, so that all threads of a phase run the same instructions.
__device__ __forceinline__ float row_dot(const float* a, const float* b, int f) {
  const float* pa = a + f * kLD;
  const float* pb = b ? b + f * kLD : pa;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t0 = 0; t0 < kTT; t0 += 8) {      // batches of 8 tokens: 16 transient registers beside the gradient accumulators
    float va[8], vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      va[t] = pa[t0 + t];
      vb[t] = pb[t0 + t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) s[t & 3] = fmaf(va[t], b ? vb[t] : 1.f, s[t & 3]);
    __builtin_amdgcn_sched_barrier(0);        // keep the batches apart: hoisting all reads to the top costs 64 registers
  }
  return (s[0] + s[1]) + (s[2] + s[3]);
}

__device__ __forceinline__ float elu1_grad(float x) { return x > 0.f ? 1.f : __expf(x); }

}  // namespace ufr
