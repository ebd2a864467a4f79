// Weight gradients of a transformer chain as ONE streaming contraction over the tokens (round 4):
//     dW[o][i] = sum_t dY[o][t] X[i][t]        for every dense matrix of the chain,
// plus the gradients that are row sums / diagonals of such products (biases, LayerNorm gamma / beta, the view token).
// Operands come from the tile buffers of bwd_tape.h: X tiles from the forward tape, dY tiles from the data-gradient kernel.
//
// gfx950 mapping.  The contraction index is the TOKEN, i.e. the MFMA's k, while a stored tile has tokens on the lanes
// (lane (g, j): features 4g..4g+3 of token column j) -- the transpose of what an MFMA operand wants.  It is transposed ON
// THE MATRIX CORE: the lane's four values, split into bf16 planes, are the A operand (m = token j, k = feature 4g + i) of one
// v_mfma_f32_16x16x32_bf16 against a constant selector B[k][n] = (k == n), whose result D[token][feature] -- exact, every
// sum has one non-zero term -- lands as lane (g', j') = feature j', tokens 4g'..4g'+3: the operand layout of the
// contraction MFMA (k slot 8g' + i <-> token 4g' + (i & 3) of column tile i >> 2, the same for both operands).  No LDS, no
// barrier, no cross-lane VALU: every wave is an independent worker that streams coalesced 1 KiB tile loads, which is what
// an HBM-bound kernel wants (the matrix pipe idles at ~15 %).  fp32 mode: both operands as bf16 hi + lo planes, three
// products per fp32 product (16 significand bits per operand: 1e-5 relative, the tolerance is 1e-3 of scale); 16-bit
// mode: the hi planes only.
// Work split: a wave owns a fixed set of (dY tile, X tile) accumulator tiles in registers -- a ROLE, by MATRIX: q, k, v,
// merge, the four quadrants of mlp0, ... -- for a contiguous range of token blocks, and flushes them once with float atomics
// into the reference-layout gradient tensors (undoing the row maps of ufr_layout.h).
#include <cstdlib>
#include "bwd_common.h"     // GradPtrs, atomic_add_f32
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"   // f16x8 / bf16x8, mfma_planes

namespace ufr {
namespace wgs {

enum FMap : int { FM_NAT = 0, FM_SLOT20, FM_RW0, FM_NAT88, FM_QUAD11, FM_HEAD11K };
// feature held by row m = 4g + r of tile t of a tensor with that row map (-1: padding)
__host__ __device__ constexpr int fmap(int kind, int t, int m, int dim) {
  const int g = m >> 2, r = m & 3;
  int v = -1;
  switch (kind) {
    case FM_NAT: v = 16 * t + m; break;
    case FM_SLOT20: v = 20 * g + 4 * t + r; break;
    case FM_RW0: v = t < 5 ? 16 * t + m : ((r == 0 && g < 3) ? 80 + g : -1); break;   // [y 80 | dir 3] (COL_RW0)
    case FM_NAT88: v = nat88(t, g, r); break;
    case FM_QUAD11: v = quad11(t, g, r); break;
    case FM_HEAD11K: v = head11_slot(m) >= 0 ? 11 * t + head11_slot(m) : -1; break;   // head tile t
  }
  return (v >= 0 && v < dim) ? v : -1;
}

enum Kind : int { K_FULL = 0, K_DIAG, K_ROWSUM };
enum Buf : int { B_TAPE = 0, B_DY = 1 };
// One job = all accumulator tiles (dY tile ai, X tile bi) of a block of a matrix.
//   dY tiles  a_tile .. a_tile + a_n   of the dY buffer, holding tiles a_t0.. of a tensor with row map a_map and a_dim rows
//   X tiles   b_tile .. b_tile + b_n   of buffer b_buf,  holding tiles b_t0.. of a tensor with row map b_map and b_dim rows
//   K_FULL: a_n x b_n tiles -> param[o * ld + i_off + i];  K_DIAG: tile (ai, ai), its diagonal -> param[o];
//   K_ROWSUM: dY tile x ones -> param[o]
struct Job { int kind, a_tile, a_n, a_t0, a_map, a_dim, b_buf, b_tile, b_n, b_t0, b_map, b_dim, param, ld, i_off; };
__host__ __device__ constexpr int job_slots(const Job& j) { return j.kind == K_FULL ? j.a_n * j.b_n : j.a_n; }

constexpr int kMaxJobs = 8;
struct Role { int n; Job j[kMaxJobs]; };
__host__ __device__ constexpr int role_slots(const Role& r) {
  int s = 0;
  for (int i = 0; i < r.n; ++i) s += job_slots(r.j[i]);
  return s;
}
__host__ __device__ constexpr int role_slot0(const Role& r, int job) {
  int s = 0;
  for (int i = 0; i < job; ++i) s += job_slots(r.j[i]);
  return s;
}

// ---- view transformer: 3 workgroup types x 4 waves
constexpr Job full(int a_tile, int a_n, int a_t0, int a_map, int a_dim, int b_buf, int b_tile, int b_n, int b_t0, int b_map, int b_dim,
                   int param, int ld, int i_off = 0) {
  return Job{K_FULL, a_tile, a_n, a_t0, a_map, a_dim, b_buf, b_tile, b_n, b_t0, b_map, b_dim, param, ld, i_off};
}
constexpr Job diag(int a_tile, int a_n, int a_dim, int b_tile, int param) {
  return Job{K_DIAG, a_tile, a_n, 0, FM_NAT, a_dim, B_TAPE, b_tile, a_n, 0, FM_NAT, a_dim, param, 0, 0};
}
constexpr Job rowsum(int a_tile, int a_n, int a_dim, int param) {
  return Job{K_ROWSUM, a_tile, a_n, 0, FM_NAT, a_dim, B_TAPE, 0, 0, 0, FM_NAT, 0, param, 0, 0};
}
constexpr int kViewTypes = 3;
constexpr Role kViewRoles[kViewTypes * 4] = {
    // type 0: the attention projections and merge, one 5 x 5 matrix per wave
    {1, {full(DV_Q, 5, 0, FM_SLOT20, 80, B_TAPE, TV_X, 5, 0, FM_NAT, 80, P_VT_Q, 80)}},
    {1, {full(DV_K, 5, 0, FM_SLOT20, 80, B_TAPE, TV_X, 5, 0, FM_NAT, 80, P_VT_K, 80)}},
    {1, {full(DV_V, 5, 0, FM_SLOT20, 80, B_TAPE, TV_X, 5, 0, FM_NAT, 80, P_VT_V, 80)}},
    {1, {full(DV_MPRE, 5, 0, FM_NAT, 80, B_TAPE, TV_MSG, 5, 0, FM_SLOT20, 80, P_VT_MERGE, 80)}},
    // type 1: mlp0 (160 x [x 80 | m 80]) in quadrants
    {1, {full(DV_HID, 5, 0, FM_NAT, 160, B_TAPE, TV_X, 5, 0, FM_NAT, 80, P_VT_MLP0, 160, 0)}},
    {1, {full(DV_HID, 5, 0, FM_NAT, 160, B_TAPE, TV_M, 5, 0, FM_NAT, 80, P_VT_MLP0, 160, 80)}},
    {1, {full(DV_HID + 5, 5, 5, FM_NAT, 160, B_TAPE, TV_X, 5, 0, FM_NAT, 80, P_VT_MLP0, 160, 0)}},
    {1, {full(DV_HID + 5, 5, 5, FM_NAT, 160, B_TAPE, TV_M, 5, 0, FM_NAT, 80, P_VT_MLP0, 160, 80)}},
    // type 2: mlp2 (80 x 160) in three column ranges of the hidden layer, and the radiance MLP with its biases
    // (LayerNorm gamma / beta and the view token are reduced inside view_dgrad.hip)
    {1, {full(DV_OPRE, 5, 0, FM_NAT, 80, B_TAPE, TV_HID, 4, 0, FM_NAT, 160, P_VT_MLP2, 160)}},
    {1, {full(DV_OPRE, 5, 0, FM_NAT, 80, B_TAPE, TV_HID + 4, 3, 4, FM_NAT, 160, P_VT_MLP2, 160)}},
    {1, {full(DV_OPRE, 5, 0, FM_NAT, 80, B_TAPE, TV_HID + 7, 3, 7, FM_NAT, 160, P_VT_MLP2, 160)}},
    {6, {full(DV_H1, 1, 0, FM_NAT, 16, B_TAPE, TV_Y, 6, 0, FM_RW0, 83, P_RW_W0, 83), rowsum(DV_H1, 1, 16, P_RW_B0),
         full(DV_H2, 1, 0, FM_NAT, 8, B_TAPE, TV_H1, 1, 0, FM_NAT, 16, P_RW_W2, 16), rowsum(DV_H2, 1, 8, P_RW_B2),
         full(DV_LG, 1, 0, FM_NAT, 1, B_TAPE, TV_H2, 1, 0, FM_NAT, 8, P_RW_W4, 8), rowsum(DV_LG, 1, 1, P_RW_B4)}},
};

// an operand tile pair (both column tiles of a block) in contraction layout: bf16 planes, 8 k slots per lane
struct Frag { f16x8 p[2]; };

__device__ __forceinline__ unsigned hi16_pair(float a, float b) {   // bf16 bits of two values that ARE bf16 numbers
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

// raw tiles (natural layout, column tiles 0 and 1) -> Frag.  sel: the transposition selector of this lane.
template <bool LOWP>
__device__ __forceinline__ Frag make_frag(const f32x4& v0, const f32x4& v1, const f16x8& sel) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  Frag f;
  unsigned h0[2], l0[2], h1[2], l1[2];
  split_pair_bf16(v0[0], v0[1], h0[0], l0[0]);
  split_pair_bf16(v0[2], v0[3], h0[1], l0[1]);
  split_pair_bf16(v1[0], v1[1], h1[0], l1[0]);
  split_pair_bf16(v1[2], v1[3], h1[1], l1[1]);
  const f32x4 z = splat4(0.f);
  // A operand of the transposition: k slots 8g + i, i < 4 <- this lane's four features; slots i >= 4 are zero
  auto tr = [&](unsigned w0, unsigned w1) __attribute__((always_inline)) -> f32x4 {
    const f16x8 a = __builtin_bit_cast(f16x8, u32x4v{w0, w1, 0u, 0u});
    return mfma_planes<true>(a, sel, z);
  };
  const f32x4 th0 = tr(h0[0], h0[1]), th1 = tr(h1[0], h1[1]);
  f.p[0] = __builtin_bit_cast(f16x8, u32x4v{hi16_pair(th0[0], th0[1]), hi16_pair(th0[2], th0[3]), hi16_pair(th1[0], th1[1]),
                                            hi16_pair(th1[2], th1[3])});
  if constexpr (!LOWP) {
    const f32x4 tl0 = tr(l0[0], l0[1]), tl1 = tr(l1[0], l1[1]);
    f.p[1] = __builtin_bit_cast(f16x8, u32x4v{hi16_pair(tl0[0], tl0[1]), hi16_pair(tl0[2], tl0[3]), hi16_pair(tl1[0], tl1[1]),
                                              hi16_pair(tl1[2], tl1[3])});
  } else {
    f.p[1] = f.p[0];
  }
  return f;
}

// the same from tiles STORED as bf16 (16-bit mode, bwd_tape.h): the lane's 8 bytes are its hi-plane words already
__device__ __forceinline__ Frag make_frag16(const u32x2_tile& w0, const u32x2_tile& w1, const f16x8& sel) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const f32x4 z = splat4(0.f);
  const f32x4 t0 = mfma_planes<true>(__builtin_bit_cast(f16x8, u32x4v{w0[0], w0[1], 0u, 0u}), sel, z);
  const f32x4 t1 = mfma_planes<true>(__builtin_bit_cast(f16x8, u32x4v{w1[0], w1[1], 0u, 0u}), sel, z);
  Frag f;
  f.p[0] = __builtin_bit_cast(f16x8, u32x4v{hi16_pair(t0[0], t0[1]), hi16_pair(t0[2], t0[3]), hi16_pair(t1[0], t1[1]), hi16_pair(t1[2], t1[3])});
  f.p[1] = f.p[0];
  return f;
}
// raw tile of one column tile as it sits in memory, and its conversion
template <bool IS16> struct RawTile { f32x4 v; };
template <> struct RawTile<true> { u32x2_tile v; };
template <bool LOWP, bool IS16>
__device__ __forceinline__ Frag to_frag(const RawTile<IS16>& a, const RawTile<IS16>& b, const f16x8& sel) {
  if constexpr (IS16) return make_frag16(a.v, b.v, sel);
  else return make_frag<LOWP>(a.v, b.v, sel);
}

template <bool LOWP>
__device__ __forceinline__ f32x4 contract(const Frag& a, const Frag& b, f32x4 acc) {
  if constexpr (!LOWP) {   // small terms first (lo.lo is dropped)
    acc = mfma_planes<true>(a.p[1], b.p[0], acc);
    acc = mfma_planes<true>(a.p[0], b.p[1], acc);
  }
  return mfma_planes<true>(a.p[0], b.p[0], acc);
}
template <bool LOWP>
__device__ __forceinline__ f32x4 contract_ones(const Frag& a, const f16x8& ones, f32x4 acc) {
  if constexpr (!LOWP) acc = mfma_planes<true>(a.p[1], ones, acc);
  return mfma_planes<true>(a.p[0], ones, acc);
}

template <const auto& TABLE, int IDX, bool LOWP, class TAPE_L, class DY_L, bool BARRIER = false>
__device__ __forceinline__ void run_role(const float* __restrict__ tape, const float* __restrict__ dbuf, int blk0, int blk1,
                                         const GradPtrs& gp, int lane) {
  constexpr Role R = TABLE[IDX];
  constexpr int NS = role_slots(R);
  const int g = lane >> 4, j = lane & 15;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  // selector of the transposition: B[k = 8g + i][n = j] = 1 iff i < 4 and 4g + i == j (bf16 1.0 = 0x3f80)
  unsigned selw[4] = {0u, 0u, 0u, 0u};
  if ((j >> 2) == g) selw[(j & 3) >> 1] = (j & 1) ? 0x3f800000u : 0x00003f80u;
  const f16x8 sel = __builtin_bit_cast(f16x8, u32x4v{selw[0], selw[1], 0u, 0u});
  const f16x8 ones = __builtin_bit_cast(f16x8, u32x4v{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  f32x4 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = splat4(0.f);

  for (int blk = blk0; blk < blk1; ++blk) {
    // The four waves of a workgroup read overlapping tiles of the SAME block (x for q / k / v, the halves of d hid for the
    // quadrants of mlp0, d opre for the column ranges of mlp2): one barrier per block keeps them within a block of each
    // other, so the second and third reader hit in the L2 (4 MB per XCD = 64 KB per resident workgroup: less than one
    // block's tiles -- free-running waves drifted apart and every read went to HBM, 3.7 GB per launch for 2.3 GB of tiles).
    // Nothing is exchanged: no LDS, no fence.
    // Measured: view kernel 0.76 -> 0.70 ms; the ray kernel, whose roles are less even, 0.45 -> 0.51 ms: so only the former.
    if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    // scalar block bases + 32-bit lane offsets
    const char* tb = reinterpret_cast<const char*>(tape) + (size_t)blk * (TAPE_L::block_units * 512);
    const char* db = reinterpret_cast<const char*>(dbuf) + (size_t)blk * (DY_L::block_units * 512);
    static_for<R.n>([&](auto ji) __attribute__((always_inline)) {
      constexpr int jx = decltype(ji)::value;
      constexpr Job J = R.j[jx];
      constexpr int S0 = role_slot0(R, jx);
      constexpr int NB = J.kind == K_ROWSUM ? 0 : J.b_n;
      // a job's tiles share a storage format (all of one tensor): fp32, or bf16 in the 16-bit mode
      constexpr bool A16 = DY_L::is16(J.a_tile), B16 = NB > 0 && TAPE_L::is16(J.b_tile);
      static_assert(J.b_buf == B_TAPE, "X operands come from the tape");
      auto raw_a = [&](int tile, int c) __attribute__((always_inline)) -> RawTile<A16> {
        RawTile<A16> r;
        r.v = *reinterpret_cast<const decltype(r.v)*>(db + tile_byte_offset<DY_L>(tile, c, lane));
        return r;
      };
      auto raw_b = [&](int tile, int c) __attribute__((always_inline)) -> RawTile<B16> {
        RawTile<B16> r;
        r.v = *reinterpret_cast<const decltype(r.v)*>(tb + tile_byte_offset<TAPE_L>(tile, c, lane));
        return r;
      };
      // the X tiles of the job first (they stay in registers as fragments), the dY tiles one tile ahead of their use: the tile
      // loads are the kernel's bottleneck, the matrix work hides behind them
      RawTile<B16> rb[NB > 0 ? NB : 1][2];
#pragma unroll
      for (int i = 0; i < NB; ++i) { rb[i][0] = raw_b(J.b_tile + i, 0); rb[i][1] = raw_b(J.b_tile + i, 1); }
      RawTile<A16> ra[2][2];
      ra[0][0] = raw_a(J.a_tile, 0);
      ra[0][1] = raw_a(J.a_tile, 1);
      Frag fb[NB > 0 ? NB : 1];
#pragma unroll
      for (int i = 0; i < NB; ++i) fb[i] = to_frag<LOWP, B16>(rb[i][0], rb[i][1], sel);
      static_for<J.a_n>([&](auto aii) __attribute__((always_inline)) {
        constexpr int ai = decltype(aii)::value;
        if constexpr (ai + 1 < J.a_n) {
          ra[(ai + 1) & 1][0] = raw_a(J.a_tile + ai + 1, 0);
          ra[(ai + 1) & 1][1] = raw_a(J.a_tile + ai + 1, 1);
        }
        const Frag fa = to_frag<LOWP, A16>(ra[ai & 1][0], ra[ai & 1][1], sel);
        if constexpr (J.kind == K_FULL) {
#pragma unroll
          for (int bi = 0; bi < NB; ++bi) acc[S0 + ai * NB + bi] = contract<LOWP>(fa, fb[bi], acc[S0 + ai * NB + bi]);
        } else if constexpr (J.kind == K_DIAG) {
          acc[S0 + ai] = contract<LOWP>(fa, fb[ai], acc[S0 + ai]);
        } else {
          acc[S0 + ai] = contract_ones<LOWP>(fa, ones, acc[S0 + ai]);
        }
      });
    });
  }

  // flush: accumulator element (row m = 4g + r, column n = j) of tile (ai, bi) -> dW[feature of dY row m][feature of X row n]
  static_for<R.n>([&](auto ji) __attribute__((always_inline)) {
    constexpr int jx = decltype(ji)::value;
    constexpr Job J = R.j[jx];
    constexpr int S0 = role_slot0(R, jx);
    float* dst = gp.p[J.param];
    static_for<job_slots(J)>([&](auto si) __attribute__((always_inline)) {
      constexpr int s = decltype(si)::value;
      constexpr int nb1 = J.b_n > 0 ? J.b_n : 1;
      constexpr int ai = J.kind == K_FULL ? s / nb1 : s, bi = J.kind == K_FULL ? s % nb1 : s;
      const f32x4 a = acc[S0 + s];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = fmap(J.a_map, J.a_t0 + ai, 4 * g + r, J.a_dim);
        if (o < 0) continue;
        if constexpr (J.kind == K_FULL) {
          const int i = fmap(J.b_map, J.b_t0 + bi, j, J.b_dim);
          if (i >= 0) atomic_add_f32(dst + (size_t)o * J.ld + J.i_off + i, a[r]);
        } else if constexpr (J.kind == K_DIAG) {
          if (4 * g + r == j) atomic_add_f32(dst + o, a[r]);
        } else {
          if (j == 0) atomic_add_f32(dst + o, a[r]);
        }
      }
    });
  });
}

// grid: n_chunks x kViewTypes workgroups of 4 waves; workgroup (type, chunk) runs roles 4 type .. 4 type + 3 over the
// chunk's blocks.  Types are ordered by weight (the heaviest first) so that the long workgroups start first.
template <bool LOWP>
__global__ void __launch_bounds__(256, 2) view_wgrad_kernel(const float* __restrict__ tape, const float* __restrict__ dbuf,
                                                            int n_blocks, int n_chunks, GradPtrs gp) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifndef UFR_VW_XCD
#define UFR_VW_XCD 0   // measured round 6 (tools/dev/train_lib_ab2.sh, two runs): 0.748 vs 0.708 ms fp32 mode (slower), 0.384 vs 0.391 in the 16-bit mode: off
#endif
  int type, chunk;
  if constexpr (UFR_VW_XCD != 0) {
    // The three workgroup types of a chunk read overlapping tiles (x: q / k / v and mlp0; d hid, d opre): dispatched
    // type-major they ran a third of the launch apart and on unrelated XCDs, so every type fetched the chunk from HBM again.
    // Workgroups go round-robin over the 8 XCDs: in groups of 8 chunks x 3 types, workgroup 24 G + 8 t + c is chunk 8 G + c of
    // type t on XCD c -- a chunk's three readers share one L2 and start within 24 dispatch slots of each other.
    const int b = blockIdx.x, full = (n_chunks / 8) * 24;
    if (b < full) { type = (b % 24) / 8; chunk = (b / 24) * 8 + b % 8; }
    else { const int rem = n_chunks % 8, w = b - full; type = w / rem; chunk = (n_chunks / 8) * 8 + w % rem; }
  } else {
    type = blockIdx.x / n_chunks; chunk = blockIdx.x - type * n_chunks;
  }
  const int per = (n_blocks + n_chunks - 1) / n_chunks;
  const int blk0 = chunk * per, blk1 = min(n_blocks, blk0 + per);
  switch (__builtin_amdgcn_readfirstlane(type * 4 + wave)) {
#ifdef UFR_WG_ONLY
#define UFR_ROLE(i) case i: if constexpr (i == UFR_WG_ONLY) run_role<kViewRoles, i, LOWP, ViewTapeLayout<LOWP>, ViewGradLayout<LOWP>, true>(tape, dbuf, blk0, blk1, gp, lane); break;
#else
#define UFR_ROLE(i) case i: run_role<kViewRoles, i, LOWP, ViewTapeLayout<LOWP>, ViewGradLayout<LOWP>, true>(tape, dbuf, blk0, blk1, gp, lane); break;
#endif
    UFR_ROLE(0) UFR_ROLE(1) UFR_ROLE(2) UFR_ROLE(3) UFR_ROLE(4) UFR_ROLE(5) UFR_ROLE(6) UFR_ROLE(7)
    UFR_ROLE(8) UFR_ROLE(9) UFR_ROLE(10) UFR_ROLE(11)
#undef UFR_ROLE
    default: break;
  }
}

// ---- ray transformer: 4 workgroup types x 4 waves.  X tiles: RT_X (nat88), RT_MSG (quad-packed), RT_M, RT_HID, RT_O, RT_D1,
// RT_D2; dY tiles: DR_Q (quad-packed rows), DR_K / DR_V (one 16-slot tile per head), DR_MPRE, DR_HID, DR_OPRE, DR_D1, DR_D2, DR_SR
constexpr int kRayTypes = 4;
constexpr Role kRayRoles[kRayTypes * 4] = {
    // type 0: q (6 x 6), merge (6 x 6), k in two halves of its head tiles (4 x 6 each)
    {1, {full(DR_Q, 6, 0, FM_QUAD11, 88, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_Q, 88)}},
    {1, {full(DR_MPRE, 6, 0, FM_NAT88, 88, B_TAPE, RT_MSG, 6, 0, FM_QUAD11, 88, P_RT_MERGE, 88)}},
    {1, {full(DR_K, 4, 0, FM_HEAD11K, 88, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_K, 88)}},
    {1, {full(DR_K + 4, 4, 4, FM_HEAD11K, 88, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_K, 88)}},
    // type 1: v in two halves, mlp2 (88 x 176) in two column ranges of the hidden layer
    {1, {full(DR_V, 4, 0, FM_HEAD11K, 88, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_V, 88)}},
    {1, {full(DR_V + 4, 4, 4, FM_HEAD11K, 88, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_V, 88)}},
    {1, {full(DR_OPRE, 6, 0, FM_NAT88, 88, B_TAPE, RT_HID, 6, 0, FM_NAT, 176, P_RT_MLP2, 176)}},
    {1, {full(DR_OPRE, 6, 0, FM_NAT88, 88, B_TAPE, RT_HID + 6, 5, 6, FM_NAT, 176, P_RT_MLP2, 176)}},
    // type 2: mlp0 (176 x [x 88 | m 88]), hidden rows 0..95 against x and m, rows 96..175 against x and m
    {1, {full(DR_HID, 6, 0, FM_NAT, 176, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_MLP0, 176, 0)}},
    {1, {full(DR_HID, 6, 0, FM_NAT, 176, B_TAPE, RT_M, 6, 0, FM_NAT88, 88, P_RT_MLP0, 176, 88)}},
    {1, {full(DR_HID + 6, 5, 6, FM_NAT, 176, B_TAPE, RT_X, 6, 0, FM_NAT88, 88, P_RT_MLP0, 176, 0)}},
    {1, {full(DR_HID + 6, 5, 6, FM_NAT, 176, B_TAPE, RT_M, 6, 0, FM_NAT88, 88, P_RT_MLP0, 176, 88)}},
    // type 3: the DensityMLP and its biases (one wave; the other three idle)
    {6, {full(DR_D1, 2, 0, FM_NAT, 32, B_TAPE, RT_O, 6, 0, FM_NAT88, 88, P_DM_W0, 88), rowsum(DR_D1, 2, 32, P_DM_B0),
         full(DR_D2, 1, 0, FM_NAT, 16, B_TAPE, RT_D1, 2, 0, FM_NAT, 32, P_DM_W2, 32), rowsum(DR_D2, 1, 16, P_DM_B2),
         full(DR_SR, 1, 0, FM_NAT, 1, B_TAPE, RT_D2, 1, 0, FM_NAT, 16, P_DM_W4, 16), rowsum(DR_SR, 1, 1, P_DM_B4)}},
    {0, {}}, {0, {}}, {0, {}},
};

template <bool LOWP>
__global__ void __launch_bounds__(256, 2) ray_wgrad_kernel(const float* __restrict__ tape, const float* __restrict__ dbuf,
                                                           int n_blocks, int n_chunks, GradPtrs gp) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int type = blockIdx.x / n_chunks, chunk = blockIdx.x - type * n_chunks;
  const int per = (n_blocks + n_chunks - 1) / n_chunks;
  const int blk0 = chunk * per, blk1 = min(n_blocks, blk0 + per);
  switch (__builtin_amdgcn_readfirstlane(type * 4 + wave)) {
#define UFR_ROLE(i) case i: run_role<kRayRoles, i, LOWP, RayTapeLayout<LOWP>, RayGradLayout<LOWP>>(tape, dbuf, blk0, blk1, gp, lane); break;
    UFR_ROLE(0) UFR_ROLE(1) UFR_ROLE(2) UFR_ROLE(3) UFR_ROLE(4) UFR_ROLE(5) UFR_ROLE(6) UFR_ROLE(7)
    UFR_ROLE(8) UFR_ROLE(9) UFR_ROLE(10) UFR_ROLE(11) UFR_ROLE(12)
#undef UFR_ROLE
    default: break;
  }
}

}  // namespace wgs

hipError_t launch_ray_wgrad(const float* tape, const float* dbuf, int n_blocks, const GradPtrs& gp, bool lowp, hipStream_t s) {
  if (n_blocks <= 0) return hipErrorInvalidValue;
  int n_chunks = (n_blocks + 15) / 16;
  static const int cap = getenv("UFR_RW_CHUNKS") ? atoi(getenv("UFR_RW_CHUNKS")) : 384;
  if (n_chunks > cap) n_chunks = cap;
  const dim3 grid(n_chunks * wgs::kRayTypes), block(256);
  if (lowp) hipLaunchKernelGGL(wgs::ray_wgrad_kernel<true>, grid, block, 0, s, tape, dbuf, n_blocks, n_chunks, gp);
  else hipLaunchKernelGGL(wgs::ray_wgrad_kernel<false>, grid, block, 0, s, tape, dbuf, n_blocks, n_chunks, gp);
  return hipGetLastError();
}

hipError_t launch_view_wgrad(const float* tape, const float* dbuf, int n_blocks, const GradPtrs& gp, bool lowp, hipStream_t s) {
  if (n_blocks <= 0) return hipErrorInvalidValue;
  // a workgroup flushes ~30 k atomics per wave (0.6 GB of write traffic per launch at 512 chunks): at least 32 blocks of
  // work each, and two rounds of the 512 resident slots (341 chunks x 3 types; 512 chunks = three rounds measured 5 %
  // slower, 170 = one round as slow as 512).  UFR_VW_CHUNKS / UFR_RW_CHUNKS: development overrides.
  int n_chunks = (n_blocks + 31) / 32;
  static const int cap = getenv("UFR_VW_CHUNKS") ? atoi(getenv("UFR_VW_CHUNKS")) : 341;
  if (n_chunks > cap) n_chunks = cap;
  const dim3 grid(n_chunks * wgs::kViewTypes), block(256);
  if (lowp) hipLaunchKernelGGL(wgs::view_wgrad_kernel<true>, grid, block, 0, s, tape, dbuf, n_blocks, n_chunks, gp);
  else hipLaunchKernelGGL(wgs::view_wgrad_kernel<false>, grid, block, 0, s, tape, dbuf, n_blocks, n_chunks, gp);
  return hipGetLastError();
}

}  // namespace ufr
