// Tile buffers of the streaming backward (round 4): what the three kernels of a transformer's backward hand to each other
// through HBM.
//
//   A1  the forward kernel again, with TAPE: every activation the backward needs, written as it is produced
//   A2  the data-gradient chain (view_dgrad.hip): reads the tape, writes the layer-output cotangents ("dY tiles")
//   B   the weight-gradient contraction over tokens (wgrad_stream.hip): dW = sum_t dY[:, t] X[:, t]^T for every matrix
//
// A TILE is one fp32 accumulator tile of the chained-MFMA layout -- 16 features x 16 token columns, lane l = (g, j) holding
// features 4g..4g+3 (register r) of column j -- stored exactly as the lanes hold it: 64 lanes x float4 = 1 KiB, one fully
// coalesced wave store / load.  A BLOCK is what one wave iteration of A1 / A2 covers: kBlockCols = 2 column tiles = 32
// token columns; a buffer is [block][tile][column tile][lane] float4.  Which feature a (tile, lane group, register) holds
// is the producing layer's row map (ufr_layout.h): kernel B carries the maps in its job tables and undoes them when it
// flushes.  Columns that hold no token (padding of a column tile, groups past the end) carry zeros in every dY tile, so
// they contribute nothing; X tiles there are finite.
#pragma once
#include "ufr_device.h"

namespace ufr {

constexpr int kBlockCols = 2;                      // column tiles per block (= UFR_VT_C of the forward kernel)
constexpr int kTileFloats = 256;                   // 64 lanes x float4

// 16-bit mode (UFR_PRECISION_16BIT): the tiles that only feed the weight-gradient contraction -- which rounds its operands
// to one bf16 plane anyway -- are STORED as bf16 (lane: 4 x bf16 = 8 bytes, 512 B per tile): the three kernels are bound by
// the bytes they move.  What the data-gradient chain reads back with full precision (Q', K', V, the LayerNorm inputs) stays
// fp32.  A buffer's layout is therefore a per-tile table: offsets in units of 512 B within a block.
// ---- view transformer tape (A1 = view_transformer_kernel<.., TAPE>)
enum ViewTape : int {
  TV_X = 0,        // 5  token inputs                                   natural rows 16t + 4g + r        (bf16 in 16-bit mode)
  TV_Q = 5,        // 5  Q' = elu(q) + 1                                 slot20 rows (ROW_SLOT20)
  TV_K = 10,       // 5  K' = elu(k) + 1                                 slot20
  TV_V = 15,       // 5  values / v_length                               slot20
  TV_MSG = 20,     // 5  attention message                               slot20                           (bf16)
  TV_XH1 = 25,     // 5  LayerNorm1 normalised input                     natural
  TV_M = 30,       // 5  LayerNorm1 output                               natural                          (bf16)
  TV_HID = 35,     // 10 relu(mlp0)                                      natural                          (bf16)
  TV_XH2 = 45,     // 5  LayerNorm2 normalised input                     natural
  TV_Y = 50,       // 6  layer output y (5) | dir (tile 5: register 0 of lane groups 0..2 = COL_RW0)     (bf16)
  TV_H1 = 56,      // 1  relu(rw0)                                                                        (bf16)
  TV_H2 = 57,      // 1  relu(rw2), rows 0..7                                                             (bf16)
  TV_MISC = 58,    // 1  {ReLU bits 0..31, bits 32..47, a, b}: bit 4 t + r = hidden unit (t, r) of this lane > 0 for t < 10, then h1
                   //    (40..43), h2 (44..47); (a, b) = (rstd1, rstd2) of the column in lane group 0, (logit before the mask, 0)
                   //    in lane group 1 -- the data-gradient kernel reads masks, not the hidden activations
  TV_COUNT = 59
};
// ---- view transformer cotangents (A2 = view_dgrad_kernel); all bf16 in 16-bit mode except the scratch tiles
enum ViewGrad : int {
  DV_Q = 0,        // 5  d q (after elu')      slot20
  DV_K = 5,        // 5  d k                   slot20
  DV_V = 10,       // 5  d v                   slot20
  DV_MPRE = 15,    // 5  d (merge output)      natural
  DV_HID = 20,     // 10 d (mlp0 output, after the ReLU mask)
  DV_OPRE = 30,    // 5  d (mlp2 output)
  DV_H1 = 35,      // 1  d (rw0 output, after the mask)
  DV_H2 = 36,      // 1  d (rw2 output, after the mask), rows 0..7
  DV_LG = 37,      // 1  d logit in row 0
  DV_SCR = 38,     // 5  the data-gradient kernel's own scratch: the parts of d x that wait for the projections' share (fp32):
  DV_SCR2 = 43,    // 5  d y (residual) in the first set, the x half of d cat in the second
  DV_COUNT = 48
};

template <bool LOWP>
struct ViewTapeLayout {
  static constexpr int count = TV_COUNT;
  static constexpr bool is16(int t) {
    return LOWP && ((t >= TV_X && t < TV_Q) || (t >= TV_MSG && t < TV_XH1) || (t >= TV_M && t < TV_XH2) || (t >= TV_Y && t < TV_MISC));
  }
  static constexpr int off(int t) {   // 512-byte units from the block's start
    int o = 0;
    for (int u = 0; u < t; ++u) o += is16(u) ? 2 : 4;
    return o;
  }
  static constexpr int block_units = off(TV_COUNT);
};
template <bool LOWP>
struct ViewGradLayout {
  static constexpr int count = DV_COUNT;
  static constexpr bool is16(int t) { return LOWP && t < DV_SCR; }
  static constexpr int off(int t) {
    int o = 0;
    for (int u = 0; u < t; ++u) o += is16(u) ? 2 : 4;
    return o;
  }
  static constexpr int block_units = off(DV_COUNT);
};

// ---- ray transformer tape (A1 = ray_transformer_kernel<.., TAPE>): per 16-token tile of a ray; a block = two consecutive
// tiles of the ray (block index = ray * ceil(n_tiles / 2) + tile / 2, column tile = tile & 1; an odd tile count leaves a
// padding tile: finite X tiles, zero dY tiles)
enum RayTape : int {
  RT_X = 0,        // 6  [token-0 feature 80 | order PE 8]                nat88 rows                       (bf16 in 16-bit mode)
  RT_Q = 6,        // 6  Q' = elu(q) + 1                                   quad-packed rows (ROW_QUAD11)
  RT_MSG = 12,     // 6  attention message                                 quad-packed                      (bf16)
  RT_ZS = 18,      // 2  Z * SN of (token, head): heads 0..3 | 4..7, the same in every lane group
  RT_XH1 = 20,     // 6  LayerNorm1 normalised input                       nat88
  RT_M = 26,       // 6  LayerNorm1 output                                 nat88                            (bf16)
  RT_HID = 32,     // 11 relu(mlp0)                                        natural (176)                    (bf16)
  RT_XH2 = 43,     // 6  LayerNorm2 normalised input                       nat88
  RT_O = 49,       // 6  layer output                                      nat88                            (bf16)
  RT_D1 = 55,      // 2  relu(dm0)                                         natural (32)                     (bf16)
  RT_D2 = 57,      // 1  relu(dm2)                                         natural (16)                     (bf16)
  RT_MISC = 58,    // 1  per column: {rstd1, rstd2, ReLU bits 0..31, bits 32..59}: bit 4 t + r = hidden unit (t, r) > 0 for
                   //    t < 11, then d1 (tiles 11, 12), d2 (tile 13) -- this lane's rows of the column
  RT_COUNT = 59
};
constexpr int kRayStateTiles = 16;   // per ray: KV_h (8 heads) then KV_h^T (8), fp32 16 x 16 tiles in accumulator layout
// ---- ray transformer cotangents (A2 = ray_dgrad_kernel)
enum RayGrad : int {
  DR_Q = 0,        // 6  d q     quad-packed
  DR_K = 6,        // 8  d k     one 16-slot tile per head (ROW_HEAD11K as rows)
  DR_V = 14,       // 8  d v
  DR_MPRE = 22,    // 6  d (merge output)   nat88
  DR_HID = 28,     // 11 d (mlp0 output, after the ReLU mask)
  DR_OPRE = 39,    // 6  d (mlp2 output)    nat88
  DR_D1 = 45,      // 2  d (dm0 output, masked)
  DR_D2 = 47,      // 1  d (dm2 output, masked)
  DR_SR = 48,      // 1  d srdf in row 0
  DR_SCR = 49,     // 6  scratch: the d x parts of sweep 1 waiting for sweep 2 (fp32): d o (residual),
  DR_SCR2 = 55,    // 6  the x half of d cat,
  DR_SCR3 = 61,    // 6  the query projection's share
  DR_COUNT = 67
};
template <bool LOWP>
struct RayTapeLayout {
  static constexpr int count = RT_COUNT;
  static constexpr bool is16(int t) {
    return LOWP && ((t >= RT_X && t < RT_Q) || (t >= RT_MSG && t < RT_ZS) || (t >= RT_M && t < RT_XH2) || (t >= RT_O && t < RT_MISC));
  }
  static constexpr int off(int t) {
    int o = 0;
    for (int u = 0; u < t; ++u) o += is16(u) ? 2 : 4;
    return o;
  }
  static constexpr int block_units = off(RT_COUNT);
};
template <bool LOWP>
struct RayGradLayout {
  static constexpr int count = DR_COUNT;
  static constexpr bool is16(int t) { return LOWP && t < DR_SCR; }
  static constexpr int off(int t) {
    int o = 0;
    for (int u = 0; u < t; ++u) o += is16(u) ? 2 : 4;
    return o;
  }
  static constexpr int block_units = off(DR_COUNT);
};

typedef unsigned u32x2_tile __attribute__((ext_vector_type(2)));
// byte offset of (tile, column tile c, lane) inside a block
template <class LAYOUT>
__host__ __device__ constexpr unsigned tile_byte_offset(int tile, int c, int lane) {
  return LAYOUT::is16(tile) ? (unsigned)(LAYOUT::off(tile) * 512 + c * 512 + lane * 8)
                            : (unsigned)(LAYOUT::off(tile) * 512 + c * 1024 + lane * 16);
}
__device__ __forceinline__ u32x2_tile pack_tile_bf16(const f32x4& v) {   // round to nearest even
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  return u32x2_tile{__builtin_bit_cast(unsigned, __builtin_convertvector(f2{v[0], v[1]}, bf2)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v[2], v[3]}, bf2))};
}
__device__ __forceinline__ f32x4 unpack_tile_bf16(const u32x2_tile& w) {
  return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
               __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
}
// store / load one tile of a block whose base is `blk` (bytes)
template <class LAYOUT>
__device__ __forceinline__ void tile_store(char* blk, int tile, int c, int lane, const f32x4& v) {
  char* p = blk + tile_byte_offset<LAYOUT>(tile, c, lane);
  if (LAYOUT::is16(tile)) *reinterpret_cast<u32x2_tile*>(p) = pack_tile_bf16(v);
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <class LAYOUT>
__device__ __forceinline__ f32x4 tile_load(const char* blk, int tile, int c, int lane) {
  const char* p = blk + tile_byte_offset<LAYOUT>(tile, c, lane);
  if (LAYOUT::is16(tile)) return unpack_tile_bf16(*reinterpret_cast<const u32x2_tile*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}

}  // namespace ufr
