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
  TV_MISC = 58,    // 1  per column: {rstd1, rstd2, logit (before the mask), 0} in every lane group
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
  DV_SCR = 38,     // 5  the data-gradient kernel's own scratch: the parts of d x that wait for the projections' share (fp32)
  DV_COUNT = 43
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
