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

// ---- view transformer tape (A1 = view_transformer_kernel<.., TAPE>)
enum ViewTape : int {
  TV_X = 0,        // 5  token inputs                                   natural rows 16t + 4g + r
  TV_Q = 5,        // 5  Q' = elu(q) + 1                                 slot20 rows (ROW_SLOT20)
  TV_K = 10,       // 5  K' = elu(k) + 1                                 slot20
  TV_V = 15,       // 5  values / v_length                               slot20
  TV_MSG = 20,     // 5  attention message                               slot20
  TV_XH1 = 25,     // 5  LayerNorm1 normalised input                     natural
  TV_M = 30,       // 5  LayerNorm1 output                               natural
  TV_HID = 35,     // 10 relu(mlp0)                                      natural
  TV_XH2 = 45,     // 5  LayerNorm2 normalised input                     natural
  TV_Y = 50,       // 6  layer output y (5) | dir (tile 5: register 0 of lane groups 0..2 = COL_RW0)
  TV_H1 = 56,      // 1  relu(rw0)
  TV_H2 = 57,      // 1  relu(rw2), rows 0..7
  TV_MISC = 58,    // 1  per column: {rstd1, rstd2, logit (before the mask), 0} in every lane group
  TV_COUNT = 59
};
// ---- view transformer cotangents (A2 = view_dgrad_kernel)
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
  DV_YLN = 38,     // 5  d (LayerNorm2 output) = d y              -> norm2 gamma / beta; also the residual part of d x
  DV_MLN = 43,     // 5  d (LayerNorm1 output) = d cat[80..159]   -> norm1 gamma / beta
  DV_X0 = 48,      // 5  d x of the view-token columns (zero elsewhere) -> view token; scratch for d cat[0..79] before that
  DV_COUNT = 53
};

__device__ __forceinline__ size_t tile_offset(int n_tiles, size_t block, int tile, int c) {
  return ((block * n_tiles + tile) * kBlockCols + c) * kTileFloats;
}

}  // namespace ufr
