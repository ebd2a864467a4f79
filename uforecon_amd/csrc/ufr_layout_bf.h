// Split-precision ("bf16x6") weight layout of the view-transformer chain.
//
// Measured on MI355X (tools/dev/mfma_valu2.*): v_mfma_f32_16x16x4_f32 does NOT overlap with VALU
// work -- it occupies the vector ALU for its 32 cycles, so every LayerNorm / attention / elu
// instruction of an fp32-MFMA kernel is lost MFMA time -- while the bf16 matrix-core MFMA
// (v_mfma_f32_16x16x32_bf16, ~17 cycles for 8x the k-depth) does overlap.  The dense layers therefore
// run on the bf16 cores with fp32-grade accuracy: every fp32 operand is split EXACTLY into three bf16
// planes (hi + mid + lo = x bit for bit: 3 x 8 mantissa bits), and a product is the six plane pairs
// with i + j <= 4, accumulated in fp32 by the MFMA:
//     w*x ~= w_mid*x_mid + w_hi*x_lo + w_lo*x_hi + w_hi*x_mid + w_mid*x_hi + w_hi*x_hi
// (dropped terms <= 2^-24 |w||x|; measured rms error 9e-8 vs 2.3e-7 for an fp32 fma GEMM, K = 160).
// 6 MFMAs of ~17 cycles replace 8 fp32 MFMAs of 32 cycles: 2.5x on the matrix work, and the VALU
// phases now hide behind it.
//
// A fragment here = (panel, out tile `to`, plane p): 64 lanes x 8 bf16 = 1 KiB; lane l holds
// W_p[row(to, l&15)][k = 8*(l>>4) + i], i = 0..7, where the 32 k-slots of a panel are the two
// accumulator tiles (2s, 2s+1) of the producing layer: slot 8g+i <-> tile 2s + (i>>2), feature
// col_map(tile, g, i&3) -- exactly what a lane holds after splitting its two fp32 accumulator tiles.
// A PANEL is one k-step (32 input features) of one matrix: all its out tiles x 3 planes, stored in
// consumption order; the stream is the sequence of panels the kernel walks.
#pragma once
#include "ufr_layout.h"

namespace ufr {

constexpr int kPlanes = 3;
constexpr int kBfChunkFrags = 24;  // 24 KiB chunks: 8 (tile,3 planes) triples; splits evenly over 4 or 8 waves

struct Panel { int mat, s; };

// consumption order of the view-transformer chain: q and k interleaved per k-step (x is split once per step),
// then v (after the attention scores are reduced to L numbers per head, q and k are dead: fewer live registers)
constexpr int kVtPanels = 3 * 3 + 3 + 5 + 5 + 3 + 1 + 1;
__host__ __device__ constexpr Panel vt_panel(int i) {
  if (i < 6) return {i % 2 == 0 ? M_VT_Q : M_VT_K, i / 2};
  if (i < 9) return {M_VT_V, i - 6};
  i -= 9;
  if (i < 3) return {M_VT_MERGE, i};
  i -= 3;
  if (i < 5) return {M_VT_MLP0, i};
  i -= 5;
  if (i < 5) return {M_VT_MLP2, i};
  i -= 5;
  if (i < 3) return {M_RW0, i};
  i -= 3;
  return {i == 0 ? M_RW2 : M_RW4, 0};
}
__host__ __device__ constexpr int ksteps(int m) { return (mat_desc(m).n_in + 1) / 2; }
__host__ __device__ constexpr int panel_frags(int i) { return mat_desc(vt_panel(i).mat).n_out * kPlanes; }
__host__ __device__ constexpr int panel_start(int i) {  // first fragment of panel i in the stream
  int o = 0;
  for (int j = 0; j < i; ++j) o += panel_frags(j);
  return o;
}
__host__ __device__ constexpr int panel_index(int m, int s) {
  for (int i = 0; i < kVtPanels; ++i)
    if (vt_panel(i).mat == m && vt_panel(i).s == s) return i;
  return -1;
}
constexpr int kVtbFrags = panel_start(kVtPanels);
constexpr int kVtbChunksRaw = (kVtbFrags + kBfChunkFrags - 1) / kBfChunkFrags;
constexpr int kVtbChunks = kVtbChunksRaw + (kVtbChunksRaw & 1);       // even: static LDS slot parity
constexpr int kVtbFragsPadded = kVtbChunks * kBfChunkFrags;
constexpr int kVtbHalfwords = kVtbFragsPadded * 512;                  // bf16 elements in the region
constexpr int kVtbBytes = kVtbFragsPadded * 1024;

// input feature of k-slot (g, i) of panel step s (or -1): accumulator tiles 2s and 2s+1 of the producer
__host__ __device__ constexpr int bf_col(int m, int s, int g, int i) {
  const MatDesc d = mat_desc(m);
  const int tile = 2 * s + (i >> 2);
  return tile < d.n_in ? col_map(d.cm, tile, g, i & 3, d.in_dim) : -1;
}

// source of halfword h of the bf16 region: parameter, element, plane (param -1 = zero)
__host__ __device__ inline void plan_entry_bf(int h, int* param, int* elem, int* plane) {
  *param = -1; *elem = 0; *plane = 0;
  int f = h >> 9;                 // fragment
  const int lane = (h >> 3) & 63, i = h & 7;
  if (f >= kVtbFrags) return;     // tail padding
  int pi = 0;
  while (f >= panel_frags(pi)) { f -= panel_frags(pi); ++pi; }
  const Panel p = vt_panel(pi);
  const MatDesc d = mat_desc(p.mat);
  const int to = f / kPlanes;
  *plane = f % kPlanes;
  const int row = row_map(d.rm, to, lane & 15, d.out_dim);
  const int col = bf_col(p.mat, p.s, lane >> 4, i);
  if (row >= 0 && col >= 0) { *param = d.param; *elem = row * d.k_raw + col; }
}

}  // namespace ufr
