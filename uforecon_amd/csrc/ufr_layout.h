// Packed-weight blob layout shared by the host plan builder (ufr_api.hip) and the MFMA kernels.
//
// A dense layer  y = W x  (W row-major [out][in], the nn.Linear layout) is executed as chained
// v_mfma_f32_16x16x4_f32 with the TOKENS as the 16 MFMA columns.  The accumulator tile of one
// layer (lane l holds rows 4*(l>>4)+r, r=0..3, of column l&15) is fed unchanged as the B operand
// of the next layer: MFMA step r then consumes "input features" {4g+r : g=l>>4}, so the A
// fragment of (out tile t, in tile t', step r) must be  W[row(t, l&15)][col(t', l>>4, r)].
// The four steps of an in-tile are one float4 per lane; a FRAGMENT is 64 lanes x float4 = 1 KiB.
// row()/col() also carry the feature permutations/paddings that make the attention heads land in
// convenient lanes (see the maps below).
//
// Fragments are stored in CONSUMPTION ORDER, one contiguous STREAM per kernel sweep, so a workgroup
// can pull them through LDS in fixed-size chunks with lane-linear LDS-DMA (global_load_lds) and all
// its waves read each fragment from LDS instead of L2 (weight_stream.h).
#pragma once
#include <stdint.h>

namespace ufr {

enum RowMap : int { ROW_NAT = 0, ROW_SLOT20, ROW_HEAD11K, ROW_NAT88, ROW_QUAD11, ROW_CAT88 };
enum ColMap : int { COL_NAT = 0, COL_SLOT20, COL_NAT88, COL_QUAD11, COL_RW0, COL_CAT88, COL_HEAD11K };

// The ray transformer's 8 heads of 11 dims (ray_transformer.hip).  In a 16-slot head tile, slot i = 4g + r holds head
// dim 3g + r for r < 3 (slots with r = 3, and slot 14, are padding): the k-index of the per-head fp32 MFMAs is the lane
// group g, so MFMA r of a head contracts dims {r, 3 + r, 6 + r, 9 + r} and THREE of them cover the 11 dims.
//   ROW_HEAD11K  K and V (swapped operands: a head tile's column j = slot j): one 16-column tile per head.
//   ROW_QUAD11   Q: only the 3 live registers of a head are computed -- "quad" 3h + q (the 4 rows one register index
//                contributes to a tile, one per lane group) is register (3h + q) & 3 of tile (3h + q) >> 2: 24 quads =
//                6 tiles instead of 8, addressed at compile time.
//   COL_QUAD11   merge: its input registers are the message's live registers in the same quad order (6 input tiles = 3
//                k-steps of the 16-bit MFMA instead of 4).
__host__ __device__ constexpr int head11_slot(int i) {   // head dim of slot i of a head tile, or -1
  return ((i & 3) < 3 && 3 * (i >> 2) + (i & 3) < 11) ? 3 * (i >> 2) + (i & 3) : -1;
}
__host__ __device__ constexpr int quad11(int t, int g, int r) {   // feature of register r, lane group g of quad-packed tile t
  return 3 * g + (4 * t + r) % 3 < 11 ? 11 * ((4 * t + r) / 3) + 3 * g + (4 * t + r) % 3 : -1;
}

// nat88: 88 features in 6 tiles; tile 5 keeps its 8 real features in registers r<2 of every lane
// group (feature 80+2g+r) so only 2 of its 4 MFMA steps are real.
__host__ __device__ constexpr int nat88(int t, int g, int r) {
  return t < 5 ? 16 * t + 4 * g + r : (r < 2 ? 80 + 2 * g + r : -1);
}
__host__ __device__ constexpr int row_map(int rm, int t, int i, int out_dim) {
  int g = i >> 2, r = i & 3, v = -1;
  switch (rm) {
    case ROW_NAT: v = 16 * t + i; break;
    case ROW_SLOT20: v = 20 * g + 4 * t + r; break;          // lane group g owns heads 2g,2g+1 (10 dims each)
    case ROW_HEAD11K: v = head11_slot(i) >= 0 ? 11 * t + head11_slot(i) : -1; break;
    case ROW_NAT88: v = nat88(t, g, r); break;
    case ROW_QUAD11: v = quad11(t, g, r); break;
    case ROW_CAT88: v = t < 6 ? nat88(t, g, r) : (nat88(t - 6, g, r) < 0 ? -1 : 88 + nat88(t - 6, g, r)); break;   // [x 88 | m 88]
  }
  return (v >= 0 && v < out_dim) ? v : -1;
}
__host__ __device__ constexpr int col_map(int cm, int t, int g, int r, int in_dim) {
  int v = -1;
  switch (cm) {
    case COL_NAT: v = 16 * t + 4 * g + r; break;
    case COL_SLOT20: v = 20 * g + 4 * t + r; break;
    case COL_NAT88: v = nat88(t, g, r); break;
    case COL_QUAD11: v = quad11(t, g, r); break;
    case COL_RW0: v = t < 5 ? 16 * t + 4 * g + r : ((r == 0 && g < 3) ? 80 + g : -1); break;  // [token 80 | dir 3]
    case COL_CAT88: v = t < 6 ? nat88(t, g, r) : (nat88(t - 6, g, r) < 0 ? -1 : 88 + nat88(t - 6, g, r)); break;
    case COL_HEAD11K: v = head11_slot(4 * g + r) >= 0 ? 11 * t + head11_slot(4 * g + r) : -1; break;   // head tile t, slot 4g + r
  }
  return (v >= 0 && v < in_dim) ? v : -1;
}

// index of each parameter inside ufr_raw_weights viewed as an array of const float*
enum Param : int {
  P_PS_W0 = 0, P_PS_B0, P_PS_W2, P_PS_B2, P_PS_W4, P_PS_B4,
  P_VT_Q, P_VT_K, P_VT_V, P_VT_MERGE, P_VT_MLP0, P_VT_MLP2, P_VT_N1W, P_VT_N1B, P_VT_N2W, P_VT_N2B,
  P_RT_Q, P_RT_K, P_RT_V, P_RT_MERGE, P_RT_MLP0, P_RT_MLP2, P_RT_N1W, P_RT_N1B, P_RT_N2W, P_RT_N2B,
  P_DM_W0, P_DM_B0, P_DM_W2, P_DM_B2, P_DM_W4, P_DM_B4,
  P_RW_W0, P_RW_B0, P_RW_W2, P_RW_B2, P_RW_W4, P_RW_B4,
  P_VIEW_TOKEN, P_VARIANCE, P_COUNT
};

// trans: the matrix is used TRANSPOSED (data-gradient chains of the backward kernels): out = W^T in, i.e. element
// (row, col) of the operand is raw[col * k_raw + row]; out_dim / in_dim are those of the transposed product.
struct MatDesc { int param, k_raw, n_out, n_in, rm, cm, out_dim, in_dim, trans; };
struct VecDesc { int param, n_tiles, rm, dim; };

enum Mat : int {
  M_VT_Q = 0, M_VT_K, M_VT_V, M_VT_MERGE, M_VT_MLP0, M_VT_MLP2,
  M_RT_Q, M_RT_K, M_RT_V, M_RT_MERGE, M_RT_MLP0, M_RT_MLP2,
  M_DM0, M_DM2, M_DM4, M_RW0, M_RW2, M_RW4, M_COUNT,
  // transposed operands of the data-gradient chain (view_dgrad.hip); not part of the forward streams
  M_RW2T = M_COUNT, M_RW0T, M_VT_MLP2T, M_VT_MLP0T, M_VT_MERGET, M_VT_QT, M_VT_KT, M_VT_VT,
  // ... and of the ray transformer's (ray_dgrad.hip)
  M_DM2T, M_DM0T, M_RT_MLP2T, M_RT_MLP0T, M_RT_MERGET, M_RT_QT, M_RT_KT, M_RT_VT, M_ALL_COUNT
};
enum Vec : int {
  V_VT_N1W = 0, V_VT_N1B, V_VT_N2W, V_VT_N2B, V_RT_N1W, V_RT_N1B, V_RT_N2W, V_RT_N2B,
  V_DM_B0, V_DM_B2, V_DM_B4, V_RW_B0, V_RW_B2, V_RW_B4, V_VIEW_TOKEN, V_RW_W4, V_DM_W4, V_COUNT
};

__host__ __device__ constexpr MatDesc mat_desc(int m) {
  switch (m) {
    case M_VT_Q: return {P_VT_Q, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80};
    case M_VT_K: return {P_VT_K, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80};
    case M_VT_V: return {P_VT_V, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80};
    case M_VT_MERGE: return {P_VT_MERGE, 80, 5, 5, ROW_NAT, COL_SLOT20, 80, 80};
    case M_VT_MLP0: return {P_VT_MLP0, 160, 10, 10, ROW_NAT, COL_NAT, 160, 160};
    case M_VT_MLP2: return {P_VT_MLP2, 160, 5, 10, ROW_NAT, COL_NAT, 80, 160};
    case M_RT_Q: return {P_RT_Q, 88, 6, 6, ROW_QUAD11, COL_NAT88, 88, 88};
    case M_RT_K: return {P_RT_K, 88, 8, 6, ROW_HEAD11K, COL_NAT88, 88, 88};
    case M_RT_V: return {P_RT_V, 88, 8, 6, ROW_HEAD11K, COL_NAT88, 88, 88};
    case M_RT_MERGE: return {P_RT_MERGE, 88, 6, 6, ROW_NAT88, COL_QUAD11, 88, 88};
    case M_RT_MLP0: return {P_RT_MLP0, 176, 11, 12, ROW_NAT, COL_CAT88, 176, 176};
    case M_RT_MLP2: return {P_RT_MLP2, 176, 6, 11, ROW_NAT88, COL_NAT, 88, 176};
    case M_DM0: return {P_DM_W0, 88, 2, 6, ROW_NAT, COL_NAT88, 32, 88};
    case M_DM2: return {P_DM_W2, 32, 1, 2, ROW_NAT, COL_NAT, 16, 32};
    case M_DM4: return {P_DM_W4, 16, 1, 1, ROW_NAT, COL_NAT, 1, 16};
    case M_RW0: return {P_RW_W0, 83, 1, 6, ROW_NAT, COL_RW0, 16, 83};
    case M_RW2: return {P_RW_W2, 16, 1, 1, ROW_NAT, COL_NAT, 8, 16};
    case M_RW4: return {P_RW_W4, 8, 1, 1, ROW_NAT, COL_NAT, 1, 8};
    // d in = W^T d out: rows follow the layout of the layer's INPUT tiles, columns that of its OUTPUT tiles
    case M_RW2T: return {P_RW_W2, 16, 1, 1, ROW_NAT, COL_NAT, 16, 8, 1};
    case M_RW0T: return {P_RW_W0, 83, 5, 1, ROW_NAT, COL_NAT, 80, 16, 1};          // the 80 feature rows (dir has no consumer)
    case M_VT_MLP2T: return {P_VT_MLP2, 160, 10, 5, ROW_NAT, COL_NAT, 160, 80, 1};
    case M_VT_MLP0T: return {P_VT_MLP0, 160, 10, 10, ROW_NAT, COL_NAT, 160, 160, 1};
    case M_VT_MERGET: return {P_VT_MERGE, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80, 1};
    case M_VT_QT: return {P_VT_Q, 80, 5, 5, ROW_NAT, COL_SLOT20, 80, 80, 1};
    case M_VT_KT: return {P_VT_K, 80, 5, 5, ROW_NAT, COL_SLOT20, 80, 80, 1};
    case M_VT_VT: return {P_VT_V, 80, 5, 5, ROW_NAT, COL_SLOT20, 80, 80, 1};
    case M_DM2T: return {P_DM_W2, 32, 2, 1, ROW_NAT, COL_NAT, 32, 16, 1};
    case M_DM0T: return {P_DM_W0, 88, 6, 2, ROW_NAT88, COL_NAT, 88, 32, 1};
    case M_RT_MLP2T: return {P_RT_MLP2, 176, 11, 6, ROW_NAT, COL_NAT88, 176, 88, 1};
    case M_RT_MLP0T: return {P_RT_MLP0, 176, 12, 11, ROW_CAT88, COL_NAT, 176, 176, 1};
    case M_RT_MERGET: return {P_RT_MERGE, 88, 6, 6, ROW_QUAD11, COL_NAT88, 88, 88, 1};   // d msg in the quad-packed layout
    case M_RT_QT: return {P_RT_Q, 88, 6, 6, ROW_NAT88, COL_QUAD11, 88, 88, 1};
    case M_RT_KT: return {P_RT_K, 88, 6, 8, ROW_NAT88, COL_HEAD11K, 88, 88, 1};          // d k / d v: one 16-slot tile per head
    case M_RT_VT: return {P_RT_V, 88, 6, 8, ROW_NAT88, COL_HEAD11K, 88, 88, 1};
  }
  return {0, 0, 0, 0, 0, 0, 0, 0, 0};
}
__host__ __device__ constexpr VecDesc vec_desc(int v) {
  switch (v) {
    case V_VT_N1W: return {P_VT_N1W, 5, ROW_NAT, 80};
    case V_VT_N1B: return {P_VT_N1B, 5, ROW_NAT, 80};
    case V_VT_N2W: return {P_VT_N2W, 5, ROW_NAT, 80};
    case V_VT_N2B: return {P_VT_N2B, 5, ROW_NAT, 80};
    case V_RT_N1W: return {P_RT_N1W, 6, ROW_NAT88, 88};
    case V_RT_N1B: return {P_RT_N1B, 6, ROW_NAT88, 88};
    case V_RT_N2W: return {P_RT_N2W, 6, ROW_NAT88, 88};
    case V_RT_N2B: return {P_RT_N2B, 6, ROW_NAT88, 88};
    case V_DM_B0: return {P_DM_B0, 2, ROW_NAT, 32};
    case V_DM_B2: return {P_DM_B2, 1, ROW_NAT, 16};
    case V_DM_B4: return {P_DM_B4, 1, ROW_NAT, 1};
    case V_RW_B0: return {P_RW_B0, 1, ROW_NAT, 16};
    case V_RW_B2: return {P_RW_B2, 1, ROW_NAT, 8};
    case V_RW_B4: return {P_RW_B4, 1, ROW_NAT, 1};
    case V_VIEW_TOKEN: return {P_VIEW_TOKEN, 5, ROW_NAT, 80};
    case V_RW_W4: return {P_RW_W4, 1, ROW_NAT, 8};     // the last radiance-MLP layer as a vector (view_dgrad.hip)
    case V_DM_W4: return {P_DM_W4, 1, ROW_NAT, 16};    // the last DensityMLP layer as a vector (ray_dgrad.hip)
  }
  return {0, 0, 0, 0};
}

// ------------------------------------------------------------------ weight streams
// S_VT : the view-transformer kernel's per-iteration layer chain
// S_RT1: ray transformer sweep 1 (K, V per column tile)      S_RT2: sweep 2 (Q .. DensityMLP)
enum Stream : int { S_VT = 0, S_RT1, S_RT2, S_COUNT };
constexpr int kChunkMaxFrags = 32;  // LDS slot = 32 KiB; two slots per workgroup

__host__ __device__ constexpr int stream_len(int s) { return s == S_VT ? 9 : (s == S_RT1 ? 2 : 7); }
#ifndef UFR_VT_OT
#define UFR_VT_OT 1
#endif
__host__ __device__ constexpr int stream_ot(int s) { return s == S_VT ? UFR_VT_OT : 2; }  // out tiles interleaved per stage
__host__ __device__ constexpr int stream_chunks(int s) { return s == S_RT1 ? 4 : 10; }  // even: static slot parity
__host__ __device__ constexpr int stream_mat(int s, int i) {
  switch (s) {
    case S_VT: {
      constexpr int m[9] = {M_VT_Q, M_VT_K, M_VT_V, M_VT_MERGE, M_VT_MLP0, M_VT_MLP2, M_RW0, M_RW2, M_RW4};
      return m[i];
    }
    case S_RT1: return i == 0 ? M_RT_K : M_RT_V;
    default: {
      constexpr int m[7] = {M_RT_Q, M_RT_MERGE, M_RT_MLP0, M_RT_MLP2, M_DM0, M_DM2, M_DM4};
      return m[i];
    }
  }
}
__host__ __device__ constexpr int mat_frags(int m) { return mat_desc(m).n_out * mat_desc(m).n_in; }
// first fragment of the i-th matrix of stream s
__host__ __device__ constexpr int stream_mat_start(int s, int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += mat_frags(stream_mat(s, j));
  return o;
}
__host__ __device__ constexpr int stream_frags(int s) { return stream_mat_start(s, stream_len(s)); }
// streams are zero-padded to a multiple of 4 fragments so that chunks split evenly over the 4 fetching waves
constexpr int kFetchSplit = 8;  // chunk sizes are multiples of this: 4- and 8-wave workgroups split a fetch evenly
__host__ __device__ constexpr int stream_frags_padded(int s) { return (stream_frags(s) + kFetchSplit - 1) / kFetchSplit * kFetchSplit; }
__host__ __device__ constexpr int stream_base_floats(int s) {
  int o = 0;
  for (int t = 0; t < s; ++t) o += stream_frags_padded(t) * 256;
  return o;
}
// fragment (out tile `to`, in tile `ti`) of a matrix whose stages interleave OT out tiles:
// order = group-major, then in tile, then the tile inside the group
__host__ __device__ constexpr int frag_in_mat(int m, int ot, int to, int ti) {
  const MatDesc d = mat_desc(m);
  const int gi = to / ot, o = to % ot;
  const int no = (d.n_out - gi * ot) < ot ? (d.n_out - gi * ot) : ot;
  return gi * ot * d.n_in + ti * no + o;
}
// chunk boundaries (in fragments): nearly equal, multiples of 4 (one quarter per fetching wave; never
// splits the two fragments of an OT=2 stage)
__host__ __device__ constexpr int chunk_begin(int s, int c) {
  return c >= stream_chunks(s) ? stream_frags_padded(s)
                               : ((c * stream_frags_padded(s) / stream_chunks(s)) / kFetchSplit) * kFetchSplit;
}
__host__ __device__ constexpr int chunk_of(int s, int f) {
  int c = 0;
  while (c + 1 < stream_chunks(s) && chunk_begin(s, c + 1) <= f) ++c;
  return c;
}
__host__ __device__ constexpr bool chunks_fit(int s) {
  for (int c = 0; c < stream_chunks(s); ++c)
    if (chunk_begin(s, c + 1) - chunk_begin(s, c) > kChunkMaxFrags) return false;
  return true;
}
static_assert(chunks_fit(S_VT) && chunks_fit(S_RT1) && chunks_fit(S_RT2), "a weight chunk exceeds the LDS slot");

// where matrix m lives: (stream, index in stream)
__host__ __device__ constexpr int mat_stream(int m) {
  for (int s = 0; s < S_COUNT; ++s)
    for (int i = 0; i < stream_len(s); ++i)
      if (stream_mat(s, i) == m) return s;
  return -1;
}
__host__ __device__ constexpr int mat_stream_index(int m) {
  const int s = mat_stream(m);
  for (int i = 0; i < stream_len(s); ++i)
    if (stream_mat(s, i) == m) return i;
  return -1;
}

// sizes / offsets in floats
__host__ __device__ constexpr int vec_floats(int v) { return vec_desc(v).n_tiles * 16; }
__host__ __device__ constexpr int vec_region_offset() { return stream_base_floats(S_COUNT); }
__host__ __device__ constexpr int vec_offset(int v) {
  int o = vec_region_offset();
  for (int i = 0; i < v; ++i) o += vec_floats(i);
  return o;
}
// The SCALE TABLE closes the vector region (so it reaches LDS with it): one float4 per forward matrix M,
//   {xs, dsc, asc, ws} = {2^a_M, 2^-(s_M + a_M), 2^(s_M + a_M), 2^s_M}
// s_M: exponent the fp16 planes of W_M carry (from max |w|), a_M: exponent the planes of the layer's INPUT carry (from an
// analytic bound of the input's magnitude) -- both chosen by ufr_weights_pack (prep.hip: weight_scale_kernel), read by the
// kernels at run time (weight_stream_f16.h: ScalarFile).  Not a function of the parameters alone: plan_entry maps it to zero.
// Behind the per-matrix entries: the scalars each transformer kernel actually consumes, derived ones included (epsilons,
// log2(e) multiples, producer-descale x consumer-scale products), in the order of the kernel's enum below -- a kernel reads
// its list into ONE vector register (lane k = scalar k) and takes a scalar with v_readlane where it needs it.
constexpr int kKernelScalars = 32;
// ... and behind the two lists the STATISTICS the exponents were derived from (prep.hip: weight_stats_kernel: max |w| and
// infinity norm per matrix, max |.| per vector parameter) plus, in slot kStatBoundSlot, the input bound the table currently
// serves.  They stay, so that the table can be re-derived for a frame whose measured feature bound is larger
// (ufr_weights_fit_frame: prep.hip weight_refit_kernel) without touching the parameters again.
constexpr int kStatFloats = 3 * kKernelScalars;
constexpr int kStatBoundSlot = 2 * kKernelScalars;      // the statistics proper occupy the slots below it (static_assert in prep.hip)
constexpr int kScaleFloats = M_COUNT * 4 + 2 * kKernelScalars + kStatFloats;
enum ViewScalar : int {   // view_transformer.hip (forward, tape)
  VS_XS_X = 0, VS_Q_DSC, VS_Q_L2E, VS_K_DSC, VS_K_L2E, VS_V_DSC, VS_M_XS, VS_EPS1, VS_M_ASC, VS_MLP0_DSC, VS_M_MLP2, VS_EPS2,
  VS_MLP2_ASC, VS_RW0_XS, VS_RW0_ASC, VS_RW0_DSC, VS_M_RW2, VS_RW2_ASC, VS_RW2_DSC, VS_M_RW4, VS_RW4_ASC, VS_RW4_DSC, VS_COUNT
};
enum RayScalar : int {    // ray_transformer.hip (forward, tape), ray_dgrad.hip (the k / v recompute)
  RS_XS_X = 0, RS_K_DSC, RS_K_L2E, RS_V_DSC, RS_V_ASC, RS_Q_DSC, RS_Q_L2E, RS_M_XS, RS_EPS1, RS_M_ASC, RS_MLP0_DSC, RS_M_MLP2,
  RS_EPS2, RS_MLP2_ASC, RS_DM0_XS, RS_DM0_ASC, RS_DM0_DSC, RS_M_DM2, RS_DM2_ASC, RS_DM2_DSC, RS_M_DM4, RS_DM4_ASC, RS_DM4_DSC,
  RS_COUNT
};
static_assert(VS_COUNT <= kKernelScalars && RS_COUNT <= kKernelScalars, "one register's worth of scalars per kernel");
__host__ __device__ constexpr int scale_table_offset() { return vec_offset(V_COUNT); }
__host__ __device__ constexpr int view_scalars_offset() { return scale_table_offset() + M_COUNT * 4; }
__host__ __device__ constexpr int ray_scalars_offset() { return view_scalars_offset() + kKernelScalars; }
__host__ __device__ constexpr int stats_offset() { return ray_scalars_offset() + kKernelScalars; }
__host__ __device__ constexpr int vec_region_floats() { return scale_table_offset() + kScaleFloats - vec_region_offset(); }
__host__ __device__ constexpr int blob_floats() { return scale_table_offset() + kScaleFloats; }
__host__ __device__ constexpr int mat_offset(int m) {  // first float of matrix m inside the blob
  return stream_base_floats(mat_stream(m)) + stream_mat_start(mat_stream(m), mat_stream_index(m)) * 256;
}

// Source of packed float i: parameter id (-1 = zero padding) and flat element index.
__host__ __device__ inline void plan_entry(int i, int* param, int* elem) {
  *param = -1;
  *elem = 0;
  if (i < vec_region_offset()) {
    int s = 0;
    while (s + 1 < S_COUNT && stream_base_floats(s + 1) <= i) ++s;
    const int j = i - stream_base_floats(s);
    const int r = j & 3, lane = (j >> 2) & 63;
    int f = j >> 8;  // fragment inside the stream
    if (f >= stream_frags(s)) return;  // tail padding
    int mi = 0;
    while (f >= mat_frags(stream_mat(s, mi))) { f -= mat_frags(stream_mat(s, mi)); ++mi; }
    const int m = stream_mat(s, mi), ot = stream_ot(s);
    const MatDesc d = mat_desc(m);
    const int full = d.n_out / ot;  // complete groups of `ot` out tiles
    int gi, ti, o;
    if (f < full * ot * d.n_in) {
      gi = f / (ot * d.n_in);
      const int rem = f % (ot * d.n_in);
      ti = rem / ot;
      o = rem % ot;
    } else {
      gi = full;
      const int rem = f - full * ot * d.n_in, no = d.n_out - full * ot;
      ti = rem / no;
      o = rem % no;
    }
    const int to = gi * ot + o;
    const int row = row_map(d.rm, to, lane & 15, d.out_dim);
    const int col = col_map(d.cm, ti, lane >> 4, r, d.in_dim);
    if (row >= 0 && col >= 0) { *param = d.param; *elem = row * d.k_raw + col; }
    return;
  }
  int off = vec_region_offset();
  for (int v = 0; v < V_COUNT; ++v) {
    const VecDesc d = vec_desc(v);
    const int n = d.n_tiles * 16;
    if (i < off + n) {
      int j = i - off;
      int r = j & 3, g = (j >> 2) & 3, t = j >> 4;
      int row = row_map(d.rm, t, 4 * g + r, d.dim);
      if (row >= 0) { *param = d.param; *elem = row; }
      return;
    }
    off += n;
  }
}

struct RawPtrs { const float* p[P_COUNT]; };

}  // namespace ufr
