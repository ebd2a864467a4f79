// Split-precision ("fp16x3") weight layout of the view- and ray-transformer chains.
//
// Measured on MI355X (tools/dev/mfma_valu2.*): v_mfma_f32_16x16x4_f32 does NOT overlap with VALU
// work -- it occupies the vector ALU for its 32 cycles, so every LayerNorm / attention / elu
// instruction of an fp32-MFMA kernel is lost MFMA time -- while the 16-bit matrix-core MFMAs
// (v_mfma_f32_16x16x32_{f16,bf16}, ~17 cycles for 8x the k-depth) do overlap.  The dense layers therefore
// run on the fp16 cores with fp32-grade accuracy: every fp32 operand is split into two fp16 planes,
//     hi = fp16(x),  lo = fp16(x - hi)        (round to nearest even; x - hi is exact in fp32)
// so hi + lo carries 22 significand bits and a sign trick's worth more (|x - hi - lo| <= 2^-23 |x|), and a
// product is three plane pairs accumulated in fp32 by the MFMA:
//     w*x ~= w_lo*x_hi + w_hi*x_lo + w_hi*x_hi        (dropped: w_lo*x_lo <= 2^-22 |w||x|)
// fp16's narrow exponent is handled with power-of-two scales (below): the planes hold 2^s_M w and 2^a_M x, exponents chosen
// per MATRIX when the weights are packed -- s_M from max |w_M|, a_M from an analytic bound of the layer's input -- which
// keeps the lo planes of weights and activations in the normal range (the MFMA honours fp16 subnormals --
// tools/dev/f16_probe.hip -- so values far below the bound lose precision gradually, with an absolute floor of
// 2^-25-a_M per activation); the accumulator carries 2^(s_M + a_M) and is descaled exactly where its consumer
// multiplies anyway.
// Measured (tests/accuracy_report.py, K = 80..176 layers of both chains against the fp32 oracle): the same
// 4e-7 as the three-plane bf16 scheme ("bf16x6": six products) this replaces, at half the matrix
// instructions, two thirds of the LDS weight traffic and 4 instead of 9 VALU instructions per split pair.
// 3 MFMAs of ~17 cycles replace 8 fp32 MFMAs of 32 cycles.
//
// A fragment here = (panel, out tile `to`, plane p): 64 lanes x 8 fp16 = 1 KiB; lane l holds
// W_p[row(to, l&15)][k = 8*(l>>4) + i], i = 0..7, where the 32 k-slots of a panel are the two
// accumulator tiles (2s, 2s+1) of the producing layer: slot 8g+i <-> tile 2s + (i>>2), feature
// col_map(tile, g, i&3) -- exactly what a lane holds after splitting its two fp32 accumulator tiles.
// The same fragment serves as the A operand (weights x activations: [feature][token] result) and as
// the B operand (activations x weights, "swapped": [token][feature] result) -- the two lane layouts
// of v_mfma_f32_16x16x32_f16 coincide.
// A PANEL is one k-step (32 input features) of one matrix: all its out tiles x 3 planes, stored in
// consumption order; a STREAM is the sequence of panels a kernel phase walks:
//   B_VT   view transformer  q0 k0 q1 k1 q2 k2 | v | merge | mlp0 | mlp2 | rw0 rw2 rw4
//   B_RT1  ray transformer sweep 1  k0 v0 k1 v1 k2 v2          (swapped operands)
//   B_RT2  ray transformer sweep 2  q | merge | mlp0 | mlp2 | dm0 dm2 dm4
#pragma once
#include "ufr_layout.h"

namespace ufr {

constexpr int kPlanes = 2;
// fp16 planes carry power-of-two scales so that the low planes stay clear of the fp16 subnormal range.  The exponents are
// DATA, not constants (ufr_layout.h: the scale table at the end of the vector region):
//   s_M  weights of matrix M are packed as 2^s_M w with 2^s_M max|w_M| in [2^14, 2^15): any finite weight fits
//   a_M  the layer's input is split as 2^a_M x with 2^a_M B_M <= 2^15, B_M an upper bound of |x| derived at pack time from
//        the bound of the token inputs the caller states (ufr_weights_pack_for; default kDefaultInputAbsMax), the
//        matrices' infinity norms, the LayerNorm gains and the biases (prep.hip: weight_scale_kernel)
// A layer's accumulator is 2^(s_M + a_M) times the true value.  kScaleExp*: the clamps of the exponents -- with them
// every accumulator stays below 2^15 2^15 K < 2^38 and every table entry is a normal fp32 number.
constexpr int kScaleExpWMin = -40, kScaleExpWMax = 24, kScaleExpXMin = -40, kScaleExpXMax = 12;
// The bound of the token features when the caller states none: the range the round-3 fixed scale 2^4 accepted (|x| < 4094).
// It is only the FLOOR: ufr_frame_prepare measures the frame's feature maps and volume features, and the table is re-derived
// for a frame beyond it (ufr_weights_fit_frame) -- a checkpoint loads and renders without side information (main.py:186-190).
constexpr float kDefaultInputAbsMax = 4094.f;
#ifndef UFR_F16_CHUNK
#define UFR_F16_CHUNK 12
#endif
constexpr int kF16ChunkFrags = UFR_F16_CHUNK;  // KiB per chunk: whole (tile, 2 planes) pairs; splits evenly over the 4 fetching waves
#ifndef UFR_F16_SLOTS
#define UFR_F16_SLOTS 3
#endif
// LDS ring depth: kF16Slots-1 chunks in flight.  Measured with the fp16x3 kernels (tools/dev/ab_kernels.sh, 4 rounds):
// 3 x 12 KiB runs the view / ray transformer 4.5 % / 2 % faster than 2 x 24 KiB and needs 12 KiB less LDS per
// workgroup (which the gather kernels of other chunks use when they share the CU); 4 x 12, 6 x 8 and 12 x 4 KiB are
// no better.
constexpr int kF16Slots = UFR_F16_SLOTS;
#ifndef UFR_F16_DEPTH
#define UFR_F16_DEPTH 2
#endif
// stages (one out tile's plane fragments = 3 C MFMAs of cover) the LDS weight reads run ahead of their use
constexpr int kF16Depth = UFR_F16_DEPTH;
static_assert(kF16Depth >= 1 && kF16Depth * kPlanes <= kF16ChunkFrags, "read-ahead must stay inside one chunk");

// B_COUNT forward streams (fp16 planes, this header's arithmetic), then the data-gradient streams of the backward kernels:
// the same fragment / panel / chunk format, but TRANSPOSED matrices (MatDesc::trans) as bf16 planes without a scale
// (hi = bf16(w), lo = bf16(w - hi): gradients need the exponent range, and 16 significand bits per operand are ample for
// the 1e-3 gradient tolerance) -- they follow the forward region in the blob, so one stream base formula serves both.
//   B_VTB   view transformer backwards  rw2^T | rw0^T | mlp2^T | mlp0^T | merge^T | q^T | k^T | v^T
//   B_RTB1  ray transformer backwards, sweep 1 (per tile)  dm2^T | dm0^T | mlp2^T | mlp0^T | merge^T | q^T
//   B_RTB2  ... sweep 2: k0 v0 k1 v1 k2 v2 (the FORWARD fp16 planes again: k, v are recomputed in the plain orientation) |
//           k^T | v^T
// The plane type is a property of the MATRIX (a transposed one = bf16), so a stream may mix both.
enum F16Stream { B_VT = 0, B_RT1 = 1, B_RT2 = 2, B_COUNT = 3, B_VTB = 3, B_RTB1 = 4, B_RTB2 = 5, B_ALL = 6 };
__host__ __device__ constexpr bool f16_mat_is_bf16(int m) { return m >= M_COUNT; }

struct Panel { int mat, s; };

__host__ __device__ constexpr int ksteps(int m) { return (mat_desc(m).n_in + 1) / 2; }

__host__ __device__ constexpr int f16_n_panels(int S) {
  return S == B_VT ? 3 * 3 + 3 + 5 + 5 + 3 + 1 + 1 : S == B_RT1 ? 6 : S == B_RT2 ? 3 + 3 + 6 + 6 + 3 + 1 + 1
       : S == B_VTB ? 1 + 1 + 3 + 5 + 3 + 3 * 3
       : S == B_RTB1 ? 1 + 1 + 3 + 6 + 3 + 3
       : 6 + 4 + 4;   // B_RTB2
}
// consumption order.  q and k (view) / k and v (ray) are interleaved per k-step: x is split once per step.
__host__ __device__ constexpr Panel f16_panel(int S, int i) {
  if (S == B_VT) {
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
  if (S == B_RT1) return {i % 2 == 0 ? M_RT_K : M_RT_V, i / 2};
  if (S == B_VTB) {
    if (i < 1) return {M_RW2T, 0};
    i -= 1;
    if (i < 1) return {M_RW0T, 0};
    i -= 1;
    if (i < 3) return {M_VT_MLP2T, i};
    i -= 3;
    if (i < 5) return {M_VT_MLP0T, i};
    i -= 5;
    if (i < 3) return {M_VT_MERGET, i};
    i -= 3;
    return {i < 3 ? M_VT_QT : i < 6 ? M_VT_KT : M_VT_VT, i % 3};
  }
  if (S == B_RTB1) {
    if (i < 1) return {M_DM2T, 0};
    i -= 1;
    if (i < 1) return {M_DM0T, 0};
    i -= 1;
    if (i < 3) return {M_RT_MLP2T, i};
    i -= 3;
    if (i < 6) return {M_RT_MLP0T, i};
    i -= 6;
    if (i < 3) return {M_RT_MERGET, i};
    i -= 3;
    return {M_RT_QT, i};
  }
  if (S == B_RTB2) {
    if (i < 6) return {i % 2 == 0 ? M_RT_K : M_RT_V, i / 2};
    i -= 6;
    return {i < 4 ? M_RT_KT : M_RT_VT, i % 4};
  }
  if (i < 3) return {M_RT_Q, i};
  i -= 3;
  if (i < 3) return {M_RT_MERGE, i};
  i -= 3;
  if (i < 6) return {M_RT_MLP0, i};
  i -= 6;
  if (i < 6) return {M_RT_MLP2, i};
  i -= 6;
  if (i < 3) return {M_DM0, i};
  i -= 3;
  return {i == 0 ? M_DM2 : M_DM4, 0};
}
__host__ __device__ constexpr int f16_mat_stream(int m) {
  return m >= M_DM2T ? ((m == M_RT_KT || m == M_RT_VT) ? B_RTB2 : B_RTB1)
         : m >= M_COUNT ? B_VTB
         : (m == M_RT_K || m == M_RT_V) ? B_RT1
         : (m == M_RT_Q || m == M_RT_MERGE || m == M_RT_MLP0 || m == M_RT_MLP2 || m == M_DM0 || m == M_DM2 || m == M_DM4)
             ? B_RT2
             : B_VT;
}
__host__ __device__ constexpr int f16_panel_frags(int S, int i) { return mat_desc(f16_panel(S, i).mat).n_out * kPlanes; }
__host__ __device__ constexpr int f16_panel_start(int S, int i) {  // first fragment of panel i within stream S
  int o = 0;
  for (int j = 0; j < i; ++j) o += f16_panel_frags(S, j);
  return o;
}
__host__ __device__ constexpr int f16_panel_index(int m, int s, int S = -1) {  // within stream S (default: the matrix's own)
  if (S < 0) S = f16_mat_stream(m);
  for (int i = 0; i < f16_n_panels(S); ++i)
    if (f16_panel(S, i).mat == m && f16_panel(S, i).s == s) return i;
  return -1;
}
__host__ __device__ constexpr int f16_stream_frags(int S) { return f16_panel_start(S, f16_n_panels(S)); }
// chunks that hold fragments, and the stream's length in the region: padded to a multiple of the ring depth so
// that a chunk's LDS slot does not depend on the pass; the kernels open the padding chunks explicitly
// (wstream_f16_finish) to keep the fetch schedule uniform
__host__ __device__ constexpr int f16_stream_real_chunks(int S) { return (f16_stream_frags(S) + kF16ChunkFrags - 1) / kF16ChunkFrags; }
__host__ __device__ constexpr int f16_stream_chunks(int S) { return (f16_stream_real_chunks(S) + kF16Slots - 1) / kF16Slots * kF16Slots; }
__host__ __device__ constexpr int f16_stream_base_frags(int S) {   // first fragment of stream S in the fp16 plane region
  int o = 0;
  for (int j = 0; j < S; ++j) o += f16_stream_chunks(j) * kF16ChunkFrags;
  return o;
}
constexpr int kF16FragsPadded = f16_stream_base_frags(B_COUNT);
constexpr int kF16Halfwords = kF16FragsPadded * 512;                  // fp16 elements in the region
constexpr int kF16Bytes = kF16FragsPadded * 1024;
constexpr int kBwdFragsPadded = f16_stream_base_frags(B_ALL) - kF16FragsPadded;   // the bf16 streams behind it
constexpr int kBwdHalfwords = kBwdFragsPadded * 512;
constexpr int kBwdBytes = kBwdFragsPadded * 1024;

// input feature of k-slot (g, i) of panel step s (or -1): accumulator tiles 2s and 2s+1 of the producer
__host__ __device__ constexpr int f16_col(int m, int s, int g, int i) {
  const MatDesc d = mat_desc(m);
  const int tile = 2 * s + (i >> 2);
  return tile < d.n_in ? col_map(d.cm, tile, g, i & 3, d.in_dim) : -1;
}

// source of halfword h of the plane regions (h < kF16Halfwords: fp16 forward streams; beyond: the bf16 backward streams):
// parameter, element, plane (param -1 = zero)
__host__ __device__ inline void plan_entry_f16(int h, int* param, int* elem, int* plane, int* is_bf16 = nullptr, int* mat = nullptr) {
  *param = -1; *elem = 0; *plane = 0;
  int f = h >> 9;                 // fragment
  const int lane = (h >> 3) & 63, i = h & 7;
  int S = 0;
  while (S + 1 < B_ALL && f >= f16_stream_base_frags(S + 1)) ++S;
  if (is_bf16) *is_bf16 = 0;
  if (mat) *mat = -1;
  f -= f16_stream_base_frags(S);
  if (f >= f16_stream_frags(S)) return;  // tail padding of the stream's last chunk
  int pi = 0;
  while (f >= f16_panel_frags(S, pi)) { f -= f16_panel_frags(S, pi); ++pi; }
  const Panel p = f16_panel(S, pi);
  if (is_bf16) *is_bf16 = f16_mat_is_bf16(p.mat) ? 1 : 0;
  if (mat) *mat = p.mat;
  const MatDesc d = mat_desc(p.mat);
  const int to = f / kPlanes;
  *plane = f % kPlanes;
  const int row = row_map(d.rm, to, lane & 15, d.out_dim);
  const int col = f16_col(p.mat, p.s, lane >> 4, i);
  if (row >= 0 && col >= 0) { *param = d.param; *elem = d.trans ? col * d.k_raw + row : row * d.k_raw + col; }
}

}  // namespace ufr
