// Weight gradients of the frustum U-Nets' 3x3x3 layers on the 16-bit matrix cores (round 6).
//   CostRegNetWeight  code1/encoder_utils/fmt/module.py:502-543   (`feature_volume.cost_reg_2`, the producer the reference trains)
//
//     dW[a][b][k] += sum_p TP[p][a] TQ[S p + k - 1][b]
// with (TP, TQ) = (d_out, in) for the convolutions -- dW = [cout][cin][27] -- and (in, d_out) for the transposed one -- dW =
// [cin][cout][27]: conv3d.hip's formulation, the reference's parameter layouts.  The contraction index is the VOXEL, i.e.
// the MFMA's k, while channel-last tensors put a voxel's channels on a lane.  conv3d.hip's fp32-MFMA kernels fetch every
// operand from global memory per tap (four voxel pairs per 32-cycle instruction, 27 bounds-checked neighbour fetches per
// voxel: 10 of the 23 ms of a training step with cost_reg_2).  Here
//  * the TQ halo of a brick of P voxels is staged ONCE through LDS as two bf16 planes (hi, lo: 16 significand bits per
//    operand, no scale -- gradients need the exponent range; wgrad_stream.hip's precision), [plane][voxel][16 channels];
//  * an operand tile (lane (g, j) = channels 4g..4g+3 of voxel j, exactly what a channel-last 16-byte load or an 8-byte LDS
//    read delivers) is TRANSPOSED ON THE MATRIX CORE: one v_mfma_f32_16x16x32_bf16 per plane against an identity selector
//    puts voxels along k (wgrad_stream.hip: make_frag) -- no LDS transpose, no unaligned windows for the x taps (a tap is a
//    voxel offset of the LDS read);
//  * a chunk = 32 consecutive x of a row = two column tiles = ONE contraction instruction per plane product; the TP
//    fragment of a chunk is made once and meets all the block's taps; the accumulator tiles (one per tap and 16 x 16
//    channel pair) stay in registers for the whole launch and are flushed once (LDS reduction over the four waves, one
//    atomic per value and block);
//  * the bias gradient (channel sums of d_out) rides as a contraction with a ones operand.
// A block's job = (a range of bricks) x (a group of 16-channel TP tiles) x (a 16-channel block of TQ) x (a group of taps):
// blockIdx.y enumerates the job kinds so that a kind's accumulators fit the registers.
#include <hip/hip_runtime.h>

#include "ufr_device.h"
#include "ufr_internal.h"
#include "weight_stream_f16.h"   // f16x8 / bf16x8, mfma_planes, split_pair_bf16, static_for

namespace ufr {

namespace {

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));

struct WgPlanesArgs {
  const float* tp2;    // HEADS: a second, one-channel TP tensor [B][Dp][Hp][Wp] = row CA of the product (CostRegNetWeight's
  float* dw2;          //        `weights` head beside `features`: one pass over the shared input), its gradient [1][CB][27]
  const float* tp;     // [B][Dp][Hp][Wp][CA]
  const float* tq;     // [B][Dq][Hq][Wq][CB]
  float* dw;           // [CA][CB][27], accumulated into
  float* dbias;        // [CA] (nullable): += sum_p TP[p][a]
  int B, Dp, Hp, Wp, Dq, Hq, Wq;
  int nbx, nby, nbz;
};

__device__ __forceinline__ unsigned hi16x2(float a, float b) {   // bf16 bits of two values that ARE bf16 numbers
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
struct WFrag { f16x8 p[2]; };   // both column tiles of a chunk in contraction layout: bf16 planes, 8 k slots per lane

// (hi, lo) words of the lane's four channels of column tiles 0 / 1 -> fragment; sel: this lane's transposition selector
__device__ __forceinline__ WFrag frag_from_planes(u32x2w h0, u32x2w l0, u32x2w h1, u32x2w l1, const f16x8& sel) {
  const f32x4 z = splat4(0.f);
  auto tr = [&](u32x2w w) __attribute__((always_inline)) -> f32x4 {
    return mfma_planes<true>(__builtin_bit_cast(f16x8, u32x4w{w[0], w[1], 0u, 0u}), sel, z);
  };
  const f32x4 a = tr(h0), b = tr(h1), c = tr(l0), d = tr(l1);
  WFrag f;
  f.p[0] = __builtin_bit_cast(f16x8, u32x4w{hi16x2(a[0], a[1]), hi16x2(a[2], a[3]), hi16x2(b[0], b[1]), hi16x2(b[2], b[3])});
  f.p[1] = __builtin_bit_cast(f16x8, u32x4w{hi16x2(c[0], c[1]), hi16x2(c[2], c[3]), hi16x2(d[0], d[1]), hi16x2(d[2], d[3])});
  return f;
}
// the same in two steps, so that the transposition MFMAs of one operand can be issued a slot ahead of the VALU that packs them
struct WTr { f32x4 a, b, c, d; };
__device__ __forceinline__ WTr frag_transpose(u32x2w h0, u32x2w l0, u32x2w h1, u32x2w l1, const f16x8& sel) {
  const f32x4 z = splat4(0.f);
  auto tr = [&](u32x2w w) __attribute__((always_inline)) -> f32x4 {
    return mfma_planes<true>(__builtin_bit_cast(f16x8, u32x4w{w[0], w[1], 0u, 0u}), sel, z);
  };
  return WTr{tr(h0), tr(h1), tr(l0), tr(l1)};
}
__device__ __forceinline__ WFrag frag_pack(const WTr& t) {
  WFrag f;
  f.p[0] = __builtin_bit_cast(f16x8, u32x4w{hi16x2(t.a[0], t.a[1]), hi16x2(t.a[2], t.a[3]), hi16x2(t.b[0], t.b[1]), hi16x2(t.b[2], t.b[3])});
  f.p[1] = __builtin_bit_cast(f16x8, u32x4w{hi16x2(t.c[0], t.c[1]), hi16x2(t.c[2], t.c[3]), hi16x2(t.d[0], t.d[1]), hi16x2(t.d[2], t.d[3])});
  return f;
}
__device__ __forceinline__ void planes_of(const f32x4& v, u32x2w& h, u32x2w& l) {
  unsigned h0, l0, h1, l1;
  split_pair_bf16(v[0], v[1], h0, l0);
  split_pair_bf16(v[2], v[3], h1, l1);
  h = u32x2w{h0, h1};
  l = u32x2w{l0, l1};
}
__device__ __forceinline__ f32x4 contract3(const WFrag& a, const WFrag& b, f32x4 acc) {   // small terms first (lo.lo dropped)
  acc = mfma_planes<true>(a.p[1], b.p[0], acc);
  acc = mfma_planes<true>(a.p[0], b.p[1], acc);
  return mfma_planes<true>(a.p[0], b.p[0], acc);
}

// CA: TP channels (1, 8, 16, 32, 64); CB: TQ channels (8, 16, 32, 64); S: 1 | 2
// NA: 16-channel TP tiles per block; NTAP: taps per block (CB = 8: 27 taps as 14 pairs, NTAP = 27)
template <int CA, int CB, int S, int NA, int NTAP>
struct WgCfg {
  static constexpr int CBL = CB < 16 ? CB : 16;                 // TQ channels staged per block
  static constexpr bool PAIRS = CB == 8;                        // two taps share a 16-column tile
  static constexpr bool ONE = CB == 1;                          // sixteen taps share it (conv0: a one-channel input)
  static constexpr int TPT = ONE ? 16 : PAIRS ? 2 : 1;          // taps per tile
  static constexpr int NSLOT = (NTAP + TPT - 1) / TPT;          // accumulator tiles per TP tile
  static constexpr int TX = 32, TY = 4, TZ = S == 1 ? 2 : 1;    // brick of P voxels: TY TZ rows of one chunk
  static constexpr int HX = S * (TX - 1) + 3, HY = S * (TY - 1) + 3, HZ = S * (TZ - 1) + 3;
  static constexpr int halo = HX * HY * HZ;
  static constexpr int plane_bytes = halo * CBL * 2;
  static constexpr int a_groups = (CA + 16 * NA - 1) / (16 * NA), b_blocks = CB / CBL, tap_groups = 27 / NTAP;
  static constexpr int kinds = a_groups * b_blocks * tap_groups;
  static constexpr int red_bytes = (NA * NSLOT + NA) * 1024 > (NA * 16 * CBL * NTAP + NA * 16) * 4 ? (NA * NSLOT + NA) * 1024
                                                                                                     : (NA * 16 * CBL * NTAP + NA * 16) * 4;   // the closing reduction / the flush stage
  static constexpr int lds = 2 * plane_bytes > red_bytes ? 2 * plane_bytes : red_bytes;
};

#ifndef UFR_WGP_ABL
#define UFR_WGP_ABL 0   // development ablations (timing only): 1 = no slot loop, 2 = no halo staging, 3 = no flush
#endif
template <int CA, int CB, int S, int NA, int NTAP, bool HEADS = false>
__global__ void __launch_bounds__(256, 2) conv3d_wgrad_planes_kernel(WgPlanesArgs a) {
  static_assert(!HEADS || (CA == 8 && NA == 1), "the two heads: 8 + 1 rows of one tile");
  typedef WgCfg<CA, CB, S, NA, NTAP> Cf;
  constexpr int CBL = Cf::CBL, NSLOT = Cf::NSLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const q_hi = smem;                       // [halo voxel][CBL] bf16
  char* const q_lo = smem + Cf::plane_bytes;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;

  // ---- job kind: (TP tile group, TQ channel block, tap group)
  const int kind = blockIdx.y;
  const int tg = kind % Cf::tap_groups, bb = (kind / Cf::tap_groups) % Cf::b_blocks, ag = kind / (Cf::tap_groups * Cf::b_blocks);
  const int tap0 = tg * NTAP, b0 = bb * CBL, a0 = ag * 16 * NA;
  const bool do_bias = a.dbias != nullptr && tg == 0 && bb == 0;

  // selector of the transposition: B[k = 8g + i][n = j] = 1 iff i < 4 and 4g + i == j (bf16 1.0 = 0x3f80)
  unsigned selw[2] = {0u, 0u};
  if ((j >> 2) == g) selw[(j & 3) >> 1] = (j & 1) ? 0x3f800000u : 0x00003f80u;
  const f16x8 sel = __builtin_bit_cast(f16x8, u32x4w{selw[0], selw[1], 0u, 0u});
  const f16x8 ones = __builtin_bit_cast(f16x8, u32x4w{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

  f32x4 acc[NA][NSLOT], accb[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    accb[t] = splat4(0.f);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc[t][s] = splat4(0.f);
  }

  const size_t p_frame = (size_t)a.Dp * a.Hp * a.Wp * CA * 4, q_frame = (size_t)a.Dq * a.Hq * a.Wq * CB * 4;
  const int n_bricks = a.nbx * a.nby * a.nbz * a.B;
  constexpr int G4 = Cf::ONE ? 1 : CBL / 4, items = Cf::halo * G4, kRounds = (items + 2047) / 2048;
  constexpr bool kPre = kRounds == 1;      // (two rounds = 64 more registers: the 16 x 16 layer, whose 27 accumulator tiles are 108, spilled 68)
  f32x4 pre[kPre ? kRounds * 8 : 1];
  auto item_off = [&](int kb, int it, bool& ok) __attribute__((always_inline)) -> unsigned {
    int kk = kb;
    const int bx = kk % a.nbx; kk /= a.nbx;
    const int by = kk % a.nby; kk /= a.nby;
    const int bz = kk % a.nbz;
    const int hv = it / G4, c4 = it - hv * G4;
    const int hx = hv % Cf::HX, hy = (hv / Cf::HX) % Cf::HY, hz = hv / (Cf::HX * Cf::HY);
    const int ix = S * bx * Cf::TX + hx - 1, iy = S * by * Cf::TY + hy - 1, iz = S * bz * Cf::TZ + hz - 1;
    ok = it < items && ix >= 0 && ix < a.Wq && iy >= 0 && iy < a.Hq && iz >= 0 && iz < a.Dq;
    return (unsigned)((((iz * a.Hq + iy) * a.Wq + ix) * CB + b0 + 4 * c4) * 4);
  };
  auto ld_item = [&](const __amdgpu_buffer_rsrc_t& rq, unsigned off) __attribute__((always_inline)) -> f32x4 {
    if constexpr (Cf::ONE) return f32x4{buf_ld1(rq, off), 0.f, 0.f, 0.f};      // one channel per voxel
    else return buf_ld4(rq, off);
  };
  auto rq_of = [&](int kb) __attribute__((always_inline)) {
    const int bi = kb / (a.nbx * a.nby * a.nbz);
    return buf_rsrc(reinterpret_cast<const char*>(a.tq) + (size_t)bi * q_frame, (unsigned)q_frame);
  };
  auto put_item = [&](int it, const f32x4& v) __attribute__((always_inline)) {
    if (it < items) {
      u32x2w h, l;
      planes_of(v, h, l);
      if constexpr (Cf::ONE) {
        *reinterpret_cast<unsigned short*>(q_hi + (size_t)it * 2) = (unsigned short)(h[0] & 0xffffu);      // [voxel] bf16
        *reinterpret_cast<unsigned short*>(q_lo + (size_t)it * 2) = (unsigned short)(l[0] & 0xffffu);
      } else {
        *reinterpret_cast<u32x2w*>(q_hi + (size_t)it * 8) = h;       // [voxel][4-channel group] = [voxel][CBL] bf16
        *reinterpret_cast<u32x2w*>(q_lo + (size_t)it * 8) = l;
      }
    }
  };
  auto pre_load = [&](int kb) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rq = rq_of(kb);
#pragma unroll
    for (int u = 0; u < (kPre ? kRounds * 8 : 0); ++u) {
      bool ok;
      const unsigned off = item_off(kb, 256 * u + tid, ok);
      pre[u] = ld_item(rq, ok ? off : kBufOut);
    }
  };
  auto pre_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < (kPre ? kRounds * 8 : 0); ++u) put_item(256 * u + tid, pre[u]);
  };
  auto stage_direct = [&](int kb) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rq = rq_of(kb);
#pragma unroll 1
    for (int i0 = 0; i0 < items; i0 += 2048) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bool ok;
        const unsigned off = item_off(kb, i0 + 256 * u + tid, ok);
        v[u] = ld_item(rq, ok ? off : kBufOut);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) put_item(i0 + 256 * u + tid, v[u]);
    }
  };
  // the TP (d_out) tiles of this wave's rows, requested one BRICK ahead: loaded at the top of a row they cost a memory round
  // trip per row with nothing to hide it (the full-resolution 8 x 8 layer: 16 k cycles per brick for 7 k of matrix work)
  constexpr int kRows = Cf::TY * Cf::TZ / 4;
  f32x4 tpn[kRows][NA][2];
  auto tp_load = [&](int kb) __attribute__((always_inline)) {
    int kk = kb;
    const int bx = kk % a.nbx; kk /= a.nbx;
    const int by = kk % a.nby; kk /= a.nby;
    const int bz = kk % a.nbz, bi = kk / a.nbz;
    const __amdgpu_buffer_rsrc_t rp = buf_rsrc(reinterpret_cast<const char*>(a.tp) + (size_t)bi * p_frame, (unsigned)p_frame);
    const __amdgpu_buffer_rsrc_t rp2 = buf_rsrc(reinterpret_cast<const char*>(HEADS ? a.tp2 : a.tp) + (size_t)bi * (p_frame / CA), (unsigned)(p_frame / CA));
#pragma unroll
    for (int rr = 0; rr < kRows; ++rr) {
      const int row = wave + 4 * rr;
      const int py = by * Cf::TY + row % Cf::TY, pz = bz * Cf::TZ + row / Cf::TY;
      const bool rok = py < a.Hp && pz < a.Dp;
#pragma unroll
      for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int px = bx * Cf::TX + 16 * c + j;
          const int ch = a0 + 16 * t + 4 * g;
          const unsigned vo = (unsigned)(((pz * a.Hp + py) * a.Wp + px) * CA);
          if constexpr (HEADS) {      // lane groups 0, 1: the 8-channel head; lane group 2, register 0: the 1-channel one
            f32x4 v = buf_ld4(rp, (rok && px < a.Wp && g < 2) ? (vo + ch) * 4u : kBufOut);
            const float w1 = buf_ld1(rp2, (rok && px < a.Wp && g == 2) ? (vo / CA) * 4u : kBufOut);
            if (g == 2) v = f32x4{w1, 0.f, 0.f, 0.f};
            tpn[rr][t][c] = v;
          } else if constexpr (CA % 4 == 0) {
            tpn[rr][t][c] = buf_ld4(rp, (rok && px < a.Wp && ch < CA) ? (vo + ch) * 4u : kBufOut);
          } else {      // CA = 1 (the weights head): one channel, in lane group 0
            tpn[rr][t][c] = f32x4{buf_ld1(rp, (rok && px < a.Wp && ch == 0) ? vo * 4u : kBufOut), 0.f, 0.f, 0.f};
          }
        }
    }
  };
  // Bricks in CONTIGUOUS runs, XCD x taking the x-th eighth (workgroups are dealt round-robin over the XCDs; gridDim.x is a
  // multiple of 8): neighbouring bricks' halos overlap 3.2 x, and dealt out strided every workgroup fetched its halos through
  // another L2 -- the full-resolution 8 x 8 layer moved 1.05 GB for 0.5 GB of tensors and was bound by that (0.58 ms).
  int k_begin, k_end;
  {
    const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8, per_xcd = gridDim.x / 8;
    const int q = n_bricks / 8, r = n_bricks % 8;
    const int x_begin = xcd * q + (xcd < r ? xcd : r), x_count = q + (xcd < r ? 1 : 0);
    const int q2 = x_count / per_xcd, r2 = x_count % per_xcd;
    k_begin = x_begin + idx * q2 + (idx < r2 ? idx : r2);
    k_end = k_begin + q2 + (idx < r2 ? 1 : 0);
  }
  if (k_begin < k_end) tp_load(k_begin);
  for (int k = k_begin; k < k_end; ++k) {
    // ---- stage the TQ halo of this block's 16 (8) channels: (voxel, 4-channel group) per thread, two bf16 planes.  When a
    // halo is one round of eight loads per thread (the full-resolution 8-channel layers), the NEXT brick's loads are
    // in flight while this one computes (pre[]); else it is staged in place, round by round.
    if constexpr (UFR_WGP_ABL == 2) { if (k == k_begin) stage_direct(k); }
    else if constexpr (!kPre) stage_direct(k);
    else if (k == k_begin) { pre_load(k); pre_store(); }
    __syncthreads();
    if constexpr (kPre && UFR_WGP_ABL != 2) {
      if (k + 1 < k_end) pre_load(k + 1);
    }
    int kk = k;
    const int bx = kk % a.nbx; kk /= a.nbx;
    const int by = kk % a.nby; kk /= a.nby;
    const int bz = kk % a.nbz, bi = kk / a.nbz;
    const int x0 = bx * Cf::TX, y0 = by * Cf::TY, z0 = bz * Cf::TZ;

    // ---- this wave's rows of the brick: one chunk (32 consecutive x = two column tiles) per row
    f32x4 tpc[kRows][NA][2];
#pragma unroll
    for (int rr = 0; rr < kRows; ++rr)
#pragma unroll
      for (int t = 0; t < NA; ++t) { tpc[rr][t][0] = tpn[rr][t][0]; tpc[rr][t][1] = tpn[rr][t][1]; }
    if (k + 1 < k_end) tp_load(k + 1);
    (void)bi;
    static_for<kRows>([&](auto rri) __attribute__((always_inline)) {
      constexpr int rr = decltype(rri)::value;
      const int row = wave + 4 * rr;
      const int ty = row % Cf::TY, tz = row / Cf::TY;
      // (rows past the volume's edge were loaded as zeros: they add nothing)
      // TP fragments: lane (g, j) = channels a0 + 16 t + 4g .. + 3 of voxel x0 + 16 c + j
      WFrag fa[NA];
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        u32x2w h0, l0, h1, l1;
        planes_of(tpc[rr][t][0], h0, l0);
        planes_of(tpc[rr][t][1], h1, l1);
        fa[t] = frag_from_planes(h0, l0, h1, l1, sel);
        if (do_bias) {
          accb[t] = mfma_planes<true>(fa[t].p[1], ones, accb[t]);
          accb[t] = mfma_planes<true>(fa[t].p[0], ones, accb[t]);
        }
      }
      // halo voxel of (tile c, lane j) at tap (0,0,0)
      const int vb0 = ((S * tz) * Cf::HY + S * ty) * Cf::HX + S * j, vb1 = vb0 + S * 16;
      // Slots in a three-stage pipeline: the LDS reads of slot s + 2, the four transposition MFMAs of slot s + 1, then the
      // packing VALU + the three contraction MFMAs of slot s (scheduling fences keep the order).  Written slot by slot, a
      // slot is one dependent chain -- LDS read -> MFMA -> VALU on its result -> MFMA -- and with two waves per SIMD nothing
      // hides it: the full-resolution 8 x 8 layer took 0.56 ms for 0.19 ms of matrix work.
      u32x2w rd[3][4];
      WTr trs[2];
      auto read_slot = [&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        if constexpr (Cf::ONE) {
          // lane group g, element i <-> tap 16 s + 4 g + i of the one input channel: four 2-byte reads per plane and column tile
          unsigned short hh[2][4], ll[2][4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int tap = tap0 + 16 * s + 4 * g + i;
            const int tt = tap < 27 ? tap : 0;
            const int toff = ((tt / 9) * Cf::HY + (tt / 3) % 3) * Cf::HX + tt % 3;
            hh[0][i] = *reinterpret_cast<const unsigned short*>(q_hi + (vb0 + toff) * 2);
            ll[0][i] = *reinterpret_cast<const unsigned short*>(q_lo + (vb0 + toff) * 2);
            hh[1][i] = *reinterpret_cast<const unsigned short*>(q_hi + (vb1 + toff) * 2);
            ll[1][i] = *reinterpret_cast<const unsigned short*>(q_lo + (vb1 + toff) * 2);
          }
          auto pk = [](const unsigned short (&v)[4]) { return u32x2w{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16)}; };
          rd[s % 3][0] = pk(hh[0]); rd[s % 3][1] = pk(ll[0]); rd[s % 3][2] = pk(hh[1]); rd[s % 3][3] = pk(ll[1]);
          return;
        }
        int tap, cofs;
        if constexpr (Cf::PAIRS) { tap = tap0 + 2 * s + (g >> 1); cofs = 4 * (g & 1); }
        else { tap = tap0 + s; cofs = 4 * g; }
        const bool live = tap < 27;
        const int tt = live ? tap : 0;
        const int toff = ((tt / 9) * Cf::HY + (tt / 3) % 3) * Cf::HX + tt % 3;
        // a padding tap (27: the second half of the last pair) reads the brick's own voxel with a ZERO selector result:
        // its products are dropped at the flush (tap < 27), and finite operands keep the accumulators finite
        const int o0 = ((vb0 + toff) * CBL + cofs) * 2, o1 = ((vb1 + toff) * CBL + cofs) * 2;
        rd[s % 3][0] = *reinterpret_cast<const u32x2w*>(q_hi + o0);
        rd[s % 3][1] = *reinterpret_cast<const u32x2w*>(q_lo + o0);
        rd[s % 3][2] = *reinterpret_cast<const u32x2w*>(q_hi + o1);
        rd[s % 3][3] = *reinterpret_cast<const u32x2w*>(q_lo + o1);
      };
      read_slot(std::integral_constant<int, 0>{});
      if constexpr (NSLOT > 1) read_slot(std::integral_constant<int, 1>{});
      trs[0] = frag_transpose(rd[0][0], rd[0][1], rd[0][2], rd[0][3], sel);
      static_for<(UFR_WGP_ABL == 1 ? 1 : NSLOT)>([&](auto si) __attribute__((always_inline)) {
        constexpr int s = decltype(si)::value;
        if constexpr (s + 2 < NSLOT) read_slot(std::integral_constant<int, s + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 1 < NSLOT)
          trs[(s + 1) & 1] = frag_transpose(rd[(s + 1) % 3][0], rd[(s + 1) % 3][1], rd[(s + 1) % 3][2], rd[(s + 1) % 3][3], sel);
        __builtin_amdgcn_sched_barrier(0);
        const WFrag fb = frag_pack(trs[s & 1]);
#pragma unroll
        for (int t = 0; t < NA; ++t) acc[t][s] = contract3(fa[t], fb, acc[t][s]);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    __syncthreads();     // every wave is done with this brick's planes
    if constexpr (kPre && UFR_WGP_ABL != 2) {
      if (k + 1 < k_end) pre_store();
    }
  }

  // ---- the four waves' partial tiles -> wave 0 through LDS, then one atomic per value
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NA; ++t) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) red[(t * NSLOT + s) * 64 + lane] = acc[t][s];
        red[(NA * NSLOT + t) * 64 + lane] = accb[t];
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) acc[t][s] += red[(t * NSLOT + s) * 64 + lane];
#pragma unroll
      for (int t = 0; t < NA; ++t) accb[t] += red[(NA * NSLOT + t) * 64 + lane];
    }
  }
  // ---- flush.  Wave 0 writes its sums into LDS IN THE PARAMETER'S ORDER ([a][b][tap], the block's taps contiguous per (a, b)),
  // then all four waves add runs of consecutive floats: a wave-instruction's 64 adds fall into a few cache lines = a few L2
  // transactions.  Lane by lane from the accumulator layout every add was its own transaction -- 27 floats apart -- and the
  // 16 x 16 layer spent 0.23 of its 0.36 ms there (512 workgroups x 6 912 same-address atomics: 15 G / s, the L2's rate).
  __syncthreads();
  float* const stage = reinterpret_cast<float*>(smem);                  // [NA 16 rows][CBL][NTAP] (+ bias [NA 16])
  constexpr int kRowF = CBL * NTAP, kStageF = NA * 16 * kRowF;
  if (wave == 0 && !(UFR_WGP_ABL == 3 && acc[0][0][0] != 12345.f)) {
#pragma unroll
    for (int t = 0; t < NA; ++t) {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        int tl, bl;                                   // tap within the block's group, TQ channel within the block
        if constexpr (Cf::ONE) { tl = 16 * s + j; bl = 0; }
        else if constexpr (Cf::PAIRS) { tl = 2 * s + (j >> 3); bl = j & 7; }
        else { tl = s; bl = j; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (tl < NTAP) stage[(16 * t + 4 * g + r) * kRowF + bl * NTAP + tl] = acc[t][s][r];
      }
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[kStageF + 16 * t + 4 * g + r] = accb[t][r];
      }
    }
  }
  __syncthreads();
  if (UFR_WGP_ABL == 3 && acc[0][0][0] != 12345.f) return;
  for (int e = tid; e < kStageF; e += 256) {
    const int row = e / kRowF, rem = e - row * kRowF, bl = rem / NTAP, tl = rem - bl * NTAP;
    const int ach = a0 + row, bch = b0 + bl, tap = tap0 + tl;
    const float x = stage[e];
    if (ach < CA && bch < CB && tap < 27 && x != 0.f) atomicAdd(a.dw + ((size_t)ach * CB + bch) * 27 + tap, x);
    if (HEADS && ach == CA && bch < CB && tap < 27 && x != 0.f) atomicAdd(a.dw2 + (size_t)bch * 27 + tap, x);
  }
  if (do_bias && tid < NA * 16 && a0 + tid < CA) atomicAdd(a.dbias + a0 + tid, stage[kStageF + tid]);
}

template <int CA, int CB, int S, int NA, int NTAP, bool HEADS = false>
hipError_t launch_wgp_t(WgPlanesArgs a, hipStream_t s) {
  typedef WgCfg<CA, CB, S, NA, NTAP> Cf;
  static LdsAttrOnce lds_attr;
  if (const hipError_t e = lds_attr.set(reinterpret_cast<const void*>(&conv3d_wgrad_planes_kernel<CA, CB, S, NA, NTAP, HEADS>), Cf::lds); e != hipSuccess)
    return e;
  a.nbx = (a.Wp + Cf::TX - 1) / Cf::TX; a.nby = (a.Hp + Cf::TY - 1) / Cf::TY; a.nbz = (a.Dp + Cf::TZ - 1) / Cf::TZ;
  const long long bricks = (long long)a.nbx * a.nby * a.nbz * a.B;
  if (bricks <= 0 || bricks > 0x7fffffffLL) return hipErrorInvalidValue;
  // resident slots (two workgroups per CU) shared by the job kinds; every block flushes NA NSLOT 256 atomics, so no more
  // blocks per kind than bricks / 4
  long long bx = 512 / Cf::kinds;
  if (bx > (bricks + 3) / 4) bx = (bricks + 3) / 4;
  bx = (bx + 7) / 8 * 8;                     // a multiple of the 8 XCDs (contiguous runs per XCD)
  hipLaunchKernelGGL((conv3d_wgrad_planes_kernel<CA, CB, S, NA, NTAP, HEADS>), dim3((unsigned)bx, Cf::kinds), dim3(256), Cf::lds, s, a);
  return hipGetLastError();
}

}  // namespace

// hipErrorInvalidValue: not a shape of this kernel family (the caller falls back to conv3d.hip's kernels)
hipError_t launch_conv3d_wgrad_planes(const float* tp, const float* tq, float* dw, float* dbias, int B, int Dp, int Hp, int Wp, int Dq,
                                      int Hq, int Wq, int ca, int cb, int S, hipStream_t s) {
  // both tensors through 31-bit buffer offsets (per batch element)
  if ((long long)Dp * Hp * Wp * ca * 4 >= (1ll << 31) || (long long)Dq * Hq * Wq * cb * 4 >= (1ll << 31)) return hipErrorInvalidValue;
  WgPlanesArgs a;
  a.tp2 = nullptr; a.dw2 = nullptr;
  a.tp = tp; a.tq = tq; a.dw = dw; a.dbias = dbias; a.B = B; a.Dp = Dp; a.Hp = Hp; a.Wp = Wp; a.Dq = Dq; a.Hq = Hq; a.Wq = Wq;
  a.nbx = a.nby = a.nbz = 0;
#define UFR_WGP(CA_, CB_, S_, NA_, NT_) if (ca == CA_ && cb == CB_ && S == S_) return launch_wgp_t<CA_, CB_, S_, NA_, NT_>(a, s);
  UFR_WGP(8, 1, 1, 1, 27)      // conv0 (one input channel: sixteen taps per tile)
  UFR_WGP(8, 8, 1, 1, 27)      // features head
  UFR_WGP(1, 8, 1, 1, 27)      // weights head
  UFR_WGP(16, 16, 1, 1, 27)    // conv2
  UFR_WGP(32, 32, 1, 2, 9)     // conv4
  UFR_WGP(64, 64, 1, 2, 9)     // conv6
  UFR_WGP(16, 8, 2, 1, 27)     // conv1, conv11
  UFR_WGP(32, 16, 2, 2, 9)     // conv3, conv9
  UFR_WGP(64, 32, 2, 2, 9)     // conv5, conv7
#undef UFR_WGP
  return hipErrorInvalidValue;
}

// CostRegNetWeight's two heads on one input (module.py:541-543): d `features.weight` [8][8][27] and d `weights.weight`
// [1][8][27] from d_out (B,D,H,W,8), d_out2 (B,D,H,W,1) and the shared input in ONE pass (the 1-channel head alone costs a
// whole pass of its own: 0.49 ms at full resolution)
hipError_t launch_conv3d_wgrad_heads(const float* in, const float* d_out, const float* d_out2, float* dw, float* dw2, int B, int D, int H,
                                     int W, hipStream_t s) {
  if ((long long)D * H * W * 8 * 4 >= (1ll << 31)) return hipErrorInvalidValue;
  WgPlanesArgs a;
  a.tp = d_out; a.tp2 = d_out2; a.tq = in; a.dw = dw; a.dw2 = dw2; a.dbias = nullptr;
  a.B = B; a.Dp = D; a.Hp = H; a.Wp = W; a.Dq = D; a.Hq = H; a.Wq = W; a.nbx = a.nby = a.nbz = 0;
  return launch_wgp_t<8, 8, 1, 1, 27, true>(a, s);
}

}  // namespace ufr
