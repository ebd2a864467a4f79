// bf16x6 split-precision GEMM core on the bf16 matrix cores, fed by an LDS weight stream
// (layout: ufr_layout_bf.h; streaming scheme: weight_stream.h).
#pragma once
#include "ufr_layout_bf.h"
#include "weight_stream.h"

namespace ufr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBfRingBytes = 2 * kBfChunkFrags * 1024;  // two 24 KiB slots
constexpr int kBfLdsBytes = kBfRingBytes + kVecBytes;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // RNE, a -> low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// exact 3-way split of the 4 fp32 values a lane holds of one accumulator tile:
// pl[p][q] = plane p of values (2q, 2q+1), packed; hi + mid + lo == value
__device__ __forceinline__ void split_tile(const f32x4& v, unsigned (&pl)[kPlanes][2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float a = v[2 * q], b = v[2 * q + 1];
    const unsigned h = pack_bf16(a, b);
    a -= __builtin_bit_cast(float, h << 16);
    b -= __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack_bf16(a, b);
    a -= __builtin_bit_cast(float, m << 16);
    b -= __builtin_bit_cast(float, m & 0xffff0000u);
    pl[0][q] = h;
    pl[1][q] = m;
    pl[2][q] = pack_bf16(a, b);
  }
}

// B operands of one k-step (accumulator tiles ta, tb of one column tile): one bf16x8 per plane
struct BStep { bf16x8 p[kPlanes]; };
__device__ __forceinline__ BStep make_bstep(const f32x4& ta, const f32x4& tb) {
  unsigned pa[kPlanes][2], pb[kPlanes][2];
  split_tile(ta, pa);
  split_tile(tb, pb);
  BStep s;
#pragma unroll
  for (int p = 0; p < kPlanes; ++p) s.p[p] = __builtin_bit_cast(bf16x8, u32x4{pa[p][0], pa[p][1], pb[p][0], pb[p][1]});
  return s;
}

struct WStreamBf {
  const char* src;     // bf16 region of the packed blob (global, wave-uniform)
  char* ring;          // LDS: two chunk slots
  const f32x4* vecs;   // LDS: vector fragments (fp32)
  int wave, lane;
  bf16x8 pre[kPlanes]; // planes of the next stage's fragment, read one stage ahead when the chunk allows
};

template <int NWAVES>
__device__ __forceinline__ WStreamBf wstream_bf_begin(const float* __restrict__ packed, char* smem) {
  WStreamBf ws;
  ws.lane = threadIdx.x & 63;
  ws.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ws.src = reinterpret_cast<const char*>(packed) + (size_t)blob_floats() * 4;
  ws.ring = smem;
  f32x4* v = reinterpret_cast<f32x4*>(smem + kBfRingBytes);
  ws.vecs = v;
  constexpr int voff = vec_region_offset(), n4 = vec_region_floats() / 4;
  const f32x4* vs = reinterpret_cast<const f32x4*>(packed + voff);
  for (int i = threadIdx.x; i < n4; i += NWAVES * 64) v[i] = vs[i];
  return ws;
}

template <int NWAVES, int CHK>
__device__ __forceinline__ void wstream_bf_fetch(const WStreamBf& ws) {
  static_assert(kBfChunkFrags % NWAVES == 0, "chunk must split evenly over the fetching waves");
  constexpr size_t goff = (size_t)CHK * kBfChunkFrags * 1024;
  constexpr int soff = (CHK & 1) * (kBfChunkFrags * 1024);
  int zero = 0;
  asm volatile("" : "+s"(zero));  // keep the loop-invariant source address out of LICM's hands
  const char* g = ws.src + zero + goff + ws.wave * 1024 + ws.lane * 16;
  char* slot = ws.ring + soff + ws.wave * 1024;
#pragma unroll
  for (int k = 0; k < kBfChunkFrags / NWAVES; ++k)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + k * NWAVES * 1024), (lds_ptr_t)(slot + k * NWAVES * 1024), 16, 0, 0);
}

template <int NWAVES, int CHK>
__device__ __forceinline__ void wstream_bf_open(const WStreamBf& ws, bool wrap) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (CHK + 1 < kVtbChunks) {
    wstream_bf_fetch<NWAVES, CHK + 1>(ws);
  } else {
    if (wrap) wstream_bf_fetch<NWAVES, 0>(ws);
  }
}

template <int V>
__device__ __forceinline__ f32x4 vec_frag(const WStreamBf& ws, int t, int g) {
  constexpr int base = (vec_offset(V) - vec_region_offset()) / 4;
  return ws.vecs[base + t * 4 + g];
}

__device__ __forceinline__ f32x4 mfma_bf(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One panel (k-step S of matrix M): out[c][to] += W[:, 32 S .. 32 S + 31] x b[c], all out tiles.
// Panels must be executed in stream order (ufr_layout_bf.h: vt_panel).
template <int M, int S, int C, int NWAVES>
__device__ __forceinline__ void gemm_bf_panel(WStreamBf& ws, const BStep (&b)[C], f32x4 (&out)[C][mat_desc(M).n_out],
                                              bool wrap) {
  constexpr int n_out = mat_desc(M).n_out;
  constexpr int F0 = panel_start(panel_index(M, S));
  static_assert(panel_index(M, S) >= 0, "not a panel of the stream");
  const bf16x8* lds = reinterpret_cast<const bf16x8*>(ws.ring) + ws.lane;
  static_for<n_out>([&](auto ti) __attribute__((always_inline)) {
    constexpr int to = decltype(ti)::value;
    constexpr int f = F0 + to * kPlanes;                 // first of the stage's three fragments
    constexpr int chk = f / kBfChunkFrags, in_chk = f % kBfChunkFrags;
    constexpr int base = ((chk & 1) * kBfChunkFrags + in_chk) * 64;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[kPlanes];
    if constexpr (in_chk == 0) {                         // chunk boundary: hand-off, then read this stage now
      wstream_bf_open<NWAVES, chk>(ws, wrap);
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) a[p] = lds[base + p * 64];
    } else {
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) a[p] = ws.pre[p];
    }
    if constexpr (in_chk + kPlanes < kBfChunkFrags && f + kPlanes < kVtbFrags) {  // next stage lies in the open chunk
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) ws.pre[p] = lds[base + (kPlanes + p) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    // six plane pairs with i + j <= 4, small terms first (0 = hi, 1 = mid, 2 = lo)
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[1], b[c].p[1], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[0], b[c].p[2], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[2], b[c].p[0], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[0], b[c].p[1], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[1], b[c].p[0], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c][to] = mfma_bf(a[0], b[c].p[0], out[c][to]);
#pragma unroll
    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(out[c][to]));  // pin (see weight_stream.h)
  });
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace ufr
