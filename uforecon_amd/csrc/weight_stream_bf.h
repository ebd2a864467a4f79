// bf16x6 split-precision GEMM core on the bf16 matrix cores, fed by an LDS weight stream
// (layout: ufr_layout_bf.h; streaming scheme: weight_stream.h).
#pragma once
#include <type_traits>

#include "ufr_layout_bf.h"
#include "weight_stream.h"

namespace ufr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBfRingBytes = kBfSlots * kBfChunkFrags * 1024;  // kBfSlots 24 KiB slots
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
  // the view transformer reads the view-token fragment before its first chunk barrier: without this barrier a
  // wave that starts early reads LDS another wave has not filled yet (seen only when gather workgroups share
  // the CU and stagger the waves' start: 8 wrong points per late workgroup)
  __syncthreads();
  return ws;
}

// fetch chunk CHK of stream S into its ring slot: this wave's share of the fragments
template <int S, int NWAVES, int CHK>
__device__ __forceinline__ void wstream_bf_fetch(const WStreamBf& ws) {
  static_assert(kBfChunkFrags % NWAVES == 0, "chunk must split evenly over the fetching waves");
  constexpr size_t goff = ((size_t)bf_stream_base_frags(S) + (size_t)CHK * kBfChunkFrags) * 1024;
  constexpr int soff = (CHK % kBfSlots) * (kBfChunkFrags * 1024);
  int zero = 0;
  asm volatile("" : "+s"(zero));  // keep the loop-invariant source address out of LICM's hands
  const char* g = ws.src + zero + goff + ws.wave * 1024;   // wave-uniform: scalar base + 32-bit lane offset
  char* slot = ws.ring + soff + ws.wave * 1024;
  const unsigned lane_off = ws.lane * 16;
#pragma unroll
  for (int k = 0; k < kBfChunkFrags / NWAVES; ++k)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + k * NWAVES * 1024 + lane_off), (lds_ptr_t)(slot + k * NWAVES * 1024), 16, 0, 0);
}

// open chunk CHK: the ring keeps kBfSlots-1 chunks in flight, so at most the (kBfSlots-2) younger fetches
// of this wave may still be outstanding when chunk CHK must have landed
template <int S, int NWAVES, int CHK>
__device__ __forceinline__ void wstream_bf_open(const WStreamBf& ws, bool wrap) {
#ifdef UFR_ABL_NOBARRIER  // ablation builds (timing only, results are garbage): no hand-off at all / barrier without fetch
  (void)ws; (void)wrap;
  return;
#endif
#ifdef UFR_ABL_NODMA
  __syncthreads();
  return;
#endif
  constexpr int per_chunk = kBfChunkFrags / NWAVES, ahead = kBfSlots - 1, n_chunks = bf_stream_chunks(S);
  constexpr int younger = (kBfSlots - 2) * per_chunk;
  if constexpr (younger == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if constexpr (CHK + ahead <= n_chunks) {          // every younger fetch was issued unconditionally
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger) : "memory");
  } else {                                                 // the younger fetches were wrap-around ones
    if (wrap) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if constexpr (CHK + ahead < n_chunks) {
    wstream_bf_fetch<S, NWAVES, CHK + ahead>(ws);
  } else {
    if (wrap) wstream_bf_fetch<S, NWAVES, (CHK + ahead) % n_chunks>(ws);
  }
}

// end of a pass over stream S: open its padding chunks (none for most streams) so the wrap-around fetches go out
template <int S, int NWAVES>
__device__ __forceinline__ void wstream_bf_finish(const WStreamBf& ws, bool wrap) {
  constexpr int real = bf_stream_real_chunks(S), pad = bf_stream_chunks(S) - real;
  static_for<pad>([&](auto ci) __attribute__((always_inline)) { wstream_bf_open<S, NWAVES, real + decltype(ci)::value>(ws, wrap); });
}

// start of a pass over stream S: its first kBfSlots-1 chunks.  The slots must be free: at kernel start, or
// after every wave has passed the barrier that opened the previous stream's last chunk with wrap == false
// (then slot 0.. are no longer read; the last chunk's own slot is (n_chunks-1) % kBfSlots = kBfSlots-1).
template <int S, int NWAVES>
__device__ __forceinline__ void wstream_bf_prime(const WStreamBf& ws) {
  static_for<kBfSlots - 1>([&](auto ci) __attribute__((always_inline)) { wstream_bf_fetch<S, NWAVES, decltype(ci)::value>(ws); });
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
// hook(integral_constant<to>) is VALU work without dependence on this panel (the split of the NEXT k-step's
// operands): it is interleaved with the stage's MFMAs -- the bf16 matrix pipe runs ~2 independent VALU
// instructions per MFMA for free (tools/dev/mfma_valu2), so the split costs nothing once it sits there.
#ifndef UFR_HOOK_VALU
#define UFR_HOOK_VALU 2   // VALU instructions of the hook issued after each MFMA
#endif
struct NoHook {
  template <class T> __device__ __forceinline__ void operator()(T) const {}
};
template <int M, int S, int C, int NWAVES, bool SWAP = false, class Hook = NoHook>
__device__ __forceinline__ void gemm_bf_panel(WStreamBf& ws, const BStep (&b)[C], f32x4 (&out)[C][mat_desc(M).n_out],
                                              bool wrap, Hook&& hook = NoHook{}) {
  constexpr int n_out = mat_desc(M).n_out, ST = bf_mat_stream(M);
  static_assert(bf_panel_index(M, S) >= 0, "not a panel of the stream");
  constexpr int F0 = bf_panel_start(ST, bf_panel_index(M, S));
  const bf16x8* lds = reinterpret_cast<const bf16x8*>(ws.ring) + ws.lane;
  static_for<n_out>([&](auto ti) __attribute__((always_inline)) {
    constexpr int to = decltype(ti)::value;
    constexpr int f = F0 + to * kPlanes;                 // first of the stage's three fragments
    constexpr int chk = f / kBfChunkFrags, in_chk = f % kBfChunkFrags;
    constexpr int base = ((chk % kBfSlots) * kBfChunkFrags + in_chk) * 64;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a[kPlanes];
    if constexpr (in_chk == 0) {                         // chunk boundary: hand-off, then read this stage now
      wstream_bf_open<ST, NWAVES, chk>(ws, wrap);
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) a[p] = lds[base + p * 64];
    } else {
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) a[p] = ws.pre[p];
    }
    if constexpr (in_chk + kPlanes < kBfChunkFrags && f + kPlanes < bf_stream_frags(ST)) {  // next stage lies in the open chunk
#pragma unroll
      for (int p = 0; p < kPlanes; ++p) ws.pre[p] = lds[base + (kPlanes + p) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    hook(ti);
    // six plane pairs with i + j <= 4, small terms first (0 = hi, 1 = mid, 2 = lo); SWAP: activations in the A slot
    static_for<6>([&](auto pi) __attribute__((always_inline)) {
      constexpr int pw[6] = {1, 0, 2, 0, 1, 0}, px[6] = {1, 2, 0, 1, 0, 0};
      constexpr int w = pw[decltype(pi)::value], x = px[decltype(pi)::value];
#pragma unroll
      for (int c = 0; c < C; ++c)
        out[c][to] = SWAP ? mfma_bf(b[c].p[x], a[w], out[c][to]) : mfma_bf(a[w], b[c].p[x], out[c][to]);
    });
    if constexpr (!std::is_same<std::decay_t<Hook>, NoHook>::value) {
      // issue order: one MFMA, then up to two of the hook's VALU instructions, repeated
#pragma unroll
      for (int i = 0; i < 6 * C; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, UFR_HOOK_VALU, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(out[c][to]));  // pin: pure MFMAs are otherwise sunk past later LDS reads
  });
  __builtin_amdgcn_sched_barrier(0);
}

// ---- pipelined operand split: unit u = (c, tile half, value pair q) of a k-step, 9 VALU instructions
template <int C>
struct BWords { unsigned w[C][kPlanes][4]; };

__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(a, b);
  a -= __builtin_bit_cast(float, h << 16);
  b -= __builtin_bit_cast(float, h & 0xffff0000u);
  m = pack_bf16(a, b);
  a -= __builtin_bit_cast(float, m << 16);
  b -= __builtin_bit_cast(float, m & 0xffff0000u);
  l = pack_bf16(a, b);
}

// units [U0, U1) of k-step S of the tiles in[c][0..NIN)
template <int S, int U0, int U1, int C, int NIN>
__device__ __forceinline__ void split_units(const f32x4 (&in)[C][NIN], BWords<C>& bw) {
  static_for<U1 - U0>([&](auto ui) __attribute__((always_inline)) {
    constexpr int u = U0 + decltype(ui)::value;
    constexpr int c = u / 4, half = (u >> 1) & 1, q = u & 1, tile = 2 * S + half;
    if constexpr (tile < NIN) {
      split_pair(in[c][tile][2 * q], in[c][tile][2 * q + 1], bw.w[c][0][2 * half + q], bw.w[c][1][2 * half + q],
                 bw.w[c][2][2 * half + q]);
    } else {
      bw.w[c][0][2 * half + q] = bw.w[c][1][2 * half + q] = bw.w[c][2][2 * half + q] = 0u;
    }
  });
}

template <int C>
__device__ __forceinline__ void bwords_to_bstep(const BWords<C>& bw, BStep (&b)[C]) {
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int p = 0; p < kPlanes; ++p)
      b[c].p[p] = __builtin_bit_cast(bf16x8, u32x4{bw.w[c][p][0], bw.w[c][p][1], bw.w[c][p][2], bw.w[c][p][3]});
}

// out += W_M x in over all k-steps of M: in[c][0..NIN) are the producer's fp32 accumulator tiles.
// The exact bf16 split of k-step s+1 is interleaved with the MFMAs of k-step s (only step 0's is exposed).
template <int M, int C, int NWAVES, int NIN>
__device__ __forceinline__ void gemm_bf(WStreamBf& ws, const f32x4 (&in)[C][NIN], f32x4 (&out)[C][mat_desc(M).n_out],
                                        bool wrap) {
  static_assert(NIN == mat_desc(M).n_in, "input tile count");
  constexpr int n_out = mat_desc(M).n_out, NU = 4 * C;
  BWords<C> cur;
  split_units<0, 0, NU>(in, cur);
  static_for<ksteps(M)>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    BStep b[C];
    bwords_to_bstep(cur, b);
#ifdef UFR_NO_HOOK
    constexpr bool use_hook = false;
#else
    constexpr bool use_hook = true;
#endif
    if constexpr (use_hook && s + 1 < ksteps(M) && n_out >= 2) {
      BWords<C> nxt;
      gemm_bf_panel<M, s, C, NWAVES, false>(ws, b, out, wrap, [&](auto ti) __attribute__((always_inline)) {
        constexpr int to = decltype(ti)::value;
        split_units<s + 1, to * NU / n_out, (to + 1) * NU / n_out>(in, nxt);
      });
      cur = nxt;
    } else {
      gemm_bf_panel<M, s, C, NWAVES>(ws, b, out, wrap);
      if constexpr (s + 1 < ksteps(M)) split_units<s + 1, 0, NU>(in, cur);
    }
  });
}

}  // namespace ufr
