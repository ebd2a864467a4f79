// Weight streaming through LDS + the chained-MFMA GEMM that consumes it.
//
// Every wave of a workgroup walks the same static sequence of 1 KiB A-fragments (a STREAM of
// ufr_layout.h).  Instead of each wave pulling its own copy from L2 (measured: 81-84 % of the MFMA
// rate at 2 waves/SIMD because 256 CUs x 8 waves hammer the same L2 lines), the workgroup stages the
// stream through LDS in chunks of <= 32 fragments:
//
//   chunk c -> LDS slot c&1, fetched with lane-linear LDS-DMA (global_load_lds_dwordx4: no VGPRs),
//   each of the 4 waves issuing a quarter of the next chunk right after the barrier that opens the
//   current one; the fetch then has a whole chunk of MFMAs (~16k cycles) to land.  One barrier per
//   chunk: it proves (a) every wave's share of chunk c has landed (each waits vmcnt(0) first) and
//   (b) every wave is done reading chunk c-1, whose slot the next fetch overwrites.
//
// All positions (fragment index, chunk, slot, LDS offset) are compile-time constants of the fully
// unrolled layer chain; only the A fragments move: LDS -> 3-deep register ring -> MFMA.
#pragma once
#include <utility>

#include "ufr_device.h"

namespace ufr {

constexpr int kLdsPrefetch = 1;                          // register ring depth for LDS fragment reads
constexpr int kRingBytes = 2 * kChunkMaxFrags * 1024;    // two 32 KiB slots
constexpr int kVecBytes = vec_region_floats() * 4;       // bias / LayerNorm / view-token fragments, resident
constexpr int kStreamLdsBytes = kRingBytes + kVecBytes;

// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N) -- every layout number below
// must be a constant expression (the optimiser does not fold the constexpr table walks on its own)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct WStream {
  const char* src;    // packed blob (global, wave-uniform)
  char* ring;         // LDS: two chunk slots
  const f32x4* vecs;  // LDS: vector fragments
  int wave;           // wave index inside the workgroup (wave-uniform)
  int lane;
};

// workgroup prologue: vector fragments -> LDS (plain copy), first chunk of stream S in flight
template <int S, int NWAVES>
__device__ __forceinline__ WStream wstream_begin(const float* __restrict__ packed, char* smem) {
  WStream ws;
  ws.lane = threadIdx.x & 63;
  ws.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ws.src = reinterpret_cast<const char*>(packed);
  ws.ring = smem;
  f32x4* v = reinterpret_cast<f32x4*>(smem + kRingBytes);
  ws.vecs = v;
  constexpr int voff = vec_region_offset(), n4 = vec_region_floats() / 4;
  const f32x4* vs = reinterpret_cast<const f32x4*>(packed + voff);
  for (int i = threadIdx.x; i < n4; i += NWAVES * 64) v[i] = vs[i];
  __syncthreads();  // vector fragments may be read before the first chunk barrier
  return ws;
}

// fetch chunk CHK of stream S into its slot: this wave's share of the fragments
template <int S, int NWAVES, int CHK>
__device__ __forceinline__ void wstream_fetch(const WStream& ws) {
  constexpr int b = chunk_begin(S, CHK), n = chunk_begin(S, CHK + 1) - b;
  constexpr size_t goff = (size_t)stream_base_floats(S) * 4 + (size_t)b * 1024;
  constexpr int soff = (CHK & 1) * (kChunkMaxFrags * 1024);
  // the source addresses are loop invariant: keep them opaque (scalar zero re-read here) so they are
  // formed at the fetch instead of being hoisted out of the tile loop as ~80 live VGPR pairs
  int zero = 0;
  asm volatile("" : "+s"(zero));
  const char* g = ws.src + zero + goff + ws.wave * 1024 + ws.lane * 16;
  char* slot = ws.ring + soff + ws.wave * 1024;
  static_assert(n % NWAVES == 0, "chunks must split evenly over the fetching waves");
#pragma unroll
  for (int k = 0; k < n / NWAVES; ++k)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + k * NWAVES * 1024), (lds_ptr_t)(slot + k * NWAVES * 1024), 16, 0, 0);
}

// open chunk CHK: own DMA landed -> barrier -> start fetching chunk CHK+1 (wrapping to chunk 0 when `wrap`)
template <int S, int NWAVES, int CHK>
__device__ __forceinline__ void wstream_open(const WStream& ws, bool wrap) {
#ifdef UFR_ABL_NOBARRIER  // ablation build: no chunk hand-off at all (weights are garbage, timing only)
  (void)ws; (void)wrap;
  return;
#endif
#ifdef UFR_ABL_NODMA      // ablation build: barrier but no fetch
  __syncthreads();
  return;
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (CHK + 1 < stream_chunks(S)) {
    wstream_fetch<S, NWAVES, CHK + 1>(ws);
  } else {
    if (wrap) wstream_fetch<S, NWAVES, 0>(ws);
  }
}

template <int V>
__device__ __forceinline__ f32x4 vec_frag(const WStream& ws, int t, int g) {
  constexpr int base = (vec_offset(V) - vec_region_offset()) / 4;
  return ws.vecs[base + t * 4 + g];
}

// ---- static description of one GEMM inside its stream
template <int M>
struct GemmStages {
  static constexpr MatDesc d = mat_desc(M);
  static constexpr int S = mat_stream(M);
  static constexpr int OT = stream_ot(S);
  static constexpr int F0 = stream_mat_start(S, mat_stream_index(M));  // first fragment in the stream
  static constexpr int n_groups = (d.n_out + OT - 1) / OT;
  static constexpr int n_stages = n_groups * d.n_in;
  __host__ __device__ static constexpr int to0(int s) { return (s / d.n_in) * OT; }
  __host__ __device__ static constexpr int ti(int s) { return s % d.n_in; }
  __host__ __device__ static constexpr int no(int s) { return (d.n_out - to0(s)) < OT ? (d.n_out - to0(s)) : OT; }
  __host__ __device__ static constexpr int frag(int s) { return F0 + frag_in_mat(M, OT, to0(s), ti(s)); }
  __host__ __device__ static constexpr int chunk(int s) { return chunk_of(S, frag(s)); }
  __host__ __device__ static constexpr bool opens_chunk(int s) { return frag(s) == chunk_begin(S, chunk(s)); }
  // stage at which the LDS read of stage t is issued: PF ahead, but never before its chunk is open
  __host__ __device__ static constexpr int issue_stage(int t) {
    int u = t;
    while (u > 0 && u > t - kLdsPrefetch && chunk(u - 1) == chunk(t)) --u;
    return u;
  }
};

// out[c][to] += W_M x in[c][*] with the A fragments coming from the LDS stream.
//   C: token column tiles sharing each fragment; SWAP: activations in the A slot ([token][feature] result)
//   wrap: whether the stream restarts after its last chunk (another iteration follows)
template <int M, int C, int NWAVES, bool SWAP = false>
__device__ __forceinline__ void gemm_lds(const WStream& ws, const f32x4 (&in)[C][mat_desc(M).n_in],
                                         f32x4 (&out)[C][mat_desc(M).n_out], bool wrap) {
  using G = GemmStages<M>;
  constexpr MatDesc d = mat_desc(M);
  constexpr int OT = G::OT, PF = kLdsPrefetch;
  f32x4 ring[PF + 1][OT];
  const f32x4* lds = reinterpret_cast<const f32x4*>(ws.ring) + ws.lane;
  static_for<G::n_stages>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G::opens_chunk(s)) wstream_open<G::S, NWAVES, G::chunk(s)>(ws, wrap);
    static_for<PF + 1>([&](auto di) __attribute__((always_inline)) {
      constexpr int t = s + decltype(di)::value;
      if constexpr (t < G::n_stages) {
        if constexpr (G::issue_stage(t) == s) {
          constexpr int f = G::frag(t), c = G::chunk(t);
          constexpr int off = ((c & 1) * kChunkMaxFrags + (f - chunk_begin(G::S, c))) * 64;
          constexpr int no = G::no(t);
#pragma unroll
          for (int o = 0; o < no; ++o) ring[t % (PF + 1)][o] = lds[off + o * 64];
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    constexpr int to = G::to0(s), ti = G::ti(s), slot = s % (PF + 1), no = G::no(s), steps = in_steps(d.cm, ti);
#pragma unroll
    for (int r = 0; r < steps; ++r) {
#pragma unroll
      for (int o = 0; o < no; ++o) {
#pragma unroll
        for (int c = 0; c < C; ++c)
          out[c][to + o] = SWAP ? mfma16(in[c][ti][r], ring[slot][o][r], out[c][to + o])
                                : mfma16(ring[slot][o][r], in[c][ti][r], out[c][to + o]);
      }
    }
    // pin this stage's MFMAs here: they are pure, so IR-level sinking may otherwise drift them past the
    // following stages' LDS reads (seen at C = 1: fragments read, spilled, consumed hundreds of lines later)
#pragma unroll
    for (int o = 0; o < no; ++o)
#pragma unroll
      for (int c = 0; c < C; ++c) asm volatile("" : "+v"(out[c][to + o]));
  });
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace ufr
