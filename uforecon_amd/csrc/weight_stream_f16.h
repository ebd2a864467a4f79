// fp16x3 split-precision GEMM core on the 16-bit matrix cores, fed by an LDS weight stream
// (layout and arithmetic: ufr_layout_f16.h; streaming scheme: weight_stream.h).
#pragma once
#include <type_traits>

#include "ufr_layout_f16.h"
#include "weight_stream.h"

namespace ufr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_16x16x32_f16: 8 k-slots per lane
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kF16RingBytes = kF16Slots * kF16ChunkFrags * 1024;  // the ring: kF16Slots (3) slots of kF16ChunkFrags (12) KiB
constexpr int kF16LdsBytes = kF16RingBytes + kVecBytes;

// Split of a value pair into the fp16 planes of m x, m a power of two from the scale table (ScalarFile): hi = fp16(m x),
// lo = fp16(m x - hi) -- four v_fma_mix{lo,hi}_f16 (the scaling rides on the conversion's multiplier; m x - hi is exact in
// fp32).  For a layer's VALUES m = 2^a_M; for a producer's RAW accumulators (2^(s_P + a_P) times the value) m = 2^a_M 2^-(s_P
// + a_P): the exact descale costs nothing.  (hipcc builds the same arithmetic from C source with 5..7 instructions: it
// converts the second value twice rather than read a register's upper half.)
// The planes hold m |x| < 65504 only: beyond it hi overflows to inf, lo to -inf, and EVERY output of the layer for that
// token is NaN (the products w_hi hi and w_hi lo are infinities of opposite sign, or 0 x inf) -- probe_gemm.  a_M comes
// from an upper bound of the layer's input (prep.hip: weight_scale_kernel), so this happens only when the caller's
// stated input bound (ufr_weights_pack_for) was wrong.
#ifndef UFR_RANGE_MODE
#define UFR_RANGE_MODE 1   // 0: no range tracking (timing ablation)
#endif
__device__ __forceinline__ void split_pair(float a, float b, float m, unsigned& h, unsigned& l) {
  if (__builtin_constant_p(a) && __builtin_constant_p(b) && a == 0.f && b == 0.f) {   // padding registers of a tile: the
    h = l = 0u;                                                                      // compiler cannot fold the asm below
    return;
  }
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(m));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(m));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(m), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(m), "v"(h));
}

// B operands of one k-step (accumulator tiles 2s, 2s+1 of one column tile): one f16x8 per plane
struct BStep { f16x8 p[kPlanes]; };

// the same split into bf16 planes WITHOUT a scale (backward data-gradient chains: cotangents need the exponent range, and
// hi + lo = 16 significand bits are ample for the 1e-3 gradient tolerance): v_cvt_pk_bf16_f32, two shifts / masks to get
// hi back as fp32, two subtractions, v_cvt_pk_bf16_f32 -- 6 VALU instructions per pair
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_pair_bf16(float a, float b, unsigned& h, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
  const float ah = __builtin_bit_cast(float, h << 16), bh = __builtin_bit_cast(float, h & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a - ah, b - bh}, bf16x2));
}

// LOWP: the reduced-precision mode (ufr_set_matrix_precision): one 16-bit plane per operand, one MFMA per product
// BF16: the stream holds bf16 planes (ufr_layout_f16.h: B_VTB ...), the activations are split with split_pair_bf16
template <bool LOWP, bool BF16 = false>
struct WStreamF16T {
  static constexpr bool lowp = LOWP;
  static constexpr bool bf16 = BF16;
  const char* src;     // fp16 plane region of the packed blob (global, wave-uniform)
  char* ring;          // LDS: the ring of kF16Slots chunk slots
  unsigned ring_lds;   // ... as an LDS byte address (scalar)
  i32x4 rsrc;          // buffer descriptor of the plane region shifted by this wave's 1 KiB (lds_dma_piece)
  unsigned lds_wave;   // ring_lds + this wave's 1 KiB
  const f32x4* vecs;   // LDS: vector fragments (fp32)
  int wave, lane;
  unsigned long long bad_in, bad_out;   // sticky wave masks (scalar registers): probe_gemm / track_external
  f16x8 pre[kF16Depth][kPlanes];  // plane fragments of the next kF16Depth stages, in flight from LDS (stage s in slot s % depth)
};

typedef WStreamF16T<false> WStreamF16;
template <int NWAVES, bool LOWP = false, bool BF16 = false>
__device__ __forceinline__ WStreamF16T<LOWP, BF16> wstream_f16_begin(const float* __restrict__ packed, char* smem) {
  WStreamF16T<LOWP, BF16> ws;
  ws.lane = threadIdx.x & 63;
  ws.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ws.bad_in = ws.bad_out = 0ull;
  // a constexpr VARIABLE: left as a call, hipcc walks the layout tables on the device at every workgroup start (a
  // two-level scalar loop with dependent s_loads: ~1 500 scalar instructions, ~5 % of a 4-iteration workgroup's life)
  constexpr size_t kPlaneRegion = (size_t)blob_floats() * 4;
  ws.src = reinterpret_cast<const char*>(packed) + kPlaneRegion;
  ws.ring = smem;
  ws.ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  {
    const unsigned long long base = reinterpret_cast<unsigned long long>(ws.src) + (unsigned long long)ws.wave * 1024ull;
    ws.rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffull));
    ws.rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffffull));   // stride 0: raw buffer
    ws.rsrc[2] = -1;                                                                          // no bound (offsets are layout constants)
    ws.rsrc[3] = 0x00020000;                                                                  // gfx9 family: 32-bit data format
    ws.lds_wave = ws.ring_lds + (unsigned)ws.wave * 1024u;
  }
#ifdef UFR_ABL_NOLDS
  for (int d = 0; d < kF16Depth; ++d)
    for (int p = 0; p < kPlanes; ++p) ws.pre[d][p] = __builtin_bit_cast(f16x8, u32x4{0x3c003c00u + ws.lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
#endif
  f32x4* v = reinterpret_cast<f32x4*>(smem + kF16RingBytes);
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

// One LDS-DMA piece: 64 lanes x 16 bytes from g_uniform + lane_off (scalar base, 32-bit lane offset: no address
// VALU) to LDS address lds_addr + 16 lane.  Issued as inline assembly ON PURPOSE: with the compiler-visible
// __builtin_amdgcn_global_load_lds in the loop, hipcc's s_waitcnt insertion gives up counting LDS reads and drains
// lgkmcnt to 0 before every use of a weight fragment -- the read-ahead was worth nothing and the kernels ran
// 15..22 % slower (the "no DMA" ablation).  Untracked vector-memory operations only make the compiler's own
// vmcnt(N) waits conservative (loads return in order); the hand-off barrier waits for the pieces explicitly.
__device__ __forceinline__ void lds_dma_16(const char* g_uniform, unsigned lane_off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(g_uniform), "v"(lane_off), "s"(lds_addr) : "memory");
}
// The same piece through the BUFFER form: rsrc = the plane region shifted by the wave's 1 KiB, GOFF / LOFF = the piece's
// byte offsets in the region / in the ring as literals.  Two scalar instructions per piece (m0, soffset) instead of
// the eight to ten the 64-bit address of the global form costs (s_mov, s_ashr, three 64-bit adds, m0, s_nop): ~550
// fewer issue slots per view-transformer iteration.  The soffset move sits between the m0 write and the DMA, which is
// the wait state the hardware wants there.
#ifndef UFR_DMA_BUFFER
#define UFR_DMA_BUFFER 1
#endif
template <unsigned GOFF, unsigned LOFF>
__device__ __forceinline__ void lds_dma_piece(const i32x4& rsrc, unsigned lds_wave, unsigned lane_off) {
  unsigned soff;
  asm volatile("s_add_u32 m0, %3, %5\n\ts_mov_b32 %0, %4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
               : "=&s"(soff)
               : "v"(lane_off), "s"(rsrc), "s"(lds_wave), "n"(GOFF), "n"(LOFF)
               : "memory", "scc");
}

// fetch pieces [P0, P1) of this wave's share of chunk CHK of stream S into the chunk's ring slot (a piece = one
// 1 KiB LDS-DMA wave instruction; the wave's share is every NWAVES-th fragment)
template <int S, int NWAVES, int CHK, int P0 = 0, int P1 = kF16ChunkFrags / NWAVES, class WS>
__device__ __forceinline__ void wstream_f16_fetch(const WS& ws) {
  static_assert(kF16ChunkFrags % NWAVES == 0, "chunk must split evenly over the fetching waves");
  if constexpr (P1 > P0) {
    constexpr size_t goff = ((size_t)f16_stream_base_frags(S) + (size_t)CHK * kF16ChunkFrags) * 1024;
    constexpr int soff = (CHK % kF16Slots) * (kF16ChunkFrags * 1024);
    const unsigned lane_off = ws.lane * 16;
#if UFR_DMA_BUFFER
    static_assert(goff + (size_t)kF16ChunkFrags * 1024 < (1ull << 31), "buffer offsets are 32-bit");
    static_for<P1 - P0>([&](auto ki) __attribute__((always_inline)) {
      constexpr unsigned k = P0 + decltype(ki)::value;
      lds_dma_piece<(unsigned)goff + k * NWAVES * 1024u, (unsigned)soff + k * NWAVES * 1024u>(ws.rsrc, ws.lds_wave, lane_off);
    });
#else
    int zero = 0;
    asm volatile("" : "+s"(zero));  // keep the loop-invariant source address out of LICM's hands
    const char* g = ws.src + zero + goff + ws.wave * 1024;   // wave-uniform: scalar base + 32-bit lane offset
#pragma unroll
    for (int k = P0; k < P1; ++k) lds_dma_16(g + k * NWAVES * 1024, lane_off, ws.ring_lds + soff + ws.wave * 1024 + k * NWAVES * 1024);
#endif
  }
}

// hand-off barrier of chunk CHK: its fragments have landed and every wave is done with the chunk before it.
// The ring keeps kF16Slots-1 chunks in flight, so at most the (kF16Slots-2) younger fetches of this wave may still
// be outstanding when chunk CHK must have landed.
template <int S, int NWAVES, int CHK, class WS>
__device__ __forceinline__ void wstream_f16_barrier(const WS& ws, bool wrap) {
#ifdef UFR_ABL_NOBARRIER  // ablation builds (timing only, results are garbage): no hand-off at all / barrier without fetch
  (void)ws; (void)wrap;
  return;
#endif
#ifdef UFR_ABL_NODMA
  __syncthreads();
  return;
#endif
  constexpr int per_chunk = kF16ChunkFrags / NWAVES, ahead = kF16Slots - 1, n_chunks = f16_stream_chunks(S);
  constexpr int younger = (kF16Slots - 2) * per_chunk;
  // lgkmcnt(0): this wave's reads of the chunk whose slot is refilled next have returned (the stages are read
  // kF16Depth ahead, across the chunk boundary: the previous chunk's last stages now live in registers)
  if constexpr (younger == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  } else if constexpr (CHK + ahead <= n_chunks) {          // every younger fetch was issued unconditionally
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(younger) : "memory");
  } else {                                                 // the younger fetches were wrap-around ones
    if (wrap) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(younger) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// after the barrier of chunk CHK the slot of chunk CHK-1 is free: pieces [P0, P1) of the refill (chunk CHK + ring
// depth - 1, or its wrap-around into the next pass).  The pieces of one refill are spread over the stages of chunk
// CHK: an LDS-DMA instruction stalls its wave for 60..185 cycles (MI355X_MICROARCH.md), several in a row for longer.
template <int S, int NWAVES, int CHK, int P0 = 0, int P1 = kF16ChunkFrags / NWAVES, class WS>
__device__ __forceinline__ void wstream_f16_refill(const WS& ws, bool wrap) {
#if defined(UFR_ABL_NOBARRIER) || defined(UFR_ABL_NODMA)
  (void)ws; (void)wrap;
  return;
#endif
  constexpr int ahead = kF16Slots - 1, n_chunks = f16_stream_chunks(S);
  if constexpr (CHK + ahead < n_chunks) {
    wstream_f16_fetch<S, NWAVES, CHK + ahead, P0, P1>(ws);
  } else {
    if (wrap) wstream_f16_fetch<S, NWAVES, (CHK + ahead) % n_chunks, P0, P1>(ws);
  }
}

template <int S, int NWAVES, int CHK, class WS>
__device__ __forceinline__ void wstream_f16_open(const WS& ws, bool wrap) {
  wstream_f16_barrier<S, NWAVES, CHK>(ws, wrap);
  wstream_f16_refill<S, NWAVES, CHK>(ws, wrap);
}

// end of a pass over stream S: open its padding chunks (none for most streams) so the wrap-around fetches go out
template <int S, int NWAVES, class WS>
__device__ __forceinline__ void wstream_f16_finish(const WS& ws, bool wrap) {
  constexpr int real = f16_stream_real_chunks(S), pad = f16_stream_chunks(S) - real;
  static_for<pad>([&](auto ci) __attribute__((always_inline)) { wstream_f16_open<S, NWAVES, real + decltype(ci)::value>(ws, wrap); });
}

// start of a pass over stream S: its first kF16Slots-1 chunks.  The slots must be free: at kernel start, or
// after every wave has passed the barrier that opened the previous stream's last chunk with wrap == false
// (then slot 0.. are no longer read; the last chunk's own slot is (n_chunks-1) % kF16Slots = kF16Slots-1).
template <int S, int NWAVES, class WS>
__device__ __forceinline__ void wstream_f16_prime(const WS& ws) {
  static_for<kF16Slots - 1>([&](auto ci) __attribute__((always_inline)) { wstream_f16_fetch<S, NWAVES, decltype(ci)::value>(ws); });
}

// Range tracking of the dense-layer inputs.  An input beyond the planes' range (or infinite) makes every accumulator of its token
// column non-finite (split_pair), so one accumulator element per column and GEMM is classified (one v_cmp_class per column
// tile) right after the last k-step -- before a ReLU, whose v_max_f32 returns the other operand for a NaN -- and the
// wave's verdict is OR-ed into a SCALAR sticky mask: the transformer kernels sit at their 256-register budget, a vector
// register carried across the layer chain costs 26..100 spilled registers.  (Round 3 first scanned the inputs: one
// v_max3 per value pair, 4 % of the view kernel's vector instructions.)  The kernel looks at the mask once at its end
// and raises the device's sticky range status (include/ufr.h: ufr_status_poll): an overflow is reported, never rendered.
// NaN from OUTSIDE is told apart by testing the externally supplied tiles where they are loaded (track_external): one
// unordered compare per value pair.  Internally every divisor is positive and every exponent non-positive.
template <int C, int N, class WS>
__device__ __forceinline__ void probe_gemm(const f32x4 (&out)[C][N], WS& ws) {
#if UFR_RANGE_MODE != 0
  bool bad = false;
#pragma unroll
  for (int c = 0; c < C; ++c) bad |= __builtin_amdgcn_class(out[c][0][0], 0x207);   // sNaN | qNaN | -inf | +inf
  ws.bad_in |= __builtin_amdgcn_ballot_w64(bad);
#endif
}
template <int C, int N, class WS>
__device__ __forceinline__ void track_external(const f32x4 (&t)[C][N], WS& ws) {
#if UFR_RANGE_MODE != 0
  bool nan = false;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i)
      nan |= __builtin_isunordered(t[c][i][0], t[c][i][1]) | __builtin_isunordered(t[c][i][2], t[c][i][3]);
  ws.bad_out |= __builtin_amdgcn_ballot_w64(nan);
#endif
}

// end of a kernel: raise the sticky range status (bit 0: a dense layer produced non-finite accumulators, i.e. one of its
// inputs left the planes' range; bit 1: NaN among the kernel's inputs -- which also raises bit 0).  One atomic per offending wave.
template <class WS>
__device__ __forceinline__ void wstream_report_range(const WS& ws, int* __restrict__ status) {
  const int bits = (ws.bad_in ? 1 : 0) | (ws.bad_out ? 2 : 0);
  if (bits && ws.lane == 0) atomicOr(status, bits);
}

template <int V, class WS>
__device__ __forceinline__ f32x4 vec_frag(const WS& ws, int t, int g) {
  constexpr int base = (vec_offset(V) - vec_region_offset()) / 4;
  return ws.vecs[base + t * 4 + g];
}

// A kernel's scalar list (ufr_layout.h: ViewScalar / RayScalar) in ONE vector register: lane k = scalar k; a scalar is
// taken with v_readlane where it is needed.
// LOCAL = false: the reads are loop-invariant, the compiler hoists them and the scalars live in scalar registers for the
// whole launch (22 of them: fine where the kernel has the room).
// LOCAL = true: every read goes through an opaque copy of the register, so it stays where it is written and the scalar
// lives for one phase.  The straddling L = 6 view kernel needs this: with 22 more long-lived scalars it spilled 23
// scalar registers, and the machine scheduler, seeing the scalar pressure above the budget, fell back to its
// register-saving mode -- which serialised the attention's cross-lane exchanges through a single temporary
// (339 instead of 53 s_waitcnt lgkmcnt(0) per iteration; the kernel ran 15 % slower).
__device__ __forceinline__ float uniform_f32(float v) {   // a wave-uniform value into a scalar register
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <bool LOCAL>
struct ScalarFile {
  float sv;
  __device__ __forceinline__ float operator[](int k) const {
    float v = sv;
    if constexpr (LOCAL) asm volatile("" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
  }
};
template <bool LOCAL, int OFFSET, class WS>
__device__ __forceinline__ ScalarFile<LOCAL> scalar_file(const WS& ws) {
  constexpr int base = OFFSET - vec_region_offset();
  return ScalarFile<LOCAL>{reinterpret_cast<const float*>(ws.vecs)[base + (ws.lane & (kKernelScalars - 1))]};
}

__device__ __forceinline__ f32x4 mfma_f16(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// the operands' 16-bit words are fp16 (BF = false) or bf16 (BF = true); same lane layout, same rate
template <bool BF>
__device__ __forceinline__ f32x4 mfma_planes(const f16x8& a, const f16x8& b, const f32x4& c) {
  if constexpr (BF)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// One panel (k-step S of matrix M): out[c][to] += W[:, 32 S .. 32 S + 31] x b[c], all out tiles.
// Panels must be executed in stream order (ufr_layout_f16.h: vt_panel).
// hook(integral_constant<to>) is VALU work without dependence on this panel (the split of the NEXT k-step's
// operands): it is interleaved with the stage's MFMAs -- the 16-bit matrix pipe runs ~2 independent VALU
// instructions per MFMA for free (tools/dev/mfma_valu2), so the split costs nothing once it sits there.
#ifndef UFR_HOOK_VALU
#define UFR_HOOK_VALU 6   // VALU instructions of the hook issued after each MFMA (2 until round 6: see UFR_VT_SCORES_HOOK)
#endif
constexpr int kProducts = 3;   // MFMAs per fp32 product
#ifndef UFR_SETPRIO
#define UFR_SETPRIO 1          // s_setprio level of a wave inside a GEMM panel (0 outside)
#endif
#ifndef UFR_F16_SPREAD
#define UFR_F16_SPREAD 1       // 1: a chunk's refill pieces are spread over its stages; 0: all right after the barrier
#endif
struct NoHook {
  template <class T> __device__ __forceinline__ void operator()(T) const {}
};
// STREAM: the stream the panel is read from (default: the matrix's own; the ray transformer's k / v panels also sit in the
// backward stream B_RTB2).  The planes' type -- fp16 with scales, or bf16 -- is the matrix's (f16_mat_is_bf16).
template <int M, int S, int C, int NWAVES, bool SWAP = false, int STREAM = -1, class Hook = NoHook, class WS = WStreamF16>
__device__ __forceinline__ void gemm_f16_panel(WS& ws, const BStep (&b)[C], f32x4 (&out)[C][mat_desc(M).n_out],
                                              bool wrap, Hook&& hook = NoHook{}) {
  constexpr bool LOWP = WS::lowp;
  constexpr bool BF = f16_mat_is_bf16(M);
  constexpr int n_planes = LOWP ? 1 : kPlanes, n_products = LOWP ? 1 : kProducts;
  constexpr int n_out = mat_desc(M).n_out, ST = STREAM >= 0 ? STREAM : f16_mat_stream(M);
  static_assert(f16_panel_index(M, S, ST) >= 0, "not a panel of the stream");
  constexpr int F0 = f16_panel_start(ST, f16_panel_index(M, S, ST));
  const f16x8* lds = reinterpret_cast<const f16x8*>(ws.ring) + ws.lane;
  // inside a GEMM panel the wave wins issue arbitration over a partner that is in a VALU-only phase (measured: view
  // transformer -1.7 %, ray transformer -1.4 % alone, 0.1 % on the whole frame with the gather beside them; priority 3
  // no better, the inverse scheme no effect)
  __builtin_amdgcn_s_setprio(UFR_SETPRIO);
  static_for<n_out>([&](auto ti) __attribute__((always_inline)) {
    constexpr int to = decltype(ti)::value;
    constexpr int f = F0 + to * kPlanes;                 // first of the stage's plane fragments
    constexpr int sidx = f / kPlanes, n_stages = f16_stream_frags(ST) / kPlanes;
    __builtin_amdgcn_sched_barrier(0);
    // stage s of the pass is read from LDS while stage s - kF16Depth computes; a chunk is opened (hand-off barrier)
    // right before its first stage is read, i.e. kF16Depth stages before it is needed
    auto read_stage = [&](auto si) __attribute__((always_inline)) {
      constexpr int s2 = decltype(si)::value, f2 = s2 * kPlanes;
      constexpr int chk = f2 / kF16ChunkFrags, in_chk = f2 % kF16ChunkFrags;
      constexpr int base = ((chk % kF16Slots) * kF16ChunkFrags + in_chk) * 64;
      // the chunk's barrier before its first stage is read; the refill it allows goes out piecewise with the stages
      constexpr int in_frags = f16_stream_frags(ST) - chk * kF16ChunkFrags;
      constexpr int n_st = (in_frags < kF16ChunkFrags ? in_frags : kF16ChunkFrags) / kPlanes, j = in_chk / kPlanes;
      constexpr int pieces = kF16ChunkFrags / NWAVES;
      if constexpr (in_chk == 0) wstream_f16_barrier<ST, NWAVES, chk>(ws, wrap);
      wstream_f16_refill<ST, NWAVES, chk, UFR_F16_SPREAD ? j * pieces / n_st : (j == 0 ? 0 : pieces),
                         UFR_F16_SPREAD ? (j + 1) * pieces / n_st : pieces>(ws, wrap);
#ifndef UFR_ABL_NOLDS   // ablation (timing only): the weight fragments are never read from LDS
#pragma unroll
      for (int p = 0; p < n_planes; ++p) ws.pre[s2 % kF16Depth][p] = lds[base + p * 64];
#endif
    };
    if constexpr (sidx == 0)                             // start of a pass: nothing is in flight yet
      static_for<kF16Depth>([&](auto di) __attribute__((always_inline)) { read_stage(di); });
    f16x8 a[kPlanes];
#pragma unroll
    for (int p = 0; p < n_planes; ++p) a[p] = ws.pre[sidx % kF16Depth][p];
    if constexpr (sidx + kF16Depth < n_stages) read_stage(std::integral_constant<int, sidx + kF16Depth>{});
    __builtin_amdgcn_sched_barrier(0);
    hook(ti);
    // three plane pairs (lo.lo is dropped), small terms first (0 = hi, 1 = lo); SWAP: activations in the A slot
    // the three products of a column tile back to back (same accumulator, same results bit for bit): 1.5 % faster than
    // alternating the column tiles between the products -- the accumulate chain stays in the matrix pipe
#pragma unroll
    for (int c = 0; c < C; ++c)
      static_for<n_products>([&](auto pi) __attribute__((always_inline)) {
        constexpr int pw[kProducts] = {1, 0, 0}, px[kProducts] = {0, 1, 0};
        constexpr int w = LOWP ? 0 : pw[decltype(pi)::value], x = LOWP ? 0 : px[decltype(pi)::value];
        out[c][to] = SWAP ? mfma_planes<BF>(b[c].p[x], a[w], out[c][to]) : mfma_planes<BF>(a[w], b[c].p[x], out[c][to]);
      });
    if constexpr (!std::is_same<std::decay_t<Hook>, NoHook>::value) {
      // issue order: one MFMA, then up to two of the hook's VALU instructions, repeated
#pragma unroll
      for (int i = 0; i < n_products * C; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, UFR_HOOK_VALU, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(out[c][to]));  // pin: pure MFMAs are otherwise sunk past later LDS reads
  });
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(0);
}

// ---- pipelined operand split: unit u = (c, tile half, value pair q) of a k-step, 4 VALU instructions
template <int C>
struct BWords { unsigned w[C][kPlanes][4]; };

// units [U0, U1) of k-step S of the tiles in[c][0..NIN), scaled by m (split_pair; bf16 planes carry no scale)
template <int S, int U0, int U1, bool BF = false, int C, int NIN>
__device__ __forceinline__ void split_units(const f32x4 (&in)[C][NIN], BWords<C>& bw, float m) {
  static_for<U1 - U0>([&](auto ui) __attribute__((always_inline)) {
    constexpr int u = U0 + decltype(ui)::value;
    constexpr int c = u / 4, half = (u >> 1) & 1, q = u & 1, tile = 2 * S + half;
    if constexpr (tile < NIN) {
      if constexpr (BF) split_pair_bf16(in[c][tile][2 * q], in[c][tile][2 * q + 1], bw.w[c][0][2 * half + q], bw.w[c][1][2 * half + q]);
      else split_pair(in[c][tile][2 * q], in[c][tile][2 * q + 1], m, bw.w[c][0][2 * half + q], bw.w[c][1][2 * half + q]);
    } else {
      bw.w[c][0][2 * half + q] = bw.w[c][1][2 * half + q] = 0u;
    }
  });
}

template <int C>
__device__ __forceinline__ void bwords_to_bstep(const BWords<C>& bw, BStep (&b)[C]) {
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int p = 0; p < kPlanes; ++p)
      b[c].p[p] = __builtin_bit_cast(f16x8, u32x4{bw.w[c][p][0], bw.w[c][p][1], bw.w[c][p][2], bw.w[c][p][3]});
}

// elu(value) + 1 of a raw accumulator a = value / dsc (dsc a power of two; dsc_l2e = dsc log2(e), exact): bit-identical
// to elu1(dsc * a), the scale rides on the fma / on the exponent's log2(e) multiply
__device__ __forceinline__ float elu1_acc(float a, float dsc, float dsc_l2e) {
#ifdef UFR_ACCURATE_EXP
  return a > 0.f ? __builtin_fmaf(a, dsc, 1.f) : expf(a * dsc);
#else
  return a > 0.f ? __builtin_fmaf(a, dsc, 1.f) : __builtin_amdgcn_exp2f(dsc_l2e * a);
#endif
}
constexpr float kLog2e = 0x1.715476p+0f;

// out += 2^(s_M + a_M) W_M x in over all k-steps of M, probed for the range (probe_gemm); the output stays a RAW
// accumulator.  in[c][0..NIN) are the producer's fp32 tiles and m the multiplier that turns them into the planes'
// 2^a_M x: 2^a_M for values, xs * (the producer's dsc) for the producer's raw accumulators (ReLU commutes with
// the scale; LayerNorm takes raw accumulators with a scaled epsilon, elu / the attention fold the factor into a multiply
// they do anyway: no layer of the two chains pays for a descale pass).  Callers that start from a bias pass it
// pre-multiplied by 2^(s_M + a_M) (exact).  bf16 matrices (the transposed ones of the backward): no scales, m unused.
// The fp16 split of k-step s+1 is interleaved with the MFMAs of k-step s (only step 0's is exposed).
template <int M, int C, int NWAVES, int STREAM = -1, int NIN, class WS, class UHook = NoHook>
__device__ __forceinline__ void gemm_f16(WS& ws, const f32x4 (&in)[C][NIN], f32x4 (&out)[C][mat_desc(M).n_out],
                                        bool wrap, float m = 1.f, UHook&& uhook = NoHook{}) {
  // uhook(integral_constant<k-step * n_out + out tile>): the CALLER's VALU work that does not depend on this GEMM, issued
  // with the stage's MFMAs like the operand split (round 6 experiment: the attention scores under the v GEMM)
  static_assert(NIN == mat_desc(M).n_in, "input tile count");
  constexpr int n_out = mat_desc(M).n_out, NU = 4 * C;
  constexpr bool BF = f16_mat_is_bf16(M);   // bf16 planes: no scales, so the output is exact as it stands and never probed
  constexpr bool kUser = !std::is_same<std::decay_t<UHook>, NoHook>::value;
  BWords<C> cur;
  split_units<0, 0, NU, BF>(in, cur, m);
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
      gemm_f16_panel<M, s, C, NWAVES, false, STREAM>(ws, b, out, wrap, [&](auto ti) __attribute__((always_inline)) {
        constexpr int to = decltype(ti)::value;
        split_units<s + 1, to * NU / n_out, (to + 1) * NU / n_out, BF>(in, nxt, m);
        if constexpr (kUser) uhook(std::integral_constant<int, s * n_out + to>{});
      });
      cur = nxt;
    } else if constexpr (kUser) {
      gemm_f16_panel<M, s, C, NWAVES, false, STREAM>(ws, b, out, wrap, [&](auto ti) __attribute__((always_inline)) {
        uhook(std::integral_constant<int, s * n_out + decltype(ti)::value>{});
      });
      if constexpr (s + 1 < ksteps(M)) split_units<s + 1, 0, NU, BF>(in, cur, m);
    } else {
      gemm_f16_panel<M, s, C, NWAVES, false, STREAM>(ws, b, out, wrap);
      if constexpr (s + 1 < ksteps(M)) split_units<s + 1, 0, NU, BF>(in, cur, m);
    }
  });
  if constexpr (!BF) probe_gemm(out, ws);
}

}  // namespace ufr
