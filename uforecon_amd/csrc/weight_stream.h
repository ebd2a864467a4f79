// Shared pieces of the LDS weight streams (the streaming scheme itself: weight_stream_f16.h).
//
// Every wave of a workgroup walks the same static sequence of 1 KiB operand fragments.  Instead of each wave
// pulling its own copy from L2 (measured: 81-84 % of the MFMA rate at 2 waves/SIMD because 256 CUs x 8 waves
// hammer the same L2 lines), the workgroup stages the stream through LDS in chunks:
//
//   chunk c -> LDS slot c % slots, fetched with lane-linear LDS-DMA (global_load_lds_dwordx4: no VGPRs),
//   each of the 4 waves issuing a quarter of the next chunk right after the barrier that opens the
//   current one.  One barrier per chunk: it proves (a) every wave's share of chunk c has landed (each
//   waits vmcnt first) and (b) every wave is done reading the chunk whose slot the next fetch overwrites.
//
// All positions (fragment index, chunk, slot, LDS offset) are compile-time constants of the fully
// unrolled layer chain; only the fragments move: LDS -> registers -> MFMA.
// (The first version of the kernels streamed fp32 A-fragments for v_mfma_f32_16x16x4_f32 through the same
// scheme; ufr_layout.h still describes that fragment order -- the CPU layout tests use it -- but the kernels
// now read only the vector fragments of that region.)
#pragma once
#include <utility>

#include "ufr_device.h"

namespace ufr {

constexpr int kVecBytes = vec_region_floats() * 4;       // bias / LayerNorm / view-token fragments, resident in LDS

// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N) -- every layout number
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

}  // namespace ufr
