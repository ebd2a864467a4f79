// Backward of pre_sim_mlp (the 8 -> 32 -> 32 -> 16 MLP on the pair similarities, code1/ray_transformer.py:128-132, 268):
// weight gradients only -- its input (sim8) is saved by the forward and carries no gradient.  The backward of the
// along-ray aggregation itself moved to ray_dgrad.hip / wgrad_stream.hip in round 4 (bwd_tape.h); this small kernel keeps
// the LDS-tile machinery of bwd_common.h (tiles of 32 points, fp32 MFMA, register-resident weight-gradient tiles).
#include "bwd_common.h"
#include "ufr_internal.h"

namespace ufr {

namespace pb {
enum : int { O_S8 = 0, O_A1 = 8, O_A2 = 40, O_DO = 72, O_DA2 = 88, O_DA1 = 120, O_END = 152 };
constexpr WgMat kMats[] = {{P_PS_W4, 16, 32, O_DO, O_A2}, {P_PS_W2, 32, 32, O_DA2, O_A1}, {P_PS_W0, 32, 8, O_DA1, O_S8}};
constexpr auto kList = make_wglist(kMats);   // 2 + 4 + 2 = 8 tiles: two slots per wave
}  // namespace pb

template <bool LOWP>
__global__ void __launch_bounds__(kBwdThreads) presim_bwd_kernel(RawPtrs wp, GradPtrs gp, const float* __restrict__ sim8,
                                                                 const float* __restrict__ d_pv, int P) {
  using namespace pb;
  __shared__ float lds[O_END * kLD];
  const int tid0 = threadIdx.x, wave = tid0 >> 6, lane = tid0 & 63;
  int tid = tid0;   // re-laundered after every barrier (bwd_common.h: opaque)
  f32x4 acc[2] = {splat4(0.f), splat4(0.f)};   // 8 tiles over the waves: at most 2 per wave
  const auto wg_tab = wgrad_table<pb::kList, 3, 2>(wave, lane);
  float accB = 0.f;
  auto R = [&](int row) -> float* { return lds + row * kLD; };
  const int n_tiles = (P + kTT - 1) / kTT;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int p0 = tile * kTT;
    for (int idx = tid; idx < kTT * 24; idx += kBwdThreads) {
      if (idx < kTT * 8) {
        const int c = idx >> 3, i = idx & 7, p = p0 + c;
        R(O_S8 + i)[c] = p < P ? sim8[(size_t)p * 8 + i] : 0.f;
      } else {
        const int c = (idx - kTT * 8) >> 4, i = (idx - kTT * 8) & 15, p = p0 + c;
        R(O_DO + i)[c] = p < P ? d_pv[(size_t)p * 40 + 24 + i] : 0.f;
      }
    }
    auto pf14 = gemm_prefetch<32, 8, false>(wp.p[P_PS_W0], 8, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 8, false, LOWP>(pf14, wp.p[P_PS_W0], 8, R(O_S8), wave, lane,
                           [&](int r, int c, float v) { R(O_A1 + r)[c] = fmaxf(v + wp.p[P_PS_B0][r], 0.f); });
    auto pf15 = gemm_prefetch<32, 32, false>(wp.p[P_PS_W2], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 32, false, LOWP>(pf15, wp.p[P_PS_W2], 32, R(O_A1), wave, lane,
                            [&](int r, int c, float v) { R(O_A2 + r)[c] = fmaxf(v + wp.p[P_PS_B2][r], 0.f); });
    auto pf16 = gemm_prefetch<32, 16, true>(wp.p[P_PS_W4], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 16, true, LOWP>(pf16, wp.p[P_PS_W4], 32, R(O_DO), wave, lane,
                           [&](int r, int c, float v) { R(O_DA2 + r)[c] = R(O_A2 + r)[c] > 0.f ? v : 0.f; });
    auto pf17 = gemm_prefetch<32, 32, true>(wp.p[P_PS_W2], 32, wave, lane, 0);
    __syncthreads();
    tid = opaque(tid0);
    gemm_compute<32, 32, true, LOWP>(pf17, wp.p[P_PS_W2], 32, R(O_DA2), wave, lane,
                           [&](int r, int c, float v) { R(O_DA1 + r)[c] = R(O_A1 + r)[c] > 0.f ? v : 0.f; });
    __syncthreads();
    tid = opaque(tid0);
    if (tid < 16) accB += row_dot(R(O_DO + tid), nullptr, 0);
    else if (tid < 48) accB += row_dot(R(O_DA2 + (tid - 16)), nullptr, 0);
    else if (tid < 80) accB += row_dot(R(O_DA1 + (tid - 48)), nullptr, 0);
    wgrad_all<2, 0, LOWP>(acc, lds, wg_tab, lane);
    __syncthreads();
    tid = opaque(tid0);
  }
  wgrad_flush_all<pb::kList, 3, 2, 0>(acc, gp, wave, lane);
  if (tid < 16) atomic_add_f32(gp.p[P_PS_B4] + tid, accB);
  else if (tid < 48) atomic_add_f32(gp.p[P_PS_B2] + (tid - 16), accB);
  else if (tid < 80) atomic_add_f32(gp.p[P_PS_B0] + (tid - 48), accB);
}

hipError_t launch_presim_bwd(const RawPtrs& wp, const GradPtrs& gp, const float* sim8, const float* d_pv, int P, bool lowp,
                             hipStream_t s) {
  const int n_tiles = (P + kTT - 1) / kTT;
  const dim3 grid(n_tiles < 256 ? n_tiles : 256), block(kBwdThreads);
  if (lowp) hipLaunchKernelGGL(presim_bwd_kernel<true>, grid, block, 0, s, wp, gp, sim8, d_pv, P);
  else hipLaunchKernelGGL(presim_bwd_kernel<false>, grid, block, 0, s, wp, gp, sim8, d_pv, P);
  return hipGetLastError();
}

}  // namespace ufr
