// One post-norm layer of the feature-matching transformer, d_model = 32, 8 heads of 4:
//   EncoderLayer.forward    code1/encoder_utils/fmt/FMT.py:99-113   x = LN1(x + out_proj(attn(x, src, src)));
//                                                                    out = LN2(x + linear2(relu(linear1(x))))
//   AttentionLayer.forward  FMT.py:63-79                             q/k/v projections with bias
//   LinearAttention.forward FMT.py:25-38                             phi = elu + 1; KV_h = sum_s phi(k_s)^T v_s (4x4 per head),
//                                                                    out_l = phi(q_l) KV_h / (phi(q_l) . sum_s phi(k_s) + eps)
// The layer is a global reduction over the source tokens (the 8 x (16 + 4) state per sample) followed by a purely
// per-token map, so it is two kernels (+ a tiny fixed-order sum): `fmt_state_kernel` (every thread folds a few source tokens
// into 160 register accumulators, wave reduction, one PARTIAL per wave -- summed in a fixed order by `fmt_state_sum_kernel`,
// not with float atomics: the frame encoder must give the same bits in every process, or a frame rendered by two ranks
// would not equal the same frame rendered by one) and `fmt_apply_kernel` (one token per thread: 6 272 FMAs with
// the weights broadcast from LDS).  d = 32 is too narrow for the matrix cores to pay (a 16x16x4 tile chain would be all
// latency); on the VALU the whole stage of a 512x640 3-view frame (about 70 layer passes over 20 480 tokens) is ~9 GFMA.
#include "ufr_device.h"
#include "ufr_internal.h"

namespace ufr {

constexpr int kFmtD = 32, kFmtH = 8, kFmtHD = 4, kFmtFF = 64;
constexpr int kFmtState = kFmtH * (kFmtHD * kFmtHD + kFmtHD);   // 160: KV_h[m][d] then sum K'_h[d]


__device__ __forceinline__ float dot32(const float* __restrict__ w /* LDS, 32 floats */, const float (&x)[32]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i += 4) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + i);     // same address in every lane: LDS broadcast
    s = fmaf(w4[0], x[i], s);
    s = fmaf(w4[1], x[i + 1], s);
    s = fmaf(w4[2], x[i + 2], s);
    s = fmaf(w4[3], x[i + 3], s);
  }
  return s;
}

__device__ __forceinline__ void load_token(const float* __restrict__ p, float (&x)[32]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = ld4(p + 4 * i);
    x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
  }
}

constexpr int kFmtTokPerBlock = 1024;   // source tokens folded by one workgroup of fmt_state_kernel

// state[n][h*16 + m*4 + d] = sum_s phi(k_s)[4h+d] v_s[4h+m];  state[n][128 + 4h + d] = sum_s phi(k_s)[4h+d]
// thread = (token slot, head): 8 adjacent lanes share a token (one broadcast load) and each owns one head's 16 + 4 sums
// partial[n][part = 4 blockIdx.x + wave][160]
__global__ void __launch_bounds__(256) fmt_state_kernel(FmtWeights w, const float* __restrict__ src, int S,
                                                         float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sw[2 * kFmtD * kFmtD + 2 * kFmtD];
  float* swk = sw;
  float* swv = sw + kFmtD * kFmtD;
  float* sbk = swv + kFmtD * kFmtD;
  float* sbv = sbk + kFmtD;
  for (int i = threadIdx.x; i < kFmtD * kFmtD; i += 256) { swk[i] = w.wk[i]; swv[i] = w.wv[i]; }
  if (threadIdx.x < kFmtD) { sbk[threadIdx.x] = w.bk[threadIdx.x]; sbv[threadIdx.x] = w.bv[threadIdx.x]; }
  __syncthreads();
  const int n = blockIdx.y, h = threadIdx.x & 7, slot = threadIdx.x >> 3;
  float kv[16], ks[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) kv[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) ks[i] = 0.f;
  const int t_end = min(S, (int)(blockIdx.x + 1) * kFmtTokPerBlock);
  for (int t = blockIdx.x * kFmtTokPerBlock + slot; t < t_end; t += 32) {
    float x[32];
    load_token(src + ((size_t)n * S + t) * kFmtD, x);
    float kp[kFmtHD], vv[kFmtHD];
#pragma unroll
    for (int d = 0; d < kFmtHD; ++d) {
      kp[d] = elu1(dot32(swk + (kFmtHD * h + d) * kFmtD, x) + sbk[kFmtHD * h + d]);
      vv[d] = dot32(swv + (kFmtHD * h + d) * kFmtD, x) + sbv[kFmtHD * h + d];
    }
#pragma unroll
    for (int m = 0; m < kFmtHD; ++m)
#pragma unroll
      for (int d = 0; d < kFmtHD; ++d) kv[m * 4 + d] = fmaf(kp[d], vv[m], kv[m * 4 + d]);
#pragma unroll
    for (int d = 0; d < kFmtHD; ++d) ks[d] += kp[d];
  }
  // sum over the wave's 8 token slots (lane bits 3..5), then one atomic per value from the slot-0 lanes
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    kv[i] += __shfl_xor(kv[i], 8);
    kv[i] += __shfl_xor(kv[i], 16);
    kv[i] += __shfl_xor(kv[i], 32);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ks[i] += __shfl_xor(ks[i], 8);
    ks[i] += __shfl_xor(ks[i], 16);
    ks[i] += __shfl_xor(ks[i], 32);
  }
  if ((threadIdx.x & 63) < 8) {
    float* dst = partial + ((size_t)n * (gridDim.x * 4) + blockIdx.x * 4 + (threadIdx.x >> 6)) * kFmtState;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[h * 16 + i] = kv[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[128 + 4 * h + i] = ks[i];
  }
}

// state[n][i] = sum over the parts in index order
__global__ void __launch_bounds__(kFmtState) fmt_state_sum_kernel(const float* __restrict__ partial, int parts, float* __restrict__ state) {
  const int n = blockIdx.x, i = threadIdx.x;
  const float* p = partial + (size_t)n * parts * kFmtState + i;
  float s = 0.f;
  for (int k = 0; k < parts; ++k) s += p[(size_t)k * kFmtState];
  state[(size_t)n * kFmtState + i] = s;
}

template <int D>
__device__ __forceinline__ void layer_norm_inplace(float (&y)[D], const float* g, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) s += y[i];
  const float mean = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float c = y[i] - mean;
    q = fmaf(c, c, q);
  }
  const float rstd = 1.f / sqrtf(q * (1.f / D) + 1e-5f);
#pragma unroll
  for (int i = 0; i < D; ++i) y[i] = fmaf((y[i] - mean) * rstd, g[i], b[i]);
}

__global__ void __launch_bounds__(256) fmt_apply_kernel(FmtWeights w, const float* __restrict__ x_in, int T,
                                                         const float* __restrict__ state, float* __restrict__ out) {
  // LDS: wq | wo | w1 | w2 | bq bo b1 b2 | n1w n1b n2w n2b | state
  __shared__ __attribute__((aligned(16))) float sw[2 * 1024 + 2 * 2048 + 32 + 32 + 64 + 32 + 4 * 32 + kFmtState];
  float* swq = sw;
  float* swo = swq + 1024;
  float* sw1 = swo + 1024;
  float* sw2 = sw1 + 2048;
  float* sbq = sw2 + 2048;
  float* sbo = sbq + 32;
  float* sb1 = sbo + 32;
  float* sb2 = sb1 + 64;
  float* sn = sb2 + 32;           // n1w n1b n2w n2b
  float* sst = sn + 128;
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < 1024; i += 256) { swq[i] = w.wq[i]; swo[i] = w.wo[i]; }
  for (int i = threadIdx.x; i < 2048; i += 256) {
    sw1[i] = w.w1[i];
    sw2[(i & 63) * 32 + (i >> 6)] = w.w2[i];      // transposed to [j][o]: the 32 outputs of hidden unit j are contiguous
  }
  if (threadIdx.x < 32) {
    const int i = threadIdx.x;
    sbq[i] = w.bq[i]; sbo[i] = w.bo[i]; sb2[i] = w.b2[i];
    sn[i] = w.n1w[i]; sn[32 + i] = w.n1b[i]; sn[64 + i] = w.n2w[i]; sn[96 + i] = w.n2b[i];
  }
  if (threadIdx.x < 64) sb1[threadIdx.x] = w.b1[threadIdx.x];
  if (threadIdx.x < kFmtState) sst[threadIdx.x] = state[(size_t)n * kFmtState + threadIdx.x];
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float x[32];
  load_token(x_in + ((size_t)n * T + t) * kFmtD, x);
  // attention message of this token (FMT.py:36-38)
  float att[32];
#pragma unroll
  for (int h = 0; h < kFmtH; ++h) {
    float qp[kFmtHD], den = 1e-6f;
#pragma unroll
    for (int d = 0; d < kFmtHD; ++d) {
      qp[d] = elu1(dot32(swq + (kFmtHD * h + d) * kFmtD, x) + sbq[kFmtHD * h + d]);
      den = fmaf(qp[d], sst[128 + 4 * h + d], den);
    }
    const float z = 1.f / den;
#pragma unroll
    for (int m = 0; m < kFmtHD; ++m) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < kFmtHD; ++d) s = fmaf(qp[d], sst[h * 16 + m * 4 + d], s);
      att[kFmtHD * h + m] = s * z;
    }
  }
  // x = LN1(x + out_projection(att))
  float y[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) y[o] = x[o] + dot32(swo + o * kFmtD, att) + sbo[o];
  layer_norm_inplace<32>(y, sn, sn + 32);
  // out = LN2(y + linear2(relu(linear1(y))))
  float hsum[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) hsum[o] = y[o] + sb2[o];
#pragma unroll 4
  for (int j = 0; j < kFmtFF; ++j) {
    const float hj = fmaxf(dot32(sw1 + j * kFmtD, y) + sb1[j], 0.f);
#pragma unroll
    for (int o = 0; o < 32; o += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(sw2 + j * 32 + o);
      hsum[o] = fmaf(w4[0], hj, hsum[o]);
      hsum[o + 1] = fmaf(w4[1], hj, hsum[o + 1]);
      hsum[o + 2] = fmaf(w4[2], hj, hsum[o + 2]);
      hsum[o + 3] = fmaf(w4[3], hj, hsum[o + 3]);
    }
  }
  layer_norm_inplace<32>(hsum, sn + 64, sn + 96);
  float* dst = out + ((size_t)n * T + t) * kFmtD;
#pragma unroll
  for (int i = 0; i < 8; ++i) st4(dst + 4 * i, f32x4{hsum[4 * i], hsum[4 * i + 1], hsum[4 * i + 2], hsum[4 * i + 3]});
}

int fmt_state_parts(int S) { return 4 * ((S + kFmtTokPerBlock - 1) / kFmtTokPerBlock); }

// state: N x 160 floats followed by N x fmt_state_parts(S) x 160 floats of per-wave partials
hipError_t launch_fmt_layer(const FmtWeights& w, const float* x, const float* src, int N, int T, int S, float* out,
                            float* state, hipStream_t s) {
  const int parts = fmt_state_parts(S);
  float* partial = state + (size_t)N * kFmtState;
  hipLaunchKernelGGL(fmt_state_kernel, dim3(parts / 4, N), dim3(256), 0, s, w, src, S, partial);
  hipLaunchKernelGGL(fmt_state_sum_kernel, dim3(N), dim3(kFmtState), 0, s, partial, parts, state);
  hipLaunchKernelGGL(fmt_apply_kernel, dim3((T + 255) / 256, N), dim3(256), 0, s, w, x, T, state, out);
  return hipGetLastError();
}

}  // namespace ufr
