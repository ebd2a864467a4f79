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
// the weights broadcast from LDS).  Those are the round-3 vector kernels (kept behind UFR_FMT_MFMA = 0 as the A/B); since
// round 5 both run on the fp32 matrix cores with tokens as MFMA columns (`fmt_state_mfma_kernel`, `fmt_apply_mfma_kernel`
// below: 3.5 -> 1.1 ms per frame), same fixed-order partial sums.
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

// ---- the same two kernels on the fp32 matrix cores (round 5).  "d = 32 is too narrow for the matrix cores" was wrong the
// way the vector kernels were built: one token per thread is 6 272 dependent FMAs behind 1 568 LDS weight reads with ONE
// wave per SIMD (61 440 tokens = 960 waves), i.e. all latency: 52 us per call for 0.4 GFLOP.  With tokens as the 16 MFMA
// columns (four column tiles = 64 tokens per wave), a layer's accumulator tile -- lane (g, j): features 16 t + 4 g + r of
// token j -- IS the B operand of the next layer's k-step (tile t, r), the chain pre_sim_mlp runs in the gather kernel
// (gather.hip: presim_block); the A operands are the weights in LDS, stored in fragment order (lane (g, j): row 16 to + j,
// column 16 ti + 4 g + r).  A lane's four values of a tile are the four dimensions of ONE head (h = 4 t + g), so the linear
// attention -- phi(q) . KV_h, the normaliser -- is lane-local; LayerNorm reduces over the four lane groups with two
// shuffles.  v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 sums.
template <int OUT, int IN>
__device__ __forceinline__ void fmt_stage_frags(float* __restrict__ dst, const float* __restrict__ w, int tid, int nthreads) {
  // dst[((to * (IN / 16) + ti) * 4 + r) * 64 + lane] = w[(16 to + j) * IN + 16 ti + 4 g + r], lane = 16 g + j: the matrix is READ
  // in its own order (coalesced; a strided gather per fragment element was most of the kernel's prologue) and scattered into LDS
  for (int i = tid; i < OUT * IN; i += nthreads) {
    const int o = i / IN, c = i - o * IN;
    const int to = o >> 4, j = o & 15, ti = c >> 4, g = (c >> 2) & 3, r = c & 3;
    dst[((to * (IN / 16) + ti) * 4 + r) * 64 + 16 * g + j] = w[i];
  }
}
// out[u][to] += W in[u]:  NI input tiles, NO output tiles, four column tiles
template <int NO, int NI>
__device__ __forceinline__ void fmt_gemm(const float* __restrict__ frags, int lane, const f32x4 (&in)[4][NI], f32x4 (&out)[4][NO]) {
#pragma unroll
  for (int ti = 0; ti < NI; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int to = 0; to < NO; ++to) {
        const float a = frags[((to * NI + ti) * 4 + r) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u][to] = mfma16(a, in[u][ti][r], out[u][to]);
      }
}
// LayerNorm over the 32 features of a token held as y[2] (tiles) x 4 registers in each of the four lane groups of column j
__device__ __forceinline__ void fmt_layer_norm(f32x4 (&y)[2], const float* __restrict__ gam, const float* __restrict__ bet, int g) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += y[t][r];
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s * (1.f / 32);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float c = y[t][r] - mean;
      q = fmaf(c, c, q);
    }
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float rstd = 1.f / sqrtf(q * (1.f / 32) + 1e-5f);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 gg = ld4(gam + 16 * t + 4 * g), bb = ld4(bet + 16 * t + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[t][r] = fmaf((y[t][r] - mean) * rstd, gg[r], bb[r]);
  }
}

constexpr int kFmtTokPerBlockMfma = 256;    // one 64-token step per wave: four times the workgroups of the vector kernel
__global__ void __launch_bounds__(256) fmt_state_mfma_kernel(FmtWeights w, const float* __restrict__ src, int S,
                                                              float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sw[2 * 1024 + 64 + 4 * kFmtState];
  float* fk = sw;
  float* fv = sw + 1024;
  float* sbk = fv + 1024;
  float* sbv = sbk + 32;
  float* red = sbv + 32;          // [wave][160]
  fmt_stage_frags<32, 32>(fk, w.wk, threadIdx.x, 256);
  fmt_stage_frags<32, 32>(fv, w.wv, threadIdx.x, 256);
  if (threadIdx.x < 32) { sbk[threadIdx.x] = w.bk[threadIdx.x]; sbv[threadIdx.x] = w.bv[threadIdx.x]; }
  __syncthreads();
  const int n = blockIdx.y, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  // this lane's heads: h = 4 to + g (to = 0, 1); kv[to][m][d], ks[to][d] over the lane's token columns
  float kv[2][4][4], ks[2][4];
#pragma unroll
  for (int to = 0; to < 2; ++to)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      ks[to][d] = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) kv[to][m][d] = 0.f;
    }
  const int t0 = blockIdx.x * kFmtTokPerBlockMfma + wave * 64;
  if (t0 < S) {
    f32x4 x[4][2], kk[4][2], vv[4][2];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + 16 * u + j;
      ok[u] = t < S;
      const float* px = src + ((size_t)n * S + (ok[u] ? t : S - 1)) * kFmtD + 4 * g;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        x[u][ti] = ld4(px + 16 * ti);
        kk[u][ti] = ld4(sbk + 16 * ti + 4 * g);
        vv[u][ti] = ld4(sbv + 16 * ti + 4 * g);
      }
    }
    fmt_gemm<2, 2>(fk, lane, x, kk);
    fmt_gemm<2, 2>(fv, lane, x, vv);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float kp = ok[u] ? elu1(kk[u][to][d]) : 0.f;
          ks[to][d] += kp;
#pragma unroll
          for (int m = 0; m < 4; ++m) kv[to][m][d] = fmaf(kp, vv[u][to][m], kv[to][m][d]);
        }
  }
  // sum over the 16 token columns of the lane group, the four waves through LDS in wave order: one PARTIAL per workgroup
#pragma unroll
  for (int to = 0; to < 2; ++to)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ks[to][d] += __shfl_xor(ks[to][d], o);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) kv[to][m][d] += __shfl_xor(kv[to][m][d], o);
    }
  if (j == 0) {
    float* dst = red + wave * kFmtState;
#pragma unroll
    for (int to = 0; to < 2; ++to) {
      const int h = 4 * to + g;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) dst[h * 16 + m * 4 + d] = kv[to][m][d];
#pragma unroll
      for (int d = 0; d < 4; ++d) dst[128 + 4 * h + d] = ks[to][d];
    }
  }
  __syncthreads();
  if (threadIdx.x < kFmtState) {
    const int i = threadIdx.x;
    partial[((size_t)n * gridDim.x + blockIdx.x) * kFmtState + i] =
        ((red[i] + red[kFmtState + i]) + red[2 * kFmtState + i]) + red[3 * kFmtState + i];
  }
}

// (the state arrives as the per-workgroup partials of fmt_state_mfma_kernel: every workgroup sums them itself, in index order --
// the fixed order the separate sum kernel used, one launch and 19 us per layer less)
__global__ void __launch_bounds__(256) fmt_apply_mfma_kernel(FmtWeights w, const float* __restrict__ x_in, int T,
                                                              const float* __restrict__ partial, int parts,
                                                              float* __restrict__ out) {
  // LDS: fragments of wq | wo | w1 | w2, then bq bo b1 b2 | n1w n1b n2w n2b | state
  __shared__ __attribute__((aligned(16))) float sw[2 * 1024 + 2 * 2048 + 32 + 32 + 64 + 32 + 4 * 32 + kFmtState];
  float* fq = sw;
  float* fo = fq + 1024;
  float* f1 = fo + 1024;
  float* f2 = f1 + 2048;
  float* sbq = f2 + 2048;
  float* sbo = sbq + 32;
  float* sb1 = sbo + 32;
  float* sb2 = sb1 + 64;
  float* sn = sb2 + 32;           // n1w n1b n2w n2b
  float* sst = sn + 128;
  const int n = blockIdx.y;
  fmt_stage_frags<32, 32>(fq, w.wq, threadIdx.x, 256);
  fmt_stage_frags<32, 32>(fo, w.wo, threadIdx.x, 256);
  fmt_stage_frags<64, 32>(f1, w.w1, threadIdx.x, 256);
  fmt_stage_frags<32, 64>(f2, w.w2, threadIdx.x, 256);
  if (threadIdx.x < 32) {
    const int i = threadIdx.x;
    sbq[i] = w.bq[i]; sbo[i] = w.bo[i]; sb2[i] = w.b2[i];
    sn[i] = w.n1w[i]; sn[32 + i] = w.n1b[i]; sn[64 + i] = w.n2w[i]; sn[96 + i] = w.n2b[i];
  }
  if (threadIdx.x < 64) sb1[threadIdx.x] = w.b1[threadIdx.x];
  if (threadIdx.x < kFmtState) {
    const float* pp = partial + (size_t)n * parts * kFmtState + threadIdx.x;
    float acc = 0.f;
#pragma unroll 16      // (sixteen loads in flight; the additions stay in index order)
    for (int k = 0; k < parts; ++k) acc += pp[(size_t)k * kFmtState];
    sst[threadIdx.x] = acc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wave) * 64;
  if (t0 >= T) return;
  f32x4 x[4][2], q[4][2];
  int tok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = t0 + 16 * u + j;
    tok[u] = t < T ? t : -1;
    const float* px = x_in + ((size_t)n * T + (t < T ? t : T - 1)) * kFmtD + 4 * g;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      x[u][ti] = ld4(px + 16 * ti);
      q[u][ti] = ld4(sbq + 16 * ti + 4 * g);
    }
  }
  fmt_gemm<2, 2>(fq, lane, x, q);
  // attention message (FMT.py:36-38): this lane's head of tile t is h = 4 t + g
  f32x4 att[4][2];
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    const int h = 4 * to + g;
    float kvh[4][4], ksh[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 r4 = ld4(sst + h * 16 + m * 4);
#pragma unroll
      for (int d = 0; d < 4; ++d) kvh[m][d] = r4[d];
    }
    {
      const f32x4 r4 = ld4(sst + 128 + 4 * h);
#pragma unroll
      for (int d = 0; d < 4; ++d) ksh[d] = r4[d];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float qp[4], den = 1e-6f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        qp[d] = elu1(q[u][to][d]);
        den = fmaf(qp[d], ksh[d], den);
      }
      const float z = 1.f / den;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float sacc = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) sacc = fmaf(qp[d], kvh[m][d], sacc);
        att[u][to][m] = sacc * z;
      }
    }
  }
  // y = LN1(x + out_projection(att))
  f32x4 y[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int to = 0; to < 2; ++to) y[u][to] = x[u][to] + ld4(sbo + 16 * to + 4 * g);
  fmt_gemm<2, 2>(fo, lane, att, y);
#pragma unroll
  for (int u = 0; u < 4; ++u) fmt_layer_norm(y[u], sn, sn + 32, g);
  // out = LN2(y + linear2(relu(linear1(y))))
  f32x4 hid[4][4], o[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int to = 0; to < 4; ++to) hid[u][to] = ld4(sb1 + 16 * to + 4 * g);
#pragma unroll
    for (int to = 0; to < 2; ++to) o[u][to] = y[u][to] + ld4(sb2 + 16 * to + 4 * g);
  }
  fmt_gemm<4, 2>(f1, lane, y, hid);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int to = 0; to < 4; ++to)
#pragma unroll
      for (int r = 0; r < 4; ++r) hid[u][to][r] = fmaxf(hid[u][to][r], 0.f);
  fmt_gemm<2, 4>(f2, lane, hid, o);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    fmt_layer_norm(o[u], sn + 64, sn + 96, g);
    if (tok[u] >= 0) {
      float* dst = out + ((size_t)n * T + tok[u]) * kFmtD + 4 * g;
      st4(dst, o[u][0]);
      st4(dst + 16, o[u][1]);
    }
  }
}

#ifndef UFR_FMT_MFMA
#define UFR_FMT_MFMA 1
#endif

int fmt_state_parts(int S) { return 4 * ((S + kFmtTokPerBlock - 1) / kFmtTokPerBlock); }

// state: N x 160 floats followed by N x fmt_state_parts(S) x 160 floats of per-wave partials
hipError_t launch_fmt_layer(const FmtWeights& w, const float* x, const float* src, int N, int T, int S, float* out,
                            float* state, hipStream_t s) {
  const int parts = fmt_state_parts(S);
  float* partial = state + (size_t)N * kFmtState;
#if UFR_FMT_MFMA
  const int blocks = (S + kFmtTokPerBlockMfma - 1) / kFmtTokPerBlockMfma;      // <= parts: the workspace holds them
  if (blocks > parts) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fmt_state_mfma_kernel, dim3(blocks, N), dim3(256), 0, s, w, src, S, partial);
  hipLaunchKernelGGL(fmt_apply_mfma_kernel, dim3((T + 255) / 256, N), dim3(256), 0, s, w, x, T, partial, blocks, out);
#else
  hipLaunchKernelGGL(fmt_state_kernel, dim3(parts / 4, N), dim3(256), 0, s, w, src, S, partial);
  hipLaunchKernelGGL(fmt_state_sum_kernel, dim3(N), dim3(kFmtState), 0, s, partial, parts, state);
  hipLaunchKernelGGL(fmt_apply_kernel, dim3((T + 255) / 256, N), dim3(256), 0, s, w, x, T, state, out);
#endif
  return hipGetLastError();
}

}  // namespace ufr
