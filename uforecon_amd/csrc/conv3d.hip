// 3-D convolutions of the frustum construction's U-Nets:
//   CostRegNet        code1/encoder_utils/fmt/module.py:469-500   (Conv3d / Deconv3d = conv + BatchNorm + ReLU, :110-187)
//   CostRegNetWeight  code1/encoder_utils/fmt/module.py:502-543   (plain nn.Conv3d / nn.ConvTranspose3d with bias)
// Every layer is 3x3x3, padding 1: stride 1, stride 2, or transposed stride 2 with output_padding 1; channels 1..64.
//
// MI355X mapping.  fp32 (the cascade's winner-take-all depth feeds the next stage's hypotheses: parity wants fp32-grade
// sums), and in fp32 the packed vector FMA and the fp32 MFMA have the same 157 TFLOP/s peak, so this is a direct
// convolution on the vector ALU:
//  * volumes are channel-last [B][D][H][W][C]: a voxel's channels are one contiguous vector (float4 loads), threads of
//    a wave are consecutive voxels along W (coalesced), every thread owns R output voxels x all COUT channels in registers;
//  * the weights of a tap chunk live in LDS as [tap][cin][cout]: every lane of a wave needs the same weight, so they are
//    read with broadcast ds_read_b128 -- 4 output channels per read, reused for the thread's R voxels;
//  * bias / folded BatchNorm (eval mode: y = conv * scale + shift, applied as one fma after the sum like the reference's
//    separate BatchNorm op) / ReLU / the U-Net's skip addition are fused into the store;
//  * a transposed convolution is 8 ordinary convolutions, one per output parity class (pz,py,px), with 1..8 taps each:
//    blockIdx.y selects the class, so a wave never diverges over taps.
// The head layers write the reference's (B,C,D,H,W) layout directly (feature frustum 8 channels + sigmoid weight frustum).
#include "ufr_internal.h"
#include "weight_stream.h"   // static_for

namespace ufr {

namespace {

enum ConvMode : int { kConvS1 = 0, kConvS2 = 1, kDeconvS2 = 2 };

struct Conv3dArgs {
  const float* in;      // [B][D][H][W][CIN]
  const float* weight;  // reference layout: conv [COUT_REAL][CIN][27]; transposed conv [CIN][COUT_REAL][27]
  const float* weight2; // heads: second conv's weight [COUT2][CIN][27] appended after the first's channels (nullable)
  const float* bias;    // [COUT_REAL] (nullable)
  const float* scale;   // [COUT_REAL] folded BatchNorm (nullable)
  const float* shift;
  const float* skip;    // [B][Do][Ho][Wo][COUT] added after the activation (nullable)
  float* out;           // channel-last [B][Do][Ho][Wo][COUT_REAL], or NCDHW when ncdhw != 0
  float* out2;          // heads: NCDHW output of the second conv (sigmoid applied)
  int B, D, H, W;       // input extent
  int Do, Ho, Wo;       // output extent
  int cout_real, cout2; // real output channels of weight / weight2
  int relu, ncdhw;
};

// LDS weights: [tap in chunk][cin][COUT]
template <int CIN, int COUT>
constexpr int taps_per_chunk() {
  return (8192 / (CIN * COUT)) >= 27 ? 27 : ((8192 / (CIN * COUT)) < 1 ? 1 : 8192 / (CIN * COUT));   // <= 32 KiB
}

template <int CIN, int COUT, int MODE, int R>
__global__ void __launch_bounds__(256) conv3d_kernel(Conv3dArgs a) {
  constexpr int TC = taps_per_chunk<CIN, COUT>();
  __shared__ __attribute__((aligned(16))) float wlds[TC * CIN * COUT];
  const int tid = threadIdx.x;

  // ---- which output voxels, and the tap list of this block
  int pz = 0, py = 0, px = 0;            // transposed: parity class of the outputs
  int Ds = a.Do, Hs = a.Ho, Ws = a.Wo;   // extent of the (sub-)grid this block's threads enumerate
  if (MODE == kDeconvS2) {
    pz = (blockIdx.y >> 2) & 1; py = (blockIdx.y >> 1) & 1; px = blockIdx.y & 1;
    Ds = a.Do / 2; Hs = a.Ho / 2; Ws = a.Wo / 2;
  }
  const long long n_sub = (long long)a.B * Ds * Hs * Ws;
  // taps: conv: all 27 (kz,ky,kx); transposed, per dimension: parity 0 -> {k=1, i=m}; parity 1 -> {k=0, i=m+1}, {k=2, i=m}
  const int nz = MODE == kDeconvS2 ? 1 + pz : 3, ny = MODE == kDeconvS2 ? 1 + py : 3, nx = MODE == kDeconvS2 ? 1 + px : 3;
  const int n_taps = nz * ny * nx;
  auto tap_k = [&](int t, int n, int par) -> int {      // kernel index of local tap t in one dimension
    if (MODE != kDeconvS2) return t;
    return par == 0 ? 1 : (t == 0 ? 0 : 2);
  };
  auto tap_di = [&](int t, int par) -> int {            // input offset of that tap relative to the base index
    if (MODE == kConvS1 || MODE == kConvS2) return t - 1;
    return par == 0 ? 0 : (t == 0 ? 1 : 0);
  };

  int vb[R], vz[R], vy[R], vx[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    long long n = ((long long)blockIdx.x * R + r) * 256 + tid;
    live[r] = n < n_sub;
    if (!live[r]) n = 0;
    vx[r] = (int)(n % Ws); n /= Ws;
    vy[r] = (int)(n % Hs); n /= Hs;
    vz[r] = (int)(n % Ds);
    vb[r] = (int)(n / Ds);
  }
  float acc[R][COUT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[r][c] = 0.f;

  for (int t0 = 0; t0 < n_taps; t0 += TC) {
    const int tn = (n_taps - t0) < TC ? (n_taps - t0) : TC;
    __syncthreads();   // the previous chunk's readers are done
    // ---- stage the chunk's weights: LDS [tap][ci][co] <- reference layout (strided, L2-resident, a few KB)
    for (int i = tid; i < tn * CIN * COUT; i += 256) {
      const int co = i % COUT, ci = (i / COUT) % CIN, tl = i / (COUT * CIN);
      const int t = t0 + tl;
      const int tz = t / (ny * nx), ty = (t / nx) % ny, tx = t % nx;
      const int k = (tap_k(tz, nz, pz) * 3 + tap_k(ty, ny, py)) * 3 + tap_k(tx, nx, px);
      float w = 0.f;
      if (co < a.cout_real) {
        w = MODE == kDeconvS2 ? a.weight[((size_t)ci * a.cout_real + co) * 27 + k] : a.weight[((size_t)co * CIN + ci) * 27 + k];
      } else if (a.weight2 && co < a.cout_real + a.cout2) {
        w = a.weight2[((size_t)(co - a.cout_real) * CIN + ci) * 27 + k];
      }
      wlds[i] = w;
    }
    __syncthreads();
    for (int tl = 0; tl < tn; ++tl) {
      const int t = t0 + tl;
      const int tz = t / (ny * nx), ty = (t / nx) % ny, tx = t % nx;
      const int dz = tap_di(tz, pz), dy = tap_di(ty, py), dx = tap_di(tx, px);
      float x[R][CIN];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int s = MODE == kConvS2 ? 2 : 1;
        const int iz = vz[r] * s + dz, iy = vy[r] * s + dy, ix = vx[r] * s + dx;
        const bool ok = live[r] && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const float* p = a.in + ((((size_t)vb[r] * a.D + (ok ? iz : 0)) * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * CIN;
        if constexpr (CIN % 4 == 0) {
#pragma unroll
          for (int c4 = 0; c4 < CIN / 4; ++c4) {
            f32x4 v = ld4(p + 4 * c4);
            if (!ok) v = splat4(0.f);
            x[r][4 * c4 + 0] = v[0]; x[r][4 * c4 + 1] = v[1]; x[r][4 * c4 + 2] = v[2]; x[r][4 * c4 + 3] = v[3];
          }
        } else {
#pragma unroll
          for (int c = 0; c < CIN; ++c) x[r][c] = ok ? p[c] : 0.f;
        }
      }
      const f32x4* w4 = reinterpret_cast<const f32x4*>(wlds + (size_t)tl * CIN * COUT);
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4) {
          const f32x4 w = w4[ci * (COUT / 4) + c4];          // same address in every lane: LDS broadcast
#pragma unroll
          for (int r = 0; r < R; ++r) {
            acc[r][4 * c4 + 0] = fmaf(w[0], x[r][ci], acc[r][4 * c4 + 0]);
            acc[r][4 * c4 + 1] = fmaf(w[1], x[r][ci], acc[r][4 * c4 + 1]);
            acc[r][4 * c4 + 2] = fmaf(w[2], x[r][ci], acc[r][4 * c4 + 2]);
            acc[r][4 * c4 + 3] = fmaf(w[3], x[r][ci], acc[r][4 * c4 + 3]);
          }
        }
    }
  }

  // ---- epilogue: bias, folded BatchNorm, ReLU, skip, store
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!live[r]) continue;
    const int oz = MODE == kDeconvS2 ? 2 * vz[r] + pz : vz[r], oy = MODE == kDeconvS2 ? 2 * vy[r] + py : vy[r],
              ox = MODE == kDeconvS2 ? 2 * vx[r] + px : vx[r];
    const size_t vox = (((size_t)vb[r] * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
    float y[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      float v = acc[r][c];
      if (c < a.cout_real) {
        if (a.bias) v += a.bias[c];
        if (a.scale) v = fmaf(v, a.scale[c], a.shift[c]);
        if (a.relu) v = fmaxf(v, 0.f);
      }
      y[c] = v;
    }
    if (a.ncdhw) {   // heads: (B,C,D,H,W); second conv -> sigmoid -> out2
      const size_t plane = (size_t)a.Do * a.Ho * a.Wo, sp = vox - (size_t)vb[r] * plane;
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        if (c < a.cout_real) a.out[((size_t)vb[r] * a.cout_real + c) * plane + sp] = y[c];
        else if (c < a.cout_real + a.cout2)
          a.out2[((size_t)vb[r] * a.cout2 + (c - a.cout_real)) * plane + sp] = 1.f / (1.f + expf(-y[c]));
      }
    } else {
      float* o = a.out + vox * COUT;
      const float* sk = a.skip ? a.skip + vox * COUT : nullptr;
      if constexpr (COUT % 4 == 0) {
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4) {
          f32x4 v = {y[4 * c4], y[4 * c4 + 1], y[4 * c4 + 2], y[4 * c4 + 3]};
          if (sk) v += ld4(sk + 4 * c4);
          st4(o + 4 * c4, v);
        }
      }
    }
  }
}

template <int CIN, int COUT, int MODE, int R>
hipError_t launch_conv_t(const Conv3dArgs& a, hipStream_t s) {
  const bool de = MODE == kDeconvS2;
  const long long n_sub = (long long)a.B * (de ? a.Do / 2 : a.Do) * (de ? a.Ho / 2 : a.Ho) * (de ? a.Wo / 2 : a.Wo);
  const long long blocks = (n_sub + 256LL * R - 1) / (256LL * R);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((conv3d_kernel<CIN, COUT, MODE, R>), dim3((unsigned)blocks, de ? 8 : 1), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace

// cout_pad: COUT of the instantiation (cout_real (+ cout2) rounded up to a multiple of 4)
hipError_t launch_conv3d(const float* in, const float* weight, const float* weight2, const float* bias, const float* scale,
                         const float* shift, const float* skip, float* out, float* out2, int B, int D, int H, int W,
                         int cin, int cout, int cout2, int mode, int relu, int ncdhw, hipStream_t s) {
  Conv3dArgs a;
  a.in = in; a.weight = weight; a.weight2 = weight2; a.bias = bias; a.scale = scale; a.shift = shift; a.skip = skip;
  a.out = out; a.out2 = out2; a.B = B; a.D = D; a.H = H; a.W = W;
  a.cout_real = cout; a.cout2 = cout2; a.relu = relu; a.ncdhw = ncdhw;
  if (mode == kConvS1) { a.Do = D; a.Ho = H; a.Wo = W; }
  else if (mode == kConvS2) { a.Do = (D + 1) / 2; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2; }   // k3 p1 s2: floor((n-1)/2)+1
  else { a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; }                                           // k3 p1 s2 output_padding 1
  const int ct = cout + cout2;
#define UFR_CONV_CASE(CI, CO, MO, RR) \
  if (cin == CI && ct <= CO && ct > CO - 4 && mode == MO) return launch_conv_t<CI, CO, MO, RR>(a, s);
  // the layers of CostRegNet / CostRegNetWeight with base_channels = 8 (module.py:469-543)
  UFR_CONV_CASE(1, 8, kConvS1, 2)       // conv0
  UFR_CONV_CASE(8, 16, kConvS2, 2)      // conv1
  UFR_CONV_CASE(16, 16, kConvS1, 2)     // conv2
  UFR_CONV_CASE(16, 32, kConvS2, 2)     // conv3
  UFR_CONV_CASE(32, 32, kConvS1, 2)     // conv4
  UFR_CONV_CASE(32, 64, kConvS2, 1)     // conv5
  UFR_CONV_CASE(64, 64, kConvS1, 1)     // conv6
  UFR_CONV_CASE(64, 32, kDeconvS2, 2)   // conv7
  UFR_CONV_CASE(32, 16, kDeconvS2, 2)   // conv9
  UFR_CONV_CASE(16, 8, kDeconvS2, 2)    // conv11
  UFR_CONV_CASE(8, 4, kConvS1, 2)       // prob (1 channel)
  UFR_CONV_CASE(8, 12, kConvS1, 2)      // features (8) + weights (1) heads
  UFR_CONV_CASE(8, 8, kConvS1, 2)       // features alone
#undef UFR_CONV_CASE
  return hipErrorInvalidValue;
}

}  // namespace ufr
