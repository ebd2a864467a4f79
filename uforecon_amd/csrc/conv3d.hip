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

#ifndef UFR_C3_FOLD
#define UFR_C3_FOLD 1
#endif

namespace ufr {

namespace {

enum ConvMode : int { kConvS1 = 0, kConvS2 = 1, kDeconvS2 = 2 };
// kernel-internal: the transposed mode with the x parity FOLDED into the channel axis (conv3d_kernel below)
constexpr int kDeconvS2F = 3;

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
  unsigned* out_absmax; // (nullable) raised to max |out| of the channel-last stores of the vector kernel: the bound the plane
                        // kernels (conv3d_planes.hip) want of their input, taken here instead of in a pass of its own
  int flip;             // stride-1 only: `weight` is the layer's FORWARD weight [cin of this launch][cout_real][27] and the taps
                        // are mirrored -- the data gradient of a stride-1 convolution (launch_conv3d_bwd_data)
};

// LDS weights: [tap in chunk][cin][COUT]
template <int CIN, int COUT>
constexpr int taps_per_chunk() {
  return (8192 / (CIN * COUT)) >= 27 ? 27 : ((8192 / (CIN * COUT)) < 1 ? 1 : 8192 / (CIN * COUT));   // <= 32 KiB
}

template <int CIN, int COUT, int MODE, int R>
__global__ void __launch_bounds__(256) conv3d_kernel(Conv3dArgs a) {
  // kDeconvS2F (round 5): a thread owns BOTH x parities of its output pair (2 m, 2 m + 1): template COUT = 2 x the layer's
  // channels, [parity 0 | parity 1].  The per-class form wrote every other voxel of a row from a block -- 32-byte pieces at a
  // 64-byte stride, the other halves arriving later from another block, usually on another XCD: two partial writes per line.
  // Folded, a lane's store is its pair's 2 x cout contiguous floats and a wave's is one contiguous run.  Taps along x become
  // the two inputs m (kx = 1 for parity 0, kx = 2 for parity 1) and m + 1 (kx = 0 for parity 1; a zero block for parity 0:
  // a quarter of the multiply-adds idle, on layers that are bound by their stores).
  constexpr bool FOLD = MODE == kDeconvS2F;
  constexpr bool DE = MODE == kDeconvS2 || FOLD;
  constexpr int CO = FOLD ? COUT / 2 : COUT;       // channels of one output voxel
  constexpr int TC = taps_per_chunk<CIN, COUT>();
  __shared__ __attribute__((aligned(16))) float wlds[TC * CIN * COUT];
  const int tid = threadIdx.x;

  // ---- which output voxels, and the tap list of this block
  int pz = 0, py = 0, px = 0;            // transposed: parity class of the outputs
  int Ds = a.Do, Hs = a.Ho, Ws = a.Wo;   // extent of the (sub-)grid this block's threads enumerate
  if (DE) {
    if (FOLD) { pz = (blockIdx.y >> 1) & 1; py = blockIdx.y & 1; px = 0; }
    else { pz = (blockIdx.y >> 2) & 1; py = (blockIdx.y >> 1) & 1; px = blockIdx.y & 1; }
    Ds = a.Do / 2; Hs = a.Ho / 2; Ws = a.Wo / 2;
  }
  const long long n_sub = (long long)a.B * Ds * Hs * Ws;
  // taps: conv: all 27 (kz,ky,kx); transposed, per dimension: parity 0 -> {k=1, i=m}; parity 1 -> {k=0, i=m+1}, {k=2, i=m}
  const int nz = DE ? 1 + pz : 3, ny = DE ? 1 + py : 3, nx = FOLD ? 2 : (DE ? 1 + px : 3);
  const int n_taps = nz * ny * nx;
  auto tap_k = [&](int t, int n, int par) -> int {      // kernel index of local tap t in one dimension
    if (!DE) return t;
    return par == 0 ? 1 : (t == 0 ? 0 : 2);
  };
  auto tap_di = [&](int t, int par) -> int {            // input offset of that tap relative to the base index
    if (!DE) return t - 1;
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
  // accumulators as PAIRS of output channels: the multiply-adds below are written on 2-vectors so that they become
  // v_pk_fma_f32 (two IEEE fmas per instruction: the same bits as the scalar form; no MFMA runs beside this kernel for the
  // packed form to disturb).  Measured round 5: halving the FMA instructions changes NOTHING (heads 0.58 ms either way),
  // and neither do four voxels per thread (half the LDS weight reads per FMA): the full-resolution layers are bound by
  // neither the VALU nor the LDS -- the 27 bounds-checked neighbour loads per voxel and their address arithmetic remain
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[R][COUT / 2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) acc[r][c] = f32x2{0.f, 0.f};

  for (int t0 = 0; t0 < n_taps; t0 += TC) {
    const int tn = (n_taps - t0) < TC ? (n_taps - t0) : TC;
    __syncthreads();   // the previous chunk's readers are done
    // ---- stage the chunk's weights: LDS [tap][ci][co] <- reference layout (strided, L2-resident, a few KB)
    for (int i = tid; i < tn * CIN * COUT; i += 256) {
      const int co = i % COUT, ci = (i / COUT) % CIN, tl = i / (COUT * CIN);
      const int t = t0 + tl;
      const int tz = t / (ny * nx), ty = (t / nx) % ny, tx = t % nx;
      int k = (tap_k(tz, nz, pz) * 3 + tap_k(ty, ny, py)) * 3 + tap_k(tx, nx, px);
      float w = 0.f;
      if (FOLD) {     // co = [x parity][channel]; input m (tx = 0): kx = 1 | 2, input m + 1 (tx = 1): none | 0
        const int par = co / CO, cr = co - par * CO;
        const int kx = tx == 0 ? 1 + par : (par == 0 ? -1 : 0);
        k = (tap_k(tz, nz, pz) * 3 + tap_k(ty, ny, py)) * 3 + kx;
        if (kx >= 0 && cr < a.cout_real) w = a.weight[((size_t)ci * a.cout_real + cr) * 27 + k];
      } else if (co < a.cout_real) {
        if (MODE == kConvS1 && a.flip) w = a.weight[((size_t)ci * a.cout_real + co) * 27 + (26 - k)];
        else w = DE ? a.weight[((size_t)ci * a.cout_real + co) * 27 + k] : a.weight[((size_t)co * CIN + ci) * 27 + k];
      } else if (a.weight2 && co < a.cout_real + a.cout2) {
        w = a.weight2[((size_t)(co - a.cout_real) * CIN + ci) * 27 + k];
      }
      wlds[i] = w;
    }
    __syncthreads();
    for (int tl = 0; tl < tn; ++tl) {
      const int t = t0 + tl;
      const int tz = t / (ny * nx), ty = (t / nx) % ny, tx = t % nx;
      const int dz = tap_di(tz, pz), dy = tap_di(ty, py), dx = FOLD ? tx : tap_di(tx, px);
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
          const f32x2 wlo = {w[0], w[1]}, whi = {w[2], w[3]};
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const f32x2 xx = {x[r][ci], x[r][ci]};
            acc[r][2 * c4 + 0] = __builtin_elementwise_fma(wlo, xx, acc[r][2 * c4 + 0]);
            acc[r][2 * c4 + 1] = __builtin_elementwise_fma(whi, xx, acc[r][2 * c4 + 1]);
          }
        }
    }
  }

  // ---- epilogue: bias, folded BatchNorm, ReLU, skip, store
  float omax = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!live[r]) continue;
    const int oz = DE ? 2 * vz[r] + pz : vz[r], oy = DE ? 2 * vy[r] + py : vy[r], ox = DE ? 2 * vx[r] + px : vx[r];
    const size_t vox = (((size_t)vb[r] * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
    float y[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      float v = acc[r][c >> 1][c & 1];
      const int cr = FOLD ? c % CO : c;      // the layer's channel of slot c
      if (cr < a.cout_real) {
        if (a.bias) v += a.bias[cr];
        if (a.scale) v = fmaf(v, a.scale[cr], a.shift[cr]);
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
      float* o = a.out + vox * CO;          // FOLD: the pair's 2 CO floats are contiguous (x is the fastest voxel axis)
      const float* sk = a.skip ? a.skip + vox * CO : nullptr;
      if constexpr (COUT % 4 == 0) {
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4) {
          f32x4 v = {y[4 * c4], y[4 * c4 + 1], y[4 * c4 + 2], y[4 * c4 + 3]};
          if (sk) v += ld4(sk + 4 * c4);
          st4(o + 4 * c4, v);
          omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
      }
    }
  }
  if (a.out_absmax) {      // one guarded atomic per wave (prep.hip: wave_abs_max)
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) omax = fmaxf(omax, __shfl_xor(omax, s));
    const unsigned bits = __builtin_bit_cast(unsigned, omax);
    if ((threadIdx.x & 63) == 0 && bits > __atomic_load_n(a.out_absmax, __ATOMIC_RELAXED)) atomicMax(a.out_absmax, bits);
  }
}


template <int CIN, int COUT, int MODE, int R>
hipError_t launch_conv_t(const Conv3dArgs& a, hipStream_t s) {
  const bool de = MODE == kDeconvS2 || MODE == kDeconvS2F;
  const long long n_sub = (long long)a.B * (de ? a.Do / 2 : a.Do) * (de ? a.Ho / 2 : a.Ho) * (de ? a.Wo / 2 : a.Wo);
  const long long blocks = (n_sub + 256LL * R - 1) / (256LL * R);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((conv3d_kernel<CIN, COUT, MODE, R>), dim3((unsigned)blocks, MODE == kDeconvS2F ? 4 : (de ? 8 : 1)), dim3(256), 0,
                     s, a);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// The 16..64-channel layers on the matrix cores (round 5): an implicit GEMM on v_mfma_f32_16x16x4_f32 -- EXACT fp32
// products and sums (no planes, no scales: the winner-take-all depth of the cascade and the parity bounds stay where
// they are), rows = 16 output channels, columns = 16 output voxels, k = 4 input channels of one tap per instruction.
//  * a wave owns VT tiles of 16 consecutive voxels of the (sub-)grid; lane (g, j) loads channels 16 cb + 4g .. + 3 of its
//    voxel j's tap neighbour with ONE 16-byte load -- the B operands of four MFMAs (k = lane group g, r = 0..3);
//  * the A operands come from LDS: one chunk = ALL 27 taps of one (16-channel input block, 16-channel output tile) pair,
//    27.6 KiB, staged from the checkpoint's layout in runs of 432 contiguous floats per output channel (coalesced; the
//    per-tap chunks of the vector kernel read with a stride of 27 floats) and stored in operand order, so a lane's four
//    k-values of a tap are one ds_read_b128;
//  * loop order: output tile > input block (stage, barrier) > tap > voxel tile: the accumulators of one output tile are
//    all that lives across the taps (4 registers per voxel tile), the tile is finished and stored before the next starts;
//    the activations are re-read once per output tile (L1 / L2), one load per four MFMAs;
//  * bias / folded BatchNorm / ReLU / skip ride in the store: the accumulator tile's lane (g, j) holds channels 4g .. 4g + 3
//    of voxel j -- one 16-byte channel-last store.
// The vector kernel runs these layers at 4..15 % of the fp32 peak (its 64 -> 64 layer at 1/8 resolution has 60 blocks of
// work for 256 CUs); measured per layer in tools/dev/conv3d_bwd_probe.py.
template <int CIN, int COUT, int MODE, int VT>
__global__ void __launch_bounds__(256) conv3d_mfma_kernel(Conv3dArgs a) {
  static_assert(CIN % 16 == 0 && COUT % 16 == 0, "whole 16-channel blocks");
  constexpr int NCB = CIN / 16, NTL = COUT / 16;
  __shared__ __attribute__((aligned(16))) float wlds[27 * 256];   // [tap][lane 64][r 4]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
  int pz = 0, py = 0, px = 0;
  int Ds = a.Do, Hs = a.Ho, Ws = a.Wo;
  if (MODE == kDeconvS2) {
    pz = (blockIdx.y >> 2) & 1; py = (blockIdx.y >> 1) & 1; px = blockIdx.y & 1;
    Ds = a.Do / 2; Hs = a.Ho / 2; Ws = a.Wo / 2;
  }
  const long long n_sub = (long long)a.B * Ds * Hs * Ws;
  const int nz = MODE == kDeconvS2 ? 1 + pz : 3, ny = MODE == kDeconvS2 ? 1 + py : 3, nx = MODE == kDeconvS2 ? 1 + px : 3;
  const int n_taps = nz * ny * nx;
  auto tap_k = [&](int t, int par) -> int { return MODE != kDeconvS2 ? t : (par == 0 ? 1 : (t == 0 ? 0 : 2)); };
  auto tap_di = [&](int t, int par) -> int { return MODE != kDeconvS2 ? t - 1 : (par == 0 ? 0 : (t == 0 ? 1 : 0)); };

  int vb[VT], vz[VT], vy[VT], vx[VT];
  bool live[VT];
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    long long n = (((long long)blockIdx.x * 4 + wave) * VT + v) * 16 + j;
    live[v] = n < n_sub;
    if (!live[v]) n = 0;
    vx[v] = (int)(n % Ws); n /= Ws;
    vy[v] = (int)(n % Hs); n /= Hs;
    vz[v] = (int)(n % Ds);
    vb[v] = (int)(n / Ds);
  }
  constexpr int s = MODE == kConvS2 ? 2 : 1;

  for (int t = 0; t < NTL; ++t) {
    f32x4 acc[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) acc[v] = splat4(0.f);
    for (int cb = 0; cb < NCB; ++cb) {
      __syncthreads();   // the previous chunk's readers are done
      // ---- stage W[16 t + i][16 cb + c][k], all 27 k: per output channel i one run of 16 x 27 contiguous floats
      // (conv layout [co][ci][27]; transposed / mirrored layers read [ci][co][27]: runs of 27 per (ci, co))
      for (int e = tid; e < 16 * 16 * 27; e += 256) {
        int i, c, k;
        float w;
        if (MODE == kDeconvS2 || (MODE == kConvS1 && a.flip)) {
          k = e % 27; i = (e / 27) % 16; c = e / (27 * 16);          // source [ci][co][27]: co fastest among the runs
          w = a.weight[((size_t)(16 * cb + c) * COUT + 16 * t + i) * 27 + k];
          if (MODE == kConvS1) k = 26 - k;                            // mirrored taps (the data gradient of a stride-1 layer)
        } else {
          k = e % 27; c = (e / 27) % 16; i = e / (27 * 16);          // source [co][ci][27]
          w = a.weight[((size_t)(16 * t + i) * CIN + 16 * cb + c) * 27 + k];
        }
        wlds[(k * 64 + (c >> 2) * 16 + i) * 4 + (c & 3)] = w;         // operand order: lane = (g = c / 4, i), r = c % 4
      }
      __syncthreads();
      for (int tl = 0; tl < n_taps; ++tl) {
        const int tz = tl / (ny * nx), ty = (tl / nx) % ny, tx = tl % nx;
        const int k = (tap_k(tz, pz) * 3 + tap_k(ty, py)) * 3 + tap_k(tx, px);
        const int dz = tap_di(tz, pz), dy = tap_di(ty, py), dx = tap_di(tx, px);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wlds[(k * 64 + lane) * 4]);
#pragma unroll
        for (int v = 0; v < VT; ++v) {
          const int iz = vz[v] * s + dz, iy = vy[v] * s + dy, ix = vx[v] * s + dx;
          const bool ok = live[v] && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          const float* p = a.in + ((((size_t)vb[v] * a.D + (ok ? iz : 0)) * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * CIN +
                           16 * cb + 4 * g;
          f32x4 xb = ld4(p);
          if (!ok) xb = splat4(0.f);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[v] = mfma16(wa[r], xb[r], acc[v]);
        }
      }
    }
    // ---- epilogue of output tile t: lane (g, j) holds channels 16 t + 4 g + r of voxel j
    const int c0 = 16 * t + 4 * g;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      if (!live[v]) continue;
      const int oz = MODE == kDeconvS2 ? 2 * vz[v] + pz : vz[v], oy = MODE == kDeconvS2 ? 2 * vy[v] + py : vy[v],
                ox = MODE == kDeconvS2 ? 2 * vx[v] + px : vx[v];
      const size_t vox = (((size_t)vb[v] * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
      f32x4 y = acc[v];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float q = y[r];
        if (a.bias) q += a.bias[c0 + r];
        if (a.scale) q = fmaf(q, a.scale[c0 + r], a.shift[c0 + r]);
        if (a.relu) q = fmaxf(q, 0.f);
        y[r] = q;
      }
      if (a.skip) y += ld4(a.skip + vox * COUT + c0);
      st4(a.out + vox * COUT + c0, y);
    }
  }
}

template <int CIN, int COUT, int MODE>
hipError_t launch_conv_mfma_t(const Conv3dArgs& a, hipStream_t s) {
  const bool de = MODE == kDeconvS2;
  const long long n_sub = (long long)a.B * (de ? a.Do / 2 : a.Do) * (de ? a.Ho / 2 : a.Ho) * (de ? a.Wo / 2 : a.Wo);
  const long long tiles = (n_sub + 15) / 16;
  // four voxel tiles per wave amortise the weight staging; small grids (the 1/8-resolution layers: ~1 000 tiles) take one,
  // so that every CU gets work
  if (tiles >= 16384) {
    const long long blocks = (tiles + 15) / 16;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv3d_mfma_kernel<CIN, COUT, MODE, 4>), dim3((unsigned)blocks, de ? 8 : 1), dim3(256), 0, s, a);
  } else {
    const long long blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL((conv3d_mfma_kernel<CIN, COUT, MODE, 1>), dim3((unsigned)blocks, de ? 8 : 1), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace

// cout_pad: COUT of the instantiation (cout_real (+ cout2) rounded up to a multiple of 4)
hipError_t launch_conv3d(const float* in, const float* weight, const float* weight2, const float* bias, const float* scale,
                         const float* shift, const float* skip, float* out, float* out2, int B, int D, int H, int W,
                         int cin, int cout, int cout2, int mode, int relu, int ncdhw, hipStream_t s, int flip, float* out_absmax) {
  Conv3dArgs a;
  a.flip = flip;
  a.out_absmax = reinterpret_cast<unsigned*>(out_absmax);
  a.in = in; a.weight = weight; a.weight2 = weight2; a.bias = bias; a.scale = scale; a.shift = shift; a.skip = skip;
  a.out = out; a.out2 = out2; a.B = B; a.D = D; a.H = H; a.W = W;
  a.cout_real = cout; a.cout2 = cout2; a.relu = relu; a.ncdhw = ncdhw;
  if (mode == kConvS1) { a.Do = D; a.Ho = H; a.Wo = W; }
  else if (mode == kConvS2) { a.Do = (D + 1) / 2; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2; }   // k3 p1 s2: floor((n-1)/2)+1
  else { a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; }                                           // k3 p1 s2 output_padding 1
  const int ct = cout + cout2;
#ifndef UFR_CONV3D_VALU_ONLY
  // whole 16-channel blocks on both sides, channel-last output, one head: the matrix-core kernel
  if (!ncdhw && cout2 == 0 && weight2 == nullptr && out_absmax == nullptr) {
#define UFR_MFMA_CASE(CI, CO, MO) if (cin == CI && cout == CO && mode == MO) return launch_conv_mfma_t<CI, CO, MO>(a, s);
    // where it wins (per-layer table of tools/dev/conv3d_bwd_probe.py, 3 x 8 x 512 x 640 stage): the stride-1 and stride-2
    // layers from 32 channels up -- 64 -> 64 0.48 -> 0.17 ms, 32 -> 64 0.30 -> 0.09, 32 -> 32 0.22 -> 0.17, 16 -> 32 0.125
    // -> 0.096, and the data gradients of the transposed layers, which run as stride-2 convolutions (64 -> 32: 0.31 ->
    // 0.09).  NOT the 16 -> 16 layer at half resolution (0.26 vs 0.23: bound by re-reading the activations once per tap,
    // whichever unit multiplies) and not the transposed mode (eight parity classes each staging all 27 taps: 0.27 vs 0.12).
    UFR_MFMA_CASE(32, 32, kConvS1) UFR_MFMA_CASE(64, 64, kConvS1)
    UFR_MFMA_CASE(16, 32, kConvS2) UFR_MFMA_CASE(32, 64, kConvS2)
    UFR_MFMA_CASE(32, 16, kConvS2) UFR_MFMA_CASE(64, 32, kConvS2)      // = data gradients of conv9 / conv7
#undef UFR_MFMA_CASE
  }
#endif
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
#if UFR_C3_FOLD
  // (transposed layers that write the two finest grids: the x parity folded into the channel axis, see conv3d_kernel)
  if (cin == 32 && ct == 16 && mode == kDeconvS2 && !a.ncdhw && a.Wo % 2 == 0) return launch_conv_t<32, 32, kDeconvS2F, 1>(a, s);
  if (cin == 16 && ct == 8 && mode == kDeconvS2 && !a.ncdhw && a.Wo % 2 == 0) return launch_conv_t<16, 16, kDeconvS2F, 2>(a, s);
#endif
  UFR_CONV_CASE(32, 16, kDeconvS2, 2)   // conv9
  UFR_CONV_CASE(16, 8, kDeconvS2, 2)    // conv11
  UFR_CONV_CASE(8, 4, kConvS1, 2)       // prob (1 channel)
  UFR_CONV_CASE(8, 12, kConvS1, 2)      // features (8) + weights (1) heads
  UFR_CONV_CASE(8, 8, kConvS1, 2)       // features alone
#undef UFR_CONV_CASE
  return hipErrorInvalidValue;
}


// =====================================================================================================================
// Backward of the plain (bias, no activation) layers of CostRegNetWeight -- the one producer the reference trains
// (model.py:72-87: every parameter outside `transmvsnet`; module.py:502-543).
//
// DATA gradients are the forward kernel again: the adjoint of a k3 p1 convolution is
//   stride 1            -> the stride-1 convolution with the taps mirrored and the channel roles swapped (Conv3dArgs::flip),
//   stride 2            -> the transposed stride-2 convolution, whose weight layout [in][out][27] IS the forward
//                          weight [cout][cin][27] read with the roles swapped,
//   transposed stride 2 -> the stride-2 convolution, likewise on the forward weight [cin][cout][27] as it stands;
// and the U-Net's skip additions become the fused `skip` operand (a gradient that arrives on two paths).
hipError_t launch_conv3d_bwd_data(const float* d_out, const float* weight, const float* accumulate, float* d_in, int B, int D,
                                  int H, int W, int cin, int cout, int mode, hipStream_t s) {
  if (mode == kConvS1) {    // d_out (B,D,H,W,cout) -> d_in (B,D,H,W,cin)
    // a 1-channel d_in: the instantiation computes 4 padded channels, so it stores through the (B,C,D,H,W) path, which
    // writes the real channels only -- the same bytes as (B,D,H,W,1); that path has no fused addition
    if (cin == 1 && accumulate) return hipErrorInvalidValue;
    return launch_conv3d(d_out, weight, nullptr, nullptr, nullptr, nullptr, accumulate, d_in, nullptr, B, D, H, W, cout, cin, 0,
                         kConvS1, 0, cin == 1 ? 1 : 0, s, 1);
  }
  if (mode == kConvS2)      // d_out (B,D/2,H/2,W/2,cout) -> d_in (B,D,H,W,cin)
    return launch_conv3d(d_out, weight, nullptr, nullptr, nullptr, nullptr, accumulate, d_in, nullptr, B, D / 2, H / 2, W / 2, cout,
                         cin, 0, kDeconvS2, 0, 0, s, 0);
  // transposed: d_out (B,2D,2H,2W,cout) -> d_in (B,D,H,W,cin)
  return launch_conv3d(d_out, weight, nullptr, nullptr, nullptr, nullptr, accumulate, d_in, nullptr, B, 2 * D, 2 * H, 2 * W, cout, cin,
                       0, kConvS2, 0, 0, s, 0);
}

// WEIGHT gradients.  One formulation for the three layer kinds: pairs of voxels (p, q = S p + k - 1) of a coarse grid P
// and a fine grid Q (the same grid for stride 1), tensors TP [P][CA] and TQ [Q][CB], and
//     dW[a][b][k] += sum_p TP[p][a] TQ[S p + k - 1][b]
// with (TP, TQ) = (d_out, in) for the convolutions -- dW = [cout][cin][27] -- and (in, d_out) for the transposed one --
// dW = [cin][cout][27]: the reference's parameter layout in both cases.  blockIdx.y = the tap k.
struct WgradArgs {
  const float* tp;   // [B][Dp][Hp][Wp][CA]
  const float* tq;   // [B][Dq][Hq][Wq][CB]
  float* dw;         // [CA][CB][27], accumulated into
  int B, Dp, Hp, Wp, Dq, Hq, Wq;
  long long n_p;     // B Dp Hp Wp
  int vox_per_block;
  int seg, nseg;      // the matrix-core kernels: a wave's unit = seg voxels of a row, nseg of them per row
};

namespace {

// (32-bit voxel indices: the launcher bounds n_p below 2^31 -- a 64-bit division per voxel and tap was most of these kernels' time)
__device__ __forceinline__ bool wgrad_q(const WgradArgs& a, unsigned n, int S, int kz, int ky, int kx, size_t* q) {
  const int x = (int)(n % (unsigned)a.Wp);
  unsigned m = n / (unsigned)a.Wp;
  const int y = (int)(m % (unsigned)a.Hp);
  m /= (unsigned)a.Hp;
  const int z = (int)(m % (unsigned)a.Dp), b = (int)(m / (unsigned)a.Dp);
  const int qz = z * S + kz - 1, qy = y * S + ky - 1, qx = x * S + kx - 1;
  const bool ok = qz >= 0 && qz < a.Dq && qy >= 0 && qy < a.Hq && qx >= 0 && qx < a.Wq;
  *q = ok ? (((size_t)b * a.Dq + qz) * a.Hq + qy) * a.Wq + qx : 0;
  return ok;
}

// few channels (CB <= 8 on the fine side: the full-resolution layers, where the traffic is): a thread takes one voxel p per
// step, a group of CAG coarse-side channels (blockIdx.y) and a run of NT consecutive taps (blockIdx.z), reads TP[p] once and
// the NT neighbours TQ[S p + k - 1] through the L1, and keeps the NT x CAG x CB products in registers (<= 192: two to four
// waves per SIMD); one shuffle reduction and one atomic per value and wave at the end.
// What bounds it is the L1: a wave's tap visit moves 64 x 4 CB bytes at 64 B/clk, so every fetched neighbour must meet as
// many coarse-side channels as the registers allow.  Measured at 3 x 8 x 512 x 640 voxels, features head (8 x 8 channels):
// one channel per thread and nine taps per pass (24 passes over TQ) 4.0 ms; all 27 taps per thread (446 registers, one
// wave per SIMD) worse still; one block per TAP streaming both tensors 27 times: 30 ms per training step for all layers.
template <int CAG, int CB, int NT, int S>
__global__ void __launch_bounds__(256) conv3d_wgrad_voxel_kernel(WgradArgs a, int CA) {
  static_assert(NT * CAG * CB <= 192 && 27 % NT == 0, "accumulators per thread");
  const int a0 = blockIdx.y * CAG, k0 = blockIdx.z * NT;
  float acc[NT][CAG][CB];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int i = 0; i < CAG; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[k][i][j] = 0.f;
  const unsigned first = blockIdx.x * (unsigned)a.vox_per_block;
  const unsigned last = first + a.vox_per_block < (unsigned)a.n_p ? first + a.vox_per_block : (unsigned)a.n_p;
  for (unsigned n = first + threadIdx.x; n < last; n += 256) {
    const int x = (int)(n % (unsigned)a.Wp);
    unsigned m = n / (unsigned)a.Wp;
    const int y = (int)(m % (unsigned)a.Hp);
    m /= (unsigned)a.Hp;
    const int z = (int)(m % (unsigned)a.Dp), b = (int)(m / (unsigned)a.Dp);
    float vp[CAG];
    const float* pp = a.tp + (size_t)n * CA + a0;
    if constexpr (CAG % 4 == 0) {
#pragma unroll
      for (int i = 0; i < CAG / 4; ++i) { const f32x4 v = ld4(pp + 4 * i); vp[4 * i] = v[0]; vp[4 * i + 1] = v[1]; vp[4 * i + 2] = v[2]; vp[4 * i + 3] = v[3]; }
    } else {
#pragma unroll
      for (int i = 0; i < CAG; ++i) vp[i] = pp[i];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = k0 + t;
      const int qz = z * S + k / 9 - 1, qy = y * S + (k / 3) % 3 - 1, qx = x * S + k % 3 - 1;
      const bool ok = qz >= 0 && qz < a.Dq && qy >= 0 && qy < a.Hq && qx >= 0 && qx < a.Wq;
      const float* pq = a.tq + ((((size_t)b * a.Dq + (ok ? qz : 0)) * a.Hq + (ok ? qy : 0)) * a.Wq + (ok ? qx : 0)) * CB;
      float vq[CB];
      if constexpr (CB % 4 == 0) {
#pragma unroll
        for (int j = 0; j < CB / 4; ++j) {
          f32x4 v = ld4(pq + 4 * j);
          if (!ok) v = splat4(0.f);
          vq[4 * j] = v[0]; vq[4 * j + 1] = v[1]; vq[4 * j + 2] = v[2]; vq[4 * j + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < CB; ++j) vq[j] = ok ? pq[j] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < CAG; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[t][i][j] = fmaf(vp[i], vq[j], acc[t][i][j]);
    }
  }
  // wave sums -> LDS -> ONE atomic per value and block: atomics on one address serialise in the L2 (a few hundred ns each),
  // and with an atomic per wave the few hundred addresses of a small layer took most of the kernel's time
  __shared__ float red[4][NT * CAG * CB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < CAG; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        float v = acc[t][i][j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][(t * CAG + i) * CB + j] = v;
      }
  __syncthreads();
  for (int e = threadIdx.x; e < NT * CAG * CB; e += 256) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const int j = e % CB, i = (e / CB) % CAG, t = e / (CB * CAG);
    if (v != 0.f) atomicAdd(a.dw + ((size_t)(a0 + i) * CB + j) * 27 + k0 + t, v);
  }
}

// many channel pairs: 16 x 16 threads, each owning a (CA / 16) x (CB / 16) register tile of the tap's CA x CB products;
// 64 voxel pairs at a time are staged through LDS (coalesced loads, broadcast reads)
template <int CA, int CB, int S>
__global__ void __launch_bounds__(256) conv3d_wgrad_pair_kernel(WgradArgs a) {
  // voxel pairs per LDS stage: as many as 32 KiB hold (several 16-byte loads per thread in flight per stage)
  constexpr int TA = CA / 16, TB = CB / 16, NV = (CA + CB) <= 32 ? 256 : (CA + CB) <= 64 ? 128 : 64;
  __shared__ __attribute__((aligned(16))) float sp[NV][CA];
  __shared__ __attribute__((aligned(16))) float sq[NV][CB];
  const int k = blockIdx.y, kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
  const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;
  float acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = 0.f;
  const unsigned first = blockIdx.x * (unsigned)a.vox_per_block;
  const unsigned last = first + a.vox_per_block < (unsigned)a.n_p ? first + a.vox_per_block : (unsigned)a.n_p;
  for (unsigned n0 = first; n0 < last; n0 += NV) {
    __syncthreads();
    // stage: thread t loads float4 pieces; a voxel's row is CA / 4 (CB / 4) pieces
    for (int i = threadIdx.x; i < NV * (CA / 4); i += 256) {
      const int v = i / (CA / 4), c4 = i % (CA / 4);
      const unsigned n = n0 + v;
      f32x4 val = splat4(0.f);
      if (n < last) val = ld4(a.tp + (size_t)n * CA + 4 * c4);
      *reinterpret_cast<f32x4*>(&sp[v][4 * c4]) = val;
    }
    for (int i = threadIdx.x; i < NV * (CB / 4); i += 256) {
      const int v = i / (CB / 4), c4 = i % (CB / 4);
      const unsigned n = n0 + v;
      f32x4 val = splat4(0.f);
      size_t q;
      if (n < last && wgrad_q(a, n, S, kz, ky, kx, &q)) val = ld4(a.tq + q * CB + 4 * c4);
      *reinterpret_cast<f32x4*>(&sq[v][4 * c4]) = val;
    }
    __syncthreads();
#pragma unroll 8
    for (int v = 0; v < NV; ++v) {
      float vp[TA], vq[TB];
#pragma unroll
      for (int i = 0; i < TA; ++i) vp[i] = sp[v][ta + 16 * i];
#pragma unroll
      for (int j = 0; j < TB; ++j) vq[j] = sq[v][tb + 16 * j];
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = fmaf(vp[i], vq[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
      if (acc[i][j] != 0.f) atomicAdd(a.dw + ((size_t)(ta + 16 * i) * CB + (tb + 16 * j)) * 27 + k, acc[i][j]);
}

// bias gradient: column sums of d_out [N][C]
template <int C>
__global__ void __launch_bounds__(256) channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, long long N, int rows_per_block) {
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  const long long first = (long long)blockIdx.x * rows_per_block;
  const long long last = first + rows_per_block < N ? first + rows_per_block : N;
#pragma unroll 4
  for (long long n = first + threadIdx.x; n < last; n += 256) {
    const float* p = x + (size_t)n * C;
    if constexpr (C % 4 == 0) {
#pragma unroll
      for (int i = 0; i < C / 4; ++i) { const f32x4 v = ld4(p + 4 * i); acc[4 * i] += v[0]; acc[4 * i + 1] += v[1]; acc[4 * i + 2] += v[2]; acc[4 * i + 3] += v[3]; }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += p[c];
    }
  }
  __shared__ float red[4][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v = acc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < C) {      // one atomic per channel and block (same-address atomics serialise in the L2)
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0.f) atomicAdd(out + threadIdx.x, v);
  }
}

// ---- weight gradients on the matrix cores (round 5): dW_k[a][b] = sum over voxel pairs of TP[p][a] TQ[q_k(p)][b] IS a
// matrix product with the VOXELS as the k axis -- v_mfma_f32_16x16x4_f32 takes four voxel pairs per instruction, exact fp32.
// Lane (i, kk) supplies TP[p0 + kk][a0 + i] as the A operand and TQ[q_k(p0 + kk)][b0 + i] as the B operand (a wave-load =
// four 64-byte channel runs); the accumulator tiles (rows a, columns b) of NT consecutive taps stay in registers for the
// whole launch, TP is read once per NT taps and the neighbours TQ[q_k] of consecutive taps hit in the L1.  No LDS stage, no
// barrier in the loop.
//
// What bounds these kernels is the address arithmetic around each 32-cycle MFMA, so the work unit is a ROW SEGMENT: a wave
// takes (b, z, y) of the P grid and `seg` consecutive x.  Everything that depends on the row -- the neighbour rows' byte
// offsets, whether z + dz / y + dy leave the volume -- is wave-uniform and computed once per unit; per step of four voxels a
// tap costs one add.  Loads go through BUFFER descriptors: an operand that does not exist (a neighbour outside the volume,
// a voxel past the segment) gets an offset past the descriptor's extent and the hardware returns 0 -- no branch, no select,
// nothing for the compiler to sink a load under (with `ok ? p[i] : 0` it moved each load into a divergent block with its
// own wait: one memory round trip per tap).  Two operand sets are in flight: the loads of the next step are issued before
// the MFMAs of the current one.  The launcher keeps both tensors below 2^30 bytes (else: the VALU kernels above).
constexpr unsigned kWgOut = 0x80000000u;     // a lane's "no such voxel"; stays out of range after adding a row offset < 2^30
constexpr unsigned kWgOutRow = 0x40000000u;  // a row's "no such row"; pushes every lane offset (< 2^30, or kWgOut) out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float wg_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// unit u -> row (b, z, y) of the P grid and the x range [xs, xe) of its segment (all wave-uniform)
struct WgradUnit {
  int b, z, y, xs, xe;
};
__device__ __forceinline__ WgradUnit wgrad_unit(const WgradArgs& a, unsigned u) {
  WgradUnit w;
  // (segment-major: the four waves of a block take the same segment of four consecutive rows y, and share the rows
  // y - 1 .. y + 4 of the fine grid in the L1)
  const unsigned rows = (unsigned)(a.B * a.Dp * a.Hp);
  const unsigned sg = u / rows;
  unsigned r = u % rows;
  w.y = (int)(r % (unsigned)a.Hp);
  r /= (unsigned)a.Hp;
  w.z = (int)(r % (unsigned)a.Dp);
  w.b = (int)(r / (unsigned)a.Dp);
  w.xs = (int)sg * a.seg;
  w.xe = w.xs + a.seg < a.Wp ? w.xs + a.seg : a.Wp;
  return w;
}

template <int CA, int CB, int S, int NT>
__global__ void __launch_bounds__(256) conv3d_wgrad_mfma_kernel(WgradArgs a) {
  static_assert(CA % 16 == 0 && CB % 16 == 0 && 27 % NT == 0, "whole 16-channel tiles, whole tap groups");
  constexpr int NA = CA / 16, NB = CB / 16;
  __shared__ __attribute__((aligned(16))) f32x4 red[NT * NA * NB][64];
  const int k0 = blockIdx.y * NT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  f32x4 acc[NT][NA][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ta = 0; ta < NA; ++ta)
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) acc[t][ta][tb] = splat4(0.f);
  const __amdgpu_buffer_rsrc_t rp = wg_rsrc(a.tp, (unsigned)a.n_p * CA * 4u);
  const __amdgpu_buffer_rsrc_t rq = wg_rsrc(a.tq, (unsigned)(a.B * a.Dq * a.Hq * a.Wq) * CB * 4u);
  struct Operands {
    float ap[NA], bq[NT][NB];
  };
  const unsigned units = (unsigned)(a.B * a.Dp * a.Hp) * (unsigned)a.nseg;
  for (unsigned u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
    const WgradUnit w = wgrad_unit(a, u);
    const unsigned prow = (unsigned)((w.b * a.Dp + w.z) * a.Hp + w.y) * (unsigned)a.Wp;      // voxel index of the P row
    unsigned qrow[NT];                                                                       // byte offset of tap t's Q row
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = k0 + t, qz = w.z * S + k / 9 - 1, qy = w.y * S + (k / 3) % 3 - 1;
      const bool ok = (unsigned)qz < (unsigned)a.Dq && (unsigned)qy < (unsigned)a.Hq;
      qrow[t] = ok ? (unsigned)((w.b * a.Dq + qz) * a.Hq + qy) * (unsigned)a.Wq * (CB * 4u) : kWgOutRow;
    }
    auto load = [&](int x0, Operands& o) {
      const int x = x0 + kk;
      const bool okp = x < w.xe;
      const unsigned poff = okp ? ((prow + (unsigned)x) * CA + i) * 4u : kWgOut;
#pragma unroll
      for (int ta = 0; ta < NA; ++ta) o.ap[ta] = wg_load(rp, poff + 64u * ta);
      unsigned qx[3];                                  // the lane's byte offset within a Q row, per dx
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int xq = x * S + d - 1;
        qx[d] = (okp && (unsigned)xq < (unsigned)a.Wq) ? ((unsigned)xq * CB + i) * 4u : kWgOut;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // (k0 is a multiple of 3 unless NT = 1, where the tap's dx is a wave-uniform pick)
        const unsigned qxt = NT % 3 == 0 ? qx[t % 3] : (k0 % 3 == 0 ? qx[0] : k0 % 3 == 1 ? qx[1] : qx[2]);
        const unsigned qoff = qxt + qrow[t];
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) o.bq[t][tb] = wg_load(rq, qoff + 64u * tb);
      }
    };
    auto contract = [&](const Operands& o) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ta = 0; ta < NA; ++ta)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) acc[t][ta][tb] = mfma16(o.ap[ta], o.bq[t][tb], acc[t][ta][tb]);
    };
    Operands A, B;
    load(w.xs, A);
    for (int x0 = w.xs; x0 < w.xe; x0 += 8) {      // (a set past the end is zeros: at most one idle contraction)
      load(x0 + 4, B);
      contract(A);
      load(x0 + 8, A);
      contract(B);
    }
  }
  // the four waves' partial tiles -> wave 0 (three rounds through one tile set of LDS), then one atomic per value
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ta = 0; ta < NA; ++ta)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) red[(t * NA + ta) * NB + tb][lane] = acc[t][ta][tb];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ta = 0; ta < NA; ++ta)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) acc[t][ta][tb] += red[(t * NA + ta) * NB + tb][lane];
    }
  }
  if (wave == 0) {     // lane (g, j): rows a = 16 ta + 4 g + r, column b = 16 tb + j
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ta = 0; ta < NA; ++ta)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = acc[t][ta][tb][r];
            if (x != 0.f) atomicAdd(a.dw + ((size_t)(16 * ta + 4 * g + r) * CB + 16 * tb + j) * 27 + k0 + t, x);
          }
  }
}

// ... and the layers with 8 channels on the fine side (the 8 x 8 head at full resolution, 16 x 8 of conv1 / conv11): TWO taps
// share an instruction -- columns (t, b) = 2 taps x 8 channels, rows a (8 of 16 used for CA = 8); the 27 taps are 14
// instructions per four voxel pairs, TP[p] is loaded once for all of them.  The lane's tap of pair m is 2 m + h (h = the
// lane's column half), so the row offsets are per lane here (14 registers per unit); which of its taps look left / right
// of x is a per-lane constant bit mask, and a step turns "x - 1 / x + 1 leaves the row" into the set of pairs to blank.
template <int CA, int S>
__global__ void __launch_bounds__(256) conv3d_wgrad_mfma8_kernel(WgradArgs a) {
  static_assert(CA == 8 || CA == 16, "rows");
  constexpr int CB = 8, NP = 14;                       // tap pairs (2 m, 2 m + 1); the last one is tap 26 alone
  __shared__ __attribute__((aligned(16))) f32x4 red[NP][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4, h = i >> 3;
  f32x4 acc[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) acc[m] = splat4(0.f);
  const __amdgpu_buffer_rsrc_t rp = wg_rsrc(a.tp, (unsigned)a.n_p * CA * 4u);
  const __amdgpu_buffer_rsrc_t rq = wg_rsrc(a.tq, (unsigned)(a.B * a.Dq * a.Hq * a.Wq) * CB * 4u);
  // bit m of mdx[d]: this lane's tap of pair m has dx = d - 1
  unsigned mdx[3] = {0u, 0u, 0u};
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    const int k = 2 * m + h;
#pragma unroll
    for (int d = 0; d < 3; ++d) mdx[d] |= (k % 3 == d ? 1u : 0u) << m;
  }
  struct Operands {
    float ap, bq[NP];
  };
  const unsigned units = (unsigned)(a.B * a.Dp * a.Hp) * (unsigned)a.nseg;
  for (unsigned u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
    const WgradUnit w = wgrad_unit(a, u);
    const unsigned prow = (unsigned)((w.b * a.Dp + w.z) * a.Hp + w.y) * (unsigned)a.Wp;
    // byte offset of (the lane's tap of pair m, channel i & 7) for the voxel x = 0 of the row; "negative" values wrap and come
    // back in range when x is added, or are blanked because x - 1 is outside
    unsigned qrow[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int k = 2 * m + h, qz = w.z * S + k / 9 - 1, qy = w.y * S + (k / 3) % 3 - 1, dx = k % 3 - 1;
      const bool ok = k < 27 && (unsigned)qz < (unsigned)a.Dq && (unsigned)qy < (unsigned)a.Hq;
      qrow[m] = ok ? ((unsigned)((w.b * a.Dq + qz) * a.Hq + qy) * (unsigned)a.Wq + (unsigned)(kk * S + dx)) * (CB * 4u) + (unsigned)(i & 7) * 4u
                   : kWgOut;
    }
    auto load = [&](int x0, Operands& o) {
      const int x = x0 + kk;
      const bool okp = x < w.xe;
      o.ap = wg_load(rp, (okp && i < CA) ? ((prow + (unsigned)x) * CA + i) * 4u : kWgOut);
      unsigned blank = 0u;
#pragma unroll
      for (int d = 0; d < 3; ++d) blank |= (okp && (unsigned)(x * S + d - 1) < (unsigned)a.Wq) ? 0u : mdx[d];
      const unsigned xb = (unsigned)x0 * (S * CB * 4u);       // wave-uniform
#pragma unroll
      for (int m = 0; m < NP; ++m) o.bq[m] = wg_load(rq, ((blank >> m) & 1u) ? kWgOut : qrow[m] + xb);
    };
    auto contract = [&](const Operands& o) {
#pragma unroll
      for (int m = 0; m < NP; ++m) acc[m] = mfma16(o.ap, o.bq[m], acc[m]);
    };
    Operands A, B;
    load(w.xs, A);
    for (int x0 = w.xs; x0 < w.xe; x0 += 8) {
      load(x0 + 4, B);
      contract(A);
      load(x0 + 8, A);
      contract(B);
    }
  }
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < NP; ++m) red[m][lane] = acc[m];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < NP; ++m) acc[m] += red[m][lane];
    }
  }
  if (wave == 0) {     // lane (g, j): rows a = 4 g + r, column j = (tap 2 m + (j >> 3), b = j & 7)
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int m = 0; m < NP; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int aa = 4 * g + r, k = 2 * m + (j >> 3);
        const float x = acc[m][r];
        if (aa < CA && k < 27 && x != 0.f) atomicAdd(a.dw + ((size_t)aa * CB + (j & 7)) * 27 + k, x);
      }
  }
}

#ifndef UFR_CONV3D_WGRAD_MFMA
#define UFR_CONV3D_WGRAD_MFMA 1
#endif
#ifndef UFR_WG_NT_SMALL
#define UFR_WG_NT_SMALL 9
#endif
#ifndef UFR_WG_NT_MID
#define UFR_WG_NT_MID 3
#endif
#ifndef UFR_WG_WAVES
#define UFR_WG_WAVES 4096       // waves in flight per launch: 4 per SIMD
#endif
#ifndef UFR_WG_WAVES8
#define UFR_WG_WAVES8 3072      // ... 3 per SIMD at the 8-channel kernel's ~150 registers
#endif
template <int CA, int CB, int S>
hipError_t launch_wgrad_t(WgradArgs a, hipStream_t s) {
#if UFR_CONV3D_WGRAD_MFMA
  // (the matrix-core kernels address both tensors with byte offsets below kWgOutRow)
  const bool small = a.n_p * CA * 4 < (1ll << 30) && (long long)a.B * a.Dq * a.Hq * a.Wq * CB * 4 < (1ll << 30);
  constexpr bool kPair = CA % 16 == 0 && CB % 16 == 0, kEight = CB == 8 && (CA == 8 || CA == 16);
  if constexpr (kPair || kEight) {
    // measured per layer at the three stages' sizes (tools/dev/conv3d_bwd_probe.py): below these volumes the per-unit setup and the
    // closing reduction outweigh the contraction, and the VALU kernels below win
    const bool large = a.n_p >= (kPair ? 100000 : 500000);
    if (small && large) {
      // units: whole rows, halved until there are a few per wave slot (never below 32 voxels = 8 steps)
      const long long rows = (long long)a.B * a.Dp * a.Hp;
      a.seg = (a.Wp + 3) / 4 * 4;
      while (rows * ((a.Wp + a.seg - 1) / a.seg) < 4 * UFR_WG_WAVES && a.seg > 32) a.seg = (a.seg / 2 + 3) / 4 * 4;
      a.nseg = (a.Wp + a.seg - 1) / a.seg;
      const long long units = rows * a.nseg;
      if constexpr (kPair) {
        constexpr int NT = CA * CB <= 256 ? UFR_WG_NT_SMALL : CA * CB <= 1024 ? UFR_WG_NT_MID : 1;   // 4 NT (CA/16) (CB/16) accumulator registers
        long long blocks = UFR_WG_WAVES / 4 * NT / 27;
        if (blocks < 16) blocks = 16;
        if (blocks > (units + 3) / 4) blocks = (units + 3) / 4;
        hipLaunchKernelGGL((conv3d_wgrad_mfma_kernel<CA, CB, S, NT>), dim3((unsigned)blocks, 27 / NT), dim3(256), 0, s, a);
      } else {
        long long blocks = UFR_WG_WAVES8 / 4;       // exactly the resident waves: the units are dealt round-robin
        if (blocks > (units + 3) / 4) blocks = (units + 3) / 4;
        hipLaunchKernelGGL((conv3d_wgrad_mfma8_kernel<CA, S>), dim3((unsigned)blocks), dim3(256), 0, s, a);
      }
      return hipGetLastError();
    }
  }
#endif
  if constexpr (CB <= 8 && CA * CB <= 128) {
    // CAG coarse-side channels per thread (CAG CB <= 64), the groups on blockIdx.y; NT taps per pass (NT CAG CB <= 192), the
    // passes on blockIdx.z
    constexpr int CAG = CA < 8 ? CA : 8;
    constexpr int NT = CAG * CB <= 8 ? 9 : 3;
    static_assert(CA % CAG == 0, "channel groups");
    long long blocks = (a.n_p + 16383) / 16384;      // >= 64 voxels per thread: the closing reduction stays a small share
    if (blocks > 512) blocks = 512;                   // ... and few same-address atomics
    if (blocks < 1) blocks = 1;
    a.vox_per_block = (int)(((a.n_p + blocks - 1) / blocks + 255) / 256 * 256);
    blocks = (a.n_p + a.vox_per_block - 1) / a.vox_per_block;
    hipLaunchKernelGGL((conv3d_wgrad_voxel_kernel<CAG, CB, NT, S>), dim3((unsigned)blocks, CA / CAG, 27 / NT), dim3(256), 0, s, a, CA);
  } else {
    // blocks per tap: enough that 27 taps fill the chip even at the coarsest level (15 k voxels), few enough that the
    // closing atomics (one per value and block) stay negligible
    long long blocks = (a.n_p + 8191) / 8192;
    if (blocks < 40) blocks = 40;
    if (blocks > 2048) blocks = 2048;
    a.vox_per_block = (int)(((a.n_p + blocks - 1) / blocks + 255) / 256 * 256);
    blocks = (a.n_p + a.vox_per_block - 1) / a.vox_per_block;
    hipLaunchKernelGGL((conv3d_wgrad_pair_kernel<CA, CB, S>), dim3((unsigned)blocks, 27), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace

// d_weight (reference layout) and d_bias (nullable) are ACCUMULATED into: the caller zeroes them (or keeps a running sum)
hipError_t launch_conv3d_bwd_weight(const float* in, const float* d_out, float* d_weight, float* d_bias, int B, int D, int H, int W,
                                    int cin, int cout, int mode, hipStream_t s) {
  WgradArgs a;
  a.dw = d_weight; a.B = B; a.vox_per_block = 0; a.seg = 0; a.nseg = 0;
  int ca, cb, S;
  long long n_out;
  if (mode == kConvS1) {
    a.tp = d_out; a.tq = in; a.Dp = D; a.Hp = H; a.Wp = W; a.Dq = D; a.Hq = H; a.Wq = W; ca = cout; cb = cin; S = 1;
    n_out = (long long)B * D * H * W;
  } else if (mode == kConvS2) {
    a.tp = d_out; a.tq = in; a.Dp = D / 2; a.Hp = H / 2; a.Wp = W / 2; a.Dq = D; a.Hq = H; a.Wq = W; ca = cout; cb = cin; S = 2;
    n_out = (long long)B * (D / 2) * (H / 2) * (W / 2);
  } else {
    a.tp = in; a.tq = d_out; a.Dp = D; a.Hp = H; a.Wp = W; a.Dq = 2 * D; a.Hq = 2 * H; a.Wq = 2 * W; ca = cin; cb = cout; S = 2;
    n_out = (long long)B * 8 * D * H * W;
  }
  a.n_p = (long long)B * a.Dp * a.Hp * a.Wp;
  if (a.n_p >= (1ll << 31) - 65536 || n_out >= (1ll << 40)) return hipErrorInvalidValue;     // the kernels index voxels with 32 bits
  hipError_t e = hipErrorInvalidValue;
#ifndef UFR_CONV3D_WGRAD_FP32_ONLY
  // round 6: the 16-bit matrix-core kernels (conv3d_wgrad_planes.hip) where they have the shape; they take the bias gradient
  // along when TP is d_out (the convolutions)
  {
    const bool bias_rides = d_bias != nullptr && mode != kDeconvS2;
    e = launch_conv3d_wgrad_planes(a.tp, a.tq, a.dw, bias_rides ? d_bias : nullptr, B, a.Dp, a.Hp, a.Wp, a.Dq, a.Hq, a.Wq, ca, cb, S, s);
    if (e == hipSuccess) {
      if (!d_bias || bias_rides) return hipSuccess;
    } else if (e != hipErrorInvalidValue) {
      return e;
    }
  }
  if (e != hipSuccess) {
#endif
  e = hipErrorInvalidValue;
#define UFR_WG_CASE(A_, B_, S_) if (ca == A_ && cb == B_ && S == S_) e = launch_wgrad_t<A_, B_, S_>(a, s);
  UFR_WG_CASE(8, 1, 1)     // conv0
  UFR_WG_CASE(16, 8, 2)    // conv1, conv11
  UFR_WG_CASE(16, 16, 1)   // conv2
  UFR_WG_CASE(32, 16, 2)   // conv3, conv9
  UFR_WG_CASE(32, 32, 1)   // conv4
  UFR_WG_CASE(64, 32, 2)   // conv5, conv7
  UFR_WG_CASE(64, 64, 1)   // conv6
  UFR_WG_CASE(8, 8, 1)     // features head
  UFR_WG_CASE(1, 8, 1)     // weights head
#undef UFR_WG_CASE
#ifndef UFR_CONV3D_WGRAD_FP32_ONLY
  }
#endif
  if (e != hipSuccess) return e;
  if (d_bias) {
    const int rows = 16384;    // 64 rows per thread, four loads in flight; ~500 blocks: few same-address atomics
    const long long blocks = (n_out + rows - 1) / rows;
#define UFR_CS_CASE(C_) if (cout == C_) hipLaunchKernelGGL((channel_sum_kernel<C_>), dim3((unsigned)blocks), dim3(256), 0, s, d_out, d_bias, n_out, rows);
    UFR_CS_CASE(1) UFR_CS_CASE(8) UFR_CS_CASE(16) UFR_CS_CASE(32) UFR_CS_CASE(64)
#undef UFR_CS_CASE
    if (cout != 1 && cout != 8 && cout != 16 && cout != 32 && cout != 64) return hipErrorInvalidValue;
    e = hipGetLastError();
  }
  return e;
}

}  // namespace ufr
