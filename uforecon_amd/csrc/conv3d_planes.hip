// The 3x3x3 layers of the frustum U-Nets (every layer with at least 8 input channels: stride 1, stride 2, transposed stride 2)
// on the 16-bit matrix cores, activations staged through LDS.
//   CostRegNetWeight  code1/encoder_utils/fmt/module.py:502-543   (`feature_volume.cost_reg_2`: plain Conv3d with bias)
//   CostRegNet        code1/encoder_utils/fmt/module.py:469-500   (Conv3d + BatchNorm + ReLU)
//
// Why a second kernel family.  The fp32 kernels of conv3d.hip run the full- and half-resolution layers (8 / 16 channels:
// two thirds of a U-Net's time) at a third of the vector peak, and neither halving their FMA instructions nor their LDS
// weight reads changes that (conv3d.hip, conv3d_kernel): every output voxel asks the L1 for its 27 neighbours again, one
// bounds-checked 16-byte load per four channels and tap.  An implicit GEMM that fetches its operands the same way inherits
// the same bound.  So here
//  * a workgroup owns a BRICK of output voxels and stages the brick's input halo ONCE: coalesced row loads through a
//    bounded buffer descriptor (zero padding = an offset past the extent), each value split into two fp16 planes
//    hi = fp16(s x), lo = fp16(s x - hi) on the way (s = a power of two from the tensor's MEASURED |max|, handed over by
//    the producing layer: `in_absmax`), 3 .. 5 fetched voxels per output instead of 27; halo planes are 8-channel-chunk
//    major, so a lane group's 16 lanes read 256 contiguous bytes whatever the channel count;
//  * the convolution is an implicit GEMM on v_mfma_f32_16x16x32_f16: rows = 16 output channels, columns = 16 consecutive
//    x of the brick, k = 32 = 4 taps x 8 channels | 2 x 16 | 1 x 32 | half a tap of 64: a lane reads ITS tap's neighbour
//    of voxel j as one ds_read_b128 per plane; the operands of step s + 2 are requested before the MFMAs of step s;
//  * a product is three plane pairs accumulated in fp32 (w_lo x_hi + w_hi x_lo + w_hi x_hi: 22 significand bits, the
//    scheme of the transformer kernels, ufr_layout_f16.h); the weights' planes [k-step][row tile][plane][lane] are made once
//    per weight version by conv3d_planes_prep (scale 2^e from max |w|) and sit in LDS (<= 32 KiB) or are read from L2;
//  * stride 2 = the same with a 2T + 1 halo; TRANSPOSED stride 2 = a 2 x 2 x 2 convolution of the INPUT grid with 8 x cout
//    output rows, one group per output parity class (structural zeros where a class has no such tap);
//  * persistent workgroups walk contiguous, XCD-aware runs of bricks with the next brick's halo loads in flight;
//  * the exact power-of-two descale, bias, folded BatchNorm, ReLU, the U-Net's skip addition, the two heads' (B,C,D,H,W)
//    layout + sigmoid ride in the store, which also raises `out_absmax` for the next layer (one guarded atomic per wave).
// The data gradients are the same kernels (stride 1: mirrored taps on the swapped weight, `flip`; strided <-> transposed on
// the forward weight as it stands).  conv0 (one input channel) stays on conv3d.hip; weight gradients: conv3d_wgrad_planes.hip.
#include <hip/hip_runtime.h>

#include "ufr_device.h"
#include "ufr_internal.h"
#include "weight_stream.h"   // static_for

namespace ufr {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int kPlanesHeader = 256;   // bytes in front of the weight planes: float [0] = 1 / s_w, [1] = max |w|

// taps of the implicit GEMM: 27 (convolutions); 8 for the TRANSPOSED stride-2 layer written as a 2 x 2 x 2 convolution of the
// input grid with 8 x cout output rows (one group per output parity class, structural zeros where a class has no such tap)
__host__ __device__ constexpr int planes_ntaps(int mode) { return mode == 2 ? 8 : 27; }
__host__ __device__ constexpr int planes_ksteps(int cin, int mode = 0) { return (planes_ntaps(mode) * cin + 31) / 32; }
__host__ __device__ constexpr int planes_tiles(int cout_total) { return (cout_total + 15) / 16; }

// the power of two s with s * m in [2^14, 2^15) (m > 0, finite); 1 for m == 0 / not finite
__device__ __forceinline__ float plane_scale(float m) {
  if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(m, &e);          // m = f 2^e, f in [0.5, 1)
  int k = 15 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return ldexpf(1.f, k);
}

// (k-step, lane group g, element h) -> (tap 0..26 or -1, input channel)
// k = 32 per instruction: 4 taps x 8 channels | 2 taps x 16 | 1 tap x 32 | half a tap (32 of 64 channels)
template <int CIN>
__host__ __device__ constexpr int planes_tap(int ks, int g) {
  return CIN == 8 ? 4 * ks + g : CIN == 16 ? 2 * ks + (g >> 1) : CIN == 32 ? ks : (ks >> 1);
}
template <int CIN>
__host__ __device__ constexpr int planes_c0(int ks, int g) {
  return CIN == 8 ? 0 : CIN == 16 ? 8 * (g & 1) : CIN == 32 ? 8 * g : 32 * (ks & 1) + 8 * g;
}

struct PrepArgs {
  const float* weight;   // conv layout [cout][cin][27]; flip: the layer's FORWARD weight [cin of this launch][cout][27], taps mirrored;
                         // transposed: [cin][cout][27]
  const float* weight2;  // heads: [cout2][cin][27] (rows cout .. cout + cout2 - 1), nullable
  char* ws;
  int cout, cout2, flip;
  int max_ready;         // 1: header word 1 already holds max |w| (measured by absmax_kernel launches: the large layers)
};
// transposed stride 2, one axis: the kernel index that links output parity p with input offset o (0 / +1), or -1
// (conv3d.hip: parity 0 <- {k = 1, i = m}; parity 1 <- {k = 0, i = m + 1}, {k = 2, i = m})
__host__ __device__ constexpr int t2_k(int p, int o) { return p == 0 ? (o == 0 ? 1 : -1) : (o == 0 ? 2 : 0); }

// grid-parallel (the 64 -> 64 layer has 110 592 weights: as one workgroup this kernel took longer than the convolution it
// prepares).  Every workgroup measures max |w| over ALL the weights itself (<= 110 592 floats, L2-resident, loads in batches
// of eight: a few microseconds) -- one launch per layer instead of memset + measure + prepare, which a training step pays
// for every layer in both directions.
template <int CIN, int MODE = 0>
__global__ void __launch_bounds__(256) conv3d_planes_prep(PrepArgs a) {
  constexpr int KS = planes_ksteps(CIN, MODE);
  const int NT = MODE == 2 ? a.cout / 2 : planes_tiles(a.cout + a.cout2);      // transposed: 8 classes x cout rows
  __shared__ float red[4];
  float m = 0.f;
  if (a.max_ready) {
    m = reinterpret_cast<const float*>(a.ws)[1];
  } else {
    const int n1 = a.cout * CIN * 27, n2 = a.weight2 ? a.cout2 * CIN * 27 : 0;
    for (int i0 = 0; i0 < n1 + n2; i0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u + threadIdx.x;
        v[u] = i < n1 ? a.weight[i] : (i < n1 + n2 ? a.weight2[i - n1] : 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  const float sw = plane_scale(m);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(a.ws)[0] = 1.f / sw;     // (word 1: only the measuring launches write it)
  _Float16* planes = reinterpret_cast<_Float16*>(a.ws + kPlanesHeader);
  const int total = KS * NT * 64 * 8;
  const int e0 = blockIdx.x * (256 * 8);
  float wv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = e0 + 256 * u + threadIdx.x;
    const int h = e & 7, lane = (e >> 3) & 63, t = (e >> 9) % NT, ks = (e >> 9) / NT;
    const int g = lane >> 4, i = lane & 15;
    const int tap = planes_tap<CIN>(ks, g), ci = planes_c0<CIN>(ks, g) + h, co = 16 * t + i;
    float w = 0.f;
    if constexpr (MODE == 2) {
      if (e < total && tap < 8) {
        const int cls = co / a.cout, cr = co - cls * a.cout;
        const int kz = t2_k((cls >> 2) & 1, (tap >> 2) & 1), ky = t2_k((cls >> 1) & 1, (tap >> 1) & 1), kx = t2_k(cls & 1, tap & 1);
        if (kz >= 0 && ky >= 0 && kx >= 0) w = a.weight[((size_t)ci * a.cout + cr) * 27 + (kz * 3 + ky) * 3 + kx];
      }
    } else if (e < total && tap < 27) {
      if (co < a.cout) w = a.flip ? a.weight[((size_t)ci * a.cout + co) * 27 + (26 - tap)] : a.weight[((size_t)co * CIN + ci) * 27 + tap];
      else if (co < a.cout + a.cout2) w = a.weight2[((size_t)(co - a.cout) * CIN + ci) * 27 + tap];
    }
    wv[u] = w;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = e0 + 256 * u + threadIdx.x;
    if (e >= total) continue;
    const int h = e & 7, lane = (e >> 3) & 63, t = (e >> 9) % NT, ks = (e >> 9) / NT;
    const float v = wv[u] * sw;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t base = ((size_t)(ks * NT + t) * 2) * 512 + lane * 8 + h;     // [ks][t][plane][lane][8]
    planes[base] = hi;
    planes[base + 512] = lo;
  }
}

struct PlanesArgs {
  const float* in;            // [B][D][H][W][CIN]
  const unsigned* in_absmax;  // bit pattern of a non-negative float >= max |in|
  const char* ws;             // conv3d_planes_prep's output
  const float* bias;          // [cout] (nullable)
  const float* scale;         // folded BatchNorm (nullable together)
  const float* shift;
  const float* skip;          // channel-last, the output's shape (nullable)
  float* out;                 // channel-last [B][D][H][W][cout], or (B,cout,D,H,W) when ncdhw
  float* out2;                // heads: (B,cout2,D,H,W), sigmoid applied
  unsigned* out_absmax;       // raised to max |out| (nullable)
  int B, D, H, W;         // input extent
  int Do, Ho, Wo;         // output extent (stride 2: ceil(n / 2))
  int cout, cout2, relu, ncdhw;
  int nbx, nby, nbz;
};

// fp16 planes of a value pair scaled by the power of two m (weight_stream_f16.h: split_pair)
__device__ __forceinline__ void split2(float a, float b, float m, unsigned& h, unsigned& l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(m));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(m));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(m), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(m), "v"(h));
}

template <int CIN, int MODE = 0>
struct Brick {   // the brick of a workgroup -- output voxels (transposed: INPUT voxels, 8 outputs each) -- and its input halo
  static constexpr int S = MODE == 1 ? 2 : 1;
  static constexpr int TX = MODE == 1 ? (CIN == 8 ? 32 : 16) : (CIN == 8 ? 64 : CIN == 16 ? 32 : 16), TY = 4,
                       TZ = (MODE == 0 && (CIN == 8 || CIN == 32)) || (MODE == 2 && CIN == 32) ? 2 : 1;
  // stride S: S (T - 1) + 3 inputs per axis from -1; transposed: offsets 0 / +1 only
  static constexpr int HX = MODE == 2 ? TX + 1 : S * (TX - 1) + 3, HY = MODE == 2 ? TY + 1 : S * (TY - 1) + 3,
                       HZ = MODE == 2 ? TZ + 1 : S * (TZ - 1) + 3;
  static constexpr int halo = HX * HY * HZ;
  static constexpr int tiles = TX / 16 * TY * TZ;          // 16-voxel column tiles
  static constexpr int per_wave = tiles / 4;
  static constexpr int plane_bytes = halo * CIN * 2;       // one fp16 plane of the halo
};

#ifndef UFR_C3P_ABL
#define UFR_C3P_ABL 0   // development ablations (timing only): 1 = one k-step, 2 = no halo loads, 3 = no stores
#endif

// Persistent workgroups (two per CU), each walking a contiguous run of bricks with the NEXT brick's halo loads in flight
// while the current one computes: as one workgroup per brick the phases of a brick -- load round trip, split + LDS write,
// MFMAs, stores -- ran one after the other and only two bricks per CU overlapped (ablations on the full-resolution 8 -> 8
// layer: 0.345 ms = 0.14 fixed + 0.09 MFMA + 0.09 stores + 0.03 loads, a sum, not a maximum).
// NT output row tiles in passes of NTG (the 64 -> 64 layer: two passes of two tiles -- four tiles' weight fragments three
// k-steps deep are 96 registers, and the kernel spilled 800; the halo is staged once, its operands are read per pass)
template <int CIN, int NT, int NTG = NT, int MODE = 0>
__global__ void __launch_bounds__(256, 2) conv3d_planes_kernel(PlanesArgs a) {
  typedef Brick<CIN, MODE> Bk;
  constexpr int S = Bk::S, O = MODE == 2 ? 0 : 1;          // stride; the halo starts O voxels in front of the brick
  constexpr bool T2 = MODE == 2;
  constexpr int KS = planes_ksteps(CIN, MODE), NTAPS = planes_ntaps(MODE);
  constexpr int VT = Bk::per_wave;
  // the weights' planes: copied into LDS once per workgroup when they fit beside the halo (<= 32 KiB: the 8- / 16-channel
  // layers), else every wave reads its fragments from global memory (108 / 432 KiB for 32 -> 32 / 64 -> 64: L2-resident, the
  // four waves of a workgroup read the same 1 KiB pieces, and those layers' grids are small)
  constexpr bool kWLds = KS * NT * 2048 <= 32768;
  constexpr int w_bytes = kWLds ? KS * NT * 2048 : 0;
  constexpr int HR = (Bk::halo + 255) / 256, C8 = CIN / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const w_lds = smem;                                  // [ks][t][plane][lane] 16 B
  // halo planes, 8-channel-chunk major: [chunk c8][halo voxel][8] fp16 -- the 16 lanes of a lane group read 16 consecutive
  // voxels' chunks = 256 contiguous bytes whatever CIN is (voxel-major, CIN = 16 .. 64 would stride the lanes by 32 .. 128
  // bytes: 2- to 8-way bank conflicts on every operand read)
  char* const x_hi = smem + w_bytes;
  char* const x_lo = x_hi + Bk::plane_bytes;
  const char* const w_glb = a.ws + kPlanesHeader;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;

  // ---- scales (wave-uniform)
  const float in_max = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)*a.in_absmax));
  const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, plane_scale(in_max))));
  const float dsc = *reinterpret_cast<const float*>(a.ws) / sx;
  const size_t frame = (size_t)a.D * a.H * a.W * CIN * 4;

  // ---- which bricks.  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2 (the grid is a multiple of
  // 8): workgroup w sits on XCD w % 8 and takes the (w / 8)-th share of that XCD's CONTIGUOUS eighth of the bricks, so that
  // bricks whose halos overlap are fetched through one L2.
  const int n_bricks = a.nbx * a.nby * a.nbz * a.B;
  int k_begin, k_end;
  {
    const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8, per_xcd = gridDim.x / 8;
    const int q = n_bricks / 8, r = n_bricks % 8;
    const int x_begin = xcd * q + (xcd < r ? xcd : r), x_count = q + (xcd < r ? 1 : 0);      // this XCD's run
    const int q2 = x_count / per_xcd, r2 = x_count % per_xcd;
    k_begin = x_begin + idx * q2 + (idx < r2 ? idx : r2);
    k_end = k_begin + q2 + (idx < r2 ? 1 : 0);
  }
  auto brick_of = [&](int k, int& bb, int& x0, int& y0, int& z0) __attribute__((always_inline)) {
    const int bx = k % a.nbx; k /= a.nbx;
    const int by = k % a.nby; k /= a.nby;
    bb = k / a.nbz;
    x0 = bx * Bk::TX; y0 = by * Bk::TY; z0 = (k % a.nbz) * Bk::TZ;
  };
  // every load of a brick's halo is issued before the first one is used (fully unrolled: a loop waits for each round trip)
  f32x4 xv[CIN <= 16 ? HR : 1][C8][2];
  auto halo_load = [&](int k) __attribute__((always_inline)) {
    int bb, x0, y0, z0;
    brick_of(k, bb, x0, y0, z0);
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(reinterpret_cast<const char*>(a.in) + (size_t)bb * frame, (unsigned)frame);
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int hv = tid + 256 * r;
      const int hx = hv % Bk::HX, hy = (hv / Bk::HX) % Bk::HY, hz = hv / (Bk::HX * Bk::HY);
      const int ix = S * x0 + hx - O, iy = S * y0 + hy - O, iz = S * z0 + hz - O;
      const bool ok = UFR_C3P_ABL != 2 && hv < Bk::halo && ix >= 0 && ix < a.W && iy >= 0 && iy < a.H && iz >= 0 && iz < a.D;
      const unsigned off = (unsigned)(((iz * a.H + iy) * a.W + ix) * (CIN * 4));
#pragma unroll
      for (int c8 = 0; c8 < C8; ++c8) {
        xv[r][c8][0] = buf_ld4(rin, ok ? off + c8 * 32 : kBufOut);
        xv[r][c8][1] = buf_ld4(rin, ok ? off + c8 * 32 + 16 : kBufOut);
      }
    }
  };
  auto halo_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int hv = tid + 256 * r;
      if (hv < Bk::halo && !(UFR_C3P_ABL == 4 && xv[r][0][0][0] != 12345.f)) {
#pragma unroll
        for (int c8 = 0; c8 < C8; ++c8) {
          const f32x4 v0 = xv[r][c8][0], v1 = xv[r][c8][1];
          unsigned h0, h1, h2, h3, l0, l1, l2, l3;
          split2(v0[0], v0[1], sx, h0, l0);
          split2(v0[2], v0[3], sx, h1, l1);
          split2(v1[0], v1[1], sx, h2, l2);
          split2(v1[2], v1[3], sx, h3, l3);
          *reinterpret_cast<u32x4v*>(x_hi + ((size_t)c8 * Bk::halo + hv) * 16) = u32x4v{h0, h1, h2, h3};
          *reinterpret_cast<u32x4v*>(x_lo + ((size_t)c8 * Bk::halo + hv) * 16) = u32x4v{l0, l1, l2, l3};
        }
      }
    }
  };

  // CIN >= 32 (small grids, a brick or two per workgroup, 128 .. 256 bytes per halo voxel): no prefetch across bricks -- the
  // halo is staged round by round (256 voxels at a time), so that only one round's loads are live
  constexpr bool kPrefetch = CIN <= 16;
  auto halo_direct = [&](int k) __attribute__((always_inline)) {
    int bb, x0, y0, z0;
    brick_of(k, bb, x0, y0, z0);
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(reinterpret_cast<const char*>(a.in) + (size_t)bb * frame, (unsigned)frame);
#pragma unroll 1
    for (int r = 0; r < HR; ++r) {
      const int hv = tid + 256 * r;
      const int hx = hv % Bk::HX, hy = (hv / Bk::HX) % Bk::HY, hz = hv / (Bk::HX * Bk::HY);
      const int ix = S * x0 + hx - O, iy = S * y0 + hy - O, iz = S * z0 + hz - O;
      const bool ok = hv < Bk::halo && ix >= 0 && ix < a.W && iy >= 0 && iy < a.H && iz >= 0 && iz < a.D;
      const unsigned off = (unsigned)(((iz * a.H + iy) * a.W + ix) * (CIN * 4));
      f32x4 t[C8][2];
#pragma unroll
      for (int c8 = 0; c8 < C8; ++c8) {
        t[c8][0] = buf_ld4(rin, ok ? off + c8 * 32 : kBufOut);
        t[c8][1] = buf_ld4(rin, ok ? off + c8 * 32 + 16 : kBufOut);
      }
      if (hv < Bk::halo) {
#pragma unroll
        for (int c8 = 0; c8 < C8; ++c8) {
          unsigned h0, h1, h2, h3, l0, l1, l2, l3;
          split2(t[c8][0][0], t[c8][0][1], sx, h0, l0);
          split2(t[c8][0][2], t[c8][0][3], sx, h1, l1);
          split2(t[c8][1][0], t[c8][1][1], sx, h2, l2);
          split2(t[c8][1][2], t[c8][1][3], sx, h3, l3);
          *reinterpret_cast<u32x4v*>(x_hi + ((size_t)c8 * Bk::halo + hv) * 16) = u32x4v{h0, h1, h2, h3};
          *reinterpret_cast<u32x4v*>(x_lo + ((size_t)c8 * Bk::halo + hv) * 16) = u32x4v{l0, l1, l2, l3};
        }
      }
    }
  };

  // ---- prologue: the weights' planes (16-byte copies) and the first brick
  if (kPrefetch && k_begin < k_end) halo_load(k_begin);
  if constexpr (kWLds) {
    constexpr int n16 = w_bytes / 16, WR = (n16 + 255) / 256;
    const u32x4v* src = reinterpret_cast<const u32x4v*>(a.ws + kPlanesHeader);
    u32x4v* dst = reinterpret_cast<u32x4v*>(w_lds);
    u32x4v wv[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r) wv[r] = src[(tid + 256 * r) < n16 ? tid + 256 * r : 0];
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (tid + 256 * r < n16) dst[tid + 256 * r] = wv[r];
  }
  if (k_begin < k_end) {
    if constexpr (kPrefetch) halo_to_lds();
    else halo_direct(k_begin);
  }
  __syncthreads();

  // ---- this wave's column tiles: tile q = wave + 4 v -> (tz, ty, tx16)
  int vbase[VT];     // halo voxel index of output voxel (tile, j) shifted by (-1,-1,-1), i.e. of tap (0,0,0)
#pragma unroll
  for (int v = 0; v < VT; ++v) {
    const int q = wave + 4 * v;
    const int tx16 = q % (Bk::TX / 16), ty = (q / (Bk::TX / 16)) % Bk::TY, tz = q / (Bk::TX / 16 * Bk::TY);
    vbase[v] = (S * tz * Bk::HY + S * ty) * Bk::HX + S * (tx16 * 16 + j);
  }
  // halo offset of this lane group's tap at a k-step (padding k: zero weights, any finite operand) and the 8-channel chunk of
  // the halo planes it reads there: per-lane tables for CIN <= 16 (the tap depends on the lane group), compile-time
  // constants + the lane group for CIN >= 32 (one tap per k-step)
  constexpr int KT = CIN <= 16 ? KS : 1;
  int toff_t[KT];
  if constexpr (CIN <= 16) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      int tap = planes_tap<CIN>(ks, g);
      tap = tap < NTAPS ? tap : 0;
      toff_t[ks] = T2 ? ((tap >> 2) * Bk::HY + ((tap >> 1) & 1)) * Bk::HX + (tap & 1) : ((tap / 9) * Bk::HY + (tap / 3) % 3) * Bk::HX + tap % 3;
    }
  }
  auto operand_off = [&](auto ksi, int vb) __attribute__((always_inline)) -> int {
    constexpr int ks = decltype(ksi)::value;
    if constexpr (CIN <= 16) {
      return ((planes_c0<CIN>(ks, g) / 8) * Bk::halo + vb + toff_t[ks]) * 16;
    } else {
      constexpr int tap = planes_tap<CIN>(ks, 0);
      constexpr int to = T2 ? ((tap >> 2) * Bk::HY + ((tap >> 1) & 1)) * Bk::HX + (tap & 1) : ((tap / 9) * Bk::HY + (tap / 3) % 3) * Bk::HX + tap % 3;
      return ((planes_c0<CIN>(ks, 0) / 8 + g) * Bk::halo + vb + to) * 16;
    }
  };
  float omax = 0.f;
  const int ct = a.cout + a.cout2;
  const size_t plane = (size_t)a.Do * a.Ho * a.Wo;

  for (int k = k_begin; k < k_end; ++k) {
    if (kPrefetch && k + 1 < k_end) halo_load(k + 1);       // in flight under this brick's MFMAs
    int bb, x0, y0, z0;
    brick_of(k, bb, x0, y0, z0);
#pragma unroll 1
    for (int pass = 0; pass < NT / NTG; ++pass) {
    f32x4 acc[VT][NTG];
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
      for (int t = 0; t < NTG; ++t) acc[v][t] = splat4(0.f);
    // One step = (k-step, column tile): two ds_read_b128 for the B planes (+ the A planes at a k-step's first tile) and
    // 3 NT MFMAs.  The operands of step s + 2 are requested before the MFMAs of step s are issued (scheduling fences keep
    // that order: left alone, the scheduler sinks every read to its use and the wave pays the LDS latency 56 times a brick
    // -- 0.31 ms for the full-resolution 8 -> 8 layer, a tenth of it in the matrix pipe).
    constexpr int kSteps = (UFR_C3P_ABL == 1 ? 1 : KS) * VT;
    f16x8 rb[3][2], ra[3][NTG][2];
    auto issue = [&](auto si) __attribute__((always_inline)) {
      constexpr int st = decltype(si)::value, ks = st / VT, v = st % VT;
      const int o = operand_off(std::integral_constant<int, ks>{}, vbase[v]);
      rb[st % 3][0] = *reinterpret_cast<const f16x8*>(x_hi + o);
      rb[st % 3][1] = *reinterpret_cast<const f16x8*>(x_lo + o);
      if constexpr (v == 0) {
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            ra[ks % 3][t][p] = *reinterpret_cast<const f16x8*>((kWLds ? w_lds : w_glb) + ((ks * NT + pass * NTG + t) * 2 + p) * 1024 + lane * 16);
      }
    };
    issue(std::integral_constant<int, 0>{});
    if constexpr (kSteps > 1) issue(std::integral_constant<int, 1>{});
    static_for<kSteps>([&](auto si) __attribute__((always_inline)) {
      constexpr int st = decltype(si)::value, ks = st / VT, v = st % VT;
      if constexpr (st + 2 < kSteps) issue(std::integral_constant<int, st + 2>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NTG; ++t) {
        acc[v][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[ks % 3][t][1], rb[st % 3][0], acc[v][t], 0, 0, 0);   // small terms first
        acc[v][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[ks % 3][t][0], rb[st % 3][1], acc[v][t], 0, 0, 0);
        acc[v][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[ks % 3][t][0], rb[st % 3][0], acc[v][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });

    // ---- epilogue: lane (g, j) holds channels 16 t + 4 g + r of voxel j of its tiles
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      const int q = wave + 4 * v;
      const int tx16 = q % (Bk::TX / 16), ty = (q / (Bk::TX / 16)) % Bk::TY, tz = q / (Bk::TX / 16 * Bk::TY);
      const int ox = x0 + tx16 * 16 + j, oy = y0 + ty, oz = z0 + tz;
      if constexpr (T2) {
        // rows 16 tile + 4 g + r = class * cout + channel: this lane's four rows are four consecutive channels of ONE parity
        // class; a wave's store covers both x parities and all channels of 32 consecutive output x: whole lines
        if (ox >= a.W || oy >= a.H || oz >= a.D) continue;
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          const int row0 = 16 * (pass * NTG + t) + 4 * g;
          const int cls = row0 / a.cout, c0 = row0 - cls * a.cout;
          const size_t vox = (size_t)bb * plane + ((size_t)(2 * oz + ((cls >> 2) & 1)) * a.Ho + 2 * oy + ((cls >> 1) & 1)) * a.Wo + 2 * ox + (cls & 1);
          f32x4 y;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float q2 = acc[v][t][r] * dsc;
            if (a.bias) q2 += a.bias[c0 + r];
            if (a.scale) q2 = fmaf(q2, a.scale[c0 + r], a.shift[c0 + r]);
            if (a.relu) q2 = fmaxf(q2, 0.f);
            y[r] = q2;
          }
          if (a.skip) y += ld4(a.skip + vox * a.cout + c0);
          st4(a.out + vox * a.cout + c0, y);
          omax = fmaxf(omax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
        }
        continue;
      }
      if (ox >= a.Wo || oy >= a.Ho || oz >= a.Do) continue;
      if (UFR_C3P_ABL == 3 && acc[v][0][0] != 12345.f) continue;
      const size_t sp = ((size_t)oz * a.Ho + oy) * a.Wo + ox;
#pragma unroll
      for (int t = 0; t < NTG; ++t) {
        const int c0 = 16 * (pass * NTG + t) + 4 * g;
        if (c0 >= ct) continue;
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float q2 = acc[v][t][r] * dsc;
          const int c = c0 + r;
          if (c < a.cout) {
            if (a.bias) q2 += a.bias[c];
            if (a.scale) q2 = fmaf(q2, a.scale[c], a.shift[c]);
            if (a.relu) q2 = fmaxf(q2, 0.f);
          }
          y[r] = q2;
        }
        if (a.ncdhw) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0 + r;
            if (c < a.cout) a.out[((size_t)bb * a.cout + c) * plane + sp] = y[r];
            else if (c < ct) a.out2[((size_t)bb * a.cout2 + (c - a.cout)) * plane + sp] = 1.f / (1.f + expf(-y[r]));
          }
        } else {
          const size_t vox = (size_t)bb * plane + sp;
          if (a.skip) y += ld4(a.skip + vox * a.cout + c0);
          st4(a.out + vox * a.cout + c0, y);
          omax = fmaxf(omax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
        }
      }
    }
    }   // pass
    __syncthreads();                           // every wave is done reading this brick's planes
    if (k + 1 < k_end) {
      if constexpr (kPrefetch) halo_to_lds();
      else halo_direct(k + 1);
    }
    __syncthreads();
  }
  if (a.out_absmax) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) omax = fmaxf(omax, __shfl_xor(omax, s));
    const unsigned bits = __builtin_bit_cast(unsigned, omax);
    if (lane == 0 && bits > __atomic_load_n(a.out_absmax, __ATOMIC_RELAXED)) atomicMax(a.out_absmax, bits);
  }
}

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, size_t n4, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = ld4(x + 4 * i);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  if (blockIdx.x == 0)
    for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
  const unsigned bits = __builtin_bit_cast(unsigned, m);
  if ((threadIdx.x & 63) == 0 && bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
}

template <int CIN, int NT, int NTG = NT, int MODE = 0>
hipError_t launch_planes_t(const PlanesArgs& a, hipStream_t s) {
  typedef Brick<CIN, MODE> Bk;
  constexpr int w_all = planes_ksteps(CIN, MODE) * NT * 2048;
  constexpr int lds = (w_all <= 32768 ? w_all : 0) + 2 * Bk::plane_bytes;
  static LdsAttrOnce lds_attr;
  if (const hipError_t e = lds_attr.set(reinterpret_cast<const void*>(&conv3d_planes_kernel<CIN, NT, NTG, MODE>), lds); e != hipSuccess) return e;
  PlanesArgs b = a;
  // bricks tile the output grid (transposed: the input grid, eight outputs per voxel)
  const int gx = MODE == 2 ? a.W : a.Wo, gy = MODE == 2 ? a.H : a.Ho, gz = MODE == 2 ? a.D : a.Do;
  b.nbx = (gx + Bk::TX - 1) / Bk::TX; b.nby = (gy + Bk::TY - 1) / Bk::TY; b.nbz = (gz + Bk::TZ - 1) / Bk::TZ;
  const long long bricks = (long long)b.nbx * b.nby * b.nbz * a.B;
  if (bricks <= 0 || bricks > 0x7fffffffLL) return hipErrorInvalidValue;
  // the resident workgroups (LDS: two per CU, one for the 64-channel halo), a multiple of the 8 XCDs; fewer when there are
  // fewer bricks than that
  long long blocks = (lds > 80 * 1024 ? 1 : 2) * 256;
  if (bricks < blocks) blocks = ((bricks + 7) / 8) * 8;
  hipLaunchKernelGGL((conv3d_planes_kernel<CIN, NT, NTG, MODE>), dim3((unsigned)blocks), dim3(256), lds, s, b);
  return hipGetLastError();
}

}  // namespace

// which (cin, cout + cout2) the plane kernels take (stride 1 only); 0 = not supported
size_t conv3d_planes_workspace_bytes(int cin, int cout, int cout2, int mode) {
  const bool s1 = mode == 0 && (((cin == 8 || cin == 16) && cout + cout2 <= 16) || (cin == 32 && cout == 32 && cout2 == 0) ||
                                (cin == 64 && cout == 64 && cout2 == 0));
  const bool s2 = mode == 1 && cout2 == 0 && ((cin == 8 && cout == 16) || (cin == 16 && cout == 32) || (cin == 32 && cout == 64));
  const bool t2 = mode == 2 && cout2 == 0 && ((cin == 16 && cout == 8) || (cin == 32 && cout == 16) || (cin == 64 && cout == 32));
  if (t2) return kPlanesHeader + (size_t)planes_ksteps(cin, 2) * (cout / 2) * 2048;
  if (!s1 && !s2) return 0;
  if (cout < 1 || cout2 < 0) return 0;
  return kPlanesHeader + (size_t)planes_ksteps(cin) * planes_tiles(cout + cout2) * 2048;
}

hipError_t launch_absmax(const float* x, size_t n, float* absmax, hipStream_t s) {
  if (!n) return hipSuccess;
  const size_t n4 = (reinterpret_cast<size_t>(x) & 15) ? 0 : n / 4;
  size_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n4, n, reinterpret_cast<unsigned*>(absmax));
  return hipGetLastError();
}

hipError_t launch_conv3d_planes(const float* in, const float* in_absmax, const float* weight, const float* weight2, const float* bias,
                                const float* scale, const float* shift, const float* skip, float* out, float* out2, float* out_absmax,
                                int B, int D, int H, int W, int cin, int cout, int cout2, int mode, int relu, int ncdhw, int flip,
                                void* ws, int planes_ready, hipStream_t s) {
  if (!conv3d_planes_workspace_bytes(cin, cout, cout2, mode) || (mode != 0 && flip) || (mode == 2 && ncdhw)) return hipErrorInvalidValue;
  // one view's tensor is a raw buffer descriptor (31-bit byte offsets, kBufOut = zero fill)
  if ((long long)D * H * W * cin * 4 >= (1ll << 31)) return hipErrorInvalidValue;
  if (!planes_ready) {
    const int ct = cout + cout2, nt = mode == 2 ? cout / 2 : planes_tiles(ct);
    PrepArgs p;
    p.weight = weight; p.weight2 = weight2; p.ws = static_cast<char*>(ws); p.cout = cout; p.cout2 = cout2; p.flip = flip;
    // small layers: every workgroup of the prep kernel measures max |w| itself (one launch); from 16 K weights up that
    // redundancy costs more than two small launches in front (64 -> 64: 216 workgroups x 110 592 loads = 0.15 ms)
    p.max_ready = (long long)cout * cin * 27 > 16384 ? 1 : 0;
    if (p.max_ready) {
      float* wmax = reinterpret_cast<float*>(ws) + 1;
      if (const hipError_t e = hipMemsetAsync(wmax, 0, 4, s); e != hipSuccess) return e;
      if (const hipError_t e = launch_absmax(weight, (size_t)cout * cin * 27, wmax, s); e != hipSuccess) return e;
    }
    const unsigned pb = (unsigned)((planes_ksteps(cin, mode) * nt * 512 + 2047) / 2048);
    if (mode == 2 && cin == 16) hipLaunchKernelGGL((conv3d_planes_prep<16, 2>), dim3(pb), dim3(256), 0, s, p);
    else if (mode == 2 && cin == 32) hipLaunchKernelGGL((conv3d_planes_prep<32, 2>), dim3(pb), dim3(256), 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL((conv3d_planes_prep<64, 2>), dim3(pb), dim3(256), 0, s, p);
    else if (cin == 8) hipLaunchKernelGGL(conv3d_planes_prep<8>, dim3(pb), dim3(256), 0, s, p);
    else if (cin == 16) hipLaunchKernelGGL(conv3d_planes_prep<16>, dim3(pb), dim3(256), 0, s, p);
    else if (cin == 32) hipLaunchKernelGGL(conv3d_planes_prep<32>, dim3(pb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv3d_planes_prep<64>, dim3(pb), dim3(256), 0, s, p);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  }
  PlanesArgs a;
  a.in = in; a.in_absmax = reinterpret_cast<const unsigned*>(in_absmax); a.ws = static_cast<const char*>(ws); a.bias = bias;
  a.scale = scale; a.shift = shift; a.skip = skip; a.out = out; a.out2 = out2; a.out_absmax = reinterpret_cast<unsigned*>(out_absmax);
  a.B = B; a.D = D; a.H = H; a.W = W; a.cout = cout; a.cout2 = cout2; a.relu = relu; a.ncdhw = ncdhw;
  a.Do = mode == 1 ? (D + 1) / 2 : D; a.Ho = mode == 1 ? (H + 1) / 2 : H; a.Wo = mode == 1 ? (W + 1) / 2 : W;   // k3 p1 s2: floor((n - 1) / 2) + 1
  if (mode == 2) { a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; }                                                   // k3 p1 s2 output_padding 1
  a.nbx = a.nby = a.nbz = 0;
  if (mode == 2) {
    if (cin == 16) return launch_planes_t<16, 4, 2, 2>(a, s);
    if (cin == 32) return launch_planes_t<32, 8, 2, 2>(a, s);
    return launch_planes_t<64, 16, 2, 2>(a, s);
  }
  if (mode == 1) {
    if (cin == 8) return launch_planes_t<8, 1, 1, 1>(a, s);
    if (cin == 16) return launch_planes_t<16, 2, 1, 1>(a, s);
    return launch_planes_t<32, 4, 2, 1>(a, s);
  }
  if (cin == 8) return launch_planes_t<8, 1>(a, s);
  if (cin == 16) return launch_planes_t<16, 1>(a, s);
  if (cin == 32) return launch_planes_t<32, 2, 1>(a, s);
  return launch_planes_t<64, 4, 2>(a, s);
}

}  // namespace ufr
