// Projection + every gather of one rendering pass, producing the view-transformer token inputs.
//   camera.get_coord_ref_ndc          code1/misc/camera.py:378-407      (projection, z>0 mask, frustum z)
//   UFORecon.query_cond_info          code1/model.py:218-305            (pairwise cosine similarity, AC=True/border)
//   UFORecon.query_depth_from_volume  code1/model.py:350-390            (3 correlation frustums, AC=True/zeros, 3-D)
//   RayTransformer.forward            code1/ray_transformer.py:185-281  (dir, 2-D gathers AC=False/zeros, depth PE,
//                                                                        pre_sim_mlp, token assembly)
// A block is 64 consecutive points (samples of one ray) x NV views.  Phase A, thread (v, p) = view v of
// point p: projection, bilinear footprints, the narrow gathers (colour, depth, frustums).  Phase B, the
// 32-channel gathers, is cooperative: 8 adjacent lanes share one footprint and each takes 4 channels, so
// one load instruction touches 8 cache lines instead of 64 -- the L1 (TCP) processes about one line per
// clock and was the kernel's limiter (r2 PMC: TCP busy ~100 %, 3.1e8 line accesses per 524 288 points).
// All maps are channel-last (prep.hip).  Cross-view reductions (pair similarities, frustum blending) go
// through LDS in the reference's summation order.
#include "ufr_device.h"
#include "ufr_internal.h"
#include "volume_sample.h"

namespace ufr {

// out = fma(v_se,se, fma(v_sw,sw, fma(v_ne,ne, v_nw*nw))) per channel; masked corners read as zero
__device__ __forceinline__ f32x4 lerp_tap4(const float* __restrict__ base, int stride, const Tap2& t, int c) {
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)   // texel index < 2^24, row stride < 2^24: one 24-bit multiply
    v[k] = t.o[k] >= 0 ? ld4(base + (__umul24((unsigned)t.o[k], (unsigned)stride) + (unsigned)c)) : splat4(0.f);
  f32x4 acc;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = mul_rn(v[0][e], t.w[0]);
    a = fmaf(v[1][e], t.w[1], a);
    a = fmaf(v[2][e], t.w[2], a);
    acc[e] = fmaf(v[3][e], t.w[3], a);
  }
  return acc;
}

#ifndef UFR_GATHER_PAIR
#define UFR_GATHER_PAIR 0       // 1: frustum lookups shared by lane pairs (volume_sample.h: sample_volume_pair) -- bit-identical, and
                                // MEASURED SLOWER (gather 0.467 vs 0.458 ms per 524 288 points, frame 123.5 vs 121.5 ms): the DPP adds and the
                                // doubled chain cost more than the halved line accesses save.  Kept as the A/B.
#endif
#ifndef UFR_GATHER_BUFFER
#define UFR_GATHER_BUFFER 1     // taps through bounded buffer loads (ufr_device.h); 0 = the conditional global loads (A/B)
#endif
// the same through a bounded descriptor: texel t.o[k] of the map that starts at byte `base` of the descriptor, channels from
// byte c4; masked corners (t.o[k] < 0) read zeros from kBufOut -- all four loads in flight at once
__device__ __forceinline__ f32x4 lerp_tap4_buf(__amdgpu_buffer_rsrc_t r, unsigned base, int stride_bytes, const Tap2& t, unsigned c4) {
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] = buf_ld4(r, t.o[k] >= 0 ? base + __umul24((unsigned)t.o[k], (unsigned)stride_bytes) + c4 : kBufOut);
  f32x4 acc;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = mul_rn(v[0][e], t.w[0]);
    a = fmaf(v[1][e], t.w[1], a);
    a = fmaf(v[2][e], t.w[2], a);
    acc[e] = fmaf(v[3][e], t.w[3], a);
  }
  return acc;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void put_tap(float* dst, const Tap2& t) {
  *reinterpret_cast<i32x4*>(dst) = i32x4{t.o[0], t.o[1], t.o[2], t.o[3]};
  *reinterpret_cast<f32x4*>(dst + 4) = f32x4{t.w[0], t.w[1], t.w[2], t.w[3]};
}
__device__ __forceinline__ Tap2 get_tap(const float* src) {
  const i32x4 o = *reinterpret_cast<const i32x4*>(src);
  const f32x4 w = *reinterpret_cast<const f32x4*>(src + 4);
  Tap2 t;
#pragma unroll
  for (int k = 0; k < 4; ++k) { t.o[k] = o[k]; t.w[k] = w[k]; }
  return t;
}

// pre_sim_mlp: Linear(8,32)-ReLU-Linear(32,32)-ReLU-Linear(32,16) (ray_transformer.py:128-132, 268) on the mean pair
// similarity of the block's 64 points, by ONE wave (round 3; a separate kernel before: it re-read the similarities and
// wrote 16 columns into each of the NV token rows of a point -- 64-byte pieces of 320-byte rows, 4.5 ms per frame).
// Points are MFMA columns (v_mfma_f32_16x16x4_f32, exact fp32 fma chains), four tiles of 16; the accumulator tile of a
// layer is the B operand of the next (lane (g,j) holds neurons 4g+r of point j, so k-step (tile, r) of the next layer
// contracts neurons 16*tile + 4g + r and the A operand is loaded with that index), biases are the accumulators'
// initial values.  sim: LDS, point q's 8 similarities at sim + q * sim_stride; out: LDS, point q's row at out + q *
// out_stride, columns 24..39.
__device__ __forceinline__ void presim_block(const PreSim& ps, const float* sim, int sim_stride, float* out, int out_stride,
                                             int lane) {
  const int g = lane >> 4, j = lane & 15;
  f32x4 h1[4][2], h2[4][2], o[4];
  float x0[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float* sq = sim + (16 * u + j) * sim_stride;
    x0[u][0] = sq[g];
    x0[u][1] = sq[4 + g];
  }
  {
    float a1[2][2];
#pragma unroll
    for (int to = 0; to < 2; ++to) {
#pragma unroll
      for (int s = 0; s < 2; ++s) a1[to][s] = ps.w0[(16 * to + j) * 8 + 4 * s + g];
      const f32x4 b1 = ld4(ps.b0 + 16 * to + 4 * g);
#pragma unroll
      for (int u = 0; u < 4; ++u) h1[u][to] = b1;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int u = 0; u < 4; ++u) h1[u][to] = mfma16(a1[to][s], x0[u][s], h1[u][to]);
  }
  {
    float a2[2][2][4];
#pragma unroll
    for (int to = 0; to < 2; ++to) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) a2[to][ti][r] = ps.w2[(16 * to + j) * 32 + 16 * ti + 4 * g + r];
      const f32x4 b2 = ld4(ps.b2 + 16 * to + 4 * g);
#pragma unroll
      for (int u = 0; u < 4; ++u) h2[u][to] = b2;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
          for (int u = 0; u < 4; ++u) h2[u][to] = mfma16(a2[to][ti][r], fmaxf(h1[u][ti][r], 0.f), h2[u][to]);
  }
  {
    float a3[2][4];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int r = 0; r < 4; ++r) a3[to][r] = ps.w4[j * 32 + 16 * to + 4 * g + r];
    const f32x4 b3 = ld4(ps.b4 + 4 * g);
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = b3;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = mfma16(a3[ti][r], fmaxf(h2[u][ti][r], 0.f), o[u]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) st4(out + (16 * u + j) * out_stride + 24 + 4 * g, o[u]);
}

// LDS layout (floats): sim[64][NPAIR][8] | { tapF[NV][64][8] | tapM[NV][64][8] } aliased with volp[64][NV-1][25] (views
// >= 1; wave 0 keeps its own in registers: the frustums are sampled AFTER the cooperative gathers, when the footprints
// are dead) | outv[64][40] -- which moves into the (then dead) similarity slots of its point when they are large enough
// (NV >= 4: a slot is NPAIR * 8 >= 48 floats).  The footprint of a block bounds the CU's occupancy -- of this kernel, and
// of the mix when it runs beside the transformer kernels of another chunk (side streams): 31 -> 29 KB at NV = 3 (no
// effect: 4 blocks per CU are enough there), 66 -> 46 KB at NV = 5 (2 -> 3 blocks per CU).
// NVT: the view count as a compile-time constant (round 5): the item / NV, item / npair divisions and the pair-index walk
// of the cooperative loops sit in the address chains of the gathers; as run-time values they were 143 32-bit multiplies
// and two ~20-instruction divisions per item.
template <int NVT>
__global__ void __launch_bounds__(64 * NVT) gather_kernel(FrameDev f, PreSim ps, const float* __restrict__ ray_o,
                                                      int o_stride, const float* __restrict__ ray_d,
                                                      const float* __restrict__ zval, int P, int SN,
                                                      float* __restrict__ x_tokens, float* __restrict__ x_point,
                                                      float* __restrict__ rgb_out,
                                                      float* __restrict__ dir_out, float* __restrict__ sim8_out,
                                                      float* __restrict__ vol24_out, float* __restrict__ xy_out,
                                                      float* __restrict__ maskz_out,
                                                      const float* __restrict__ vol24_in,
                                                      const float* __restrict__ sim8_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NV = NVT;
  constexpr int npair = NV * (NV - 1) / 2;
  float* sh_sim = smem;                        // 64*npair*8
  float* sh_tapF = sh_sim + 64 * npair * 8;    // NV*64*8: feature-map footprint (align_corners=False, zeros)
  float* sh_tapM = sh_tapF + NV * 64 * 8;      // NV*64*8: matching-map footprint (align_corners=True, border)
  float* sh_vol = sh_tapF;                     // 64*(NV-1)*25, written after the last footprint read
  constexpr int sim_slot = npair * 8;              // floats per point in sh_sim
  constexpr bool out_in_sim = sim_slot >= 48;      // [64][40] = frustum blend 24 | pre_sim_mlp 16 ...
  constexpr int region2 = 2 * NV * 64 * 8 > 64 * (NV - 1) * 25 ? 2 * NV * 64 * 8 : 64 * (NV - 1) * 25;
  float* sh_out = out_in_sim ? sh_sim + 8 : sh_tapF + region2;   // ... behind the point's 8 pre_sim_mlp inputs, or on its own
  constexpr int out_stride = out_in_sim ? sim_slot : 40;

  const int v = threadIdx.x >> 6, p = threadIdx.x & 63;
  const int vu = __builtin_amdgcn_readfirstlane(v);      // a wave = one view: per-view descriptors live in scalar registers
  // XCD-aware block -> point-group map: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so
  // giving XCD x the x-th contiguous eighth of the launch keeps the texels that neighbouring rays share in ONE L2
  // instead of eight
#ifndef UFR_GATHER_NO_XCD_MAP
  const int nb8 = (int)(gridDim.x / 8) * 8;
  const int blk = (int)blockIdx.x < nb8 ? (int)(blockIdx.x % 8) * (nb8 / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
#else
  const int blk = blockIdx.x;
#endif
  const int pidx = blk * 64 + p;
  const bool active = pidx < P;
  const int pc = active ? pidx : P - 1;
  const int ray = pc / SN;

  // ---- sample position (sampler.py:47: o + z*d, unfused like torch's mul then add)
  const float zz = zval[pc];
  const float* o = ray_o + (size_t)ray * o_stride;
  const float px = mul_add_unfused(zz, ray_d[3 * ray + 0], o[0]);
  const float py = mul_add_unfused(zz, ray_d[3 * ray + 1], o[1]);
  const float pz = mul_add_unfused(zz, ray_d[3 * ray + 2], o[2]);

  // ---- projection into view v (camera.py:384-393); k-ordered fma chain = what torch.bmm (MKL sgemm)
  // produces for a 4x4 pose, bit for bit
  const float* M = f.pose[v];
  float qx = fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))) + M[3];
  float qy = fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))) + M[7];
  float qz = fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))) + M[11];
  const float mask_z = qz > 0.f ? 1.f : 0.f;
  const float x = qx / qz, y = qy / qz;
  if (active && xy_out) {
    xy_out[((size_t)v * P + pidx) * 2 + 0] = x;
    xy_out[((size_t)v * P + pidx) * 2 + 1] = y;
  }
  if (active && maskz_out) maskz_out[(size_t)v * P + pidx] = mask_z;

  // token row of (point, view): public layout 80 columns, compact layout [feat 32 | PE 8] (ufr_internal.h)
  const int row_cols = x_point ? kViewCols : UFR_TOKEN_DIM, pe_col = x_point ? 32 : 72;
  float* xrow = x_tokens + ((size_t)pc * NV + v) * row_cols;

  // ---- 2-D gathers with align_corners=False, zeros (grid_sample.py:5-19; ray_transformer.py:222-237)
  {
    put_tap(sh_tapF + (v * 64 + p) * 8, taps_zeros(unnorm2d_nac(x, f.w), unnorm2d_nac(y, f.h), f.w, f.h));
    put_tap(sh_tapM + (v * 64 + p) * 8, taps_border(unnorm2d_ac(x, f.w), unnorm2d_ac(y, f.h), f.w, f.h));
    Tap2 tf = taps_zeros(unnorm2d_nac(x, f.W), unnorm2d_nac(y, f.H), f.W, f.H);
#if UFR_GATHER_BUFFER
    const unsigned img_px = (unsigned)(f.H * f.W);
    f32x4 c4 = lerp_tap4_buf(buf_rsrc(f.rgb + (size_t)vu * img_px * 4, img_px * 16u), 0u, 16, tf, 0u);
#else
    f32x4 c4 = lerp_tap4(f.rgb + (size_t)v * f.H * f.W * 4, 4, tf, 0);
#endif
    const float inb = (x <= 1.f && x >= -1.f && y <= 1.f && y >= -1.f) ? 1.f : 0.f;  // inclusive mask
    c4[3] = inb * mask_z;                                                             // ray_transformer.py:251-252
    if (active) st4(rgb_out + ((size_t)pidx * NV + v) * 4, c4);
    // MVS depth guide + positional encoding (ray_transformer.py:229-247, 29-73)
    float dv[4];
#if UFR_GATHER_BUFFER
    const __amdgpu_buffer_rsrc_t rdepth = buf_rsrc(f.depth + (size_t)vu * img_px, img_px * 4u);
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = buf_ld1(rdepth, tf.o[k] >= 0 ? (unsigned)tf.o[k] * 4u : kBufOut);
#else
    const float* dmap = f.depth + (size_t)v * f.H * f.W;
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[k] = tf.o[k] >= 0 ? dmap[tf.o[k]] : 0.f;
#endif
    const float dm = fmaf(dv[3], tf.w[3], fmaf(dv[2], tf.w[2], fmaf(dv[1], tf.w[1], mul_rn(dv[0], tf.w[0]))));
    const float* R = f.w2c_z[v];
    float zc = fmaf(R[2], pz, fmaf(R[1], py, mul_rn(R[0], px))) + R[3];
    float delta = dm - zc;
    float pe[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float fr = 3.14159274101257324f * (float)(1 << k);  // float32(pi) * 2^k
      pe[2 * k] = sinf(mul_rn(delta, fr));                       // torch.addcmul is a single fma
      pe[2 * k + 1] = sinf(fmaf(delta, fr, 1.57079637050628662f));  // phase pi/2 -> cosine slot
    }
    if (active) {
      st4(xrow + pe_col, f32x4{pe[0], pe[1], pe[2], pe[3]});
      st4(xrow + pe_col + 4, f32x4{pe[4], pe[5], pe[6], pe[7]});
    }
    // relative direction (ray_transformer.py:185-191)
    float ax = px - f.ref_pos[0], ay = py - f.ref_pos[1], az = pz - f.ref_pos[2];
    float bx = px - f.cam_pos[v][0], by = py - f.cam_pos[v][1], bz = pz - f.cam_pos[v][2];
    // unit vectors through one reciprocal each (v_rcp_f32, 1 ulp) instead of three IEEE divisions: the difference to the
    // reference's x / |x| is a rounding of the last bit of a direction component
    const float ra = __builtin_amdgcn_rcpf(sqrtf(ax * ax + ay * ay + az * az));
    const float rb = __builtin_amdgcn_rcpf(sqrtf(bx * bx + by * by + bz * bz));
    if (active) st4(dir_out + ((size_t)pidx * NV + v) * 4, f32x4{ax * ra - bx * rb, ay * ra - by * rb, az * ra - bz * rb, 0.f});
  }

  __syncthreads();

  // ---- cooperative 32-channel gathers: lane group of 8 = one footprint, lane c8 = channels 4*c8..4*c8+3
  const int c8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  constexpr int n_grp = 64 * NV / 8;
#if UFR_GATHER_BUFFER
  // one descriptor over all views' maps (the item's view differs per lane group): [NV][h][w][32] and [NV][h][w][match_ch]
  const unsigned map_px = (unsigned)(f.h * f.w);
  const __amdgpu_buffer_rsrc_t rfeat = buf_rsrc(f.feat, (unsigned)NV * map_px * 128u);
  const __amdgpu_buffer_rsrc_t rmatch = buf_rsrc(f.match, (unsigned)NV * map_px * (unsigned)f.match_ch * 4u);
#endif
  // image features of (point, view) -> token columns 0..31 (ray_transformer.py:222-226)
#ifdef UFR_GABL_NOCOOP  // ablation build: no 32-channel gathers (timing only)
  if (false)
#endif
  for (int item = grp; item < 64 * NV; item += n_grp) {
    const int ip = item / NV, iv = item - ip * NV;
    const int ipidx = blk * 64 + ip;
    if (ipidx < P) {
      const Tap2 t = get_tap(sh_tapF + (iv * 64 + ip) * 8);
#if UFR_GATHER_BUFFER
      st4(x_tokens + ((size_t)ipidx * NV + iv) * row_cols + 4 * c8,
          lerp_tap4_buf(rfeat, (unsigned)iv * map_px * 128u, 128, t, 16u * c8));
#else
      st4(x_tokens + ((size_t)ipidx * NV + iv) * row_cols + 4 * c8,
          lerp_tap4(f.feat + (size_t)iv * f.h * f.w * 32, 32, t, 4 * c8));
#endif
    }
  }
  // pairwise similarity (model.py:271-283): pair q = (a, b), sides (view a, chunk b) / (view b+1, chunk a);
  // lane c8 = channel group gi of model.py:278-280
#ifdef UFR_GABL_NOCOOP
  if (false)
#endif
  if (!sim8_in)
  for (int item = grp; item < 64 * npair; item += n_grp) {
    const int ip = item / npair, q = item - ip * npair;
    int a = 0, rem = q;
    while (rem >= NV - 1 - a) { rem -= NV - 1 - a; ++a; }
    const int b = a + rem;
    const int va = a, ca = b, vb = b + 1, cb = a;
    const Tap2 ta = get_tap(sh_tapM + (va * 64 + ip) * 8), tb = get_tap(sh_tapM + (vb * 64 + ip) * 8);
    const float* ba = f.match + (size_t)va * f.h * f.w * f.match_ch + 32 * ca;
    const float* bb = f.match + (size_t)vb * f.h * f.w * f.match_ch + 32 * cb;
#if UFR_GATHER_BUFFER
    const unsigned mrow = (unsigned)f.match_ch * 4u;
    f32x4 fa = lerp_tap4_buf(rmatch, (unsigned)va * map_px * mrow + 128u * ca, (int)mrow, ta, 16u * c8);
    f32x4 fb = lerp_tap4_buf(rmatch, (unsigned)vb * map_px * mrow + 128u * cb, (int)mrow, tb, 16u * c8);
#else
    f32x4 fa = lerp_tap4(ba, f.match_ch, ta, 4 * c8);
    f32x4 fb = lerp_tap4(bb, f.match_ch, tb, 4 * c8);
#endif
    float na = fmaxf(sqrtf(fa[0] * fa[0] + fa[1] * fa[1] + fa[2] * fa[2] + fa[3] * fa[3]), 1e-8f);
    float nb = fmaxf(sqrtf(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2] + fb[3] * fb[3]), 1e-8f);
    // F.normalize(x) = x / max(|x|, eps) as x * (1 / max(|x|, eps)): two reciprocals instead of eight IEEE divisions
    const float ra = __builtin_amdgcn_rcpf(na), rb = __builtin_amdgcn_rcpf(nb);
    sh_sim[(ip * npair + q) * 8 + c8] = (fa[0] * ra) * (fb[0] * rb) + (fa[1] * ra) * (fb[1] * rb) +
                                        (fa[2] * ra) * (fb[2] * rb) + (fa[3] * ra) * (fb[3] * rb);
  }
  __syncthreads();

  // ---- correlation frustums of view v (model.py:359-386); skipped when the caller supplies the blended lookup
  // (RayTransformer.forward receives it as `fea_volume`, ray_transformer.py:175, 199)
  float own[25];
  if (vol24_in) {
#pragma unroll
    for (int c = 0; c < 25; ++c) own[c] = 0.f;
  } else {
    const float zn = ((qz - f.vol_near) / (f.vol_far - f.vol_near)) * 2.f - 1.f;  // camera.py:400-401
    float fl[24], wl = 0.f;
#if UFR_GATHER_PAIR && UFR_GATHER_BUFFER && !defined(UFR_GABL_NOVOL)
    // lanes 2q, 2q + 1 (two consecutive samples of a ray, the same view) share every lookup: first the even lane's, then the
    // odd lane's, per stage -- the owner loads the x0 side and keeps the result (volume_sample.h: sample_volume_pair)
    {
      const int odd = threadIdx.x & 1;
      const float xo = pair_swap(x), yo = pair_swap(y), zo = pair_swap(zn);
#pragma unroll
      for (int s = 0; s < UFR_NUM_STAGES; ++s) {
        const unsigned vox = (unsigned)(f.vD[s] * f.vH[s] * f.vW[s]);
        const __amdgpu_buffer_rsrc_t rv = buf_rsrc(f.vol[s] + (size_t)vu * vox * kVolCh, vox * (unsigned)(kVolCh * 4));
        float ev[9], od[9];
        // the even lane's lookup: the even lane is the owner (side 0), the odd lane helps with ITS partner's coordinates
        sample_volume_pair(rv, f.vD[s], f.vH[s], f.vW[s], odd ? xo : x, odd ? yo : y, odd ? zo : zn, odd, ev);
        // the odd lane's lookup
        sample_volume_pair(rv, f.vD[s], f.vH[s], f.vW[s], odd ? x : xo, odd ? y : yo, odd ? zn : zo, odd ^ 1, od);
#pragma unroll
        for (int c = 0; c < 8; ++c) fl[8 * s + c] = odd ? od[c] : ev[c];
        const float ws = odd ? od[8] : ev[8];
        wl = s == 0 ? ws : wl + ws;                                               // :375-378
      }
    }
#else
#pragma unroll
    for (int s = 0; s < UFR_NUM_STAGES; ++s) {
      float fs[8], ws;
      const float* vol = f.vol[s] + (size_t)v * f.vD[s] * f.vH[s] * f.vW[s] * kVolCh;
#ifdef UFR_GABL_NOVOL   // ablation build: no frustum taps (timing only)
      for (int c = 0; c < 8; ++c) fs[c] = x; ws = y;
#else
#if UFR_GATHER_BUFFER
      const unsigned vox = (unsigned)(f.vD[s] * f.vH[s] * f.vW[s]);
      sample_volume_buf(buf_rsrc(f.vol[s] + (size_t)vu * vox * kVolCh, vox * (unsigned)(kVolCh * 4)), f.vD[s], f.vH[s], f.vW[s], x, y,
                        zn, fs, ws);
#else
      sample_volume(vol, f.vD[s], f.vH[s], f.vW[s], x, y, zn, fs, ws);
#endif
#endif
#pragma unroll
      for (int c = 0; c < 8; ++c) fl[8 * s + c] = fs[c];
      wl = s == 0 ? ws : wl + ws;                                                 // :375-378
    }
#endif
#pragma unroll
    for (int c = 0; c < 24; ++c) own[c] = fl[c] * wl;                             // features_L * weights_L
    own[24] = wl;
    if (v > 0) {
      float* dst = sh_vol + (p * (NV - 1) + (v - 1)) * 25;
#pragma unroll
      for (int c = 0; c < 25; ++c) dst[c] = own[c];
    }
  }
  __syncthreads();

  // ---- per-point reductions by wave 0: mean similarity, frustum blend, pre_sim_mlp
  if (v == 0) {
    float sim[8];
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
      float s = 0.f;
      for (int q = 0; q < npair; ++q) s += sh_sim[(p * npair + q) * 8 + gi];
      sim[gi] = s / (float)npair;                                                 // torch.mean over pairs
      if (sim8_in) sim[gi] = sim8_in[(size_t)pc * 8 + gi];                        // cond_info['feat_info'] given
    }
    float* ov = sh_out + p * out_stride;
    {
      float Wsum = own[24];                                                       // view 0 first, then 1..NV-1
      for (int n = 1; n < NV; ++n) Wsum += sh_vol[(p * (NV - 1) + n - 1) * 25 + 24];
      const float rW = 1.f / (Wsum + 1e-8f);                                      // one division for the 24 channels
#pragma unroll
      for (int c = 0; c < 24; ++c) {
        float G = own[c];
        for (int n = 1; n < NV; ++n) G += sh_vol[(p * (NV - 1) + n - 1) * 25 + c];
        ov[c] = G * rW;                                                           // model.py:388
        if (vol24_in) ov[c] = vol24_in[(size_t)pc * 24 + c];
      }
    }
    // pre_sim_mlp of the block's 64 points on this wave's matrix core (presim_tile below): the similarities change
    // from "thread p holds point p" to the MFMA operand layout through the point's own (now dead) sh_sim slots
    {
      float* mine = sh_sim + p * npair * 8;
      st4(mine, f32x4{sim[0], sim[1], sim[2], sim[3]});
      st4(mine + 4, f32x4{sim[4], sim[5], sim[6], sim[7]});
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      presim_block(ps, sh_sim, sim_slot, sh_out, out_stride, p);
    }
    if (active && sim8_out) {
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) sim8_out[(size_t)pidx * 8 + gi] = sim[gi];
    }
    if (active && vol24_out) {
      for (int c = 0; c < 24; ++c) vol24_out[(size_t)pidx * 24 + c] = ov[c];
    }
  }
  __syncthreads();

  // ---- token assembly: [feat 32 | vol 24 | sim 16 | depth PE 8] (ray_transformer.py:258-281)
  // (public layout: the per-point columns go into every view's row; compact: once, by view 0's thread)
  if (active && (!x_point || v == 0)) {
    const float* ov = sh_out + p * out_stride;
    float* dst = x_point ? x_point + (size_t)pidx * kPointCols : xrow + 32;
#pragma unroll
    for (int c = 0; c < 40; c += 4) st4(dst + c, ld4(ov + c));   // frustum lookup 32..55, pre_sim_mlp output 56..71
  }
}

hipError_t launch_gather(const FrameDev& f, const PreSim& ps, const float* ray_o, int o_stride, const float* ray_d,
                         const float* z, int RN, int SN, float* x_tokens, float* x_point, float* rgb, float* dir, float* sim8,
                         float* vol24, float* xy, float* mask_z, const float* vol24_in, const float* sim8_in,
                         hipStream_t s) {
  const int P = RN * SN, NV = f.NV;
  const int npair = NV * (NV - 1) / 2;
  const size_t taps = 2 * (size_t)NV * 64 * 8, volp = (size_t)64 * (NV - 1) * 25, outv = npair * 8 >= 48 ? 0 : 64 * 40;
  size_t lds = sizeof(float) * ((size_t)64 * npair * 8 + (taps > volp ? taps : volp) + outv);
  switch (NV) {
#define UFR_GATHER_CASE(N)                                                                                                   \
    case N:                                                                                                                  \
      hipLaunchKernelGGL(gather_kernel<N>, dim3((P + 63) / 64), dim3(64 * N), lds, s, f, ps, ray_o, o_stride, ray_d, z, P, SN, \
                         x_tokens, x_point, rgb, dir, sim8, vol24, xy, mask_z, vol24_in, sim8_in);                          \
      break;
    UFR_GATHER_CASE(2) UFR_GATHER_CASE(3) UFR_GATHER_CASE(4) UFR_GATHER_CASE(5) UFR_GATHER_CASE(6) UFR_GATHER_CASE(7)
#undef UFR_GATHER_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ufr
