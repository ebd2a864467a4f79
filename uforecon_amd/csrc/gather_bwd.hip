// Backward of the correlation-frustum lookup (gather.hip / UFORecon.query_depth_from_volume, code1/model.py:350-390):
// scatter-add of d vol24 into the six sampled volumes -- the gradients through which feature_volume.cost_reg_2.* trains
// (SURVEY.md appendix C).  The 2-D feature maps and matching features are frozen (model.py:82-83) and sample positions
// carry no gradient, so the frustums are the only tensors the gather differentiates into.
//
//   out[c]   = sum_n fL_n[c] wL_n / (sum_n wL_n + 1e-8),   fL_n = cat_s trilinear(feat_s,n),  wL_n = sum_s trilinear(w_s,n)
//   d fL_n[c] = d out[c] wL_n / (Wsum + 1e-8)
//   d wL_n    = sum_c d out[c] (fL_n[c] - out[c]) / (Wsum + 1e-8)
// then the adjoint of the trilinear interpolation (align_corners=True, zeros padding) spreads both over the 8 corners.
// Thread (v, p) = view v of point p, like the forward; the per-view samples are recomputed from the channel-last copy
// the forward used and exchanged through LDS for the cross-view sums.
// The scatter (round 4).  What float atomics cost on this chip is L2 TRANSACTIONS, not adds (tools/dev/atomic_probe.hip:
// 19 G/s for scattered single floats, whether the target is HBM- or cache-resident; but lanes of ONE instruction that hit
// the same line travel as one transaction: 9 lanes adding to the 9 consecutive floats of a record run at 13.7 G records/s
// = 123 G adds/s).  A corner's 8 feature channels + weight are therefore added as ONE 36-byte record of a CHANNEL-LAST
// scratch volume [view][D][H][W][12] by nine lanes of one instruction -- the surviving corners of a wave are compacted
// into LDS records and re-read lane-transposed -- instead of nine instructions into nine planes of the reference layout
// (round 3: 38 M transactions, 1.6 of the kernel's 1.76 ms; with the atomics removed the kernel takes 0.15 ms).  One
// coalesced pass (volume_unpack_kernel) then writes the reference-layout tensors (NV,8,D,H,W) / (NV,1,D,H,W).
#include "ufr_internal.h"
#include "volume_sample.h"
#include "weight_stream.h"   // static_for

#ifndef UFR_GBWD_ROUNDS
#define UFR_GBWD_ROUNDS 3   // fold aligned groups of 2, 4, 8 lanes (4: also 16, the whole DPP / shuffle row)
#endif

namespace ufr {

// value of lane + D / lane - D of the same 16-lane row (own value where that lane does not exist): __shfl_down / __shfl_up
// with width 16, as a DPP row shift -- one VALU instruction; the ds_bpermute the shuffle intrinsics compile to is an LDS
// round trip each, and this kernel issued 1 056 of them per thread with a wait behind every one
template <int D, class T>
__device__ __forceinline__ T row_down(T v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(b, b, 0x100 + D, 0xf, 0xf, false));   // row_shl:D
}
template <int D, class T>
__device__ __forceinline__ T row_up(T v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(b, b, 0x110 + D, 0xf, 0xf, false));   // row_shr:D
}

struct VolGrads {
  float* rec[UFR_NUM_STAGES];             // channel-last scratch of a stage: [NV][D][H][W][kVolCh], zero on entry
  unsigned char* touch[UFR_NUM_STAGES];   // one byte per kTouchGroup consecutive voxels of [NV][D H W]: set by the scatter
};
// The rays of a step reach a fraction of the voxels (1 024 rays: a fifth), but the reference-layout gradient tensors are
// dense.  The scatter marks the groups of 8 consecutive voxels it adds into (a plain byte store: idempotent, no atomic);
// volume_unpack_kernel reads -- and zeroes again -- only the marked groups and writes zeros for the rest, so the record
// volume is read where it was written instead of whole, and it LEAVES THE WORKSPACE ZERO: a caller that keeps the workspace
// zero-fills it once, not 0.9 GB per step (UFR_GBWD_WORKSPACE_ZEROED).
constexpr int kTouchGroup = 8;

// wave-level: the lanes with `alive` add their 9-float records val[0..8] at float offset key * kVolCh of `base`.
// slot: this wave's LDS staging area, 64 x 10 floats.
__device__ __forceinline__ void scatter_records(float* __restrict__ base, unsigned char* __restrict__ touch, bool alive, int key,
                                                const float (&val)[9], float* slot, int lane) {
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(alive);
  if (mask == 0ull) return;                                  // wave-uniform
  const int n = __builtin_popcountll(mask);
  const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
  if (alive) {
    float* r = slot + rank * 10;
    r[0] = __builtin_bit_cast(float, key);
#pragma unroll
    for (int c = 0; c < 9; ++c) r[1 + c] = val[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int sub = lane / 9, c = lane - 9 * sub;              // 7 records per instruction, lanes = channels
  for (int r0 = 0; r0 < n; r0 += 7) {
    const int rec = r0 + sub;
    if (sub < 7 && rec < n) {
      const float* r = slot + rec * 10;
      const int k = __builtin_bit_cast(int, r[0]);
      unsafeAtomicAdd(base + (size_t)k * kVolCh + c, r[1 + c]);
      if (c == 0) touch[k / kTouchGroup] = 1;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                            // the slot is reused by the next call
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(448) gather_bwd_kernel(FrameDev f, VolGrads vg, const float* __restrict__ ray_o,
                                                          int o_stride, const float* __restrict__ ray_d,
                                                          const float* __restrict__ zval, const float* __restrict__ d_pv,
                                                          const int* __restrict__ pv_row,
                                                          int P, int SN) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [64][NV][25] | [NV waves][64][10] record staging
  const int NV = f.NV;
  float* const rec_slot = smem + 64 * NV * 25 + (threadIdx.x >> 6) * 640;
  const int v = threadIdx.x >> 6, p = threadIdx.x & 63;
  const int pidx = blockIdx.x * 64 + p;
  const bool active = pidx < P;
  const int pc = active ? pidx : P - 1;
  const int ray = pc / SN;
  const float zz = zval[pc];
  const float* o = ray_o + (size_t)ray * o_stride;
  const float px = mul_add_unfused(zz, ray_d[3 * ray + 0], o[0]);
  const float py = mul_add_unfused(zz, ray_d[3 * ray + 1], o[1]);
  const float pz = mul_add_unfused(zz, ray_d[3 * ray + 2], o[2]);
  const float* M = f.pose[v];
  const float qx = fmaf(M[2], pz, fmaf(M[1], py, mul_rn(M[0], px))) + M[3];
  const float qy = fmaf(M[6], pz, fmaf(M[5], py, mul_rn(M[4], px))) + M[7];
  const float qz = fmaf(M[10], pz, fmaf(M[9], py, mul_rn(M[8], px))) + M[11];
  const float x = qx / qz, y = qy / qz;
  const float zn = ((qz - f.vol_near) / (f.vol_far - f.vol_near)) * 2.f - 1.f;

  float fl[24], wl = 0.f;
#pragma unroll
  for (int s = 0; s < UFR_NUM_STAGES; ++s) {
    float fs[8], ws;
    sample_volume(f.vol[s] + (size_t)v * f.vD[s] * f.vH[s] * f.vW[s] * kVolCh, f.vD[s], f.vH[s], f.vW[s], x, y, zn, fs, ws);
#pragma unroll
    for (int c = 0; c < 8; ++c) fl[8 * s + c] = fs[c];
    wl = s == 0 ? ws : wl + ws;
  }
  float* mine = smem + (p * NV + v) * 25;
#pragma unroll
  for (int c = 0; c < 24; ++c) mine[c] = fl[c] * wl;
  mine[24] = wl;
  __syncthreads();
  float Wsum = 0.f;
  for (int n = 0; n < NV; ++n) Wsum += smem[(p * NV + n) * 25 + 24];
  const float inv = 1.f / (Wsum + 1e-8f);
  // pv_row (nullable): row of d_pv holding slot pidx -- the two-pass step hands over ALL merged samples of a ray in one
  // launch (sorted: twice the density of either pass, so the run folding below removes more atomics) and d_pv as the pool
  const size_t pv_base = (size_t)(pv_row ? pv_row[pc] : pc) * 40;
  float dfl[24], dwl = 0.f;
#pragma unroll
  for (int c = 0; c < 24; ++c) {
    float G = 0.f;
    for (int n = 0; n < NV; ++n) G += smem[(p * NV + n) * 25 + c];
    const float dout = active ? d_pv[pv_base + c] : 0.f;
    dfl[c] = dout * wl * inv;
    dwl = fmaf(dout, (fl[c] - G * inv) * inv, dwl);
  }
  // (inactive lanes stay for the lane exchanges below; they contribute nothing)

#pragma unroll
  for (int s = 0; s < UFR_NUM_STAGES; ++s) {
    const int D = f.vD[s], H = f.vH[s], W = f.vW[s];
    const size_t plane = (size_t)D * H * W;
    float* grec = vg.rec[s] + (size_t)v * plane * kVolCh;
    unsigned char* gtouch = vg.touch[s] + (size_t)v * ((plane + kTouchGroup - 1) / kTouchGroup);
    const float ix = unnorm3d_ac(x, W), iy = unnorm3d_ac(y, H), iz = unnorm3d_ac(zn, D);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // the two corners (dz = 0, 1) of this (dy, dx) column of the cell
        float val[2][9];
        int off[2];
        bool alive[2];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
          const float cx = fx + dx, cy = fy + dy, cz = fz + dz;
          const bool ok = active && cx >= 0.f && cx <= (float)(W - 1) && cy >= 0.f && cy <= (float)(H - 1) && cz >= 0.f &&
                          cz <= (float)(D - 1);
          const float wt = ok ? wx[dx] * wy[dy] * wz[dz] : 0.f;
          off[dz] = ok ? (((int)cz * H + (int)cy) * W + (int)cx) : -1 - (int)threadIdx.x - 1024 * dz;   // invalid: never equal
#pragma unroll
          for (int c = 0; c < 8; ++c) val[dz][c] = wt * dfl[8 * s + c];
          val[dz][8] = wt * dwl;
          alive[dz] = ok;
        }
        // Neighbouring lanes are consecutive samples of one ray (both passes hand their samples over sorted), and a
        // straight line visits the cells of a grid monotonically.  Float atomics are what this kernel costs (plain stores
        // instead: 0.76 vs 4.5 ms), so equal addresses are folded in registers before any is issued:
        //  (a) across the cell boundary: when the next sample sits one cell further along z, its near corner IS this
        //      sample's far corner -- this lane hands its dz = 1 value to the next lane's dz = 0 slot;
        {
          const int n_off0 = row_down<1>(off[0]);                            // next lane's near corner
          const int n_alive0 = row_down<1>((int)alive[0]);
          const bool give = alive[1] && n_alive0 && n_off0 == off[1] && (threadIdx.x & 15) != 15;
          const int p_give = row_up<1>((int)give);                           // does the previous lane hand over?
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            const float v_up = row_up<1>(val[1][c]);
            if (p_give && (threadIdx.x & 15) != 0) val[0][c] += v_up;
          }
          if (give) alive[1] = false;
        }
        //  (b) inside a cell: equal corner offsets form contiguous runs -- aligned pairs, then quads, then octets of lanes
        //      with the same offset add up through lane shifts and only the surviving lane of each group issues atomics.
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
          static_for<UFR_GBWD_ROUNDS>([&](auto ri) __attribute__((always_inline)) {
            constexpr int d = 1 << decltype(ri)::value;
            const int k_dn = row_down<d>(off[dz]);                           // offset of lane + d (groups never leave a row)
            const int k_up = row_up<d>(off[dz]);                             // offset of lane - d
            const int a_dn = row_down<d>((int)alive[dz]), a_up = row_up<d>((int)alive[dz]);
            const int pos = threadIdx.x & (2 * d - 1);
            const bool absorb = pos == 0 && alive[dz] && a_dn && k_dn == off[dz];
            const bool absorbed = pos == d && alive[dz] && a_up && k_up == off[dz];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
              const float v_dn = row_down<d>(val[dz][c]);
              if (absorb) val[dz][c] += v_dn;
            }
            if (absorbed) alive[dz] = false;
          });
          scatter_records(grec, gtouch, alive[dz], off[dz], val[dz], rec_slot, threadIdx.x & 63);
        }
      }
  }
}

// channel-last scratch -> reference layout: feat (N,8,S) and weight (N,1,S) from rec (N,S,12); accumulate: += instead of =.
// Only the groups the scatter marked are read (and zeroed again, with their mark); the others are zeros (or, accumulating,
// nothing).  A group's 8 voxels are 8 consecutive threads of one wave: the mark is read by all of them in one instruction
// and cleared by the first after it.
__global__ void __launch_bounds__(256) volume_unpack_kernel(float* __restrict__ rec, unsigned char* __restrict__ touch,
                                                             float* __restrict__ feat, float* __restrict__ weight, int S,
                                                             int accumulate) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (s >= S) return;
  const size_t groups = ((size_t)S + kTouchGroup - 1) / kTouchGroup;
  unsigned char* mark = touch + (size_t)n * groups + s / kTouchGroup;
  const bool touched = *mark != 0;
  float v[8], w = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 0.f;
  if (touched) {
    float* r = rec + ((size_t)n * S + s) * kVolCh;
    const f32x4 a = ld4(r), b = ld4(r + 4);
    w = r[8];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    st4(r, splat4(0.f));
    st4(r + 4, splat4(0.f));
    r[8] = 0.f;
    if (s % kTouchGroup == 0) *mark = 0;     // (behind the wave's read of it: the branch above needed the value)
  }
  float* fo = feat + (size_t)n * 8 * S + s;
  float* wo = weight + (size_t)n * S + s;
  if (accumulate) {
    if (touched) {
#pragma unroll
      for (int c = 0; c < 8; ++c) fo[(size_t)c * S] += v[c];
      *wo += w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) fo[(size_t)c * S] = v[c];
    *wo = w;
  }
}

static size_t touch_bytes(const FrameDev& f, int i) {     // per stage: NV maps, each padded to a multiple of 16 bytes
  const size_t groups = ((size_t)f.vD[i] * f.vH[i] * f.vW[i] + kTouchGroup - 1) / kTouchGroup;
  return (size_t)f.NV * groups;
}
size_t gather_bwd_scratch_floats(const FrameDev& f) {
  size_t n = 0, t = 0;
  for (int i = 0; i < UFR_NUM_STAGES; ++i) {
    n += (size_t)f.NV * f.vD[i] * f.vH[i] * f.vW[i] * kVolCh;
    t += (touch_bytes(f, i) + 15) / 16 * 16;
  }
  return n + t / 4;
}

hipError_t launch_gather_bwd(const FrameDev& f, float* const* grad_feat, float* const* grad_weight, const float* ray_o,
                             int o_stride, const float* ray_d, const float* z, const float* d_pv, const int* pv_row, int RN, int SN,
                             float* scratch, bool accumulate, bool scratch_zeroed, hipStream_t s) {
  const int P = RN * SN, NV = f.NV;
  if (!scratch_zeroed) {      // 0.9 GB at the training size, 0.12 ms: a caller may zero it earlier, beside other work
    hipError_t e = hipMemsetAsync(scratch, 0, gather_bwd_scratch_floats(f) * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  VolGrads vg;
  size_t off = 0;
  for (int i = 0; i < UFR_NUM_STAGES; ++i) {
    vg.rec[i] = scratch + off;
    off += (size_t)NV * f.vD[i] * f.vH[i] * f.vW[i] * kVolCh;
  }
  unsigned char* tb = reinterpret_cast<unsigned char*>(scratch + off);
  for (int i = 0; i < UFR_NUM_STAGES; ++i) {
    vg.touch[i] = tb;
    tb += (touch_bytes(f, i) + 15) / 16 * 16;
  }
  const size_t lds = sizeof(float) * (64 * NV * 25 + NV * 640);
  hipLaunchKernelGGL(gather_bwd_kernel, dim3((P + 63) / 64), dim3(64 * NV), lds, s, f, vg, ray_o, o_stride, ray_d, z, d_pv, pv_row,
                     P, SN);
  for (int i = 0; i < UFR_NUM_STAGES; ++i) {
    const int S = f.vD[i] * f.vH[i] * f.vW[i];
    hipLaunchKernelGGL(volume_unpack_kernel, dim3((S + 255) / 256, NV), dim3(256), 0, s, vg.rec[i], vg.touch[i], grad_feat[i],
                       grad_weight[i], S, accumulate ? 1 : 0);
  }
  return hipGetLastError();
}

}  // namespace ufr
