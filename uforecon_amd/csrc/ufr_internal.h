// Internal host/device structs behind the opaque handles of include/ufr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/ufr.h"
#include "ufr_layout.h"

namespace ufr {

constexpr int kVolCh = 12;  // channel-last volume texel: 8 features + 1 weight + 3 pad = 48 B

// What ufr_frame_prepare records (lives inside the caller's ufr_frame; passed BY VALUE to kernels).
struct FrameDev {
  int32_t NV, H, W, h, w, match_ch;      // match_ch = 32*(NV-1)
  const float* feat;                     // [NV][h][w][32]
  const float* match;                    // [NV][h][w][match_ch]
  const float* rgb;                      // [NV][H][W][4]
  const float* depth;                    // [NV][H][W]   (borrowed, reference layout)
  const float* vol[UFR_NUM_STAGES];      // [NV][D][Hs][Ws][12]
  int32_t vD[UFR_NUM_STAGES], vH[UFR_NUM_STAGES], vW[UFR_NUM_STAGES];
  float pose[UFR_MAX_VIEWS][12];         // rows 0..2 of source_poses
  float cam_pos[UFR_MAX_VIEWS][3];
  float w2c_z[UFR_MAX_VIEWS][4];
  float ref_pos[3];
  float vol_near, vol_far;
  uint32_t magic;
  // device word: the largest |value| among the frame's feature maps and volume features, as a float's bit pattern --
  // measured by the re-layout kernels (prep.hip), consumed by ufr_weights_fit_frame
  const unsigned* abs_max;
};
static_assert(sizeof(FrameDev) <= sizeof(ufr_frame), "ufr_frame too small");
constexpr uint32_t kFrameMagic = 0x55465246u;  // "UFRF"

struct PreSim {  // pre_sim_mlp raw pointers (reference layout) used by the gather kernel
  const float *w0, *b0, *w2, *b2, *w4, *b4;
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set it once per (kernel, device),
// safely from any thread (the entry points are called from the caller's thread AND from autograd's workers).
struct LdsAttrOnce {
  std::once_flag flag[16];
  hipError_t err[16];
  hipError_t set(const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    std::call_once(flag[dev], [&] { err[dev] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    return err[dev];
  }
};

// kernel launchers (one per .hip file); all enqueue on `s` and return hipGetLastError()
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int S, int Cpad, hipStream_t s, unsigned* abs_max = nullptr);
hipError_t launch_volume_pack(const float* feat, const float* weight, float* out, int N, int S, hipStream_t s, unsigned* abs_max = nullptr);
hipError_t launch_ray_setup(const int64_t* ray_idx, const float* ray_d, const float* cam_ray_d, int HW, float near_z,
                            float far_z, int RN, float* rd_out, float* near_out, float* far_out, float* camz_out,
                            const float* ray_o_host, float* ray_o_out, hipStream_t s);
hipError_t launch_sample_fixed(const float* near, const float* far, const float* U, int u_stride, float* z, int RN,
                               int SN, hipStream_t s);
hipError_t launch_importance_merge(const float* weight, const float* z, const float* U2, int u_stride, float* z_fine,
                                   float* z_all, int RN, int SN, int PN, float* z_new, int* src_row, hipStream_t s);
hipError_t launch_order_pe(float* table, int SN, hipStream_t s);
hipError_t launch_points(const float* ray_o, int o_stride, const float* ray_d, const float* z, float* pts, int RN,
                         int SN, hipStream_t s);
// Token rows.  Public layout (x_point == nullptr): x_tokens (P, NV, 80) = [feat 32 | frustum lookup 24 | pre_sim_mlp 16 |
// depth PE 8] per (point, view).  Columns 32..71 are the same for all views of a point, and writing them NV times is a
// sixth of the gather kernel's time (ablation, round 3): the whole-path entry point keeps them once per point --
// compact layout (x_point != nullptr): x_tokens (P, NV, kViewCols) = [feat 32 | depth PE 8], x_point (P, kPointCols) =
// [frustum lookup 24 | pre_sim_mlp 16].
constexpr int kViewCols = 40, kPointCols = 40;
hipError_t launch_gather(const FrameDev& f, const PreSim& ps, const float* ray_o, int o_stride, const float* ray_d,
                         const float* z, int RN, int SN, float* x_tokens, float* x_point, float* rgb, float* dir, float* sim8,
                         float* vol24, float* xy, float* mask_z, const float* vol24_in, const float* sim8_in,
                         hipStream_t s);
// lowp: matrix precision of the call (include/ufr.h): false = fp32-grade split precision, true = one 16-bit plane per
// operand ("bf16" training mode of BASELINE configs[4]).  status: the device's sticky range word (ufr_status_poll).
hipError_t launch_view_transformer(const float* packed, const float* x_tokens, const float* x_point, const float* rgb,
                                   const float* dir, int P, int NV, float* token0, float* radiance, float* view_out, bool lowp, int* status,
                                   hipStream_t s);
// tok_row / rad_row (nullable): row of token0 / radiance holding sample (ray, s) -- the fine pass of the whole-path
// renderer keeps coarse and new evaluations in one pool instead of re-evaluating the coarse points
hipError_t launch_ray_transformer(const float* packed, const float* token0, const int* tok_row, const float* order_pe,
                                  int RN, int SN, float* srdf, float* ray_out, bool lowp, int* status, hipStream_t s);
hipError_t launch_composite(const float* z, const float* radiance, const int* rad_row, const float* srdf,
                            const float* variance, int RN, int SN, float* rgb, float* depth, float* opacity, float* weight,
                            const float* camz, float* depth_z, hipStream_t s);
// render_loss.hip: the training loss of a ray batch and its cotangents (model.py:552-566), one launch
hipError_t launch_render_loss(const float* rgb_c, const float* depth_c, const float* rgb_f, const float* depth_f,
                              const float* rgb_gt, const float* depth_gt, const float* near_far, int nf_stride, int B, int RN,
                              float weight_rgb, float weight_depth, float* loss, float* d_rgb_c, float* d_depth_c,
                              float* d_rgb_f, float* d_depth_f, hipStream_t s);
hipError_t launch_composite_bwd(const float* z, const float* radiance, const int* rad_row, bool accumulate, const float* srdf,
                                const float* variance, int RN, int SN, const float* d_rgb, const float* d_depth,
                                const float* d_opacity, const float* d_weight, float* d_radiance, float* d_srdf,
                                float* d_variance, hipStream_t s);
struct GradPtrs;
// The streaming view-transformer backward (round 4; bwd_tape.h): the forward again with a tape, the data-gradient chain,
// the weight-gradient contraction.  tape: view_tape_blocks(P, NV) blocks of TV_COUNT tiles; dbuf: as many blocks of DV_COUNT.
int view_tape_blocks(int P, int NV);
hipError_t launch_view_tape(const float* packed, const float* x_tokens, const float* rgb, const float* dir, int P, int NV,
                            float* token0, float* radiance, float* tape, bool lowp, int* status, hipStream_t s);
hipError_t launch_view_dgrad(const float* packed, const float* tape, const float* rgbm, const float* d_tok_a,
                             const float* d_tok_b, const float* d_radiance, int P, int NV, float* dbuf, float* d_pv,
                             const GradPtrs& gp, bool lowp, hipStream_t s);
hipError_t launch_view_wgrad(const float* tape, const float* dbuf, int n_blocks, const GradPtrs& gp, bool lowp, hipStream_t s);
// ... and the ray transformer's: tape = RN x ceil(SN / 32) blocks of RT_COUNT tiles, ray_state = RN x kRayStateTiles tiles,
// dbuf = as many blocks of DR_COUNT tiles.  d_tok_a receives the whole gradient (rows through tok_row, += when accumulate),
// d_tok_b (nullable) is zero-filled when not accumulating: the two-buffer form of the former kernel's interface.
hipError_t launch_ray_tape(const float* packed, const float* token0, const int* tok_row, const float* order_pe, int RN, int SN,
                           float* srdf, float* tape, float* ray_state, bool lowp, int* status, hipStream_t s);
hipError_t launch_ray_dgrad(const float* packed, const float* tape, const float* ray_state, const float* d_srdf, const int* tok_row,
                            bool accumulate, int RN, int SN, float* dbuf, float* d_tok_a, float* d_tok_b, const GradPtrs& gp, bool lowp,
                            hipStream_t s);
hipError_t launch_ray_wgrad(const float* tape, const float* dbuf, int n_blocks, const GradPtrs& gp, bool lowp, hipStream_t s);
hipError_t launch_presim_bwd(const RawPtrs& wp, const GradPtrs& gp, const float* sim8, const float* d_pv, int P, bool lowp,
                             hipStream_t s);
// scratch: gather_bwd_scratch_floats(f) floats (zeroed by the launcher): the channel-last record volumes of the scatter
size_t gather_bwd_scratch_floats(const FrameDev& f);
hipError_t launch_gather_bwd(const FrameDev& f, float* const* grad_feat, float* const* grad_weight, const float* ray_o,
                             int o_stride, const float* ray_d, const float* z, const float* d_pv, const int* pv_row, int RN,
                             int SN, float* scratch, bool accumulate, bool scratch_zeroed, hipStream_t s);
struct FmtWeights {  // = ufr_fmt_layer_weights
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *w1, *b1, *w2, *b2, *n1w, *n1b, *n2w, *n2b;
};
int fmt_state_parts(int S);   // per-wave partial states of one sample's source tokens
hipError_t launch_fmt_layer(const FmtWeights& w, const float* x, const float* src, int N, int T, int S, float* out,
                            float* state, hipStream_t s);
hipError_t launch_pack_weights(const RawPtrs& raw, float* packed, float input_abs_max, int* range_flag, hipStream_t s);
hipError_t launch_refit_weights(float* packed, const unsigned* frame_bound, int* range_flag, hipStream_t s);
hipError_t launch_tsdf_integrate(float* tsdf, float* weight, float* color, const int* dim, const float* origin,
                                 float voxel_size, float trunc_margin, const float* K, const float* P,
                                 const float* depth_im, const float* color_im, int im_h, int im_w, float obs_weight,
                                 int integrate_color, hipStream_t s);
// conv2d.hip: the plain 2-D convolutions of FeatureNet on channel-last tensors, epilogue fused (see the file header)
struct Conv2dArgs {
  const float* in;      // [B][H][W][CIN] (the stem: planar [B][3][H][W])
  const float* w;       // [cout][CIN][KS][KS]
  const float* scale;   // [cout] or null (= 1)
  const float* shift;   // [cout] or null (= 0)
  const float* skip;    // [B][Ho/2][Wo/2][cout] added after nearest 2x upsampling, or null
  float* out;           // [B][Ho][Wo][cout], or planar [B][cout][Ho][Wo]
  int B, H, W, Ho, Wo, cout;
  int relu, out_planar, sigmoid_from;   // sigmoid on channels >= sigmoid_from (< 0: none)
};
hipError_t launch_conv2d(const Conv2dArgs& a, int cin, int ks, int stride, int in_planar, hipStream_t s);
hipError_t launch_upsample_add(const float* reduced, const float* fine, float* out, int B, int C, int h, int w, hipStream_t s);
// dcn.hip, channel-last form: epilogue out = [relu]((dcn + bias) * scale + shift), channel-last or planar output
struct DcnEpilogue {
  const float* scale;
  const float* shift;
  int relu, out_cl;
  int om_planes;      // 0: offset [B][18][H][W] and mask [B][9][H][W]; 27: one [B][27][H][W] block, mask = offset + 18 planes
};
hipError_t launch_deform_conv3x3(const float* in_cl, const float* offset, const float* mask, const float* weight,
                                 const float* bias, float* out, int B, int C, int Cout, int H, int W, hipStream_t s,
                                 DcnEpilogue ep = DcnEpilogue{nullptr, nullptr, 0, 0, 0});
size_t conv3d_planes_workspace_bytes(int cin, int cout, int cout2, int mode);
hipError_t launch_conv3d_wgrad_heads(const float* in, const float* d_out, const float* d_out2, float* dw, float* dw2, int B, int D, int H,
                                     int W, hipStream_t s);
hipError_t launch_conv3d_wgrad_planes(const float* tp, const float* tq, float* dw, float* dbias, int B, int Dp, int Hp, int Wp, int Dq,
                                      int Hq, int Wq, int ca, int cb, int S, hipStream_t s);
hipError_t launch_absmax(const float* x, size_t n, float* absmax, hipStream_t s);
hipError_t launch_conv3d_planes(const float* in, const float* in_absmax, const float* weight, const float* weight2, const float* bias,
                                const float* scale, const float* shift, const float* skip, float* out, float* out2, float* out_absmax,
                                int B, int D, int H, int W, int cin, int cout, int cout2, int mode, int relu, int ncdhw, int flip,
                                void* ws, int planes_ready, hipStream_t s);
hipError_t launch_conv3d(const float* in, const float* weight, const float* weight2, const float* bias, const float* scale,
                         const float* shift, const float* skip, float* out, float* out2, int B, int D, int H, int W,
                         int cin, int cout, int cout2, int mode, int relu, int ncdhw, hipStream_t s, int flip = 0,
                         float* out_absmax = nullptr);
hipError_t launch_conv3d_bwd_data(const float* d_out, const float* weight, const float* accumulate, float* d_in, int B, int D,
                                  int H, int W, int cin, int cout, int mode, hipStream_t s);
hipError_t launch_conv3d_bwd_weight(const float* in, const float* d_out, float* d_weight, float* d_bias, int B, int D, int H, int W,
                                    int cin, int cout, int mode, hipStream_t s);
hipError_t launch_chw_to_hwc(const float* in, float* out, int N, int C, int S, hipStream_t s);
hipError_t launch_pixelwise_weights(const float* sim, const float* params, float* vw, float* agg, int NS, int D, int H, int W,
                                    hipStream_t s);
hipError_t launch_correlate(const float* ref_cl, const float* src_cl, const float* proj_host, int NS, const float* depth,
                            const float* vw, float* sim, float* agg, int C, int H, int W, int D, hipStream_t s);

}  // namespace ufr
