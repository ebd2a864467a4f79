/*
 * ufr.h -- C ABI of libufr.so: the MI355X (gfx950) implementation of UFORecon's per-ray
 * volume-rendering path.
 *
 * Every entry point replaces a piece of the reference's Python hot path (paths relative
 * to the upstream repo, code1/...).  Plain pointers and sizes only; all tensor pointers
 * are DEVICE pointers to contiguous fp32 (int64 for ray indices) unless marked "host".
 * Kernels are enqueued on the given HIP stream and never synchronise.  Return value:
 * 0 on success, negative ufr_status otherwise; ufr_last_error() gives the message
 * (thread-local).  Shapes use the reference's names: NV source views, RN rays, SN samples
 * per ray, P = RN*SN points, H x W image, h x w = H/4 x W/4 feature maps.
 */
#ifndef UFR_H
#define UFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ufr_stream; /* hipStream_t */

enum ufr_status {
  UFR_OK = 0,
  UFR_ERR_ARG = -1,       /* bad shape / null pointer / unsupported size */
  UFR_ERR_HIP = -2,       /* a HIP runtime call failed */
  UFR_ERR_WORKSPACE = -3, /* workspace too small */
  UFR_ERR_RANGE = -4      /* a value left the range the split-precision planes can hold (see ufr_status_poll) */
};

/* Version of this header's ABI (argument lists, struct layouts, packed-blob layout).  ufr_version() returns the value
 * the library was built with: a binding must refuse a library whose version differs (uforecon_amd/_lib.py does). */
#define UFR_ABI_VERSION 503

#define UFR_MAX_VIEWS 7
#define UFR_NUM_STAGES 3
#define UFR_TOKEN_DIM 80 /* [img feat 32 | volume 24 | similarity 16 | depth PE 8], ray_transformer.py:258-281 */
#define UFR_RAY_DIM 88   /* token dim + 8 order PE, ray_transformer.py:301-303 */

int ufr_version(void);
const char* ufr_last_error(void);

/* ------------------------------------------------------------------ weights
 * Device pointers to the per-ray parameters in the reference's own layout (nn.Linear
 * weight = row-major [out][in]); names follow the state_dict keys under
 * "ray_transformer." (parameter inventory). */
typedef struct ufr_layer_weights { /* one LoFTREncoderLayer, attention/transformer.py:7-58 */
  const float *q, *k, *v, *merge; /* [d][d]            */
  const float *mlp0;              /* [2d][2d]          */
  const float *mlp2;              /* [d][2d]           */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [d] */
} ufr_layer_weights;

typedef struct ufr_mlp3_weights { /* Linear-ReLU-Linear-ReLU-Linear with bias */
  const float *w0, *b0, *w2, *b2, *w4, *b4;
} ufr_mlp3_weights;

typedef struct ufr_raw_weights {
  ufr_mlp3_weights pre_sim;   /* pre_sim_mlp.{0,2,4}: 8->32->32->16          (ray_transformer.py:128-132) */
  ufr_layer_weights view;     /* density_view_transformer.layers.0, d=80     (ray_transformer.py:135)     */
  ufr_layer_weights ray;      /* density_ray_transformer.layers.0, d=88      (ray_transformer.py:138)     */
  ufr_mlp3_weights density;   /* DensityMLP.{0,2,4}: 88->32->16->1           (ray_transformer.py:147-150) */
  ufr_mlp3_weights radiance;  /* linear_radianceweight_1_softmax: 83->16->8->1 (ray_transformer.py:159-163) */
  const float* view_token;    /* viewToken.view_token [80]                   (ray_transformer.py:325-331) */
  const float* variance;      /* deviation_network.variance, scalar          (single_variance_network.py:8) */
} ufr_raw_weights;

/* Matrix precision of the dense layers.  Every entry point that runs dense layers takes a `precision` argument
 * (ufr_render_args has a field); the forward and the backward of one training step must be given the same value
 * (uforecon_amd/autograd.py records the forward's in the autograd context).
 *   UFR_PRECISION_DEFAULT the process default, FP32 unless ufr_set_matrix_precision changed it (a convenience for
 *                       command-line tools; library code that pairs a forward with a backward passes an explicit mode)
 *   UFR_PRECISION_FP32  fp32-grade FORWARD: every product as three fp16 plane products (22+ significand bits per operand),
 *                       fp32 accumulate -- the mode all 1e-4 parity statements refer to.  The BACKWARD of this mode splits
 *                       its operands into bf16 hi + lo planes (16 significand bits per operand, three products, fp32
 *                       accumulate: "bf16x3"): gradient-grade, not fp32-grade -- every gradient tensor within 5e-5 of its
 *                       scale of the reference's fp32 autograd (tests/test_gpu_backward.py; the bench line says
 *                       `bwd_operand_dtype`).
 *   UFR_PRECISION_16BIT the "bf16" training mode of the reference's mixed-precision recipe (BASELINE configs[4]): one
 *                       16-bit plane per operand (fp16 hi planes in the forward, bf16 operands in the backward GEMMs and
 *                       weight gradients), fp32 accumulation, LayerNorm / attention / softmax / compositor in fp32.
 *                       Tolerance: 2e-3 of scale on forward rows, 3e-2 on gradients (tests/test_gpu_backward.py). */
#define UFR_PRECISION_DEFAULT (-1)
#define UFR_PRECISION_FP32 0
#define UFR_PRECISION_16BIT 1
int ufr_set_matrix_precision(int mode); /* sets what UFR_PRECISION_DEFAULT resolves to (FP32 or 16BIT) */
int ufr_get_matrix_precision(void);

/* Sticky range status of the current device.  The dense layers run on fp16 planes whose power-of-two scales are chosen
 * when the weights are packed: per matrix from max |w| (any finite weight fits), per layer from an analytic upper bound
 * of the layer's input that starts at the bound of the token features and is carried through the matrices' infinity
 * norms, the LayerNorm gains and the biases.  The bound of the token features is MEASURED: ufr_frame_prepare takes the
 * maximum magnitude of the frame's feature maps and volume features while it re-lays them out, and ufr_render_rays /
 * ufr_weights_fit_frame re-derive the layer exponents (scalars of the table; the weight planes do not change) for a frame
 * beyond the bound the table serves -- ufr_weights_pack_for's input_abs_max, default 4094, is only the floor and an
 * optional override.  With a true input bound no layer can overflow.  The kernels never synchronise, so a violation
 * (tokens handed to ufr_aggregate / ufr_view_transform directly, beyond the floor and without a frame) raises a
 * device-side sticky flag instead of failing the launch:
 *   bit 0  a dense-layer input of a transformer kernel left the range of its planes: a token feature beyond the bound
 *          the table serves, or +-inf -- ufr_weights_fit_frame with the frame the tokens come from, or repack with a
 *          larger input_abs_max
 *   bit 1  NaN among the externally supplied inputs of a transformer kernel (token rows, dir); +-inf inputs and
 *          internally produced overflows raise bit 0
 *   bit 2  ufr_weights_pack met a parameter that is not finite
 * ufr_status_poll copies the flag to the host on `stream`; with synchronize != 0 it waits for the stream and returns
 * UFR_ERR_RANGE (message: which bits) when the flag is set, clearing it.  Without synchronize it returns what an
 * EARLIER poll / launch has already delivered.  ufr_render_rays, ufr_aggregate, ufr_view_transform and
 * ufr_ray_transform poll lazily: each enqueues the copy after its kernels and fails with UFR_ERR_RANGE on entry when a
 * previous call's copy arrived set -- one call late, without a host synchronisation in the ray loop.  `flags_out`
 * (nullable) receives the bits.  Across streams: every copy waits (on its own stream) for the last report-and-clear, so a
 * poll on another stream than the one that reported cannot miss bits raised since; a bit raised again between a report's
 * copy and its clear is reported with the next violation rather than at once (reports can be late, never spurious). */
int ufr_status_poll(ufr_stream stream, int32_t synchronize, int32_t* flags_out);
/* The same, restricted to the bits of `mask`: other bits stay set for whoever polls for them (uforecon_amd.ops.PackedWeights
 * checks a fresh pack with mask 4, so that an unrelated, still unreported activation overflow does not fail a valid pack). */
int ufr_status_poll_bits(ufr_stream stream, int32_t synchronize, int32_t mask, int32_t* flags_out);

/* Re-orders the dense matrices into MFMA A-fragment order (one 16x16 output tile x 16
 * input features = 64 lanes x float4, zero padded) so the kernels stream them with
 * contiguous 1 KiB wave loads.  Call again whenever the parameters change. */
size_t ufr_packed_weights_bytes(void);
int ufr_weights_pack(const ufr_raw_weights* raw, void* packed, ufr_stream stream);   /* input_abs_max = 4094 */
/* The same for feature maps and volume features bounded by input_abs_max (positive, finite) -- what the encoder hands to
 * ufr_frame_prepare; for ufr_aggregate / ufr_view_transform callers: columns 0..55 and 72..79 of x_tokens (the 16
 * pre-similarity columns are bounded from pre_sim_mlp's own weights, the positional columns by 1).  The bound
 * only sets the exponents the activations' planes carry (a pessimistic one costs no accuracy: the planes keep 22
 * significand bits down to 2^-17 of each layer's bound), so state it generously. */
int ufr_weights_pack_for(const ufr_raw_weights* raw, void* packed, float input_abs_max, ufr_stream stream);
/* (ufr_weights_fit_frame, declared with the frame handle below: the table follows a frame's measured bound) */
/* Host-only description of that re-ordering (for tests / other bindings): for every packed
 * float, param_id (index into the pointer list of ufr_raw_weights in declaration order, -1 =
 * zero padding) and elem (flat element index inside that parameter).  Arrays of
 * ufr_packed_weights_bytes()/4 int32 each.  No GPU needed. */
int ufr_pack_plan(int32_t* param_id, int32_t* elem);
/* The packed blob is [fp32 region: ufr_packed_fp32_floats() floats | fp16 plane region: ufr_packed_f16_halfwords()
 * 16-bit words | bf16 plane region of the backward kernels: ufr_packed_bwd_halfwords() 16-bit words | 16-byte tail].
 * The bf16 region holds the TRANSPOSED dense matrices of the data-gradient chains (hi = bf16(w), lo = bf16(w - hi), same
 * fragment order as the forward planes).  The fp16 plane region holds the dense layers of both transformer chains as two fp16
 * planes per weight (w' = 2^s_M w with the matrix's exponent s_M: hi = fp16(w'), lo = fp16(w' - hi); plane 0/1) for the
 * split-precision MFMA path;
 * ufr_pack_plan_f16 describes it like ufr_pack_plan (one entry per halfword, plus the plane).  Of the fp32 region
 * ufr_weights_pack fills only the trailing vector fragments (biases, LayerNorm, view token) and, behind them, the scale
 * table: per dense matrix M (ufr_packed_scale_table_offset() floats into the blob, 4 floats each, in the order of
 * csrc/ufr_layout.h: Mat) {2^a_M, 2^-(s_M + a_M), 2^(s_M + a_M), 2^s_M}, then the two kernels' scalar lists and the weight
 * statistics the exponents derive from (kept for ufr_weights_fit_frame); the kernels read nothing else of the region.
 * ufr_weights_pack does not synchronise: a parameter that is not finite raises bit 2 of the sticky status -- call
 * ufr_status_poll(stream, 1, ...) after packing to fail at once (uforecon_amd.ops.PackedWeights does), or let the next
 * compute entry point report it. */
size_t ufr_packed_scale_table_offset(void);
int ufr_packed_scale_table_entries(void);
size_t ufr_packed_fp32_floats(void);
size_t ufr_packed_f16_halfwords(void);
size_t ufr_packed_bwd_halfwords(void);
int ufr_pack_plan_bwd(int32_t* param_id, int32_t* elem, int32_t* plane);   /* the bf16 region, like ufr_pack_plan_f16 */
int ufr_pack_plan_f16(int32_t* param_id, int32_t* elem, int32_t* plane);

/* ------------------------------------------------------------------ frame
 * Per-frame tensors produced by the encoder (model.py:780-808), in the reference layout.
 * ufr_frame_prepare re-lays them out channel-last into `workspace` (one tap = one
 * contiguous line), measures max |value| of the feature maps and volume features on the way (a device word inside
 * `workspace`: what ufr_weights_fit_frame reads) and records the camera constants; the result is immutable during the
 * ray loop (model.py:814-823). */
typedef struct ufr_frame_desc {
  int32_t NV, H, W;               /* full-resolution image size; feature maps are H/4 x W/4          */
  const float* source_imgs;       /* (NV,3,H,W)   batch['source_imgs']                               */
  const float* depth_info;        /* (NV,H,W)     batch['depth_info'] (model.py:807-808)             */
  const float* feat;              /* (NV,32,h,w)  source_imgs_feat                                   */
  const float* match;             /* (NV,32*(NV-1),h,w) match_feature[0]; NULL: ufr_project_gather needs sim8_in */
  const float* vol_feat[UFR_NUM_STAGES];   /* (NV,8,D,Hs,Ws) feature_volume[stage]['feature_volume']; all NULL (with
                                              vol_weight): ufr_project_gather needs vol24_in              */
  const float* vol_weight[UFR_NUM_STAGES]; /* (NV,1,D,Hs,Ws) feature_volume[stage]['weight_volume']   */
  int32_t vol_D[UFR_NUM_STAGES], vol_H[UFR_NUM_STAGES], vol_W[UFR_NUM_STAGES];
  /* camera constants, HOST pointers (copied) */
  const float* source_poses;      /* (NV,4,4) normalize @ K_pad @ w2c  (dtu_test_sparse.py:409-416)  */
  const float* source_cam_pos;    /* (NV,3)   source_poses_inv[:, :3, 3] (ray_transformer.py:187)    */
  const float* ref_cam_pos;       /* (3)      ref_pose_inv[:3, 3]        (ray_transformer.py:185)    */
  const float* w2c_row2;          /* (NV,4)   w2cs[s_idx:, 2, :]         (ray_transformer.py:240-243) */
  float vol_near, vol_far;        /* batch['near_fars'][0][0]            (model.py:328)              */
} ufr_frame_desc;

/* Host-side handle filled by ufr_frame_prepare (device pointers into `workspace`, sizes, camera
 * constants).  Plain data: copy it freely; it stays valid while `workspace` and the borrowed
 * depth_info tensor are alive. */
typedef struct ufr_frame { uint64_t opaque[160]; } ufr_frame;

size_t ufr_frame_workspace_bytes(const ufr_frame_desc* d);
int ufr_frame_prepare(const ufr_frame_desc* d, void* workspace, size_t workspace_bytes, ufr_frame* out,
                      ufr_stream stream);
/* The activation exponents of `packed` follow a frame: when the feature bound ufr_frame_prepare measured for `frame`
 * exceeds the one the table was derived for, the table is re-derived on the device for the next power of two above it
 * (one 64-thread kernel; otherwise it returns at once).  Asynchronous, no host read-back.  The bound only grows until the
 * next ufr_weights_pack, so the backward of an earlier forward stays in range whatever frames are fitted in between.
 * ufr_render_rays does this itself; callers of the stepwise entry points (ufr_project_gather -> ufr_aggregate ...) call it
 * once per (frame, pack) -- uforecon_amd.ops does, in FrameHandle-taking calls.  Replaces "state input_abs_max for this
 * checkpoint": the reference loads any checkpoint without side information (main.py:186-190).
 * OWNERSHIP: the refit WRITES the scale table inside `packed` (ufr_render_rays does too, although it takes the blob as
 * const void*: the planes are const, the table is not).  A packed blob therefore belongs to ONE stream at a time: kernels of
 * another stream, or of an earlier call on another stream, that still read the table while a refit runs would race.  Fit on
 * the stream the render / aggregate calls use (stream order then serialises refit and readers), and give every
 * concurrently rendered frame stream its own packed copy.  Activations taped under an older, smaller table stay valid:
 * the bound only grows, and a backward recomputes with the table it finds. */
int ufr_weights_fit_frame(void* packed, const ufr_frame* frame, ufr_stream stream);

/* ------------------------------------------------------------------ per-op entry points
 * (each mirrors one reference function; used by the drop-in Python classes and the parity
 * tests; ufr_render_rays below chains them for whole-frame inference) */

/* FixedSampler.sample_ray with near/far (encoder_utils/sampler.py:15-50).
 * U: (SN,RN) uniforms exactly as torch.rand((SN,RN)) produced them.  z_out: (RN,SN). */
int ufr_sample_fixed(const float* near, const float* far, const float* U, float* z_out,
                     int32_t RN, int32_t SN, ufr_stream stream);

/* ImportanceSampler.sample_ray (sampler.py:74-108) fused with the coarse+fine merge
 * (model.py:466-470).  weight,z: (RN,SN); U2: (PN,RN) as drawn by torch.rand(PN,RN);
 * z_fine: (RN,PN) sorted (may be NULL); z_all: (RN,SN+PN) sorted. */
int ufr_sample_importance_merge(const float* weight, const float* z, const float* U2, float* z_fine,
                                float* z_all, int32_t RN, int32_t SN, int32_t PN, ufr_stream stream);

/* points = ray_o + z * ray_d (sampler.py:47).  ray_o: (3) or (RN,3) by `ray_o_stride` (0 or 3). */
int ufr_points(const float* ray_o, int32_t ray_o_stride, const float* ray_d, const float* z, float* points,
               int32_t RN, int32_t SN, ufr_stream stream);

/* Projection + all gathers of one pass: camera.get_coord_ref_ndc (misc/camera.py:378-407),
 * query_cond_info (model.py:218-305), query_depth_from_volume (model.py:350-390) and
 * ray_transformer.py:185-281 up to the token assembly (incl. pre_sim_mlp).
 * Outputs: x_tokens (P,NV,80); rgb (P,NV,4) [r,g,b,mask]; dir (P,NV,4) [dx,dy,dz,0].
 * Optional outputs (may be NULL): sim8 (P,8) = cond_info['feat_info'], vol24 (P,24) = the blended frustum lookup,
 * xy (NV,P,2) = points_pixel, mask_z (NV,P) = mask_valid.
 * Optional inputs (may be NULL): vol24_in (P,24) / sim8_in (P,8) replace the kernel's own frustum lookup / pair
 * similarity -- RayTransformer.forward receives them as `fea_volume` / cond_info['feat_info']
 * (ray_transformer.py:175, 199, 265); a frame prepared without volumes / matching features requires them. */
int ufr_project_gather(const ufr_frame* frame, const ufr_raw_weights* raw, const float* ray_o,
                       int32_t ray_o_stride, const float* ray_d, const float* z, int32_t RN, int32_t SN,
                       float* x_tokens, float* rgb, float* dir, float* sim8, float* vol24, float* xy,
                       float* mask_z, const float* vol24_in, const float* sim8_in, ufr_stream stream);

/* View transformer + ray transformer + SRDF / radiance heads (ray_transformer.py:283-320).
 * radiance: (P,3); srdf: (RN,SN).  workspace >= ufr_aggregate_workspace_bytes(RN,SN).
 * Optional debug outputs (may be NULL): view_out (P,NV+1,80), ray_out (P,88).
 * One call handles at most P * (NV + 1) * 80 < 2^30 token values (6.7 M points at NV = 3; the view-transformer kernel
 * addresses its buffers with 32-bit offsets): UFR_ERR_ARG beyond it -- chunk the points (ufr_render_rays does). */
size_t ufr_aggregate_workspace_bytes(int32_t RN, int32_t SN, int32_t NV);
int ufr_aggregate(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir,
                  int32_t RN, int32_t SN, int32_t NV, float* radiance, float* srdf, void* workspace,
                  float* view_out, float* ray_out, int32_t precision, ufr_stream stream);

/* VolumeRenderer.render (encoder_utils/renderer.py:7-48) with SingleVarianceNetwork
 * (single_variance_network.py:10-11).  z,srdf: (RN,SN); radiance: (RN,SN,3); variance: device scalar.
 * row (nullable, (RN,SN) int32): the colour of slot (ray, s) is radiance[row[ray][s]] -- the sample pool of the two-pass
 * step (ufr_sample_importance_pool); NULL = slot order.
 * Outputs: rgb (RN,3), depth (RN), opacity (RN), weight (RN,SN); any may be NULL except depth. */
int ufr_composite(const float* z, const float* radiance, const int32_t* row, const float* srdf, const float* variance,
                  int32_t RN, int32_t SN, float* rgb, float* depth, float* opacity, float* weight,
                  ufr_stream stream);

/* ------------------------------------------------------------------ backward (training step, BASELINE configs[4])
 * The reference gets these from autograd over code1/model.py:540-566 (training_step: infer -> losses).  Which
 * tensors receive gradients: every ray_transformer.* parameter, deviation_network.variance and the six sampled
 * volumes (through which feature_volume.cost_reg_2.* trains); not the 2-D feature maps / matching features (their
 * producer is frozen, model.py:82-83) and not the sample positions (model.py:456-457 detaches them).
 * Each *_bwd entry point is the adjoint of the forward entry point of the same name and takes that call's inputs
 * again (activations are recomputed, not stored).  Parameter gradients are ACCUMULATED (+=) into caller-owned
 * tensors of the parameters' own shapes: zero them once per step. */
#define UFR_NUM_PARAMS 40
typedef struct ufr_raw_grads {
  float* p[UFR_NUM_PARAMS]; /* one per pointer of ufr_raw_weights, in its declaration order; same shapes */
} ufr_raw_grads;

/* Adjoint of ufr_composite (autograd of renderer.py:19-46).  d_rgb (RN,3), d_depth (RN), d_opacity (RN),
 * d_weight (RN,SN): gradients of the outputs, any may be NULL (= zero).  Outputs: d_radiance (rows as `radiance`: slot
 * order, or pool rows through `row`; overwritten, or added to when accumulate != 0 -- every row is touched once per call),
 * d_srdf (RN,SN) (overwritten); d_variance: device scalar, ACCUMULATED. */
int ufr_composite_bwd(const float* z, const float* radiance, const int32_t* row, const float* srdf, const float* variance,
                      int32_t RN, int32_t SN, const float* d_rgb, const float* d_depth, const float* d_opacity,
                      const float* d_weight, float* d_radiance, int32_t accumulate, float* d_srdf, float* d_variance,
                      ufr_stream stream);

/* The training loss of a ray batch and its cotangents in one launch (code1/model.py:552-566):
 *   loss = weight_rgb (mse(rgb_c, rgb_gt) + mse(rgb_f, rgb_gt)) + weight_depth (l1(depth_c, depth_gt | valid) + l1(depth_f, ...)),
 *   valid = (depth_gt != 0) & (depth_gt >= near) & (depth_gt <= far), the two L1 terms 0 when no ray is valid.
 * rgb_* (B*RN,3), depth_* (B*RN): both passes' rendered colours / ray depths and the ground truth, batch-major;
 * near_far: near = near_far[b * nf_stride], far = near_far[b * nf_stride + 1] of batch element b (batch['near_fars'][b, 0]:
 * nf_stride = V*2).  Outputs: loss[5] = {total, rgb_c, rgb_f, depth_c, depth_f} (the reference logs all five);
 * d_rgb_c (B*RN,3), d_depth_c (B*RN), d_rgb_f, d_depth_f = d total / d input (sign(0) = 0 in the L1 terms, as torch). */
int ufr_render_loss(const float* rgb_c, const float* depth_c, const float* rgb_f, const float* depth_f, const float* rgb_gt,
                    const float* depth_gt, const float* near_far, int32_t nf_stride, int32_t B, int32_t RN, float weight_rgb,
                    float weight_depth, float* loss, float* d_rgb_c, float* d_depth_c, float* d_rgb_f, float* d_depth_f,
                    ufr_stream stream);

/* Adjoint of ufr_aggregate (autograd of ray_transformer.py:283-320).  x_tokens / rgb / dir: the forward's inputs;
 * token0 (P,80): the view transformer's token-0 output = the first RN*SN*80 floats of the forward's workspace;
 * d_radiance (P,3), d_srdf (RN,SN): gradients of the forward's outputs.  Accumulates the gradients of the view / ray
 * transformer, DensityMLP, radiance-weight MLP and view-token parameters into `grads`; writes d_pv (P,40): gradient
 * w.r.t. token columns 32..71 (24 frustum features | 16 pre_sim_mlp outputs) summed over the NV view tokens of a
 * point -- the input of ufr_project_gather_bwd.  packed_weights: ufr_weights_pack of the same parameters (the view
 * transformer's backward re-runs the forward kernel with a tape and walks the chain backwards on transposed weight planes
 * of the packed blob; its weight gradients are one streaming contraction over the tokens).  The workspace holds the tape
 * and the cotangent tiles: ~15 KB per point at NV = 3.  (ABI 500: the `debug_ray` dump argument of earlier versions, ignored
 * since the round-4 kernels, is gone.) */
size_t ufr_aggregate_bwd_workspace_bytes(int32_t RN, int32_t SN, int32_t NV);
int ufr_aggregate_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights, const float* x_tokens,
                      const float* rgb, const float* dir, const float* token0, int32_t RN, int32_t SN, int32_t NV,
                      const float* d_radiance, const float* d_srdf, float* d_pv, void* workspace,
                      int32_t precision, ufr_stream stream);

/* Adjoint of ufr_project_gather w.r.t. the sampled volumes and pre_sim_mlp (autograd of model.py:350-390 and
 * ray_transformer.py:268).  sim8 (P,8): the forward's `sim8` output; d_pv (P,40) from ufr_aggregate_bwd.
 * row (nullable, (RN,SN) int32): d_pv / sim8 are POOL tensors and slot (ray, s) owns row row[ray][s] of d_pv (the two-pass
 * step hands over all merged samples of a ray in one call; sim8 is only summed over, any row order).
 * grad_vol_feat[s] (NV,8,D,Hs,Ws) / grad_vol_weight[s] (NV,1,D,Hs,Ws): reference layout, written whole -- overwritten, or
 * added to when accumulate != 0 (the scatter itself goes into a channel-last record volume in `workspace`,
 * ufr_project_gather_bwd_workspace_bytes(frame) bytes = 1.33 x the volumes: nine lanes of one atomic instruction add the
 * nine values of a voxel corner as ONE L2 transaction, then one coalesced pass writes these tensors);
 * both arrays NULL: only the pre_sim_mlp gradients (accumulated into `grads`) are computed, no workspace needed.
 * `accumulate` is a bit set: UFR_GBWD_ACCUMULATE, and UFR_GBWD_WORKSPACE_ZEROED = `workspace` is all zero on entry
 * (ordered before this call on `stream`) -- the library then skips its own memset (0.9 GB at the configs[4] size).
 * The call LEAVES THE WORKSPACE ZERO: the scatter marks the groups of 8 voxels it adds into, and the pass that writes the
 * gradient tensors reads -- and zeroes again -- only those (a step's rays reach a fraction of the voxels).  A caller that
 * keeps the workspace between calls therefore zero-fills it once and passes UFR_GBWD_WORKSPACE_ZEROED from then on (a
 * call that returned an error leaves it undefined: fill it again). */
#define UFR_GBWD_ACCUMULATE 1
#define UFR_GBWD_WORKSPACE_ZEROED 2
#define UFR_GBWD_NO_PRESIM 4 /* the volume scatter only: the pre_sim_mlp gradients come from another call (both arrays NULL),
                                which a caller may put on another stream -- the two halves share no output */
size_t ufr_project_gather_bwd_workspace_bytes(const ufr_frame* frame);
int ufr_project_gather_bwd(const ufr_frame* frame, const ufr_raw_weights* raw, const ufr_raw_grads* grads,
                           const float* ray_o, int32_t ray_o_stride, const float* ray_d, const float* z, int32_t RN,
                           int32_t SN, const float* sim8, const float* d_pv, const int32_t* row,
                           float* const* grad_vol_feat, float* const* grad_vol_weight, int32_t accumulate, void* workspace,
                           int32_t precision, ufr_stream stream);

/* ------------------------------------------------------------------ the two halves of ufr_aggregate, and the sample pool
 * The fine pass of `infer` (model.py:455-473) re-evaluates all SN+PN merged samples, but a sample's gathers and
 * view-transformer output depend on its own position only: evaluating the PN NEW samples and re-using the coarse pass's
 * rows gives the same numbers (ufr_render_rays does that internally).  These entry points expose the pieces so that the
 * training step can do the same, forwards and backwards:
 *   ufr_sample_importance_pool = ufr_sample_importance_merge that also returns the new positions z_new (RN,PN), sorted
 *     along the ray, and for every merged slot its row in the pool [RN*SN coarse rows (ray-major) | RN*PN new rows]: row (RN,SN+PN) int32;
 *   ufr_view_transform / ufr_ray_transform = the view-transformer and ray-transformer halves of ufr_aggregate
 *     (token0 (P,80), radiance (P,3) | token0 rows of the RN*SN samples -> srdf (RN,SN)); `row` (nullable, (RN,SN) int32)
 *     maps slot (ray, s) to its row of token0 (NULL = slot order), so the fine pass reads the pool in place;
 *   ufr_view_transform_bwd / ufr_ray_transform_bwd = the corresponding halves of ufr_aggregate_bwd (d_token0 comes out
 *     in d_token0_a, written at the same rows `row` names -- overwritten, or added to when accumulate != 0 (the coarse
 *     pass adds its cotangents onto the fine pass's coarse rows); d_token0_b (nullable) is the second partial buffer of
 *     the interface's earlier two-sweep form: zero-filled when not accumulating, so that a + b is the gradient; pass
 *     both, or a and NULL, to ufr_view_transform_bwd). */
int ufr_sample_importance_pool(const float* weight, const float* z, const float* U2, float* z_all, float* z_new,
                               int32_t* row, int32_t RN, int32_t SN, int32_t PN, ufr_stream stream);
int ufr_view_transform(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir, int32_t P,
                       int32_t NV, float* token0, float* radiance, int32_t precision, ufr_stream stream);
size_t ufr_ray_transform_workspace_bytes(int32_t SN);
int ufr_ray_transform(const void* packed_weights, const float* token0, const int32_t* row, int32_t RN, int32_t SN,
                      float* srdf, void* workspace, int32_t precision, ufr_stream stream);
size_t ufr_ray_transform_bwd_workspace_bytes(int32_t RN, int32_t SN);   /* tape, per-ray attention state, cotangent tiles */
int ufr_ray_transform_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights, const float* token0,
                          const int32_t* row, int32_t RN, int32_t SN, const float* d_srdf, float* d_token0_a, float* d_token0_b,
                          int32_t accumulate, void* workspace, int32_t precision, ufr_stream stream);
size_t ufr_view_transform_bwd_workspace_bytes(int32_t P, int32_t NV);
int ufr_view_transform_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                           const float* x_tokens, const float* rgb, const float* dir, const float* d_token0_a,
                           const float* d_token0_b, const float* d_radiance, int32_t P, int32_t NV, float* d_pv, void* workspace,
                           int32_t precision, ufr_stream stream);
/* The same in stages, for callers that overlap them with independent work on other streams (uforecon_amd/autograd.py does):
 *   TAPE   the forward again, recording the activations into the workspace -- needs only the forward's inputs, so it can run
 *          beside the ray transformer's backward (pointers of later stages may be NULL);
 *   DGRAD  the data-gradient chain: reads the tape, d_token0_a/b, d_radiance; writes d_pv and the cotangent tiles;
 *   WGRAD  the weight-gradient contractions: read tape and cotangent tiles, add into `grads` -- nothing downstream waits for
 *          them, so they can run beside ufr_project_gather_bwd (which needs d_pv only).
 * The SAME workspace must be passed to every stage, and the stages must execute in this order (stream events are the
 * caller's business).  stages = UFR_BWD_STAGE_ALL is ufr_view_transform_bwd. */
#define UFR_BWD_STAGE_TAPE 1
#define UFR_BWD_STAGE_DGRAD 2
#define UFR_BWD_STAGE_WGRAD 4
#define UFR_BWD_STAGE_ALL 7
int ufr_view_transform_bwd_stages(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                                  const float* x_tokens, const float* rgb, const float* dir, const float* d_token0_a,
                                  const float* d_token0_b, const float* d_radiance, int32_t P, int32_t NV, float* d_pv,
                                  void* workspace, int32_t stages, int32_t precision, ufr_stream stream);
/* Training forward WITH the tape: what ufr_view_transform / ufr_ray_transform compute (same numbers), recorded into the
 * workspace of the matching backward call, whose TAPE stage is then skipped (stages = DGRAD | WGRAD): nothing is computed
 * twice.  View: the points [p0, p0 + P) of a pool of P_total points that ONE backward will walk; x_tokens / rgb / dir /
 * token0 / radiance point at the range's first row; p0 (and P, unless the range closes the pool) must be multiples of
 * ufr_view_tape_block_points(NV); workspace: ufr_view_transform_bwd_workspace_bytes(P_total, NV).  Ray: one pass;
 * workspace: ufr_ray_transform_bwd_workspace_bytes(RN, SN). */
int32_t ufr_view_tape_block_points(int32_t NV);
int ufr_view_transform_tape(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir, int32_t P,
                            int32_t NV, float* token0, float* radiance, void* workspace, int32_t p0, int32_t P_total,
                            int32_t precision, ufr_stream stream);
int ufr_ray_transform_tape(const void* packed_weights, const float* token0, const int32_t* row, int32_t RN, int32_t SN,
                           float* srdf, void* workspace, int32_t precision, ufr_stream stream);
int ufr_ray_transform_bwd_stages(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                                 const float* token0, const int32_t* row, int32_t RN, int32_t SN, const float* d_srdf,
                                 float* d_token0_a, float* d_token0_b, int32_t accumulate, void* workspace, int32_t stages,
                                 int32_t precision, ufr_stream stream);   /* the ray transformer's backward likewise */

/* ------------------------------------------------------------------ whole-path inference
 * UFORecon.infer(extract_geometry=True) (model.py:393-478) for RN rays of one frame:
 * ray gather by index, near/far / cam_ray_d.z, coarse pass, importance sampling + merge, fine
 * pass.  Depth is the ray length (model.py:821 converts to z-depth: see depth_z). */
typedef struct ufr_render_args {
  const ufr_frame* frame;     /* from ufr_frame_prepare                                   */
  const void* packed_weights; /* from ufr_weights_pack                                    */
  const ufr_raw_weights* raw; /* same parameters, reference layout (pre_sim_mlp, variance) */
  const int64_t* ray_idx;     /* (RN) indices into H*W                      (model.py:409) */
  const float* ray_d;         /* (3,H*W) batch['ray_d']                                   */
  const float* cam_ray_d;     /* (3,H*W) batch['cam_ray_d']                (model.py:424) */
  float ray_o[3];             /* batch['ray_o']                                           */
  float near_z, far_z;        /* batch['near_fars'][0,0,:]              (model.py:416-421) */
  const float* U1;            /* (SN,RN) coarse jitter uniforms         (sampler.py:42)    */
  const float* U2;            /* (PN,RN) importance uniforms; NULL if coarse_only (sampler.py:86) */
  int32_t RN, SN, PN;         /* rays, coarse samples, fine samples                        */
  int32_t coarse_only;        /* args.test_coarse_only                  (model.py:449-452) */
  int32_t precision;          /* UFR_PRECISION_*                                           */
  /* outputs (device) */
  float* depth;               /* (RN) ray-length depth                                     */
  float* depth_z;             /* (RN) depth * cam_ray_d.z (model.py:821), may be NULL      */
  float* rgb;                 /* (RN,3)                                                    */
  float* srdf;                /* (RN,S) S = SN or SN+PN, may be NULL                        */
  float* z_all;               /* (RN,S) sample distances of the returned pass, may be NULL  */
  int32_t chunk_rays;         /* rays per internal launch group (0 = library default)       */
  int32_t n_streams;          /* >1: chunks are issued round-robin on that many library-owned side streams
                                 (forked from / joined to `stream`); needs n_streams x the workspace   */
  void* workspace;            /* >= ufr_render_workspace_bytes(chunk_rays,SN,PN,NV) (x n_streams)     */
  size_t workspace_bytes;
} ufr_render_args;

size_t ufr_render_workspace_bytes(int32_t chunk_rays, int32_t SN, int32_t PN, int32_t NV);
int32_t ufr_default_chunk_rays(void);
int ufr_render_rays(const ufr_render_args* a, ufr_stream stream);

/* Per-kernel HIP-event timing of the most recent ufr_render_rays on this thread when
 * enabled (for bench.py's roofline line).  names/ms arrays of length `cap`; returns count. */
/* ---- correlation-volume construction (first step) -------------------------------------
 * Replaces, for one frame and one cascade stage, the loop body of DepthNet.forward step 2
 * (code1/encoder_utils/fmt/TransMVSNet.py:66-97): homo_warping_trans (code1/encoder_utils/fmt/module.py:329-367)
 * of every source view onto the D depth hypotheses of the reference view, similarity = mean over channels of
 * warped x reference, and -- when pixel-wise view weights are given -- the weighted aggregate over the views.
 * The (C,D,H,W) warped volume is never materialised.
 *   ref_fea [C][H][W], src_fea [NS][C][H][W], depth_values [D][H][W], view_weights [NS][H][W] (nullable): device, fp32
 *   rel_proj: HOST array [NS][12], rows of (src_proj_new @ inverse(ref_proj_new))[:3,:4] (module.py:340-342)
 *   similarity [NS][D][H][W] (nullable), aggregated [D][H][W] (nullable; needs view_weights): device outputs
 *   C in {4,8,16,32,64}; 1 <= NS <= UFR_MAX_VIEWS                                                              */
size_t ufr_correlate_workspace_bytes(int32_t C, int32_t H, int32_t W, int32_t NS);
int ufr_frustum_correlate(const float* ref_fea, const float* src_fea, const float* rel_proj, const float* depth_values,
                          const float* view_weights, int32_t C, int32_t H, int32_t W, int32_t D, int32_t NS,
                          float* similarity, float* aggregated, void* workspace, size_t workspace_bytes,
                          ufr_stream stream);

/* ---- 3-D convolutions of the frustum construction's U-Nets ---------------------
 * One layer of CostRegNet / CostRegNetWeight (code1/encoder_utils/fmt/module.py:469-543; Conv3d / Deconv3d = convolution
 * + BatchNorm + ReLU, :110-187): 3x3x3, padding 1, fp32, volumes CHANNEL-LAST [B][D][H][W][C] (a (B,1,D,H,W) tensor is
 * both layouts at once).
 *   mode   UFR_CONV3D_S1 stride 1 | UFR_CONV3D_S2 stride 2 (output ceil(n/2)) | UFR_CONV3D_T2 transposed, stride 2,
 *          output_padding 1 (output 2n)
 *   weight the layer's weight in the checkpoint's layout: conv (cout,cin,3,3,3), transposed conv (cin,cout,3,3,3)
 *   bias (cout, nullable); bn_scale / bn_shift (cout, nullable together): eval-mode BatchNorm folded to
 *          y = conv * scale + shift with scale = gamma / sqrt(var + eps), shift = beta - mean * scale; relu != 0: ReLU
 *   skip   (nullable) channel-last tensor of the output's shape, added after the activation (the U-Net's "convK + ...")
 *   out    channel-last [B][Do][Ho][Wo][cout]; with out_ncdhw != 0 the reference's (B,cout,Do,Ho,Wo) instead, and then
 *          weight2 (cout2,cin,3,3,3) / out2 (B,cout2,Do,Ho,Wo) may name a second head on the same input whose output goes
 *          through a sigmoid (CostRegNetWeight's `features` + `weights`, module.py:541-543) -- both heads in one pass.
 *   out_absmax (nullable; ABI 503; channel-last outputs with at most 16 channels -- the vector kernel): a device float the
 *          caller has zeroed, raised to max |out| (skip included): the bound ufr_conv3d_planes wants of its input.
 * Supported (cin, cout + cout2): the layers of the two networks with base_channels 8:
 *   S1: (1,8) (16,16) (32,32) (64,64) (8,1) (8,8) (8,8+1);  S2: (8,16) (16,32) (32,64);  T2: (64,32) (32,16) (16,8).    */
#define UFR_CONV3D_S1 0
#define UFR_CONV3D_S2 1
#define UFR_CONV3D_T2 2
int ufr_conv3d(const float* in, const float* weight, const float* weight2, const float* bias, const float* bn_scale,
               const float* bn_shift, const float* skip, float* out, float* out2, int32_t B, int32_t D, int32_t H,
               int32_t W, int32_t cin, int32_t cout, int32_t cout2, int32_t mode, int32_t relu, int32_t out_ncdhw,
               float* out_absmax, ufr_stream stream);
/* Backward of the plain layers (bias, no BatchNorm / activation): CostRegNetWeight, i.e. `feature_volume.cost_reg_2` -- the
 * one producer the reference trains (model.py:72-87; module.py:502-543).  B, D, H, W, cin, cout, mode describe the FORWARD
 * layer (its input extent and channels); tensors are channel-last.
 *   ufr_conv3d_bwd_data    d_in (B,D,H,W,cin) = adjoint of the layer applied to d_out (the layer's output extent, cout
 *                          channels) [+ accumulate (same shape as d_in, nullable): a gradient that arrives on two paths,
 *                          the U-Net's skip additions].  The forward kernel itself: mirrored taps for stride 1, the
 *                          transposed / strided twin for the strided / transposed layers, all on the forward weight as it
 *                          stands.  cin = 1: no `accumulate`.
 *   ufr_conv3d_bwd_weight  d_weight (the reference's parameter layout: (cout,cin,3,3,3); transposed: (cin,cout,3,3,3)) +=
 *                          sum over voxels of d_out x in per tap; d_bias (cout, nullable) += column sums of d_out.
 *                          Both are ACCUMULATED into (zero them, or keep a running sum over frames). */
int ufr_conv3d_bwd_data(const float* d_out, const float* weight, const float* accumulate, float* d_in, int32_t B, int32_t D,
                        int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t mode, ufr_stream stream);
int ufr_conv3d_bwd_weight(const float* in, const float* d_out, float* d_weight, float* d_bias, int32_t B, int32_t D, int32_t H,
                          int32_t W, int32_t cin, int32_t cout, int32_t mode, ufr_stream stream);

/* CostRegNetWeight's two heads (module.py:541-543: `features` 8 -> 8 and `weights` 8 -> 1 on the same input) in one pass:
 * d_weight (8,8,3,3,3) += d_out x in, d_weight2 (1,8,3,3,3) += d_out2 x in per tap; in (B,D,H,W,8), d_out (B,D,H,W,8),
 * d_out2 (B,D,H,W,1) channel-last.  Both are ACCUMULATED into.  (ABI 503)                                                  */
int ufr_conv3d_bwd_weight_heads(const float* in, const float* d_out, const float* d_out2, float* d_weight, float* d_weight2, int32_t B,
                                int32_t D, int32_t H, int32_t W, ufr_stream stream);

/* ---- the same layers on the 16-bit matrix cores (ABI 503) -------------------------------------------------------
 * The stride-1 layers of the U-Nets (module.py:469-543: conv2, conv4, conv6, `prob`, `features` + `weights`; and, with
 * flip != 0, their data gradients) as an implicit GEMM on
 * v_mfma_f32_16x16x32_f16 with the input brick staged once through LDS (csrc/conv3d_planes.hip).  Every fp32 product is
 * three fp16 plane products accumulated in fp32 (22 significand bits): not bit-identical to ufr_conv3d, within ~1e-6 of it.
 *   in_absmax   device pointer to ONE float >= max |in| (> 0 unless the tensor is zero): the planes' power-of-two scale.
 *               ufr_absmax measures it; a layer's out_absmax is the next layer's in_absmax.
 *   out_absmax  (nullable, channel-last outputs only) device float the caller has zeroed; raised to max |out| (skip included)
 *   flip        0: `weight` (cout,cin,3,3,3).  1: the data gradient of a stride-1 layer -- `weight` is that layer's FORWARD
 *               weight (cin of this call, cout of this call, 3,3,3) and the taps are mirrored (ufr_conv3d_bwd_data, S1).
 *   mode        UFR_CONV3D_S1; UFR_CONV3D_S2 (conv1 / conv3 / conv5, and -- on the transposed layers' forward weights as they
 *               stand -- the data gradients of conv11 / conv9 / conv7); UFR_CONV3D_T2 (conv7 / conv9 / conv11 -- `weight`
 *               (cin,cout,3,3,3) -- and, on the strided layers' forward weights, the data gradients of conv5 / conv3 / conv1):
 *               the transposed layer runs as a 2 x 2 x 2 convolution of the input grid with 8 x cout output rows, one group
 *               per output parity class
 *   workspace   ufr_conv3d_planes_workspace_bytes(cin, cout, cout2, mode) bytes (the weights' planes); 0 = combination not
 *               supported here (use ufr_conv3d).  S1: cin in {8, 16} with cout + cout2 <= 16, (32, 32), (64, 64); S2: (8, 16),
 *               (16, 32), (32, 64); T2: (16, 8), (32, 16), (64, 32).
 *   planes_ready  0: the planes are made from `weight` (/ `weight2`) by this call (three small launches).  1: `workspace` still
 *               holds the planes a call with planes_ready = 0 made from the SAME weights, flip and channel counts (frozen
 *               weights: once per checkpoint, not once per frame); the caller vouches for that.
 * bias / bn_scale / bn_shift / relu / skip / out_ncdhw / weight2 / out2 as for ufr_conv3d.                              */
size_t ufr_conv3d_planes_workspace_bytes(int32_t cin, int32_t cout, int32_t cout2, int32_t mode);
int ufr_conv3d_planes(const float* in, const float* in_absmax, const float* weight, const float* weight2, const float* bias,
                      const float* bn_scale, const float* bn_shift, const float* skip, float* out, float* out2,
                      float* out_absmax, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cout2,
                      int32_t mode, int32_t relu, int32_t out_ncdhw, int32_t flip, void* workspace, size_t workspace_bytes, int32_t planes_ready,
                      ufr_stream stream);
/* *absmax = max(*absmax, max |x[0..n)|) (device float; zero it first).  One pass at HBM speed.                          */
int ufr_absmax(const float* x, size_t n, float* absmax, ufr_stream stream);

/* ---- TSDF fusion -------------------------------------------------------------------
 * Replaces the reference's `integrate` kernel (tsdf_fusion.py:77-152, a CUDA string compiled through pycuda) and the
 * launch loop of TSDFVolume.integrate (:240-265): one depth (+ colour) observation into the (X,Y,Z) fp32 volumes,
 * z fastest.  tsdf / weight / color, depth_im [H][W], color_im [H][W] (folded b*65536+g*256+r, nullable): device;
 * dim, origin, cam_intr (3x3 row-major), cam_pose (4x4 row-major camera-to-world): host.
 * integrate_color = 0 reproduces the reference (its colour block is unreachable).                             */
int ufr_tsdf_integrate(float* tsdf, float* weight, float* color, const int32_t* dim, const float* origin,
                       float voxel_size, float trunc_margin, const float* cam_intr, const float* cam_pose,
                       const float* depth_im, const float* color_im, int32_t im_h, int32_t im_w, float obs_weight,
                       int32_t integrate_color, ufr_stream stream);

/* Pixel-wise view weights of the first cascade stage and the weighted aggregate (DepthNet.forward,
 * code1/encoder_utils/fmt/TransMVSNet.py:80-97 with PixelwiseNet :23-41), one pass over the similarity volume:
 *   view_weights[i] = max_d sigmoid(conv2(relu(bn1(conv1(relu(bn0(conv0(similarity[i]))))))))      (1x1x1 convolutions)
 *   aggregated[d]   = (sum_i similarity[i][d] view_weights[i]) / (1e-5 + sum_i view_weights[i])       (nullable)
 * similarity [NS][D][H][W], view_weights [NS][H][W], aggregated [D][H][W]: device fp32.  params: 185 device floats
 * [a0 16 | b0 16 | W1 8x16 | a1 8 | b1 8 | w2 8 | b2 1], the eval-mode BatchNorms folded by the caller: a0 = conv0.weight *
 * scale0, b0 = shift0, W1 = conv1.weight, a1 = scale1, b1 = shift1, w2 = conv2.weight, b2 = conv2.bias. */
int ufr_pixelwise_view_weights(const float* similarity, const float* params, float* view_weights, float* aggregated, int32_t NS,
                               int32_t D, int32_t H, int32_t W, ufr_stream stream);

/* ---- deformable convolution of the feature backbone -------------------------------------------------------
 * Replaces torchvision.ops.deform_conv2d as called by DCN.forward (code1/encoder_utils/fmt/dcn.py:66-80; every use in
 * FeatureNet, code1/encoder_utils/fmt/module.py:407-440, is 3x3, stride 1, padding 1, dilation 1, one offset group,
 * modulated).  input [B][C][H][W], offset [B][18][H][W] (channel 2k = dy, 2k+1 = dx of tap k), mask [B][9][H][W]
 * (nullable), weight [Cout][C][3][3], bias [Cout] (nullable), output [B][Cout][H][W]: device, fp32.
 * C a multiple of 4, <= 32; Cout in {8, 16, 32}.                                                                */
size_t ufr_deform_conv2d_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
int ufr_deform_conv2d(const float* input, const float* offset, const float* mask, const float* weight,
                      const float* bias, float* output, int32_t B, int32_t C, int32_t Cout, int32_t H, int32_t W,
                      void* workspace, size_t workspace_bytes, ufr_stream stream);

/* ---- the feature backbone on channel-last tensors (ABI 502) --------------------------------------------------
 * FeatureNet.forward (code1/encoder_utils/fmt/module.py:388-468) is convolution -> BatchNorm -> ReLU blocks (module.py:26-62),
 * 1x1 lateral connections added to a nearest-upsampled map (:452-466), and deformable layers whose offsets / masks come from
 * a plain convolution (fmt/dcn.py:41-80).  These two entry points run that chain without a layout change and without a
 * separate elementwise pass; `uforecon_amd/featurenet.py` is the plan that calls them.
 *
 * ufr_conv2d: out = [relu]( conv(input, weight) * scale + shift ) [+ up2(skip)], zero padding ksize / 2.
 *   input [B][H][W][cin] (UFR_CONV2D_IN_PLANAR: [B][3][H][W], the image); weight [cout][cin][ksize][ksize]; scale / shift
 *   [cout], nullable (eval-mode BatchNorm and / or bias folded by the caller); skip [B][Ho/2][Wo/2][cout], nullable: added
 *   after nearest 2x upsampling (F.interpolate(scale_factor=2, mode='nearest')); output [B][Ho][Wo][cout], or planar
 *   [B][cout][Ho][Wo] with UFR_CONV2D_OUT_PLANAR; sigmoid_from >= 0: sigmoid on output channels >= sigmoid_from (the mask
 *   channels 18..26 of DCN.conv_offset_mask, whose planar output IS ufr_deform_conv2d*'s offset | mask pair), < 0: none.
 *   Shapes: the layer shapes of FeatureNet (cin in {3 planar, 8, 16, 32}, ksize 1 / 3 / 5, stride 1 / 2, cout <= 32);
 *   anything else returns UFR_ERR_ARG.  fp32 products and sums (v_mfma_f32_16x16x4_f32).
 * ufr_deform_conv2d_cl: ufr_deform_conv2d on a channel-last input [B][H][W][32] (no re-layout, no workspace) with the
 *   epilogue out = [relu]((dcn + bias) * scale + shift); offset_mask [B][27][H][W] = planes 0..17 the offsets (2k = dy,
 *   2k+1 = dx of tap k), planes 18..26 the masks (after the sigmoid) -- ufr_conv2d(..., UFR_CONV2D_OUT_PLANAR, sigmoid_from
 *   = 18) of DCN.conv_offset_mask; output channel-last [B][H][W][Cout] unless UFR_CONV2D_OUT_PLANAR. */
#define UFR_CONV2D_RELU 1
#define UFR_CONV2D_IN_PLANAR 2
#define UFR_CONV2D_OUT_PLANAR 4
int ufr_conv2d(const float* input, const float* weight, const float* scale, const float* shift, const float* skip, float* output,
               int32_t B, int32_t cin, int32_t cout, int32_t H, int32_t W, int32_t ksize, int32_t stride, int32_t flags,
               int32_t sigmoid_from, ufr_stream stream);
/* The sum the smoothing convolutions of the FMT's top-down pathway read (FMT_with_pathway, code1/encoder_utils/fmt/FMT.py:
 * 226-255: smooth(interpolate(reduce(coarse), size = fine, mode = 'bilinear') + fine)): output_cl [B][2h][2w][C] =
 * bilinear 2x of reduced_cl [B][h][w][C] (align_corners = False, torch's tap order) + fine [B][C][2h][2w] (planar, as the
 * reference holds the backbone's maps).  C in {8, 16}.  The reduction and the smoothing are ufr_conv2d calls. */
int ufr_upsample_add(const float* reduced_cl, const float* fine, float* output_cl, int32_t B, int32_t C, int32_t h, int32_t w,
                     ufr_stream stream);
int ufr_deform_conv2d_cl(const float* input_cl, const float* offset_mask, const float* weight, const float* bias,
                         const float* scale, const float* shift, float* output, int32_t B, int32_t C, int32_t Cout, int32_t H,
                         int32_t W, int32_t flags, ufr_stream stream);


/* ---- feature-matching transformer layer ----------------------------------------------
 * One EncoderLayer of the FMT (code1/encoder_utils/fmt/FMT.py:82-113: linear attention with the elu+1 feature map, FMT.py:17-39,
 * d_model 32, 8 heads; out-projection + residual, LayerNorm, 32-64-32 ReLU MLP + residual, LayerNorm; dropout 0).
 * x (N,T,32): query tokens; src (N,S,32): key/value tokens (NULL = x, self-attention); out (N,T,32) (may alias neither).
 * All nn.Linear weights row-major [out][in] as in the checkpoint.  workspace >= ufr_fmt_layer_workspace_bytes(N). */
typedef struct ufr_fmt_layer_weights {
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo; /* attention.{query,key,value,out}_projection: [32][32], [32] */
  const float *w1, *b1, *w2, *b2;                     /* linear1 [64][32], [64]; linear2 [32][64], [32]             */
  const float *n1w, *n1b, *n2w, *n2b;                 /* norm1 / norm2 weight, bias [32]                            */
} ufr_fmt_layer_weights;
size_t ufr_fmt_layer_workspace_bytes(int32_t N, int32_t S);   /* S: source tokens per sample (T for self-attention) */
int ufr_fmt_layer(const ufr_fmt_layer_weights* w, const float* x, const float* src, int32_t N, int32_t T, int32_t S,
                  float* out, void* workspace, ufr_stream stream);

void ufr_profile_enable(int on);
int ufr_profile_read(const char** names, float* ms, int32_t* launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* UFR_H */
