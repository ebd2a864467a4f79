// extern "C" surface of libufr.so (include/ufr.h): argument validation, workspace carving, kernel
// sequencing on the caller's HIP stream.  No torch types, no host<->device synchronisation.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "bwd_common.h"
#include "bwd_tape.h"
#include "ufr_internal.h"
#include "ufr_layout_f16.h"

using namespace ufr;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define UFR_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(UFR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define UFR_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return fail(UFR_ERR_ARG, __VA_ARGS__); \
  } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {  // bump allocator over a caller workspace
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  float* f32(size_t n) {
    float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(n * sizeof(float));
    return r;
  }
};

// ---- optional per-kernel timing with HIP events on the caller's stream
// process-wide (the backward entry points are called from autograd's worker thread, not the caller's)
struct ProfEntry { const char* name; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::vector<ProfEntry> g_prof;
std::mutex g_prof_mu;

struct ProfScope {
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  ProfScope(const char* n, hipStream_t st) : s(st), name(n) {
    if (g_prof_on.load(std::memory_order_relaxed)) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, s);
    }
  }
  ~ProfScope() {
    if (a) {
      hipEventRecord(b, s);
      std::lock_guard<std::mutex> lock(g_prof_mu);
      g_prof.push_back({name, a, b});
    }
  }
};

__global__ void order_pe_kernel(float* __restrict__ table, int SN) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= SN * 8) return;
  int pos = i >> 3, jj = i & 7;
  // ray_transformer.py:165-173: float64 table pos / 10000^(2*(j//2)/8), sin on even / cos on odd dims
  double ang = (double)pos / pow(10000.0, 2.0 * (double)(jj / 2) / 8.0);
  table[i] = (float)((jj & 1) ? cos(ang) : sin(ang));
}

PreSim presim_of(const ufr_raw_weights* r) {
  return PreSim{r->pre_sim.w0, r->pre_sim.b0, r->pre_sim.w2, r->pre_sim.b2, r->pre_sim.w4, r->pre_sim.b4};
}

const FrameDev* frame_of(const ufr_frame* f) {
  const FrameDev* d = reinterpret_cast<const FrameDev*>(f);
  return (f && d->magic == kFrameMagic) ? d : nullptr;
}

}  // namespace

namespace ufr {
hipError_t launch_order_pe(float* table, int SN, hipStream_t s) {
  hipLaunchKernelGGL(order_pe_kernel, dim3((SN * 8 + 255) / 256), dim3(256), 0, s, table, SN);
  return hipGetLastError();
}
}  // namespace ufr

namespace {
std::atomic<int> g_matrix_precision{UFR_PRECISION_FP32};   // what UFR_PRECISION_DEFAULT resolves to

// precision argument of an entry point -> "reduced" flag of the launchers; false + error for an unknown value
bool resolve_precision(int precision, bool* lowp) {
  if (precision == UFR_PRECISION_DEFAULT) precision = g_matrix_precision.load(std::memory_order_relaxed);
  if (precision != UFR_PRECISION_FP32 && precision != UFR_PRECISION_16BIT) return false;
  *lowp = precision == UFR_PRECISION_16BIT;
  return true;
}
#define UFR_PRECISION(arg, lowp_var, who)                                                        \
  bool lowp_var = false;                                                                          \
  if (!resolve_precision(arg, &lowp_var)) return fail(UFR_ERR_ARG, "%s: unknown precision %d", who, (int)(arg))

// ---- sticky range status (include/ufr.h: ufr_status_poll): per device, created on first use (call_once: the entry points
// run on the caller's thread and on autograd's workers), never freed (process lifetime).
//   dev[0]  the bits the kernels OR into          dev[1]  generation tag, rewritten by every report-and-clear
//   host[0..1]  pinned copy of both words: written only by the D2H copies the entry points enqueue
// A report clears the reported bits on the device and starts a new generation; a copy that was already in flight then
// still delivers the OLD tag and is ignored -- without the tag it re-armed the host word with stale bits and a later,
// healthy call failed with UFR_ERR_RANGE (round-3 advisor finding).
struct StatusSlot {
  std::once_flag once;
  int rc = UFR_OK;
  int* dev = nullptr;
  volatile int* host = nullptr;
  std::atomic<int> gen{0};
  std::mutex report_mu;
  // recorded behind every report-and-clear kernel; the copies that deliver the words wait for it on THEIR stream, so a poll
  // on another stream than the one the last report was issued on cannot fetch the old tag and miss newly raised bits
  // (round-4 advisor finding: reports could be delayed or dropped across streams)
  hipEvent_t upd = nullptr;
  std::atomic<bool> upd_recorded{false};
};
constexpr int kStatusDevices = 16;
StatusSlot g_status[kStatusDevices];
constexpr int kStatusAll = 7;

__global__ void status_update_kernel(int* dev, int keep_mask, int gen) {
  atomicAnd(dev, keep_mask);
  dev[1] = gen;
}

int status_slot(StatusSlot** out) {
  int dev = 0;
  UFR_HIP(hipGetDevice(&dev));
  UFR_REQUIRE(dev >= 0 && dev < kStatusDevices, "status: device %d out of range", dev);
  StatusSlot& sl = g_status[dev];
  std::call_once(sl.once, [&sl] {
    int* h = nullptr;
    int* d = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), 2 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d), 2 * sizeof(int)) != hipSuccess || hipMemset(d, 0, 2 * sizeof(int)) != hipSuccess) {
      sl.rc = UFR_ERR_HIP;
      return;
    }
    h[0] = h[1] = 0;
    sl.host = h;
    sl.dev = d;
    if (hipEventCreateWithFlags(&sl.upd, hipEventDisableTiming) != hipSuccess) sl.upd = nullptr;
  });
  if (sl.rc != UFR_OK) return fail(sl.rc, "status: could not allocate the device's status words");
  *out = &sl;
  return UFR_OK;
}

int status_message(int bits, const char* who) {
  return fail(UFR_ERR_RANGE, "%s: range status 0x%x:%s%s%s (include/ufr.h: ufr_status_poll)", who, bits,
              (bits & 1) ? " a dense-layer input left the range of its fp16 planes (a token feature beyond the input_abs_max the weights were packed for -- ufr_weights_pack_for -- or infinite);" : "",
              (bits & 2) ? " NaN among the token / dir inputs handed to a transformer kernel;" : "",
              (bits & 4) ? " ufr_weights_pack met a parameter that is not finite (or an input bound that is not a positive finite number);" : "");
}

// what the last delivered copy says about the CURRENT generation, restricted to `mask`; reported bits are cleared on the
// device (enqueued on `s`) and a new generation starts
int status_consume(StatusSlot* sl, hipStream_t s, int mask, const char* who, int* flags_out) {
  std::lock_guard<std::mutex> lock(sl->report_mu);
  const int bits = sl->host[0], tag = sl->host[1];
  const int cur = sl->gen.load(std::memory_order_relaxed);
  const int hit = tag == cur ? bits & mask : 0;
  if (flags_out) *flags_out = tag == cur ? bits : 0;
  if (hit == 0) return UFR_OK;
  const int next = cur + 1;
  sl->gen.store(next, std::memory_order_relaxed);
  hipLaunchKernelGGL(status_update_kernel, dim3(1), dim3(1), 0, s, sl->dev, ~hit, next);
  UFR_HIP(hipGetLastError());
  if (sl->upd && hipEventRecord(sl->upd, s) == hipSuccess) sl->upd_recorded.store(true, std::memory_order_release);
  return status_message(hit, who);
}
// entry of a compute call: report (and clear) what an earlier call's copy delivered
int status_enter(StatusSlot* sl, hipStream_t s, const char* who) { return status_consume(sl, s, kStatusAll, who, nullptr); }
// exit of a compute call: deliver the words as of the end of this call's kernels
int status_fetch(StatusSlot* sl, hipStream_t s) {
  // behind the last report-and-clear, whatever stream issued it (a completed event costs nothing to wait for)
  if (sl->upd_recorded.load(std::memory_order_acquire)) UFR_HIP(hipStreamWaitEvent(s, sl->upd, 0));
  UFR_HIP(hipMemcpyAsync(const_cast<int*>(sl->host), sl->dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  return UFR_OK;
}
int status_leave(StatusSlot* sl, hipStream_t s) { return status_fetch(sl, s); }
}  // namespace

extern "C" {

int ufr_set_matrix_precision(int mode) {
  UFR_REQUIRE(mode == UFR_PRECISION_FP32 || mode == UFR_PRECISION_16BIT, "ufr_set_matrix_precision: unknown mode %d", mode);
  g_matrix_precision.store(mode, std::memory_order_relaxed);
  return UFR_OK;
}
int ufr_get_matrix_precision(void) { return g_matrix_precision.load(std::memory_order_relaxed); }

int ufr_status_poll_bits(ufr_stream stream, int32_t synchronize, int32_t mask, int32_t* flags_out) {
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (synchronize) {
    rc = status_fetch(sl, s);
    if (rc != UFR_OK) return rc;
    UFR_HIP(hipStreamSynchronize(s));
  }
  int flags = 0;
  rc = status_consume(sl, s, mask, "ufr_status_poll", &flags);
  if (flags_out) *flags_out = flags;
  if (rc == UFR_OK && !synchronize) return status_leave(sl, s);
  return rc;
}

int ufr_status_poll(ufr_stream stream, int32_t synchronize, int32_t* flags_out) {
  return ufr_status_poll_bits(stream, synchronize, kStatusAll, flags_out);
}

int ufr_version(void) { return UFR_ABI_VERSION; }
const char* ufr_last_error(void) { return g_err; }

// ------------------------------------------------------------------ weights
// [fp32 region | fp16 plane region (forward) | bf16 plane region (backward data-gradient chains) | 16-byte tail (reserved)]
size_t ufr_packed_weights_bytes(void) { return (size_t)blob_floats() * sizeof(float) + (size_t)kF16Bytes + (size_t)kBwdBytes + 16; }
size_t ufr_packed_bwd_halfwords(void) { return (size_t)kBwdHalfwords; }
size_t ufr_packed_fp32_floats(void) { return (size_t)blob_floats(); }
size_t ufr_packed_f16_halfwords(void) { return (size_t)kF16Halfwords; }
size_t ufr_packed_scale_table_offset(void) { return (size_t)scale_table_offset(); }
int ufr_packed_scale_table_entries(void) { return M_COUNT; }

int ufr_pack_plan_f16(int32_t* param_id, int32_t* elem, int32_t* plane) {
  UFR_REQUIRE(param_id && elem && plane, "ufr_pack_plan_f16: null output");
  for (int h = 0; h < kF16Halfwords; ++h) plan_entry_f16(h, &param_id[h], &elem[h], &plane[h]);
  return UFR_OK;
}

int ufr_pack_plan_bwd(int32_t* param_id, int32_t* elem, int32_t* plane) {
  UFR_REQUIRE(param_id && elem && plane, "ufr_pack_plan_bwd: null output");
  for (int h = 0; h < kBwdHalfwords; ++h) plan_entry_f16(kF16Halfwords + h, &param_id[h], &elem[h], &plane[h]);
  return UFR_OK;
}

int ufr_pack_plan(int32_t* param_id, int32_t* elem) {
  UFR_REQUIRE(param_id && elem, "ufr_pack_plan: null output");
  for (int i = 0; i < blob_floats(); ++i) {
    int p, e;
    plan_entry(i, &p, &e);
    param_id[i] = p;
    elem[i] = e;
  }
  return UFR_OK;
}

int ufr_weights_pack_for(const ufr_raw_weights* raw, void* packed, float input_abs_max, ufr_stream stream) {
  UFR_REQUIRE(raw && packed, "ufr_weights_pack: null argument");
  static_assert(sizeof(ufr_raw_weights) == sizeof(RawPtrs), "ufr_raw_weights must be P_COUNT pointers");
  RawPtrs rp;
  memcpy(&rp, raw, sizeof(rp));
  for (int i = 0; i < P_COUNT; ++i) UFR_REQUIRE(rp.p[i], "ufr_weights_pack: parameter %d is null", i);
  UFR_REQUIRE(input_abs_max > 0.f && input_abs_max <= 3.0e38f, "ufr_weights_pack_for: input_abs_max=%g must be positive and finite",
              (double)input_abs_max);
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  // every exponent of the planes is chosen on the device from the parameters themselves (prep.hip: weight_scale_kernel):
  // any finite weight fits; a non-finite one raises bit 2 of the sticky status (no synchronisation here: training re-packs
  // after every optimizer step)
  UFR_HIP(launch_pack_weights(rp, static_cast<float*>(packed), input_abs_max, sl->dev, s));
  return status_leave(sl, s);
}

int ufr_weights_fit_frame(void* packed, const ufr_frame* frame, ufr_stream stream) {
  const FrameDev* f = frame_of(frame);
  UFR_REQUIRE(packed && f, "ufr_weights_fit_frame: null weights / frame handle not prepared");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  UFR_HIP(launch_refit_weights(static_cast<float*>(packed), f->abs_max, sl->dev, s));
  return UFR_OK;
}

int ufr_weights_pack(const ufr_raw_weights* raw, void* packed, ufr_stream stream) {
  return ufr_weights_pack_for(raw, packed, kDefaultInputAbsMax, stream);
}

// ------------------------------------------------------------------ frame
static int frame_check(const ufr_frame_desc* d) {
  UFR_REQUIRE(d, "frame desc is null");
  UFR_REQUIRE(d->NV >= 2 && d->NV <= UFR_MAX_VIEWS, "NV=%d unsupported (2..%d)", d->NV, UFR_MAX_VIEWS);
  UFR_REQUIRE(d->H >= 8 && d->W >= 8 && d->H % 4 == 0 && d->W % 4 == 0, "H,W must be multiples of 4 (got %dx%d)",
              d->H, d->W);
  UFR_REQUIRE(d->source_imgs && d->depth_info && d->feat, "null frame tensor");
  UFR_REQUIRE((long long)d->H * d->W < (1 << 24) && (long long)(d->H / 4) * (d->W / 4) * 32 * (d->NV - 1) < (1ll << 32),
              "image of %dx%d is too large for the gather's 24-bit texel indices", d->H, d->W);
  // the gathers read through raw buffer descriptors whose masked taps are sent to offset 0x80000000 (kBufOut) and rely on
  // that offset lying OUTSIDE the buffer: every descriptor must stay below 2^31 bytes (gather.hip:236-288, 358)
  {
    const long long px = (long long)d->H * d->W, mpx = (long long)(d->H / 4) * (d->W / 4), lim = 1ll << 31;
    UFR_REQUIRE(px * 16 < lim && d->NV * mpx * 128 < lim && d->NV * mpx * 32 * (d->NV - 1) * 4 < lim,
                "frame of %dx%d with %d views: an image / feature-map buffer reaches 2 GiB (the gathers' zero-fill offset)",
                d->H, d->W, d->NV);
  }
  // match / volumes may be absent: such a frame only serves ufr_project_gather calls that pass sim8_in / vol24_in
  const bool has_vol = d->vol_feat[0] != nullptr;
  for (int s = 0; s < UFR_NUM_STAGES; ++s) {
    UFR_REQUIRE((d->vol_feat[s] != nullptr) == has_vol && (d->vol_weight[s] != nullptr) == has_vol,
                "volumes must be given for all stages or for none (stage %d)", s + 1);
    if (has_vol) UFR_REQUIRE(d->vol_D[s] >= 2 && d->vol_H[s] >= 2 && d->vol_W[s] >= 2, "degenerate volume (stage %d)", s + 1);
    // the gathers index texels with 24-bit multiplies and 32-bit float offsets inside one view
    if (has_vol)
      UFR_REQUIRE((long long)d->vol_D[s] * d->vol_H[s] < (1 << 24) && (long long)d->vol_W[s] * kVolCh < (1 << 24) &&
                      (long long)d->vol_D[s] * d->vol_H[s] * d->vol_W[s] * kVolCh * 4 < (1ll << 31),
                  "volume of stage %d is too large for the gather's 31-bit byte offsets (per view: < 2 GiB)", s + 1);
  }
  UFR_REQUIRE(d->source_poses && d->source_cam_pos && d->ref_cam_pos && d->w2c_row2, "null camera constants");
  return UFR_OK;
}

size_t ufr_frame_workspace_bytes(const ufr_frame_desc* d) {
  if (frame_check(d) != UFR_OK) return 0;
  Carver c(nullptr);
  const size_t h = d->H / 4, w = d->W / 4, NV = d->NV;
  c.f32(NV * h * w * 32);
  if (d->match) c.f32(NV * h * w * 32 * (NV - 1));
  c.f32(NV * (size_t)d->H * d->W * 4);
  if (d->vol_feat[0])
    for (int s = 0; s < UFR_NUM_STAGES; ++s) c.f32(NV * (size_t)d->vol_D[s] * d->vol_H[s] * d->vol_W[s] * kVolCh);
  c.f32(64);    // the frame's measured feature bound (one word; its own 256 bytes)
  return c.off;
}

int ufr_frame_prepare(const ufr_frame_desc* d, void* workspace, size_t workspace_bytes, ufr_frame* out,
                      ufr_stream stream) {
  int rc = frame_check(d);
  if (rc != UFR_OK) return rc;
  UFR_REQUIRE(workspace && out, "ufr_frame_prepare: null workspace/out");
  if (workspace_bytes < ufr_frame_workspace_bytes(d))
    return fail(UFR_ERR_WORKSPACE, "frame workspace too small: %zu < %zu", workspace_bytes, ufr_frame_workspace_bytes(d));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int NV = d->NV, h = d->H / 4, w = d->W / 4;
  Carver c(workspace);
  const bool has_vol = d->vol_feat[0] != nullptr;
  float* feat = c.f32((size_t)NV * h * w * 32);
  float* match = d->match ? c.f32((size_t)NV * h * w * 32 * (NV - 1)) : nullptr;
  float* rgb = c.f32((size_t)NV * d->H * d->W * 4);
  float* vol[UFR_NUM_STAGES] = {};
  if (has_vol)
    for (int st = 0; st < UFR_NUM_STAGES; ++st)
      vol[st] = c.f32((size_t)NV * d->vol_D[st] * d->vol_H[st] * d->vol_W[st] * kVolCh);
  unsigned* abs_max = reinterpret_cast<unsigned*>(c.f32(64));

  // the passes that re-lay the token features out also measure their magnitude (the matching features only enter as cosine
  // similarities, the colours only the blend: neither is a dense-layer input)
  UFR_HIP(hipMemsetAsync(abs_max, 0, 256, s));
  UFR_HIP(launch_nchw_to_nhwc(d->feat, feat, NV, 32, h * w, 32, s, abs_max));
  if (match) UFR_HIP(launch_nchw_to_nhwc(d->match, match, NV, 32 * (NV - 1), h * w, 32 * (NV - 1), s));
  UFR_HIP(launch_nchw_to_nhwc(d->source_imgs, rgb, NV, 3, d->H * d->W, 4, s));
  if (has_vol)
    for (int st = 0; st < UFR_NUM_STAGES; ++st)
      UFR_HIP(launch_volume_pack(d->vol_feat[st], d->vol_weight[st], vol[st], NV, d->vol_D[st] * d->vol_H[st] * d->vol_W[st], s, abs_max));

  FrameDev f;
  memset(&f, 0, sizeof(f));
  f.NV = NV; f.H = d->H; f.W = d->W; f.h = h; f.w = w; f.match_ch = 32 * (NV - 1);
  f.feat = feat; f.match = match; f.rgb = rgb; f.depth = d->depth_info;
  for (int st = 0; st < UFR_NUM_STAGES; ++st) {
    f.vol[st] = vol[st];
    if (has_vol) { f.vD[st] = d->vol_D[st]; f.vH[st] = d->vol_H[st]; f.vW[st] = d->vol_W[st]; }
  }
  for (int v = 0; v < NV; ++v) {
    memcpy(f.pose[v], d->source_poses + 16 * v, 12 * sizeof(float));
    memcpy(f.cam_pos[v], d->source_cam_pos + 3 * v, 3 * sizeof(float));
    memcpy(f.w2c_z[v], d->w2c_row2 + 4 * v, 4 * sizeof(float));
  }
  memcpy(f.ref_pos, d->ref_cam_pos, 3 * sizeof(float));
  f.vol_near = d->vol_near; f.vol_far = d->vol_far;
  f.abs_max = abs_max;
  f.magic = kFrameMagic;
  memset(out, 0, sizeof(*out));
  memcpy(out, &f, sizeof(f));
  return UFR_OK;
}

// ------------------------------------------------------------------ per-op entry points
int ufr_sample_fixed(const float* near, const float* far, const float* U, float* z_out, int32_t RN, int32_t SN,
                     ufr_stream stream) {
  UFR_REQUIRE(near && far && U && z_out, "ufr_sample_fixed: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 2, "ufr_sample_fixed: RN=%d SN=%d", RN, SN);
  UFR_HIP(launch_sample_fixed(near, far, U, RN, z_out, RN, SN, static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

int ufr_sample_importance_merge(const float* weight, const float* z, const float* U2, float* z_fine, float* z_all,
                                int32_t RN, int32_t SN, int32_t PN, ufr_stream stream) {
  UFR_REQUIRE(weight && z && U2 && z_all, "ufr_sample_importance_merge: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 2 && SN <= 256 && PN >= 1 && PN <= 256, "ufr_sample_importance_merge: RN=%d SN=%d PN=%d",
              RN, SN, PN);
  UFR_HIP(launch_importance_merge(weight, z, U2, RN, z_fine, z_all, RN, SN, PN, nullptr, nullptr,
                                  static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

int ufr_points(const float* ray_o, int32_t ray_o_stride, const float* ray_d, const float* z, float* points, int32_t RN,
               int32_t SN, ufr_stream stream) {
  UFR_REQUIRE(ray_o && ray_d && z && points, "ufr_points: null argument");
  UFR_REQUIRE(ray_o_stride == 0 || ray_o_stride == 3, "ufr_points: ray_o_stride must be 0 or 3");
  UFR_HIP(launch_points(ray_o, ray_o_stride, ray_d, z, points, RN, SN, static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

int ufr_project_gather(const ufr_frame* frame, const ufr_raw_weights* raw, const float* ray_o, int32_t ray_o_stride,
                       const float* ray_d, const float* z, int32_t RN, int32_t SN, float* x_tokens, float* rgb,
                       float* dir, float* sim8, float* vol24, float* xy, float* mask_z, const float* vol24_in,
                       const float* sim8_in, ufr_stream stream) {
  const FrameDev* f = frame_of(frame);
  UFR_REQUIRE(f, "ufr_project_gather: frame handle not prepared");
  UFR_REQUIRE(raw && ray_o && ray_d && z && x_tokens && rgb && dir, "ufr_project_gather: null argument");
  UFR_REQUIRE(ray_o_stride == 0 || ray_o_stride == 3, "ufr_project_gather: ray_o_stride must be 0 or 3");
  UFR_REQUIRE(RN > 0 && SN > 0, "ufr_project_gather: RN=%d SN=%d", RN, SN);
  UFR_REQUIRE(f->match || sim8_in, "ufr_project_gather: the frame has no matching features: sim8_in is required");
  UFR_REQUIRE(f->vol[0] || vol24_in, "ufr_project_gather: the frame has no volumes: vol24_in is required");
  ProfScope prof("gather", static_cast<hipStream_t>(stream));
  UFR_HIP(launch_gather(*f, presim_of(raw), ray_o, ray_o_stride, ray_d, z, RN, SN, x_tokens, nullptr, rgb, dir, sim8, vol24, xy,
                        mask_z, vol24_in, sim8_in, static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

size_t ufr_aggregate_workspace_bytes(int32_t RN, int32_t SN, int32_t NV) {
  (void)NV;
  Carver c(nullptr);
  c.f32((size_t)RN * SN * UFR_TOKEN_DIM);
  c.f32((size_t)SN * 8);
  return c.off;
}

static int aggregate_impl(const void* packed, const float* x_tokens, const float* x_point, const float* rgb, const float* dir, int RN, int SN,
                          int NV, float* radiance, float* srdf, float* token0, float* order_pe, bool pe_ready,
                          float* view_out, float* ray_out, bool lowp, int* status, hipStream_t s) {
  UFR_REQUIRE((unsigned long long)RN * SN * (NV + 1) * UFR_TOKEN_DIM < (1ull << 30),
              "view transformer: %d x %d points x %d tokens exceed the 2^30 token values one call addresses; chunk the points", RN, SN, NV + 1);
  {
    ProfScope ps("view_transformer", s);
    UFR_HIP(launch_view_transformer(static_cast<const float*>(packed), x_tokens, x_point, rgb, dir, RN * SN, NV, token0, radiance,
                                    view_out, lowp, status, s));
  }
  if (!pe_ready) UFR_HIP(launch_order_pe(order_pe, SN, s));
  {
    ProfScope ps("ray_transformer", s);
    UFR_HIP(launch_ray_transformer(static_cast<const float*>(packed), token0, nullptr, order_pe, RN, SN, srdf, ray_out, lowp,
                                   status, s));
  }
  return UFR_OK;
}

int ufr_aggregate(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir, int32_t RN,
                  int32_t SN, int32_t NV, float* radiance, float* srdf, void* workspace, float* view_out,
                  float* ray_out, int32_t precision, ufr_stream stream) {
  UFR_REQUIRE(packed_weights && x_tokens && rgb && dir && radiance && srdf && workspace, "ufr_aggregate: null argument");
  UFR_REQUIRE(NV >= 2 && NV <= UFR_MAX_VIEWS, "ufr_aggregate: NV=%d unsupported", NV);
  UFR_REQUIRE(RN > 0 && SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_aggregate: SN=%d must be a multiple of 16 in [16,256]", SN);
  UFR_PRECISION(precision, lowp, "ufr_aggregate");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc == UFR_OK) rc = status_enter(sl, s, "ufr_aggregate");
  if (rc != UFR_OK) return rc;
  Carver c(workspace);
  float* token0 = c.f32((size_t)RN * SN * UFR_TOKEN_DIM);
  float* order_pe = c.f32((size_t)SN * 8);
  rc = aggregate_impl(packed_weights, x_tokens, nullptr, rgb, dir, RN, SN, NV, radiance, srdf, token0, order_pe, false, view_out,
                      ray_out, lowp, sl->dev, s);
  return rc != UFR_OK ? rc : status_leave(sl, s);
}

int ufr_composite(const float* z, const float* radiance, const int32_t* row, const float* srdf, const float* variance,
                  int32_t RN, int32_t SN, float* rgb, float* depth, float* opacity, float* weight, ufr_stream stream) {
  UFR_REQUIRE(z && radiance && srdf && variance && depth, "ufr_composite: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 2 && SN <= 256, "ufr_composite: SN=%d out of range [2,256]", SN);
  ProfScope prof("composite", static_cast<hipStream_t>(stream));
  UFR_HIP(launch_composite(z, radiance, row, srdf, variance, RN, SN, rgb, depth, opacity, weight, nullptr, nullptr,
                           static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

// ------------------------------------------------------------------ backward
static int raw_and_grads(const ufr_raw_weights* raw, const ufr_raw_grads* grads, RawPtrs& rp, GradPtrs& gp, const char* who) {
  static_assert(sizeof(ufr_raw_grads) == sizeof(GradPtrs) && UFR_NUM_PARAMS == P_COUNT, "ufr_raw_grads layout");
  UFR_REQUIRE(raw && grads, "%s: null weights / grads", who);
  memcpy(&rp, raw, sizeof(rp));
  memcpy(&gp, grads, sizeof(gp));
  for (int i = 0; i < P_COUNT; ++i) UFR_REQUIRE(rp.p[i] && gp.p[i], "%s: parameter %d has a null pointer", who, i);
  return UFR_OK;
}

int ufr_composite_bwd(const float* z, const float* radiance, const int32_t* row, const float* srdf, const float* variance,
                      int32_t RN, int32_t SN, const float* d_rgb, const float* d_depth, const float* d_opacity,
                      const float* d_weight, float* d_radiance, int32_t accumulate, float* d_srdf, float* d_variance,
                      ufr_stream stream) {
  UFR_REQUIRE(z && radiance && srdf && variance && d_radiance && d_srdf && d_variance, "ufr_composite_bwd: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 2 && SN <= 256, "ufr_composite_bwd: SN=%d out of range [2,256]", SN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("composite_bwd", s);
  UFR_HIP(launch_composite_bwd(z, radiance, row, accumulate != 0, srdf, variance, RN, SN, d_rgb, d_depth, d_opacity, d_weight,
                               d_radiance, d_srdf, d_variance, s));
  return UFR_OK;
}

int ufr_render_loss(const float* rgb_c, const float* depth_c, const float* rgb_f, const float* depth_f, const float* rgb_gt,
                    const float* depth_gt, const float* near_far, int32_t nf_stride, int32_t B, int32_t RN, float weight_rgb,
                    float weight_depth, float* loss, float* d_rgb_c, float* d_depth_c, float* d_rgb_f, float* d_depth_f,
                    ufr_stream stream) {
  UFR_REQUIRE(rgb_c && depth_c && rgb_f && depth_f && rgb_gt && depth_gt && near_far && loss && d_rgb_c && d_depth_c && d_rgb_f && d_depth_f,
              "ufr_render_loss: null argument");
  UFR_REQUIRE(B > 0 && RN > 0 && nf_stride >= 2 && (long long)B * RN < (1ll << 24), "ufr_render_loss: B=%d RN=%d nf_stride=%d", B, RN, nf_stride);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("render_loss", s);
  UFR_HIP(launch_render_loss(rgb_c, depth_c, rgb_f, depth_f, rgb_gt, depth_gt, near_far, nf_stride, B, RN, weight_rgb, weight_depth,
                             loss, d_rgb_c, d_depth_c, d_rgb_f, d_depth_f, s));
  return UFR_OK;
}

// The three kernels of the view transformer's backward (bwd_tape.h) over a caller workspace:
// [tape | dY tiles | token0 scratch | radiance scratch]
struct ViewBwdWs { float *tape, *dbuf, *token0, *radiance; int blocks; };
static ViewBwdWs carve_view_bwd(Carver& c, int P, int NV) {
  ViewBwdWs w;
  w.blocks = view_tape_blocks(P, NV);
  // sized for the fp32 layouts (the 16-bit mode's are smaller)
  w.tape = c.f32((size_t)w.blocks * ViewTapeLayout<false>::block_units * 128);
  w.dbuf = c.f32((size_t)w.blocks * ViewGradLayout<false>::block_units * 128);
  w.token0 = c.f32((size_t)P * UFR_TOKEN_DIM);
  w.radiance = c.f32((size_t)P * 3);
  return w;
}
static int view_bwd_impl(const void* packed, const GradPtrs& gp, const float* x_tokens, const float* rgb, const float* dir,
                         const float* d_tok_a, const float* d_tok_b, const float* d_radiance, int P, int NV, float* d_pv,
                         const ViewBwdWs& w, bool lowp, int* status, hipStream_t s, int stages = UFR_BWD_STAGE_ALL) {
  UFR_REQUIRE((unsigned long long)P * (NV + 1) * UFR_TOKEN_DIM < (1ull << 30),
              "view transformer backward: %d points x %d tokens exceed the 2^30 token values one call addresses; chunk the points", P, NV + 1);
  const float* pk = static_cast<const float*>(packed);
  if (stages & UFR_BWD_STAGE_TAPE) {
    ProfScope p("view_tape", s);
    UFR_HIP(launch_view_tape(pk, x_tokens, rgb, dir, P, NV, w.token0, w.radiance, w.tape, lowp, status, s));
  }
  if (stages & UFR_BWD_STAGE_DGRAD) {
    ProfScope p("view_dgrad", s);
    UFR_HIP(launch_view_dgrad(pk, w.tape, rgb, d_tok_a, d_tok_b, d_radiance, P, NV, w.dbuf, d_pv, gp, lowp, s));
  }
  if (stages & UFR_BWD_STAGE_WGRAD) {
    ProfScope p("view_wgrad", s);
    UFR_HIP(launch_view_wgrad(w.tape, w.dbuf, w.blocks, gp, lowp, s));
  }
  return UFR_OK;
}

// ... and of the ray transformer's: [order code | tape | per-ray state | dY tiles | srdf scratch]
struct RayBwdWs { float *order_pe, *tape, *state, *dbuf, *srdf; int blocks; };
static RayBwdWs carve_ray_bwd(Carver& c, int RN, int SN) {
  RayBwdWs w;
  w.blocks = RN * ((SN / 16 + 1) / 2);
  w.order_pe = c.f32((size_t)SN * 8);
  w.tape = c.f32((size_t)w.blocks * RayTapeLayout<false>::block_units * 128);
  w.state = c.f32((size_t)RN * kRayStateTiles * kTileFloats);
  w.dbuf = c.f32((size_t)w.blocks * RayGradLayout<false>::block_units * 128);
  w.srdf = c.f32((size_t)RN * SN);
  return w;
}
static int ray_bwd_impl(const void* packed, const GradPtrs& gp, const float* token0, const int* row, bool accumulate,
                        const float* d_srdf, int RN, int SN, float* d_tok_a, float* d_tok_b, const RayBwdWs& w, bool lowp,
                        int* status, hipStream_t s, int stages = UFR_BWD_STAGE_ALL) {
  const float* pk = static_cast<const float*>(packed);
  if (stages & UFR_BWD_STAGE_TAPE) {
    UFR_HIP(launch_order_pe(w.order_pe, SN, s));
    ProfScope p("ray_tape", s);
    UFR_HIP(launch_ray_tape(pk, token0, row, w.order_pe, RN, SN, w.srdf, w.tape, w.state, lowp, status, s));
  }
  if (stages & UFR_BWD_STAGE_DGRAD) {
    ProfScope p("ray_dgrad", s);
    UFR_HIP(launch_ray_dgrad(pk, w.tape, w.state, d_srdf, row, accumulate, RN, SN, w.dbuf, d_tok_a, d_tok_b, gp, lowp, s));
  }
  if (stages & UFR_BWD_STAGE_WGRAD) {
    ProfScope p("ray_wgrad", s);
    UFR_HIP(launch_ray_wgrad(w.tape, w.dbuf, w.blocks, gp, lowp, s));
  }
  return UFR_OK;
}

size_t ufr_ray_transform_bwd_workspace_bytes(int32_t RN, int32_t SN) {
  if (RN <= 0 || SN < 16 || SN % 16 != 0) return 0;
  Carver c(nullptr);
  carve_ray_bwd(c, RN, SN);
  return c.off;
}

size_t ufr_view_transform_bwd_workspace_bytes(int32_t P, int32_t NV) {
  if (P <= 0 || NV < 2 || NV > UFR_MAX_VIEWS) return 0;
  Carver c(nullptr);
  carve_view_bwd(c, P, NV);
  return c.off;
}

size_t ufr_aggregate_bwd_workspace_bytes(int32_t RN, int32_t SN, int32_t NV) {
  if (RN <= 0 || SN <= 0 || NV < 2 || NV > UFR_MAX_VIEWS) return 0;
  if (SN < 16 || SN % 16 != 0) return 0;
  Carver c(nullptr);
  c.f32((size_t)RN * SN * UFR_TOKEN_DIM);
  carve_ray_bwd(c, RN, SN);
  carve_view_bwd(c, RN * SN, NV);
  return c.off;
}

int ufr_aggregate_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights, const float* x_tokens,
                      const float* rgb, const float* dir, const float* token0, int32_t RN, int32_t SN, int32_t NV,
                      const float* d_radiance, const float* d_srdf, float* d_pv, void* workspace,
                      int32_t precision, ufr_stream stream) {
  RawPtrs rp;
  GradPtrs gp;
  int rc = raw_and_grads(raw, grads, rp, gp, "ufr_aggregate_bwd");
  if (rc != UFR_OK) return rc;
  UFR_PRECISION(precision, lowp, "ufr_aggregate_bwd");
  UFR_REQUIRE(packed_weights && x_tokens && rgb && dir && token0 && d_radiance && d_srdf && d_pv && workspace, "ufr_aggregate_bwd: null argument");
  UFR_REQUIRE(NV >= 2 && NV <= UFR_MAX_VIEWS, "ufr_aggregate_bwd: NV=%d unsupported", NV);
  UFR_REQUIRE(RN > 0 && SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_aggregate_bwd: SN=%d must be a multiple of 16 in [16,256]", SN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  float* d_tok = c.f32((size_t)RN * SN * UFR_TOKEN_DIM);
  const RayBwdWs rw = carve_ray_bwd(c, RN, SN);
  const ViewBwdWs vw = carve_view_bwd(c, RN * SN, NV);
  StatusSlot* sl = nullptr;
  rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  rc = ray_bwd_impl(packed_weights, gp, token0, nullptr, false, d_srdf, RN, SN, d_tok, nullptr, rw, lowp, sl->dev, s);
  if (rc != UFR_OK) return rc;
  return view_bwd_impl(packed_weights, gp, x_tokens, rgb, dir, d_tok, nullptr, d_radiance, RN * SN, NV, d_pv, vw, lowp, sl->dev, s);
}

size_t ufr_project_gather_bwd_workspace_bytes(const ufr_frame* frame) {
  const FrameDev* f = frame_of(frame);
  return (f && f->vol[0]) ? align_up(gather_bwd_scratch_floats(*f) * sizeof(float)) : 0;
}

int ufr_project_gather_bwd(const ufr_frame* frame, const ufr_raw_weights* raw, const ufr_raw_grads* grads,
                           const float* ray_o, int32_t ray_o_stride, const float* ray_d, const float* z, int32_t RN,
                           int32_t SN, const float* sim8, const float* d_pv, const int32_t* row,
                           float* const* grad_vol_feat, float* const* grad_vol_weight, int32_t accumulate, void* workspace,
                           int32_t precision, ufr_stream stream) {
  const FrameDev* f = frame_of(frame);
  UFR_REQUIRE(f, "ufr_project_gather_bwd: frame handle not prepared");
  UFR_PRECISION(precision, lowp, "ufr_project_gather_bwd");
  RawPtrs rp;
  GradPtrs gp;
  int rc = raw_and_grads(raw, grads, rp, gp, "ufr_project_gather_bwd");
  if (rc != UFR_OK) return rc;
  UFR_REQUIRE(ray_o && ray_d && z && sim8 && d_pv, "ufr_project_gather_bwd: null argument");
  const bool scatter = grad_vol_feat != nullptr || grad_vol_weight != nullptr;   // both NULL: pre_sim_mlp gradients only
  UFR_REQUIRE(!scatter || (grad_vol_feat && grad_vol_weight && f->vol[0]),
              "ufr_project_gather_bwd: volume gradients need both pointer arrays and a frame prepared with volumes");
  UFR_REQUIRE(ray_o_stride == 0 || ray_o_stride == 3, "ufr_project_gather_bwd: ray_o_stride must be 0 or 3");
  UFR_REQUIRE(RN > 0 && SN > 0, "ufr_project_gather_bwd: RN=%d SN=%d", RN, SN);
  for (int i = 0; scatter && i < UFR_NUM_STAGES; ++i)
    UFR_REQUIRE(grad_vol_feat[i] && grad_vol_weight[i], "ufr_project_gather_bwd: null volume gradient (stage %d)", i + 1);
  UFR_REQUIRE(!scatter || workspace, "ufr_project_gather_bwd: the volume scatter needs its workspace");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (scatter) {
    ProfScope p("gather_bwd", s);
    UFR_HIP(launch_gather_bwd(*f, grad_vol_feat, grad_vol_weight, ray_o, ray_o_stride, ray_d, z, d_pv, row, RN, SN,
                              static_cast<float*>(workspace), (accumulate & UFR_GBWD_ACCUMULATE) != 0,
                              (accumulate & UFR_GBWD_WORKSPACE_ZEROED) != 0, s));
  }
  if (!(accumulate & UFR_GBWD_NO_PRESIM)) {
    ProfScope p("presim_bwd", s);
    UFR_HIP(launch_presim_bwd(rp, gp, sim8, d_pv, RN * SN, lowp, s));
  }
  return UFR_OK;
}

// ------------------------------------------------------------------ halves of aggregate / sample pool
int ufr_sample_importance_pool(const float* weight, const float* z, const float* U2, float* z_all, float* z_new,
                               int32_t* row, int32_t RN, int32_t SN, int32_t PN, ufr_stream stream) {
  UFR_REQUIRE(weight && z && U2 && z_all && z_new && row, "ufr_sample_importance_pool: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 2 && SN <= 256 && PN >= 1 && PN <= 256, "ufr_sample_importance_pool: RN=%d SN=%d PN=%d", RN, SN, PN);
  UFR_HIP(launch_importance_merge(weight, z, U2, RN, nullptr, z_all, RN, SN, PN, z_new, row, static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

int ufr_view_transform(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir, int32_t P,
                       int32_t NV, float* token0, float* radiance, int32_t precision, ufr_stream stream) {
  UFR_REQUIRE(packed_weights && x_tokens && rgb && dir && token0 && radiance, "ufr_view_transform: null argument");
  UFR_REQUIRE(NV >= 2 && NV <= UFR_MAX_VIEWS && P > 0, "ufr_view_transform: P=%d NV=%d", P, NV);
  UFR_PRECISION(precision, lowp, "ufr_view_transform");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc == UFR_OK) rc = status_enter(sl, s, "ufr_view_transform");
  if (rc != UFR_OK) return rc;
  {
    ProfScope p("view_transformer", s);
    UFR_REQUIRE((unsigned long long)P * (NV + 1) * UFR_TOKEN_DIM < (1ull << 30),
                "ufr_view_transform: %d points x %d tokens exceed the 2^30 token values one call addresses; chunk the points", P, NV + 1);
    UFR_HIP(launch_view_transformer(static_cast<const float*>(packed_weights), x_tokens, nullptr, rgb, dir, P, NV, token0, radiance,
                                    nullptr, lowp, sl->dev, s));
  }
  return status_leave(sl, s);
}

// The forward WITH the tape (training): the TAPE instantiation of the forward kernel writes token0 / radiance like
// ufr_view_transform and records the activations into the backward's workspace, so that the backward starts at its
// data-gradient stage -- nothing is computed twice.
int32_t ufr_view_tape_block_points(int32_t NV) {
  return (NV >= 2 && NV <= UFR_MAX_VIEWS) ? (16 / (NV + 1)) * kBlockCols : 0;
}

int ufr_view_transform_tape(const void* packed_weights, const float* x_tokens, const float* rgb, const float* dir, int32_t P,
                            int32_t NV, float* token0, float* radiance, void* workspace, int32_t p0, int32_t P_total,
                            int32_t precision, ufr_stream stream) {
  UFR_REQUIRE(packed_weights && x_tokens && rgb && dir && token0 && radiance && workspace, "ufr_view_transform_tape: null argument");
  UFR_REQUIRE(NV >= 2 && NV <= UFR_MAX_VIEWS && P > 0 && p0 >= 0 && p0 + P <= P_total, "ufr_view_transform_tape: P=%d p0=%d P_total=%d NV=%d",
              P, p0, P_total, NV);
  const int ppw = ufr_view_tape_block_points(NV);
  UFR_REQUIRE(p0 % ppw == 0 && (p0 + P == P_total || P % ppw == 0),
              "ufr_view_transform_tape: the point range [%d, %d) must start and (unless it closes the pool) end on a multiple of %d points",
              p0, p0 + P, ppw);
  UFR_REQUIRE((unsigned long long)P_total * (NV + 1) * UFR_TOKEN_DIM < (1ull << 30),
              "ufr_view_transform_tape: %d points x %d tokens exceed the 2^30 token values one backward addresses", P_total, NV + 1);
  UFR_PRECISION(precision, lowp, "ufr_view_transform_tape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  const ViewBwdWs vw = carve_view_bwd(c, P_total, NV);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc == UFR_OK) rc = status_enter(sl, s, "ufr_view_transform_tape");
  if (rc != UFR_OK) return rc;
  const size_t blk_floats = (size_t)(lowp ? ViewTapeLayout<true>::block_units : ViewTapeLayout<false>::block_units) * 128;
  {
    ProfScope p("view_tape", s);
    UFR_HIP(launch_view_tape(static_cast<const float*>(packed_weights), x_tokens, rgb, dir, P, NV, token0, radiance,
                             vw.tape + (size_t)(p0 / ppw) * blk_floats, lowp, sl->dev, s));
  }
  return status_leave(sl, s);
}

int ufr_ray_transform_tape(const void* packed_weights, const float* token0, const int32_t* row, int32_t RN, int32_t SN,
                           float* srdf, void* workspace, int32_t precision, ufr_stream stream) {
  UFR_REQUIRE(packed_weights && token0 && srdf && workspace, "ufr_ray_transform_tape: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_ray_transform_tape: SN=%d must be a multiple of 16 in [16,256]", SN);
  UFR_PRECISION(precision, lowp, "ufr_ray_transform_tape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  const RayBwdWs rw = carve_ray_bwd(c, RN, SN);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc == UFR_OK) rc = status_enter(sl, s, "ufr_ray_transform_tape");
  if (rc != UFR_OK) return rc;
  UFR_HIP(launch_order_pe(rw.order_pe, SN, s));
  {
    ProfScope p("ray_tape", s);
    UFR_HIP(launch_ray_tape(static_cast<const float*>(packed_weights), token0, row, rw.order_pe, RN, SN, srdf, rw.tape, rw.state, lowp,
                            sl->dev, s));
  }
  return status_leave(sl, s);
}

size_t ufr_ray_transform_workspace_bytes(int32_t SN) { return align_up((size_t)(SN > 0 ? SN : 1) * 8 * sizeof(float)); }

int ufr_ray_transform(const void* packed_weights, const float* token0, const int32_t* row, int32_t RN, int32_t SN,
                      float* srdf, void* workspace, int32_t precision, ufr_stream stream) {
  UFR_REQUIRE(packed_weights && token0 && srdf && workspace, "ufr_ray_transform: null argument");
  UFR_REQUIRE(RN > 0 && SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_ray_transform: SN=%d must be a multiple of 16 in [16,256]", SN);
  UFR_PRECISION(precision, lowp, "ufr_ray_transform");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int rc = status_slot(&sl);
  if (rc == UFR_OK) rc = status_enter(sl, s, "ufr_ray_transform");
  if (rc != UFR_OK) return rc;
  float* order_pe = static_cast<float*>(workspace);
  UFR_HIP(launch_order_pe(order_pe, SN, s));
  {
    ProfScope p("ray_transformer", s);
    UFR_HIP(launch_ray_transformer(static_cast<const float*>(packed_weights), token0, row, order_pe, RN, SN, srdf, nullptr,
                                   lowp, sl->dev, s));
  }
  return status_leave(sl, s);
}

int ufr_ray_transform_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights, const float* token0,
                          const int32_t* row, int32_t RN, int32_t SN, const float* d_srdf, float* d_token0_a, float* d_token0_b,
                          int32_t accumulate, void* workspace, int32_t precision, ufr_stream stream) {
  return ufr_ray_transform_bwd_stages(raw, grads, packed_weights, token0, row, RN, SN, d_srdf, d_token0_a, d_token0_b, accumulate,
                                      workspace, UFR_BWD_STAGE_ALL, precision, stream);
}

int ufr_ray_transform_bwd_stages(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                                 const float* token0, const int32_t* row, int32_t RN, int32_t SN, const float* d_srdf,
                                 float* d_token0_a, float* d_token0_b, int32_t accumulate, void* workspace, int32_t stages,
                                 int32_t precision, ufr_stream stream) {
  RawPtrs rp;
  GradPtrs gp;
  int rc = raw_and_grads(raw, grads, rp, gp, "ufr_ray_transform_bwd");
  if (rc != UFR_OK) return rc;
  UFR_PRECISION(precision, lowp, "ufr_ray_transform_bwd");
  UFR_REQUIRE(stages > 0 && (stages & ~UFR_BWD_STAGE_ALL) == 0, "ufr_ray_transform_bwd_stages: stages=%d", stages);
  UFR_REQUIRE(packed_weights && workspace, "ufr_ray_transform_bwd: null argument");
  UFR_REQUIRE(!(stages & UFR_BWD_STAGE_TAPE) || token0, "ufr_ray_transform_bwd: the tape stage needs token0");
  UFR_REQUIRE(!(stages & UFR_BWD_STAGE_DGRAD) || (d_srdf && d_token0_a), "ufr_ray_transform_bwd: the data-gradient stage needs d_srdf, d_token0_a");
  UFR_REQUIRE(RN > 0 && SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_ray_transform_bwd: SN=%d must be a multiple of 16 in [16,256]", SN);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  const RayBwdWs rw = carve_ray_bwd(c, RN, SN);
  StatusSlot* sl = nullptr;
  rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  return ray_bwd_impl(packed_weights, gp, token0, row, accumulate != 0, d_srdf, RN, SN, d_token0_a, d_token0_b, rw, lowp, sl->dev, s,
                      stages);
}

int ufr_view_transform_bwd(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                           const float* x_tokens, const float* rgb, const float* dir, const float* d_token0_a,
                           const float* d_token0_b, const float* d_radiance, int32_t P, int32_t NV, float* d_pv, void* workspace,
                           int32_t precision, ufr_stream stream) {
  return ufr_view_transform_bwd_stages(raw, grads, packed_weights, x_tokens, rgb, dir, d_token0_a, d_token0_b, d_radiance, P, NV, d_pv,
                                       workspace, UFR_BWD_STAGE_ALL, precision, stream);
}

int ufr_view_transform_bwd_stages(const ufr_raw_weights* raw, const ufr_raw_grads* grads, const void* packed_weights,
                                  const float* x_tokens, const float* rgb, const float* dir, const float* d_token0_a,
                                  const float* d_token0_b, const float* d_radiance, int32_t P, int32_t NV, float* d_pv,
                                  void* workspace, int32_t stages, int32_t precision, ufr_stream stream) {
  RawPtrs rp;
  GradPtrs gp;
  int rc = raw_and_grads(raw, grads, rp, gp, "ufr_view_transform_bwd");
  if (rc != UFR_OK) return rc;
  UFR_PRECISION(precision, lowp, "ufr_view_transform_bwd");
  UFR_REQUIRE(stages > 0 && (stages & ~UFR_BWD_STAGE_ALL) == 0, "ufr_view_transform_bwd_stages: stages=%d", stages);
  UFR_REQUIRE(packed_weights && workspace, "ufr_view_transform_bwd: null argument");
  UFR_REQUIRE(!(stages & UFR_BWD_STAGE_TAPE) || (x_tokens && rgb && dir), "ufr_view_transform_bwd: the tape stage needs x_tokens, rgb, dir");
  UFR_REQUIRE(!(stages & UFR_BWD_STAGE_DGRAD) || (rgb && d_token0_a && d_radiance && d_pv),
              "ufr_view_transform_bwd: the data-gradient stage needs rgb, d_token0_a, d_radiance, d_pv");
  UFR_REQUIRE(NV >= 2 && NV <= UFR_MAX_VIEWS && P > 0, "ufr_view_transform_bwd: P=%d NV=%d", P, NV);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  const ViewBwdWs vw = carve_view_bwd(c, P, NV);
  StatusSlot* sl = nullptr;
  rc = status_slot(&sl);
  if (rc != UFR_OK) return rc;
  return view_bwd_impl(packed_weights, gp, x_tokens, rgb, dir, d_token0_a, d_token0_b, d_radiance, P, NV, d_pv, vw, lowp, sl->dev, s, stages);
}

// ------------------------------------------------------------------ whole-path inference
int32_t ufr_default_chunk_rays(void) { return 4096; }

namespace {
struct RenderWs {
  float *ray_o, *rd, *near, *far, *camz, *z1, *w1, *srdf1, *depth1, *rgb1, *z2, *srdf2, *rad, *x, *xp, *rgbm, *dir, *token0,
      *pe1, *pe2, *z_new;
  int* row;  // merged slot -> row of the [coarse | new] evaluation pool (token0, rad)
  size_t bytes;
};
RenderWs carve_render(void* ws, int R, int SN, int PN, int NV) {
  // Smax: samples per ray in the evaluation pool (coarse + new); Sg: points per ray one gather / view-transformer
  // launch handles (the fine pass evaluates only its PN new points)
  const size_t S2 = (size_t)SN + PN, Smax = S2 > (size_t)SN ? S2 : SN, Sg = (size_t)(SN > PN ? SN : PN);
  Carver c(ws);
  RenderWs r;
  r.ray_o = c.f32(4);
  r.rd = c.f32((size_t)R * 3);
  r.near = c.f32(R);
  r.far = c.f32(R);
  r.camz = c.f32(R);
  r.z1 = c.f32((size_t)R * SN);
  r.w1 = c.f32((size_t)R * SN);
  r.srdf1 = c.f32((size_t)R * SN);
  r.depth1 = c.f32(R);
  r.rgb1 = c.f32((size_t)R * 3);
  r.z2 = c.f32((size_t)R * S2);
  r.srdf2 = c.f32((size_t)R * S2);
  r.rad = c.f32((size_t)R * Smax * 3);
  r.x = c.f32((size_t)R * Sg * NV * kViewCols);   // compact token layout (ufr_internal.h): per-view columns ...
  r.xp = c.f32((size_t)R * Sg * kPointCols);      // ... and the per-point ones, once
  r.rgbm = c.f32((size_t)R * Sg * NV * 4);
  r.dir = c.f32((size_t)R * Sg * NV * 4);
  r.z_new = c.f32((size_t)R * (PN > 0 ? PN : 1));
  r.row = reinterpret_cast<int*>(c.f32((size_t)R * S2));
  r.token0 = c.f32((size_t)R * Smax * UFR_TOKEN_DIM);
  r.pe1 = c.f32((size_t)SN * 8);
  r.pe2 = c.f32(S2 * 8);
  r.bytes = c.off;
  return r;
}
}  // namespace

size_t ufr_render_workspace_bytes(int32_t chunk_rays, int32_t SN, int32_t PN, int32_t NV) {
  if (chunk_rays <= 0) chunk_rays = ufr_default_chunk_rays();
  return carve_render(nullptr, chunk_rays, SN, PN, NV).bytes;
}

namespace {
// Side streams: consecutive ray chunks are independent, so they are issued round-robin on a few
// library-owned HIP streams -- the gather kernel of one chunk (L2/latency-bound, no MFMA) then runs
// beside the transformer kernels of another (MFMA-bound) instead of in front of them.
constexpr int kMaxLanes = 4;
struct SidePool {
  hipStream_t s[kMaxLanes] = {};
  hipEvent_t fork = nullptr, join[kMaxLanes] = {};
  int n = 0;
};
constexpr int kMaxDevices = 16;
thread_local SidePool g_side_by_device[kMaxDevices];   // streams and events belong to the device they were created on

int side_pool_get(int n, SidePool** out) {
  int dev = 0;
  UFR_HIP(hipGetDevice(&dev));
  UFR_REQUIRE(dev >= 0 && dev < kMaxDevices, "side streams: device %d out of range", dev);
  SidePool& p = g_side_by_device[dev];
  if (!p.fork) UFR_HIP(hipEventCreateWithFlags(&p.fork, hipEventDisableTiming));
  for (; p.n < n; ++p.n) {
    // the ray path's chunks at the device's HIGHEST stream priority: whatever the caller runs beside a frame (the next
    // frame's producers on a default- or low-priority stream, uforecon_amd/evalset.py) then fills the gaps the ray kernels
    // leave instead of taking turns with them
    int least = 0, greatest = 0;
    UFR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // (measured round 6, configs[2] on one GPU, 24 frames, two runs each: 137.5 / 139.4 ms per frame against 140.5 / 140.6
    // with flat priorities and 145.4 / 143.8 without the overlap: tools/dev/prio_ab.sh)
    UFR_HIP(hipStreamCreateWithPriority(&p.s[p.n], hipStreamNonBlocking, greatest));
    UFR_HIP(hipEventCreateWithFlags(&p.join[p.n], hipEventDisableTiming));
  }
  *out = &p;
  return UFR_OK;
}

// one chunk of R rays starting at r0, entirely on stream s with workspace w
int render_chunk(const ufr_render_args* a, const FrameDev* f, const RenderWs& w, bool& pe_ready, int r0, int R, bool lowp,
                 int* status, hipStream_t s) {
  const int RN = a->RN, SN = a->SN, PN = a->coarse_only ? 0 : a->PN, NV = f->NV;
  const PreSim ps = presim_of(a->raw);
  const int S2 = SN + PN, HW = f->H * f->W;
  {
    ProfScope p("sampler", s);
    UFR_HIP(launch_ray_setup(a->ray_idx + r0, a->ray_d, a->cam_ray_d, HW, a->near_z, a->far_z, R, w.rd, w.near, w.far,
                             w.camz, a->ray_o, w.ray_o, s));
    UFR_HIP(launch_sample_fixed(w.near, w.far, a->U1 + r0, RN, w.z1, R, SN, s));
    if (!pe_ready) {
      UFR_HIP(launch_order_pe(w.pe1, SN, s));
      if (!a->coarse_only) UFR_HIP(launch_order_pe(w.pe2, S2, s));
      pe_ready = true;
    }
  }
  // ---- coarse pass (model.py:445)
  {
    ProfScope p("gather", s);
    UFR_HIP(launch_gather(*f, ps, w.ray_o, 0, w.rd, w.z1, R, SN, w.x, w.xp, w.rgbm, w.dir, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s));
  }
  int rc = aggregate_impl(a->packed_weights, w.x, w.xp, w.rgbm, w.dir, R, SN, NV, w.rad, w.srdf1, w.token0, w.pe1, true,
                          nullptr, nullptr, lowp, status, s);
  if (rc != UFR_OK) return rc;
  const bool last = a->coarse_only != 0;
  {
    ProfScope p("composite", s);
    UFR_HIP(launch_composite(w.z1, w.rad, nullptr, w.srdf1, a->raw->variance, R, SN, last ? a->rgb + 3 * (size_t)r0 : w.rgb1,
                             last ? a->depth + r0 : w.depth1, nullptr, w.w1, w.camz,
                             (last && a->depth_z) ? a->depth_z + r0 : nullptr, s));
  }
  if (last) {
    if (a->srdf) UFR_HIP(hipMemcpyAsync(a->srdf + (size_t)r0 * SN, w.srdf1, (size_t)R * SN * 4, hipMemcpyDeviceToDevice, s));
    if (a->z_all) UFR_HIP(hipMemcpyAsync(a->z_all + (size_t)r0 * SN, w.z1, (size_t)R * SN * 4, hipMemcpyDeviceToDevice, s));
    return UFR_OK;
  }
  // ---- importance sampling + merge (model.py:455-470), fine pass (model.py:472).  The reference re-evaluates
  // all SN+PN merged samples; a sample's gathers and view-transformer output depend on its own position only,
  // so the SN coarse evaluations (token0, radiance: rows [0, R*SN) of the pool) are kept and only the PN new
  // points go through gather + view transformer (rows [R*SN, R*(SN+PN))).  The ray transformer and the
  // compositor, which do couple the samples of a ray, run over all SN+PN through the slot -> row table.
  {
    ProfScope p("sampler", s);
    UFR_HIP(launch_importance_merge(w.w1, w.z1, a->U2 + r0, RN, nullptr, w.z2, R, SN, PN, w.z_new, w.row, s));
  }
  {
    ProfScope p("gather", s);
    UFR_HIP(launch_gather(*f, ps, w.ray_o, 0, w.rd, w.z_new, R, PN, w.x, w.xp, w.rgbm, w.dir, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s));
  }
  {
    ProfScope p("view_transformer", s);
    UFR_HIP(launch_view_transformer(static_cast<const float*>(a->packed_weights), w.x, w.xp, w.rgbm, w.dir, R * PN, NV,
                                    w.token0 + (size_t)R * SN * UFR_TOKEN_DIM, w.rad + (size_t)R * SN * 3, nullptr, lowp,
                                    status, s));
  }
  {
    ProfScope p("ray_transformer", s);
    UFR_HIP(launch_ray_transformer(static_cast<const float*>(a->packed_weights), w.token0, w.row, w.pe2, R, S2, w.srdf2,
                                   nullptr, lowp, status, s));
  }
  {
    ProfScope p("composite", s);
    UFR_HIP(launch_composite(w.z2, w.rad, w.row, w.srdf2, a->raw->variance, R, S2, a->rgb + 3 * (size_t)r0, a->depth + r0,
                             nullptr, nullptr, w.camz, a->depth_z ? a->depth_z + r0 : nullptr, s));
  }
  if (a->srdf) UFR_HIP(hipMemcpyAsync(a->srdf + (size_t)r0 * S2, w.srdf2, (size_t)R * S2 * 4, hipMemcpyDeviceToDevice, s));
  if (a->z_all) UFR_HIP(hipMemcpyAsync(a->z_all + (size_t)r0 * S2, w.z2, (size_t)R * S2 * 4, hipMemcpyDeviceToDevice, s));
  return UFR_OK;
}
}  // namespace

int ufr_render_rays(const ufr_render_args* a, ufr_stream stream) {
  UFR_REQUIRE(a, "ufr_render_rays: null args");
  const FrameDev* f = frame_of(a->frame);
  UFR_REQUIRE(f, "ufr_render_rays: frame handle not prepared");
  UFR_REQUIRE(f->match && f->vol[0], "ufr_render_rays: the frame was prepared without matching features / volumes");
  UFR_REQUIRE(a->packed_weights && a->raw && a->ray_idx && a->ray_d && a->U1 && a->depth && a->rgb && a->workspace,
              "ufr_render_rays: null argument");
  UFR_REQUIRE(a->coarse_only || a->U2, "ufr_render_rays: U2 required unless coarse_only");
  const int RN = a->RN, SN = a->SN, PN = a->coarse_only ? 0 : a->PN, NV = f->NV;
  UFR_REQUIRE(RN > 0, "ufr_render_rays: RN=%d", RN);
  UFR_REQUIRE(SN >= 16 && SN % 16 == 0 && SN <= 256, "ufr_render_rays: coarse samples %d must be a multiple of 16 in [16,256]", SN);
  UFR_REQUIRE(a->coarse_only || (PN >= 16 && (SN + PN) % 16 == 0 && SN + PN <= 256 && PN <= 256),
              "ufr_render_rays: fine samples %d unsupported", PN);
  const int chunk = a->chunk_rays > 0 ? a->chunk_rays : ufr_default_chunk_rays();
  // the transformer kernels address a launch's buffers with 32-bit offsets: validate the caller's chunk size up front, not
  // after the gather and earlier chunks were enqueued (the default 4096 rays is 160 x below the limit)
  UFR_REQUIRE((unsigned long long)chunk * (SN > PN ? SN : PN) * (NV + 1) * UFR_TOKEN_DIM < (1ull << 30),
              "ufr_render_rays: chunk_rays=%d x %d samples x %d tokens exceeds the 2^30 token values one launch addresses; use a "
              "smaller chunk", chunk, SN > PN ? SN : PN, NV + 1);
  const size_t need = ufr_render_workspace_bytes(chunk, SN, PN, NV);
  if (a->workspace_bytes < need) return fail(UFR_ERR_WORKSPACE, "render workspace too small: %zu < %zu", a->workspace_bytes, need);
  UFR_PRECISION(a->precision, lowp, "ufr_render_rays");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatusSlot* sl = nullptr;
  int src = status_slot(&sl);
  if (src == UFR_OK) src = status_enter(sl, s, "ufr_render_rays");
  if (src != UFR_OK) return src;
  // the planes' activation exponents follow THIS frame's measured feature bound (a no-op kernel unless the frame exceeds
  // what the table serves; ufr_weights_fit_frame) -- on the caller's stream, in front of the fork to the side streams
  UFR_HIP(launch_refit_weights(static_cast<float*>(const_cast<void*>(a->packed_weights)), f->abs_max, sl->dev, s));
  int lanes = a->n_streams > 1 ? a->n_streams : 1;
  if (lanes > kMaxLanes) lanes = kMaxLanes;
  if ((size_t)lanes * need > a->workspace_bytes) lanes = (int)(a->workspace_bytes / need);  // one workspace per lane
  // a small ray set (one rank's tile of a frame split over 8 GPUs) is cut finer so that every side stream still
  // gets >= 4 chunks: the last round of a round-robin over few chunks otherwise leaves streams idle
  int eff_chunk = chunk;
  if (lanes > 1) {
    // ... in multiples of 2048 rays: 512 resident workgroup slots x 4 rays fill whole rounds of the ray transformer
    int target = (RN + 4 * lanes - 1) / (4 * lanes) / 2048 * 2048;
    if (target < 2048) target = 2048;
    if (target < eff_chunk) eff_chunk = target;
  }
  const int n_chunks = (RN + eff_chunk - 1) / eff_chunk;
  if (lanes > n_chunks) lanes = n_chunks;

  if (lanes <= 1) {
    RenderWs w = carve_render(a->workspace, chunk, SN, PN, NV);
    bool pe_ready = false;
    for (int r0 = 0; r0 < RN; r0 += eff_chunk) {
      int rc = render_chunk(a, f, w, pe_ready, r0, (RN - r0) < eff_chunk ? (RN - r0) : eff_chunk, lowp, sl->dev, s);
      if (rc != UFR_OK) return rc;
    }
    return status_leave(sl, s);
  }
  SidePool* side = nullptr;
  int rc = side_pool_get(lanes, &side);
  if (rc != UFR_OK) return rc;
  UFR_HIP(hipEventRecord(side->fork, s));
  RenderWs w[kMaxLanes];
  bool pe_ready[kMaxLanes] = {};
  for (int l = 0; l < lanes; ++l) {
    w[l] = carve_render(static_cast<char*>(a->workspace) + (size_t)l * need, chunk, SN, PN, NV);
    UFR_HIP(hipStreamWaitEvent(side->s[l], side->fork, 0));
  }
  for (int c = 0; c < n_chunks && rc == UFR_OK; ++c) {
    const int l = c % lanes, r0 = c * eff_chunk;
    rc = render_chunk(a, f, w[l], pe_ready[l], r0, (RN - r0) < eff_chunk ? (RN - r0) : eff_chunk, lowp, sl->dev, side->s[l]);
  }
  // join even when a chunk failed: the caller's stream must not run ahead of (and its allocator must not recycle the
  // workspace under) side-stream kernels that were already enqueued
  for (int l = 0; l < lanes; ++l) {
    hipError_t e = hipEventRecord(side->join[l], side->s[l]);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, side->join[l], 0);
    if (e != hipSuccess) hipStreamSynchronize(side->s[l]);
  }
  return rc != UFR_OK ? rc : status_leave(sl, s);
}

// ------------------------------------------------------------------ correlation-volume construction
size_t ufr_correlate_workspace_bytes(int32_t C, int32_t H, int32_t W, int32_t NS) {
  Carver c(nullptr);
  c.f32((size_t)H * W * C);
  c.f32((size_t)NS * H * W * C);
  return c.off;
}

int ufr_frustum_correlate(const float* ref_fea, const float* src_fea, const float* rel_proj, const float* depth_values,
                          const float* view_weights, int32_t C, int32_t H, int32_t W, int32_t D, int32_t NS,
                          float* similarity, float* aggregated, void* workspace, size_t workspace_bytes,
                          ufr_stream stream) {
  UFR_REQUIRE(ref_fea && src_fea && rel_proj && depth_values && workspace, "ufr_frustum_correlate: null argument");
  UFR_REQUIRE(similarity || aggregated, "ufr_frustum_correlate: no output requested");
  UFR_REQUIRE(!aggregated || view_weights, "ufr_frustum_correlate: aggregated output needs view_weights");
  UFR_REQUIRE(C == 4 || C == 8 || C == 16 || C == 32 || C == 64, "ufr_frustum_correlate: C=%d unsupported (4,8,16,32,64)", C);
  UFR_REQUIRE(NS >= 1 && NS <= UFR_MAX_VIEWS, "ufr_frustum_correlate: NS=%d unsupported (1..%d)", NS, UFR_MAX_VIEWS);
  UFR_REQUIRE(H >= 2 && W >= 2 && D >= 1, "ufr_frustum_correlate: H=%d W=%d D=%d", H, W, D);
  const size_t need = ufr_correlate_workspace_bytes(C, H, W, NS);
  if (workspace_bytes < need) return fail(UFR_ERR_WORKSPACE, "correlate workspace too small: %zu < %zu", workspace_bytes, need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  float* ref_cl = c.f32((size_t)H * W * C);
  float* src_cl = c.f32((size_t)NS * H * W * C);
  ProfScope p("correlate", s);
  UFR_HIP(launch_chw_to_hwc(ref_fea, ref_cl, 1, C, H * W, s));
  UFR_HIP(launch_chw_to_hwc(src_fea, src_cl, NS, C, H * W, s));
  UFR_HIP(launch_correlate(ref_cl, src_cl, rel_proj, NS, depth_values, view_weights, similarity, aggregated, C, H, W, D, s));
  return UFR_OK;
}

// ------------------------------------------------------------------ 3-D convolutions of the frustum U-Nets
int ufr_conv3d(const float* in, const float* weight, const float* weight2, const float* bias, const float* bn_scale,
               const float* bn_shift, const float* skip, float* out, float* out2, int32_t B, int32_t D, int32_t H,
               int32_t W, int32_t cin, int32_t cout, int32_t cout2, int32_t mode, int32_t relu, int32_t out_ncdhw,
               float* out_absmax, ufr_stream stream) {
  UFR_REQUIRE(in && weight && out, "ufr_conv3d: null argument");
  UFR_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "ufr_conv3d: B=%d D=%d H=%d W=%d", B, D, H, W);
  UFR_REQUIRE(mode == UFR_CONV3D_S1 || mode == UFR_CONV3D_S2 || mode == UFR_CONV3D_T2, "ufr_conv3d: unknown mode %d", mode);
  UFR_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ufr_conv3d: bn_scale and bn_shift go together");
  UFR_REQUIRE(cout2 == 0 || (weight2 && out2 && out_ncdhw && mode == UFR_CONV3D_S1),
              "ufr_conv3d: a second head needs weight2, out2, out_ncdhw and stride 1");
  UFR_REQUIRE(out_ncdhw || (cout % 4 == 0 && cout2 == 0), "ufr_conv3d: channel-last outputs need cout %% 4 == 0 (got %d)", cout);
  UFR_REQUIRE(!skip || !out_ncdhw, "ufr_conv3d: skip is a channel-last tensor; not with out_ncdhw");
  UFR_REQUIRE((long long)B * D * H * W * (cin > cout ? cin : cout) < (1ll << 40), "ufr_conv3d: volume too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("conv3d", s);
  UFR_REQUIRE(!out_absmax || (!out_ncdhw && cout <= 16), "ufr_conv3d: out_absmax goes with channel-last outputs of at most 16 channels");
  const hipError_t e = launch_conv3d(in, weight, weight2, bias, bn_scale, bn_shift, skip, out, out2, B, D, H, W, cin, cout,
                                     cout2, mode, relu, out_ncdhw, s, 0, out_absmax);
  if (e == hipErrorInvalidValue)
    return fail(UFR_ERR_ARG, "ufr_conv3d: (cin %d, cout %d+%d, mode %d) is not a layer of CostRegNet / CostRegNetWeight", cin,
                cout, cout2, mode);
  UFR_HIP(e);
  return UFR_OK;
}

// the stride-1 8 / 16-channel layers on the 16-bit matrix cores (conv3d_planes.hip)
size_t ufr_conv3d_planes_workspace_bytes(int32_t cin, int32_t cout, int32_t cout2, int32_t mode) {
  return conv3d_planes_workspace_bytes(cin, cout, cout2, mode);
}

int ufr_conv3d_planes(const float* in, const float* in_absmax, const float* weight, const float* weight2, const float* bias,
                      const float* bn_scale, const float* bn_shift, const float* skip, float* out, float* out2,
                      float* out_absmax, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cout2,
                      int32_t mode, int32_t relu, int32_t out_ncdhw, int32_t flip, void* workspace, size_t workspace_bytes, int32_t planes_ready,
                      ufr_stream stream) {
  UFR_REQUIRE(in && in_absmax && weight && out && workspace, "ufr_conv3d_planes: null argument");
  UFR_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "ufr_conv3d_planes: B=%d D=%d H=%d W=%d", B, D, H, W);
  UFR_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ufr_conv3d_planes: bn_scale and bn_shift go together");
  UFR_REQUIRE(cout2 == 0 || (weight2 && out2 && out_ncdhw && !flip), "ufr_conv3d_planes: a second head needs weight2, out2, out_ncdhw");
  UFR_REQUIRE(out_ncdhw || (cout % 4 == 0 && cout2 == 0), "ufr_conv3d_planes: channel-last outputs need cout %% 4 == 0 (got %d)", cout);
  UFR_REQUIRE(!(out_ncdhw && (skip || out_absmax)), "ufr_conv3d_planes: skip / out_absmax go with channel-last outputs");
  UFR_REQUIRE(mode == UFR_CONV3D_S1 || mode == UFR_CONV3D_S2 || mode == UFR_CONV3D_T2, "ufr_conv3d_planes: unknown mode %d", mode);
  UFR_REQUIRE(!(mode != UFR_CONV3D_S1 && flip), "ufr_conv3d_planes: flip is the stride-1 data gradient");
  UFR_REQUIRE(!(mode == UFR_CONV3D_T2 && out_ncdhw), "ufr_conv3d_planes: the transposed layers write channel-last");
  const size_t need = conv3d_planes_workspace_bytes(cin, cout, cout2, mode);
  if (!need) return fail(UFR_ERR_ARG, "ufr_conv3d_planes: (cin %d, cout %d+%d, mode %d) is not a layer of this kernel family", cin, cout, cout2, mode);
  if (workspace_bytes < need) return fail(UFR_ERR_WORKSPACE, "ufr_conv3d_planes: workspace %zu < %zu", workspace_bytes, need);
  UFR_REQUIRE((long long)D * H * W * cin * 4 < (1ll << 31), "ufr_conv3d_planes: one batch element reaches 2 GiB");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p(flip ? "conv3d_dgrad" : "conv3d", s);
  UFR_HIP(launch_conv3d_planes(in, in_absmax, weight, weight2, bias, bn_scale, bn_shift, skip, out, out2, out_absmax, B, D, H, W, cin,
                               cout, cout2, mode, relu, out_ncdhw, flip, workspace, planes_ready != 0, s));
  return UFR_OK;
}

int ufr_absmax(const float* x, size_t n, float* absmax, ufr_stream stream) {
  UFR_REQUIRE(x && absmax, "ufr_absmax: null argument");
  UFR_HIP(launch_absmax(x, n, absmax, static_cast<hipStream_t>(stream)));
  return UFR_OK;
}

// backward of the plain layers (CostRegNetWeight: the producer the reference trains) -- conv3d.hip, second half
int ufr_conv3d_bwd_data(const float* d_out, const float* weight, const float* accumulate, float* d_in, int32_t B, int32_t D,
                        int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t mode, ufr_stream stream) {
  UFR_REQUIRE(d_out && weight && d_in, "ufr_conv3d_bwd_data: null argument");
  UFR_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "ufr_conv3d_bwd_data: B=%d D=%d H=%d W=%d", B, D, H, W);
  UFR_REQUIRE(mode == UFR_CONV3D_S1 || mode == UFR_CONV3D_S2 || mode == UFR_CONV3D_T2, "ufr_conv3d_bwd_data: unknown mode %d", mode);
  UFR_REQUIRE(mode != UFR_CONV3D_S2 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "ufr_conv3d_bwd_data: stride 2 needs even extents");
  UFR_REQUIRE(cin == 1 || cin % 4 == 0, "ufr_conv3d_bwd_data: cin=%d", cin);
  UFR_REQUIRE(!(cin == 1 && accumulate), "ufr_conv3d_bwd_data: no fused addition into a 1-channel gradient");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("conv3d_dgrad", s);
  const hipError_t e = launch_conv3d_bwd_data(d_out, weight, accumulate, d_in, B, D, H, W, cin, cout, mode, s);
  if (e == hipErrorInvalidValue)
    return fail(UFR_ERR_ARG, "ufr_conv3d_bwd_data: (cin %d, cout %d, mode %d) is not a layer of CostRegNetWeight", cin, cout, mode);
  UFR_HIP(e);
  return UFR_OK;
}

int ufr_conv3d_bwd_weight(const float* in, const float* d_out, float* d_weight, float* d_bias, int32_t B, int32_t D, int32_t H,
                          int32_t W, int32_t cin, int32_t cout, int32_t mode, ufr_stream stream) {
  UFR_REQUIRE(in && d_out && d_weight, "ufr_conv3d_bwd_weight: null argument");
  UFR_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "ufr_conv3d_bwd_weight: B=%d D=%d H=%d W=%d", B, D, H, W);
  UFR_REQUIRE(mode == UFR_CONV3D_S1 || mode == UFR_CONV3D_S2 || mode == UFR_CONV3D_T2, "ufr_conv3d_bwd_weight: unknown mode %d", mode);
  UFR_REQUIRE(mode != UFR_CONV3D_S2 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "ufr_conv3d_bwd_weight: stride 2 needs even extents");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("conv3d_wgrad", s);
  const hipError_t e = launch_conv3d_bwd_weight(in, d_out, d_weight, d_bias, B, D, H, W, cin, cout, mode, s);
  if (e == hipErrorInvalidValue)
    return fail(UFR_ERR_ARG, "ufr_conv3d_bwd_weight: (cin %d, cout %d, mode %d) is not a layer of CostRegNetWeight", cin, cout, mode);
  UFR_HIP(e);
  return UFR_OK;
}

int ufr_conv3d_bwd_weight_heads(const float* in, const float* d_out, const float* d_out2, float* d_weight, float* d_weight2, int32_t B,
                                int32_t D, int32_t H, int32_t W, ufr_stream stream) {
  UFR_REQUIRE(in && d_out && d_out2 && d_weight && d_weight2, "ufr_conv3d_bwd_weight_heads: null argument");
  UFR_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "ufr_conv3d_bwd_weight_heads: B=%d D=%d H=%d W=%d", B, D, H, W);
  UFR_REQUIRE((long long)D * H * W * 32 < (1ll << 31), "ufr_conv3d_bwd_weight_heads: one batch element reaches 2 GiB");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("conv3d_wgrad", s);
  UFR_HIP(launch_conv3d_wgrad_heads(in, d_out, d_out2, d_weight, d_weight2, B, D, H, W, s));
  return UFR_OK;
}

// ------------------------------------------------------------------ TSDF fusion
int ufr_tsdf_integrate(float* tsdf, float* weight, float* color, const int32_t* dim, const float* origin,
                       float voxel_size, float trunc_margin, const float* cam_intr, const float* cam_pose,
                       const float* depth_im, const float* color_im, int32_t im_h, int32_t im_w, float obs_weight,
                       int32_t integrate_color, ufr_stream stream) {
  UFR_REQUIRE(tsdf && weight && dim && origin && cam_intr && cam_pose && depth_im, "ufr_tsdf_integrate: null argument");
  UFR_REQUIRE(dim[0] > 0 && dim[1] > 0 && dim[2] > 0, "ufr_tsdf_integrate: volume %dx%dx%d", dim[0], dim[1], dim[2]);
  UFR_REQUIRE(im_h > 0 && im_w > 0, "ufr_tsdf_integrate: image %dx%d", im_h, im_w);
  UFR_REQUIRE(voxel_size > 0.f && trunc_margin > 0.f, "ufr_tsdf_integrate: voxel_size %g, trunc_margin %g", voxel_size, trunc_margin);
  UFR_REQUIRE(!integrate_color || (color && color_im), "ufr_tsdf_integrate: colour integration needs color and color_im");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("tsdf_integrate", s);
  UFR_HIP(launch_tsdf_integrate(tsdf, weight, color, dim, origin, voxel_size, trunc_margin, cam_intr, cam_pose, depth_im,
                                color_im, im_h, im_w, obs_weight, integrate_color, s));
  return UFR_OK;
}

int ufr_pixelwise_view_weights(const float* similarity, const float* params, float* view_weights, float* aggregated, int32_t NS,
                               int32_t D, int32_t H, int32_t W, ufr_stream stream) {
  UFR_REQUIRE(similarity && params && view_weights, "ufr_pixelwise_view_weights: null argument");
  UFR_REQUIRE(NS >= 1 && NS <= UFR_MAX_VIEWS && D >= 1 && H >= 1 && W >= 1, "ufr_pixelwise_view_weights: NS=%d D=%d H=%d W=%d", NS, D, H, W);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("pixelwise_view_weights", s);
  UFR_HIP(launch_pixelwise_weights(similarity, params, view_weights, aggregated, NS, D, H, W, s));
  return UFR_OK;
}

// ------------------------------------------------------------------ deformable convolution
size_t ufr_deform_conv2d_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
  Carver c(nullptr);
  c.f32((size_t)B * H * W * C);
  return c.off;
}

int ufr_deform_conv2d(const float* input, const float* offset, const float* mask, const float* weight,
                      const float* bias, float* output, int32_t B, int32_t C, int32_t Cout, int32_t H, int32_t W,
                      void* workspace, size_t workspace_bytes, ufr_stream stream) {
  UFR_REQUIRE(input && offset && weight && output && workspace, "ufr_deform_conv2d: null argument");
  UFR_REQUIRE(C > 0 && C % 4 == 0 && C <= 32, "ufr_deform_conv2d: C=%d unsupported (multiple of 4, <= 32)", C);
  UFR_REQUIRE(Cout == 8 || Cout == 16 || Cout == 32, "ufr_deform_conv2d: Cout=%d unsupported (8, 16, 32)", Cout);
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "ufr_deform_conv2d: B=%d H=%d W=%d", B, H, W);
  const size_t need = ufr_deform_conv2d_workspace_bytes(B, C, H, W);
  if (workspace_bytes < need) return fail(UFR_ERR_WORKSPACE, "deform_conv2d workspace too small: %zu < %zu", workspace_bytes, need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver c(workspace);
  float* in_cl = c.f32((size_t)B * H * W * C);
  ProfScope p("deform_conv2d", s);
  UFR_HIP(launch_chw_to_hwc(input, in_cl, B, C, H * W, s));
  UFR_HIP(launch_deform_conv3x3(in_cl, offset, mask, weight, bias, output, B, C, Cout, H, W, s));
  return UFR_OK;
}

// channel-last pipeline of the feature backbone (conv2d.hip; featurenet.py is the plan)
int ufr_conv2d(const float* input, const float* weight, const float* scale, const float* shift, const float* skip, float* output,
               int32_t B, int32_t cin, int32_t cout, int32_t H, int32_t W, int32_t ksize, int32_t stride, int32_t flags,
               int32_t sigmoid_from, ufr_stream stream) {
  UFR_REQUIRE(input && weight && output, "ufr_conv2d: null argument");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && cout <= 32, "ufr_conv2d: B=%d H=%d W=%d cout=%d", B, H, W, cout);
  UFR_REQUIRE((ksize == 1 || ksize == 3 || ksize == 5) && (stride == 1 || stride == 2), "ufr_conv2d: ksize=%d stride=%d", ksize, stride);
  UFR_REQUIRE((unsigned long long)H * W * (cin > 3 ? cin : 4) * 4ull < (1ull << 31), "ufr_conv2d: %d x %d x %d exceeds 2^31 bytes per image", H, W, cin);
  Conv2dArgs a;
  a.in = input; a.w = weight; a.scale = scale; a.shift = shift; a.skip = skip; a.out = output;
  a.B = B; a.H = H; a.W = W; a.cout = cout;
  a.Ho = (H + 2 * (ksize / 2) - ksize) / stride + 1;
  a.Wo = (W + 2 * (ksize / 2) - ksize) / stride + 1;
  a.relu = (flags & UFR_CONV2D_RELU) != 0;
  a.out_planar = (flags & UFR_CONV2D_OUT_PLANAR) != 0;
  a.sigmoid_from = sigmoid_from;
  UFR_REQUIRE(!skip || (a.Ho % 2 == 0 && a.Wo % 2 == 0 && cout % 4 == 0), "ufr_conv2d: the upsampled skip needs even output extents and cout %% 4 == 0");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("conv2d", s);
  const hipError_t e = launch_conv2d(a, cin, ksize, stride, (flags & UFR_CONV2D_IN_PLANAR) != 0, s);
  if (e == hipErrorInvalidValue)
    return fail(UFR_ERR_ARG, "ufr_conv2d: %d -> %d channels, %dx%d, stride %d is not a layer shape of FeatureNet (conv2d.hip)", cin, cout, ksize, ksize, stride);
  UFR_HIP(e);
  return UFR_OK;
}

int ufr_upsample_add(const float* reduced_cl, const float* fine, float* output_cl, int32_t B, int32_t C, int32_t h, int32_t w,
                     ufr_stream stream) {
  UFR_REQUIRE(reduced_cl && fine && output_cl, "ufr_upsample_add: null argument");
  UFR_REQUIRE((C == 8 || C == 16) && B > 0 && h > 0 && w > 0, "ufr_upsample_add: B=%d C=%d h=%d w=%d (C in {8, 16})", B, C, h, w);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("upsample_add", s);
  UFR_HIP(launch_upsample_add(reduced_cl, fine, output_cl, B, C, h, w, s));
  return UFR_OK;
}

int ufr_deform_conv2d_cl(const float* input_cl, const float* offset_mask, const float* weight, const float* bias,
                         const float* scale, const float* shift, float* output, int32_t B, int32_t C, int32_t Cout, int32_t H,
                         int32_t W, int32_t flags, ufr_stream stream) {
  const float* offset = offset_mask;
  UFR_REQUIRE(input_cl && offset && weight && output, "ufr_deform_conv2d_cl: null argument");
  const float* mask = offset_mask + (size_t)18 * H * W;
  UFR_REQUIRE(C == 32, "ufr_deform_conv2d_cl: C=%d (the channel-last form exists for the 32-channel layers of FeatureNet)", C);
  UFR_REQUIRE(Cout == 8 || Cout == 16 || Cout == 32, "ufr_deform_conv2d_cl: Cout=%d unsupported (8, 16, 32)", Cout);
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "ufr_deform_conv2d_cl: B=%d H=%d W=%d", B, H, W);
  UFR_REQUIRE((scale == nullptr) == (shift == nullptr), "ufr_deform_conv2d_cl: scale and shift come together");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("deform_conv2d", s);
  UFR_HIP(launch_deform_conv3x3(input_cl, offset, mask, weight, bias, output, B, C, Cout, H, W, s,
                                DcnEpilogue{scale, shift, (flags & UFR_CONV2D_RELU) != 0, (flags & UFR_CONV2D_OUT_PLANAR) == 0, 27}));
  return UFR_OK;
}

// ------------------------------------------------------------------ feature-matching transformer layer
size_t ufr_fmt_layer_workspace_bytes(int32_t N, int32_t S) {
  return align_up((size_t)(N > 0 ? N : 1) * 160 * sizeof(float) * (1 + (size_t)fmt_state_parts(S > 0 ? S : 1)));
}

int ufr_fmt_layer(const ufr_fmt_layer_weights* w, const float* x, const float* src, int32_t N, int32_t T, int32_t S,
                  float* out, void* workspace, ufr_stream stream) {
  static_assert(sizeof(ufr_fmt_layer_weights) == sizeof(FmtWeights), "ufr_fmt_layer_weights layout");
  UFR_REQUIRE(w && x && out && workspace, "ufr_fmt_layer: null argument");
  FmtWeights fw;
  memcpy(&fw, w, sizeof(fw));
  const float* const* pw = reinterpret_cast<const float* const*>(&fw);
  for (int i = 0; i < 16; ++i) UFR_REQUIRE(pw[i], "ufr_fmt_layer: weight pointer %d is null", i);
  if (!src) { src = x; S = T; }
  UFR_REQUIRE(N > 0 && T > 0 && S > 0, "ufr_fmt_layer: N=%d T=%d S=%d", N, T, S);
  UFR_REQUIRE(out != x && out != src, "ufr_fmt_layer: out must not alias an input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope p("fmt_layer", s);
  UFR_HIP(launch_fmt_layer(fw, x, src, N, T, S, out, static_cast<float*>(workspace), s));
  return UFR_OK;
}

// ------------------------------------------------------------------ profiling hooks
void ufr_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (auto& e : g_prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  g_prof.clear();
  g_prof_on.store(on != 0, std::memory_order_relaxed);
}

int ufr_profile_read(const char** names, float* ms, int32_t* launches, int cap) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  int n = 0;
  for (auto& e : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) continue;
    int k = 0;
    for (; k < n; ++k)
      if (strcmp(names[k], e.name) == 0) break;
    if (k == n) {
      if (n >= cap) continue;
      names[n] = e.name; ms[n] = 0.f; launches[n] = 0; ++n;
    }
    ms[k] += t;
    launches[k] += 1;
  }
  for (auto& e : g_prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  g_prof.clear();
  return n;
}

}  // extern "C"
