// extern "C" surface of libudet.so (see include/udet.h).
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/udet.h"
#include "common.h"
#include "conv_host.h"
#include "elementwise.h"

namespace udet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return UDET_ERR_HIP;
}

thread_local LaunchSink* g_launch_sink = nullptr;
bool launch_sink_next(hipEvent_t* a, hipEvent_t* b) {
  LaunchSink* s = g_launch_sink;
  if (!s || s->n >= s->cap) return false;
  if (hipEventCreate(a) != hipSuccess) return false;
  if (hipEventCreate(b) != hipSuccess) { (void)hipEventDestroy(*a); return false; }
  s->ev[2 * s->n] = *a;
  s->ev[2 * s->n + 1] = *b;
  ++s->n;
  return true;
}

// bump allocator over the caller's workspace (256-byte granules)
struct Arena {
  char* base;
  size_t cap, used;
  Arena(void* p, size_t n) : base((char*)p), cap(n), used(0) {}
  float* take(size_t floats) {
    const size_t bytes = (floats * sizeof(float) + 255) & ~(size_t)255;
    if (!base || used + bytes > cap) return nullptr;
    float* r = (float*)(base + used);
    used += bytes;
    return r;
  }
};
static inline size_t gran(size_t floats) { return (floats * sizeof(float) + 255) & ~(size_t)255; }
#define SPLITK_FLOATS ((size_t)8 << 20)  // 32 MiB of split-K scratch for the single-op entry points
}  // namespace udet

using namespace udet;

extern "C" {

int udet_version(void) { return 101; }
const char* udet_last_error(void) { return g_err; }

int udet_warp(const float* image, const float* flow, float flow_scale, float* out, int n, int h, int w, int c,
              void* stream) {
  return launch_warp(image, flow, 2, 0, flow_scale, out, n, h, w, c, nullptr, nullptr, (hipStream_t)stream);
}
int udet_warp_debug(const float* image, const float* flow, float flow_scale, float* out, int* floor_yx, float* alpha_yx,
                    int n, int h, int w, int c, void* stream) {
  return launch_warp(image, flow, 2, 0, flow_scale, out, n, h, w, c, floor_yx, alpha_yx, (hipStream_t)stream);
}
int udet_cost_volume(const float* c1, const float* warp, float* out, int n, int h, int w, int c, void* stream) {
  return launch_cost_volume(c1, warp, out, 81, 0, n, h, w, c, (hipStream_t)stream);
}

int udet_resize_bilinear_legacy_fwd(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream) {
  if (n < 1 || h < 1 || w < 1 || c < 1 || oh < 1 || ow < 1) { set_error("resize: bad shape"); return UDET_ERR_SHAPE; }
  return launch_resize_bilinear_fwd(x, c, 0, n, h, w, y, c, 0, oh, ow, c, 1.f, 1.f, (hipStream_t)stream);
}
int udet_resize_bilinear_legacy_bwd(const float* dy, float* dx, int n, int h, int w, int c, int oh, int ow, void* stream) {
  if (n < 1 || h < 1 || w < 1 || c < 1 || oh < 1 || ow < 1) { set_error("resize: bad shape"); return UDET_ERR_SHAPE; }
  return launch_resize_bilinear_bwd(dy, c, 0, n, oh, ow, dx, c, 0, h, w, c, 0, (hipStream_t)stream);
}

int udet_crop_flip_resize(const void* src, int src_is_u8, int nearest, int n, int h, int w, int c, const int* params6,
                          float* dst, int oh, int ow, float div, float add, void* stream) {
  if (n < 1 || h < 1 || w < 1 || c < 1 || oh < 1 || ow < 1 || !src || !dst) { set_error("crop_flip_resize: bad argument"); return UDET_ERR_ARG; }
  return launch_crop_flip_resize(src, src_is_u8, nearest, n, h, w, c, params6, dst, oh, ow, div, add, (hipStream_t)stream);
}

int udet_mask_stats(const float* pred_masks, const float* gt_masks, int n, int h, int w, float threshold, float gt_threshold,
                    double* stats8, void* stream) {
  if (n < 1 || h < 4 || w < 4 || !pred_masks || !gt_masks || !stats8) { set_error("mask_stats: bad argument"); return UDET_ERR_ARG; }
  return launch_mask_stats(pred_masks, gt_masks, n, h, w, threshold, gt_threshold, stats8, (hipStream_t)stream);
}

size_t udet_conv2d_workspace_bytes(int n, int h, int w, int cin, int cout, int kh, int kw, int upsample2x) {
  const int kc = round_up(cin > cout ? cin : cout, 8), ldw = round_up(cin > cout ? cin : cout, 4);
  const size_t pix_in = (size_t)n * h * w, pix_out = pix_in * (upsample2x ? 4 : 1);
  size_t b = 0;
  b += gran((size_t)kh * kw * kc * ldw);                       // packed weights
  b += gran(pix_in * round_up(cin, 8));                          // channel-padded x
  b += 2 * gran(pix_out * round_up(cout, 8));                    // channel-padded dy / y_saved
  b += gran(SPLITK_FLOATS);                                      // split-K partials
  b += gran(64 * wgrad_partial_floats_needed(kh * kw, cin, cout));
  return b + 8192 + gran(64 + UDET_MAX_TICKETS);
}

int udet_conv2d(const float* x, const float* w_hwio, const float* bias, float* y, int n, int h, int w, int cin, int cout,
                int kh, int kw, int stride, int dilation, int upsample2x, int act, float alpha, void* workspace,
                size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (kh * kw > UDET_MAX_TAPS || n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1 || stride < 1 || dilation < 1) {
    set_error("conv2d: bad shape");
    return UDET_ERR_SHAPE;
  }
  Arena ar(workspace, workspace_bytes);
  const int kc = round_up(cin, 8), ldw = round_up(cout, 4);
  float* wp = ar.take((size_t)kh * kw * kc * ldw);
  const float* xin = x;
  int ldx = cin;
  if (cin % 8 != 0) {
    float* xp = ar.take((size_t)n * h * w * kc);
    if (!xp) { set_error("conv2d: workspace too small"); return UDET_ERR_ARG; }
    UDET_HIP(hipMemsetAsync(xp, 0, (size_t)n * h * w * kc * sizeof(float), stream));
    UDET_TRY(launch_copy_channels(x, cin, 0, xp, kc, 0, (long)n * h * w, cin, 1.f, 0.f, stream));
    xin = xp;
    ldx = kc;
  }
  float* part = ar.take(SPLITK_FLOATS);
  float* zero = ar.take(64 + UDET_MAX_TICKETS);  // 16-byte zero block + split-K tickets
  if (!wp || !part || !zero) { set_error("conv2d: workspace too small"); return UDET_ERR_ARG; }
  UDET_HIP(hipMemsetAsync(zero, 0, (64 + UDET_MAX_TICKETS) * sizeof(float), stream));
  UDET_TRY(launch_pack_weights(w_hwio, wp, kh * kw, cin, cout, kc, ldw, kc, 0, 0, nullptr, stream));
  ConvParams p;
  memset(&p, 0, sizeof(p));
  const int us = upsample2x ? 1 : 0;
  conv_setup_fwd(p, n, h << us, w << us, kh, kw, stride, dilation);
  p.x = xin; p.ldx = ldx; p.x_coff = 0; p.up_shift = us;
  p.wp = wp; p.Kc = kc; p.ldw = ldw; p.bias = bias;
  p.y = y; p.ldy = cout; p.y_coff = 0; p.Cout = cout;
  p.act = act; p.alpha = alpha;
  p.partial = part; p.partial_cap = SPLITK_FLOATS;
  p.zero16 = zero;
  p.tickets = reinterpret_cast<int*>(zero + 64);
  return launch_conv(p, stream);
}

int udet_conv2d_transpose4x4s2(const float* x, const float* w_hwoi, const float* bias, float* y, int n, int h, int w,
                               int cin, int cout, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Arena ar(workspace, workspace_bytes);
  const int kc = round_up(cin, 8), ldw = round_up(cout, 4);
  float* wp = ar.take((size_t)16 * kc * ldw);
  const float* xin = x;
  int ldx = cin;
  if (cin % 8 != 0) {
    float* xp = ar.take((size_t)n * h * w * kc);
    if (!xp) { set_error("conv2d_transpose: workspace too small"); return UDET_ERR_ARG; }
    UDET_HIP(hipMemsetAsync(xp, 0, (size_t)n * h * w * kc * sizeof(float), stream));
    UDET_TRY(launch_copy_channels(x, cin, 0, xp, kc, 0, (long)n * h * w, cin, 1.f, 0.f, stream));
    xin = xp;
    ldx = kc;
  }
  float* part = ar.take(SPLITK_FLOATS);
  float* zero = ar.take(64 + UDET_MAX_TICKETS);  // 16-byte zero block + split-K tickets
  if (!wp || !part || !zero) { set_error("conv2d_transpose: workspace too small"); return UDET_ERR_ARG; }
  UDET_HIP(hipMemsetAsync(zero, 0, (64 + UDET_MAX_TICKETS) * sizeof(float), stream));
  // w is [t][cout][cin]; B operand wants [t][k=cin][n=cout]  -> mode 1 with (R=cout, C=cin)
  UDET_TRY(launch_pack_weights(w_hwoi, wp, 16, cout, cin, kc, ldw, kc, 0, 1, nullptr, stream));
  for (int cls = 0; cls < conv_dgrad_classes(2, 2 * h, 2 * w); ++cls) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    if (!conv_setup_dgrad(p, cls, n, 2 * h, 2 * w, 4, 4, 2, 1)) continue;
    p.x = xin; p.ldx = ldx; p.wp = wp; p.Kc = kc; p.ldw = ldw; p.bias = bias;
    p.y = y; p.ldy = cout; p.Cout = cout;
    p.partial = part; p.partial_cap = SPLITK_FLOATS;
    p.zero16 = zero;
    p.tickets = reinterpret_cast<int*>(zero + 64);
    UDET_TRY(launch_conv(p, stream));
  }
  return UDET_OK;
}

int udet_conv2d_backward_data(const float* dy, const float* y_saved, const float* w_hwio, float* dx, int n, int h, int w,
                              int cin, int cout, int kh, int kw, int stride, int dilation, int act, float alpha,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (stride != 1 && stride != 2) { set_error("conv2d_backward_data: stride %d unsupported", stride); return UDET_ERR_UNSUPPORTED; }
  Arena ar(workspace, workspace_bytes);
  const int kc = round_up(cout, 8), ldw = round_up(cin, 4);
  float* wp = ar.take((size_t)kh * kw * kc * ldw);
  int pt, pl, oh, ow;
  same_pad(h, kh, stride, dilation, &pt, &oh);
  same_pad(w, kw, stride, dilation, &pl, &ow);
  const float *dyin = dy, *yain = (act != UDET_ACT_NONE) ? y_saved : nullptr;
  int ldy = cout;
  if (cout % 8 != 0) {
    const size_t fl = (size_t)n * oh * ow * kc;
    float* a = ar.take(fl);
    float* b = ar.take(fl);
    if (!a || !b) { set_error("conv2d_backward_data: workspace too small"); return UDET_ERR_ARG; }
    UDET_HIP(hipMemsetAsync(a, 0, fl * sizeof(float), stream));
    UDET_TRY(launch_copy_channels(dy, cout, 0, a, kc, 0, (long)n * oh * ow, cout, 1.f, 0.f, stream));
    dyin = a;
    if (yain) {
      UDET_HIP(hipMemsetAsync(b, 0, fl * sizeof(float), stream));
      UDET_TRY(launch_copy_channels(y_saved, cout, 0, b, kc, 0, (long)n * oh * ow, cout, 1.f, 0.f, stream));
      yain = b;
    }
    ldy = kc;
  }
  float* part = ar.take(SPLITK_FLOATS);
  float* zero = ar.take(64 + UDET_MAX_TICKETS);  // 16-byte zero block + split-K tickets
  if (!wp || !part || !zero) { set_error("conv2d_backward_data: workspace too small"); return UDET_ERR_ARG; }
  UDET_HIP(hipMemsetAsync(zero, 0, (64 + UDET_MAX_TICKETS) * sizeof(float), stream));
  UDET_TRY(launch_pack_weights(w_hwio, wp, kh * kw, cin, cout, kc, ldw, kc, 0, 1, nullptr, stream));
  for (int cls = 0; cls < conv_dgrad_classes(stride, h, w); ++cls) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    if (!conv_setup_dgrad(p, cls, n, h, w, kh, kw, stride, dilation)) continue;
    p.x = dyin; p.ldx = ldy; p.xa = yain; p.xact = act; p.xalpha = alpha;
    p.wp = wp; p.Kc = kc; p.ldw = ldw; p.kreal = cout;
    p.y = dx; p.ldy = cin; p.Cout = cin;
    p.partial = part; p.partial_cap = SPLITK_FLOATS;
    p.zero16 = zero;
    p.tickets = reinterpret_cast<int*>(zero + 64);
    UDET_TRY(launch_conv(p, stream));
  }
  return UDET_OK;
}

int udet_conv2d_backward_filter(const float* x, const float* dy, const float* y_saved, float* dw_hwio, float* dbias, int n,
                                int h, int w, int cin, int cout, int kh, int kw, int stride, int dilation, int upsample2x,
                                int act, float alpha, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Arena ar(workspace, workspace_bytes);
  const int us = upsample2x ? 1 : 0;
  ConvParams g;
  memset(&g, 0, sizeof(g));
  conv_setup_fwd(g, n, h << us, w << us, kh, kw, stride, dilation);
  WgradParams p;
  memset(&p, 0, sizeof(p));
  const int cin8 = round_up(cin, 8), cout8 = round_up(cout, 8);
  const float* xin = x;
  int ldx = cin;
  if (cin % 4 != 0) {
    const size_t fl = (size_t)n * h * w * cin8;
    float* xp = ar.take(fl);
    if (!xp) { set_error("conv2d_backward_filter: workspace too small"); return UDET_ERR_ARG; }
    UDET_HIP(hipMemsetAsync(xp, 0, fl * sizeof(float), stream));
    UDET_TRY(launch_copy_channels(x, cin, 0, xp, cin8, 0, (long)n * h * w, cin, 1.f, 0.f, stream));
    xin = xp;
    ldx = cin8;
  }
  const float *dyin = dy, *yain = (act != UDET_ACT_NONE) ? y_saved : nullptr;
  int ldy = cout;
  if (cout % 4 != 0) {
    const size_t fl = (size_t)n * g.OH * g.OW * cout8;
    float* a = ar.take(fl);
    float* b = ar.take(fl);
    if (!a || !b) { set_error("conv2d_backward_filter: workspace too small"); return UDET_ERR_ARG; }
    UDET_HIP(hipMemsetAsync(a, 0, fl * sizeof(float), stream));
    UDET_TRY(launch_copy_channels(dy, cout, 0, a, cout8, 0, (long)n * g.OH * g.OW, cout, 1.f, 0.f, stream));
    dyin = a;
    if (yain) {
      UDET_HIP(hipMemsetAsync(b, 0, fl * sizeof(float), stream));
      UDET_TRY(launch_copy_channels(y_saved, cout, 0, b, cout8, 0, (long)n * g.OH * g.OW, cout, 1.f, 0.f, stream));
      yain = b;
    }
    ldy = cout8;
  }
  float* zero = ar.take(64);
  if (!zero) { set_error("conv2d_backward_filter: workspace too small"); return UDET_ERR_ARG; }
  UDET_HIP(hipMemsetAsync(zero, 0, 64 * sizeof(float), stream));
  const size_t pf = 64 * wgrad_partial_floats_needed(kh * kw, cin, cout);
  size_t avail = (ar.cap - ar.used) / sizeof(float);
  if (avail > 256) avail -= 256;
  const size_t takef = pf < avail ? pf : avail;
  float* part = ar.take(takef);
  if (!part) { set_error("conv2d_backward_filter: workspace too small"); return UDET_ERR_ARG; }
  p.x = xin; p.ldx = ldx; p.N = n; p.H = h << us; p.W = w << us; p.up_shift = us; p.Cin = cin;
  p.dy = dyin; p.ya = yain; p.ldy = ldy; p.Cout = cout; p.yact = act; p.yalpha = alpha;
  p.OH = g.OH; p.OW = g.OW; p.isy = p.isx = stride;
  p.ntaps = g.ntaps;
  memcpy(p.taps, g.taps, sizeof(g.taps));
  p.dw = dw_hwio; p.db = dbias; p.partial = part; p.partial_floats = takef;
  p.zero16 = zero;
  return launch_wgrad_T(p, kh * kw, stream);
}

/* ---- per-stage entry points of the loss / optimizer tail (SURVEY 8b minimum export list) ---------------------------- */
int udet_flow_normalize(const float* flow, float* out, int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  if (!flow || !out || n < 1 || h < 1 || w < 1) { set_error("flow_normalize: bad argument"); return UDET_ERR_ARG; }
  if (!workspace || workspace_bytes < flow_stats_doubles(n) * sizeof(double) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
    set_error("flow_normalize: workspace needs %zu bytes, 8-byte aligned", flow_stats_doubles(n) * sizeof(double));
    return UDET_ERR_ARG;
  }
  return launch_flow_normalize(flow, (double*)workspace, out, n, (long)h * w, (hipStream_t)stream);
}
size_t udet_stage_workspace_bytes(int n) {
  const size_t a = flow_stats_doubles(n) * sizeof(double), b = (loss_part_floats(n) + 5 * (size_t)n + 64) * sizeof(float);
  return (a > b ? a : b) + 256;
}
int udet_charbonnier_loss(const float* gt_flows, const float* pred_flows, const float* masks, int mask_channels, int n, int h, int w,
                          float cbn, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gt_flows || !pred_flows || !out || n < 1 || h < 1 || w < 1 || (masks && mask_channels != 1 && mask_channels != 2)) {
    set_error("charbonnier_loss: bad argument (masks: null, 1 or 2 channels)");
    return UDET_ERR_ARG;
  }
  if (!workspace || workspace_bytes < udet_stage_workspace_bytes(n)) { set_error("charbonnier_loss: workspace too small"); return UDET_ERR_ARG; }
  return launch_charbonnier(gt_flows, pred_flows, masks, masks ? mask_channels : 1, n, (long)h * w, cbn, (float*)workspace, out, (hipStream_t)stream);
}
int udet_losses_forward(const float* flow, const float* mask, const float* pred3, int b, int h, int w, float cbn, float epsilon,
                        float* losses8, float* coef, void* workspace, size_t workspace_bytes, void* stream) {
  if (!flow || !mask || !pred3 || !losses8 || !coef || b < 1 || h < 1 || w < 1) { set_error("losses_forward: bad argument"); return UDET_ERR_ARG; }
  if (!workspace || workspace_bytes < udet_stage_workspace_bytes(b)) { set_error("losses_forward: workspace too small"); return UDET_ERR_ARG; }
  float* part = (float*)workspace;
  float* sums = part + loss_part_floats(b);
  return launch_losses(flow, mask, pred3, (long)h * w, b, cbn, epsilon, (float)w * (float)h * (float)b, part, losses8, coef, sums,
                       (hipStream_t)stream);
}
int udet_losses_backward(const float* flow, const float* mask, const float* pred3, const float* coef, int which, int b, int h, int w,
                         float cbn, float* dpred, float* dmask, void* stream) {
  if (!flow || !mask || !pred3 || !dpred || b < 1 || h < 1 || w < 1) { set_error("losses_backward: bad argument"); return UDET_ERR_ARG; }
  const long HW = (long)h * w;
  if (which == 2) return launch_rec_loss_bwd(flow, mask, pred3, dpred, (long)b * HW, cbn, 1.0f / ((float)w * (float)h * (float)b), (hipStream_t)stream);
  if (which == 1) {
    if (!coef || !dmask) { set_error("losses_backward: the generator loss needs coef (udet_losses_forward) and dmask"); return UDET_ERR_ARG; }
    return launch_gen_loss_bwd(flow, mask, pred3, coef, dpred, dmask, HW, b, cbn, (hipStream_t)stream);
  }
  set_error("losses_backward: which must be 1 (generator loss) or 2 (recover loss)");
  return UDET_ERR_ARG;
}
int udet_clip_or_noise(float* g, size_t n, float clip, const float* flag2, unsigned long long seed, long step, void* stream) {
  if (!g) { set_error("clip_or_noise: null gradient"); return UDET_ERR_ARG; }
  return launch_adam(nullptr, g, nullptr, nullptr, (long)n, 0.f, 0.f, 0.f, 0.f, clip, flag2, seed, (uint64_t)step, (hipStream_t)stream, 1);
}
int udet_adam_step(float* w, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, long t,
                   void* stream) {
  if (!w || !g || !m || !v || t < 1) { set_error("adam_step: bad argument (t counts applies from 1)"); return UDET_ERR_ARG; }
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  return launch_adam(w, const_cast<float*>(g), m, v, (long)n, (float)lr_t, beta1, beta2, eps, 0.f, nullptr, 0, 0, (hipStream_t)stream, 2);
}
/* fused warp -> cost volume of one pyramid level (what the step plan launches); corr [n,h,w,81]; warped_dbg optional [n,h,w,c] */
int udet_warp_cost_volume(const float* c1, const float* c2, const float* flow, float flow_scale, float* corr, float* warped_dbg, int n,
                          int h, int w, int c, void* stream) {
  if (!c1 || !c2 || !corr) { set_error("warp_cost_volume: null argument"); return UDET_ERR_ARG; }
  return launch_warp_cost_volume(c1, c2, flow, 2, 0, flow_scale, corr, 81, 0, -1, warped_dbg, n, h, w, c, (hipStream_t)stream);
}

}  // extern "C"
