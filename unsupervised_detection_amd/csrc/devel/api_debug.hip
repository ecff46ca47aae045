// libudet_debug.so -- the test-only hooks of include/udet_debug.h.  They are NOT part of libudet.so: this small library links
// against it and reaches the (process-global) selection state of the convolution launcher through the internal C++
// interface (conv_host.h).  Only tests/ and tools/ load it; the product path never does.
#include <string.h>

#include "../../../include/udet_debug.h"
#include "../conv_host.h"
#include "../plan.h"

using namespace udet;

extern "C" {
int udet_debug_last_conv(void) { return conv_last_config(); }
void udet_debug_force_conv(int bm, int bn, int ks) { conv_force_config(bm, bn, ks); }
void udet_debug_conv_fp16(int on) { conv_debug_f16(on); }
void udet_debug_force_wgrad(int nsplit, int dma) { wgrad_force(nsplit, dma); }
int udet_debug_last_wgrad(void) { return wgrad_last_config(); }
void udet_debug_upb_min_pixels(long v) { plan_debug_upb_min_pixels(v); }
void udet_debug_force_pair(int on) { conv_force_pair(on); }
int udet_debug_last_pair(void) { return conv_last_pair(); }
// two forward convolutions of the same geometry (cin a multiple of 8; separate inputs / weights / biases / outputs, batches na / nb) through
// launch_conv_pair -- ONE launch when udet_debug_force_pair(1) is set and the LDS-DMA family can take both (tests/test_ops_gpu.py)
int udet_debug_conv2d_pair(const float* xa, const float* xb, const float* wa, const float* wb, const float* ba, const float* bb, float* ya, float* yb,
                           int na, int nb, int h, int w, int cin, int cout, int k, int stride, int dilation, int act, float alpha, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cin % 8 != 0 || k * k > UDET_MAX_TAPS) { set_error("debug_conv2d_pair: cin must be a multiple of 8"); return UDET_ERR_SHAPE; }
  const int kc = cin, ldw = round_up(cout, 4);
  const size_t wfl = ((size_t)k * k * kc * ldw + 63) & ~(size_t)63, part_fl = (size_t)4 << 20, zero_fl = 64 + UDET_MAX_TICKETS;
  if (workspace_bytes < (2 * wfl + part_fl + zero_fl) * sizeof(float)) { set_error("debug_conv2d_pair: workspace too small"); return UDET_ERR_ARG; }
  float* wpa = reinterpret_cast<float*>(workspace);
  float* wpb = wpa + wfl;
  float* part = wpb + wfl;
  float* zero = part + part_fl;
  UDET_HIP(hipMemsetAsync(zero, 0, zero_fl * sizeof(float), stream));
  UDET_TRY(launch_pack_weights(wa, wpa, k * k, cin, cout, kc, ldw, kc, 0, 0, nullptr, stream));
  UDET_TRY(launch_pack_weights(wb, wpb, k * k, cin, cout, kc, ldw, kc, 0, 0, nullptr, stream));
  ConvParams p[2];
  for (int i = 0; i < 2; ++i) {
    memset(&p[i], 0, sizeof(ConvParams));
    conv_setup_fwd(p[i], i ? nb : na, h, w, k, k, stride, dilation);
    p[i].x = i ? xb : xa; p[i].ldx = cin; p[i].wp = i ? wpb : wpa; p[i].Kc = kc; p[i].ldw = ldw; p[i].bias = i ? bb : ba;
    p[i].y = i ? yb : ya; p[i].ldy = cout; p[i].Cout = cout; p[i].act = act; p[i].alpha = alpha;
    p[i].partial = part; p[i].partial_cap = part_fl; p[i].zero16 = zero; p[i].tickets = reinterpret_cast<int*>(zero + 64);
  }
  return launch_conv_pair(p[0], p[1], stream);
}
void udet_debug_set_tuning(int on) {
  conv_set_tuning(on);
  wgrad_set_tuning(on);
}
}
