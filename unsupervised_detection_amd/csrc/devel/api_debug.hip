// libudet_debug.so -- the test-only hooks of include/udet_debug.h.  They are NOT part of libudet.so: this small library links
// against it and reaches the (process-global) selection state of the convolution launcher through the internal C++
// interface (conv_host.h).  Only tests/ and tools/ load it; the product path never does.
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
void udet_debug_set_tuning(int on) {
  conv_set_tuning(on);
  wgrad_set_tuning(on);
}
}
