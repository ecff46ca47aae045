/* libudet.so -- C ABI of the MI355X-native adversarial_learner hot path.
 *
 * The reference (antonilo/unsupervised_detection) has no FFI/plugin layer: its hot path is a
 * TF-1.13 graph assembled by models/adversarial_learner.py:72-258 from models/nets.py,
 * models/PWCNet/{model_pwcnet,core_warp,core_costvol}.py and models/utils/{convolution,loss,flow}_utils.py.  Each entry
 * point below names the reference function (file:line under /root/reference) it replaces.
 *
 * Conventions
 *  - every tensor is float32, NHWC, contiguous unless a channel stride is given; pointers are
 *    DEVICE pointers owned by the caller; the library never allocates device memory;
 *  - convolution weights are HWIO ([kh][kw][cin][cout]) exactly like the TF variables;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *    asynchronously on it; no internal threads, no global state except the thread-local error;
 *  - return value: 0 = ok, <0 = error (UDET_ERR_*); udet_last_error() describes the failure.
 */
#ifndef UDET_H
#define UDET_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UDET_OK 0
#define UDET_ERR_SHAPE (-1)
#define UDET_ERR_ALIGN (-2)
#define UDET_ERR_HIP (-3)
#define UDET_ERR_UNSUPPORTED (-4)
#define UDET_ERR_ARG (-5)

#define UDET_ACT_NONE 0
#define UDET_ACT_LEAKY 1
#define UDET_ACT_ELU 2

int udet_version(void);
const char* udet_last_error(void);

/* ---- single-op entry points (parity tests) -------------------------------------------- */

/* dense_image_warp(image, flow*flow_scale): models/PWCNet/core_warp.py:153-202 (+ :42-150);
 * model_pwcnet.py:240-245,616-617.  c % 4 == 0. */
int udet_warp(const float* image, const float* flow, float flow_scale, float* out, int n, int h, int w, int c,
              void* stream);
/* same, additionally dumping the int32 (floor_y,floor_x) and float (alpha_y,alpha_x) per pixel
 * ([n,h,w,2] each) -- the grid-index math that must be bit-exact (core_warp.py:99-115). */
int udet_warp_debug(const float* image, const float* flow, float flow_scale, float* out, int* floor_yx,
                    float* alpha_yx, int n, int h, int w, int c, void* stream);
/* cost_volume(c1, warp, search_range=4) incl. leaky 0.1: models/PWCNet/core_costvol.py:20-40. out [n,h,w,81] */
int udet_cost_volume(const float* c1, const float* warp, float* out, int n, int h, int w, int c, void* stream);

/* tf.nn.conv2d / tf.layers.conv2d, padding='SAME', + bias + activation
 * (models/utils/convolution_utils.py:46,81-84; models/PWCNet/model_pwcnet.py:161-165,484-504,562-574).
 * upsample2x != 0 first applies tf.image.resize_nearest_neighbor(x2, align_corners=True)
 * (convolution_utils.py:70-71) fused into the loader.  workspace >= udet_conv2d_workspace_bytes(). */
size_t udet_conv2d_workspace_bytes(int n, int h, int w, int cin, int cout, int kh, int kw, int upsample2x);
int udet_conv2d(const float* x, const float* w_hwio, const float* bias, float* y, int n, int h, int w, int cin, int cout,
                int kh, int kw, int stride, int dilation, int upsample2x, int act, float alpha, void* workspace,
                size_t workspace_bytes, void* stream);
/* gradients of the op above w.r.t. its input (dx, [n,h,w,cin]) and w.r.t. weights / bias
 * (tf.gradients through Conv2D: Conv2DBackpropInput / Conv2DBackpropFilter / BiasAddGrad).
 * dy is the gradient w.r.t. the activated output; y_saved is the forward output (needed when act != NONE). */
int udet_conv2d_backward_data(const float* dy, const float* y_saved, const float* w_hwio, float* dx, int n, int h, int w,
                              int cin, int cout, int kh, int kw, int stride, int dilation, int act, float alpha,
                              void* workspace, size_t workspace_bytes, void* stream);
int udet_conv2d_backward_filter(const float* x, const float* dy, const float* y_saved, float* dw_hwio, float* dbias, int n,
                                int h, int w, int cin, int cout, int kh, int kw, int stride, int dilation, int upsample2x,
                                int act, float alpha, void* workspace, size_t workspace_bytes, void* stream);
/* tf.layers.conv2d_transpose(x, cout, 4, 2, 'same') + bias: models/PWCNet/model_pwcnet.py:283-286.
 * w is [4][4][cout][cin]; y is [n,2h,2w,cout]. */
int udet_conv2d_transpose4x4s2(const float* x, const float* w_hwoi, const float* bias, float* y, int n, int h, int w,
                               int cin, int cout, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UDET_H */
