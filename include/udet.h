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
#define UDET_ERR_OVERFLOW (-6) /* conv_fp16 mode: an optimizer update was dropped because its gradients were not finite */

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
/* The fused form the step plan launches once per pyramid level (model_pwcnet.py:616-623): warped = dense_image_warp(c2,
 * flow*flow_scale) is produced tile by tile in LDS and correlated with c1 at once -- the warped tensor never exists in HBM.
 * flow == NULL: no warp (level 6 correlates c1 with c2).  corr [n,h,w,81]; warped_dbg: optional [n,h,w,c] dump of the warped
 * features (test hook).  Bit-identical to udet_warp followed by udet_cost_volume. */
int udet_warp_cost_volume(const float* c1, const float* c2, const float* flow, float flow_scale, float* corr, float* warped_dbg,
                          int n, int h, int w, int c, void* stream);

/* tf.image.resize_images(x,[oh,ow]) / tf.image.resize_bilinear, TF-1.13 legacy sampling (align_corners=False, no
 * half-pixel centres): models/adversarial_learner.py:87-90, models/nets.py:108, models/utils/convolution_utils.py:88,
 * data/davis2016_data_utils.py:86-91.  x [n,h,w,c] -> y [n,oh,ow,c]; bwd is its exact adjoint (dy [n,oh,ow,c] -> dx). */
int udet_resize_bilinear_legacy_fwd(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream);
int udet_resize_bilinear_legacy_bwd(const float* dy, float* dx, int n, int h, int w, int c, int oh, int ow, void* stream);

/* Input stage ("next" row N1): the readers' per-image pipeline in one pass -- optional uint8 -> v/div + add
 * (preprocess_image / preprocess_mask, data/davis2016_data_utils.py:86-99; identical in fbms_/segtrackv2_data_utils.py),
 * flip (data/aug_flips.py:3-16), crop window (tf.random_crop :101-127 / tf.image.central_crop :129-133), TF-1.13
 * legacy bilinear or nearest-neighbour resize (tf.image.resize_images) to [oh,ow].  src [n,h,w,c] uint8 or float32,
 * dst [n,oh,ow,c] float32, params6 device int32 [n][6] = {y0, x0, crop_h, crop_w, flip_lr, flip_td} or NULL. */
int udet_crop_flip_resize(const void* src, int src_is_u8, int nearest, int n, int h, int w, int c, const int* params6,
                          float* dst, int oh, int ow, float div, float add, void* stream);

/* Evaluation tail ("next" row N2): the per-sample sums behind compute_boundary_score / disambiguate_forw_back /
 * tf_iou_computation / compute_all_IoU (models/utils/general_utils.py:89-159) and compute_IoU / compute_mae
 * (test_generator.py:19-40).  pred_masks, gt_masks [n,h,w,1] device float32; stats8 [n][8] device doubles =
 * {border sum (two-pixel strips, corners twice), |pred|, |gt|, |pred & gt|, sum pred*|gt-1|, sum (1-pred)*|gt|,
 *  sum (1-pred)*|gt-1|, sum pred*|gt|} with pred = mask > threshold, gt = gt_mask > gt_threshold. */
int udet_mask_stats(const float* pred_masks, const float* gt_masks, int n, int h, int w, float threshold, float gt_threshold,
                    double* stats8, void* stream);

/* Post-processing stage ("next" row N4; post_processing/generate_soft_score_from_buffer.py, crf_refine.py).  The third-party
 * routines those scripts call are absent from the reference tree; each entry point names the routine it restates and the call
 * site that fixes its arguments.  Frames are small (192x384): one workgroup reductions, double accumulation like numpy float64.
 *  - udet_post_border_mean: sanity_check (:116-125): mean over the four two-pixel border strips of n masks [n,h,w] -> out[n].
 *  - udet_post_bytescale / udet_post_resample_u8 / udet_post_place: rectify_pred_mask (:98-114) = scipy.misc.imresize (scipy <=
 *    1.2: bytescale with the window's own min / max + Pillow's 8-bit BILINEAR resampler, Resample.c; kk [n_out][ksize] 22-bit
 *    fixed-point coefficients and bounds [n_out][2] = (first tap, taps) are computed by the host exactly as Pillow does; axis 1 =
 *    horizontal pass, 0 = vertical pass) and the placement on a zero canvas divided by (max + 1e-6).
 *  - udet_post_minmax_norm: pred_mask = (score - min) / (max - min + 1e-6) (:88-90).
 *  - udet_post_remap: cv2.remap(src, flow + pixel grid, None, INTER_LINEAR) with BORDER_CONSTANT 0 (propagate :166-176; OpenCV
 *    remapBilinear: coordinates rounded to 1/32 pixel, float 4-tap weights); flow_uv [h,w,2] = (u, v).
 *  - udet_post_blend: y = a * x / (max(x) + 1e-8) + b * y, optionally followed by y /= (max(y) + 1e-8) (propagate :177-184).
 *  - udet_post_gauss1d: one axis of scipy.ndimage.gaussian_filter (mode 'reflect'), weights k[2r+1] on the device (crf_refine :113).
 *  - udet_post_dense_crf: DenseCRF2D + setUnaryEnergy + addPairwiseBilateral(sxy, srgb, rgbim, compat) + inference(iters)
 *    (crf_refine.py:110-130; Kraehenbuehl & Koltun 2011, mean field with one bilateral Potts term, symmetric normalisation), the
 *    Gaussian kernel evaluated exactly inside a (2*radius+1)^2 window; unary [2,h,w] energies, image_rgb uint8 [h,w,3],
 *    q [2,h,w] marginals out. */
int udet_post_border_mean(const float* s, int n, int h, int w, double* out, void* stream);
int udet_post_bytescale(const double* src, int ld, int y0, int x0, int h, int w, unsigned char* dst, void* stream);
int udet_post_resample_u8(const unsigned char* src, int h, int w, unsigned char* dst, int oh, int ow, const int* kk, const int* bounds,
                          int ksize, int axis, void* stream);
int udet_post_place(const unsigned char* patch, int hh, int ww, int y0, int x0, int h, int w, double* canvas, void* stream);
int udet_post_minmax_norm(const double* score, int n, double* out, void* stream);
int udet_post_remap(const float* src, const float* flow_uv, float* dst, int h, int w, void* stream);
int udet_post_blend(const float* x, float a, float* y, float b, int n, int renorm, void* stream);
int udet_post_gauss1d(const double* src, double* dst, int h, int w, const double* k, int r, int axis, void* stream);
size_t udet_post_crf_workspace_bytes(int h, int w);
int udet_post_dense_crf(const float* unary, const unsigned char* image_rgb, int h, int w, float sxy, float srgb, float compat, int iters,
                        int radius, float* q, void* workspace, size_t workspace_bytes, void* stream);

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


/* ---- per-stage entry points of the loss / optimizer tail (parity tests, callers that keep their own graph) ------------
 * `workspace` (device, 8-byte aligned) >= udet_stage_workspace_bytes(n) holds the deterministic two-stage reduction partials. */
size_t udet_stage_workspace_bytes(int n);
/* preprocess_flow_batch(flow): models/utils/flow_utils.py:5-12 -- per sample and channel (f - mean) / sqrt(var), population
 * variance, no epsilon.  flow, out [n,h,w,2]. */
int udet_flow_normalize(const float* flow, float* out, int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
/* charbonnier_loss(gt_flows, pred_flows, masks, cbn): models/utils/loss_utils.py:34-51.  out[n] (device) = sum over h,w,c of
 * ((gt - pred)^2 + 0.001^2)^cbn * mask; masks [n,h,w,mask_channels] with 1 (broadcast) or 2 channels, or NULL (ones). */
int udet_charbonnier_loss(const float* gt_flows, const float* pred_flows, const float* masks, int mask_channels, int n, int h, int w,
                          float cbn, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* losses{} of models/adversarial_learner.py:141-204 from flow [b,h,w,2], mask [b,h,w,1] and the three recover predictions
 * pred3 [3b,h,w,2] (calls: masked, complement, image-only).  losses8 (device) in the order of :196-204; coef [b][4] (device) =
 * d generator_loss / d {R_b, D_b, Rc_b, Dc_b}, consumed by udet_losses_backward. */
int udet_losses_forward(const float* flow, const float* mask, const float* pred3, int b, int h, int w, float cbn, float epsilon,
                        float* losses8, float* coef, void* workspace, size_t workspace_bytes, void* stream);
/* tf.gradients of the two losses w.r.t. the predictions (models/utils/loss_utils.py:18 through :141-204):
 * which = 2: d recover_loss / d pred3 -> dpred [3b,h,w,2];  which = 1: d generator_loss / d pred3[0:2b] -> dpred [2b,h,w,2]
 * and the direct mask term -> dmask [b,h,w,1] (coef from udet_losses_forward). */
int udet_losses_backward(const float* flow, const float* mask, const float* pred3, const float* coef, int which, int b, int h, int w,
                         float cbn, float* dpred, float* dmask, void* stream);
/* train_op, second half (models/utils/loss_utils.py:22-31): g <- flag2[1] != 0 ? |U(-clip,clip)| (counter-based stream keyed by
 * (seed, step, index)) : clip(g, +-clip).  flag2: device {avg, flag} from udet_grad_absmean, or NULL (clip only). */
int udet_clip_or_noise(float* g, size_t n, float clip, const float* flag2, unsigned long long seed, long step, void* stream);
/* tf.train.AdamOptimizer(lr, beta1).apply_gradients (adversarial_learner.py:216; TF form lr_t = lr*sqrt(1-b2^t)/(1-b1^t)) on a
 * flat buffer; t = number of applies of the shared optimizer object including this one (>= 1). */
int udet_adam_step(float* w, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, long t,
                   void* stream);

/* ---- the step plan ------------------------------------------------------------------------
 * One plan = one (batch, shapes, flags) specialisation of the graph assembled by
 * AdversarialLearner.build_train_graph / build_test_graph / build_aug_test_graph
 * (models/adversarial_learner.py:72-258, 450-523, 525-592).  The caller owns one workspace of
 * udet_workspace_bytes() bytes (256-byte aligned) that holds packed weights, every activation,
 * every activation-gradient and all scratch; udet_buffer_info() names its regions so the host
 * wrapper can view them as tensors without copies.  Weights, gradients and Adam slots are flat
 * fp32 buffers per network in TF variable order (udet_param_info), owned by the caller; the
 * gradient buffers are the RCCL all-reduce payload.
 * net / which: 0 = pwcnet (frozen), 1 = generator "MaskNet", 2 = recover "FlownetS"; which=3 both. */
typedef struct udet_plan udet_plan;
typedef struct {
  int batch;                 /* frame pairs per GPU (common_flags.py:8) */
  int in_h, in_w;            /* PWC-Net input size: 384 x 640 after the readers' resize (multiples of 64) */
  int img_h, img_w;          /* generator / recover size: flags img_height/img_width = 192 x 384 */
  float flow_normalizer;     /* 80   (common_flags.py:10) */
  float cbn;                 /* 0.5  (common_flags.py:17) */
  float epsilon;             /* 75   (common_flags.py:18) */
  float lr, beta1, beta2, adam_eps; /* 1e-4, flag beta1=0.9, 0.999, 1e-8 (adversarial_learner.py:216) */
  float clip;                /* 0.2  (adversarial_learner.py:227,233) */
  unsigned long long noise_seed;    /* stream of the escape-noise branch (loss_utils.py:7-10,19-26) */
  int conv_fp16;             /* 0 (default): fp32 MFMA, the reference's arithmetic.  1: BASELINE.json configs[4] -- the convolution
                              * GEMMs (forward, backward-data, backward-filter) multiply in fp16 with fp32 accumulation
                              * (v_mfma_f32_32x32x8_f16; gradient operands scaled by 4096 against underflow); tensors, losses,
                              * reductions and the optimizer stay fp32.  Not a reference capability: parity tolerance 2e-2.
                              * Overflow guard: a gradient operand beyond 16 (65504 / 4096) becomes inf in fp16 and reaches the
                              * flat gradient buffer as inf / NaN; udet_apply counts the non-finite gradient values on the
                              * device and DROPS the update when there are any (weights / Adam slots untouched); the next
                              * forward / backward call on the plan that finds the count returns UDET_ERR_OVERFLOW -- once per
                              * report, BEFORE it enqueues anything, so the caller simply re-issues the call (what
                              * trainer.train_step does, counting the event) -- and udet_fp16_overflow_count synchronises and
                              * returns the number of dropped updates.  No report is lost when the host never synchronises:
                              * udet_apply consumes the report of the network's previous apply before it re-uses the slot.
                              * The shared Adam step count advances for a dropped update too (the host cannot know at enqueue
                              * time): one bias-correction step is skipped. */
} udet_config;

int udet_plan_create(const udet_config* cfg, udet_plan** out);
void udet_plan_destroy(udet_plan* plan);
size_t udet_workspace_bytes(const udet_plan* plan);
/* zero-fills the workspace and uploads the static tables; call once per workspace (synchronises the stream) */
int udet_plan_init(udet_plan* plan, void* workspace, void* stream);

int udet_param_count(int net);
size_t udet_param_total(int net);
int udet_param_info(int net, int index, const char** name, int* rank, int* shape4, size_t* offset_floats);
int udet_buffer_count(const udet_plan* plan);
int udet_buffer_info(const udet_plan* plan, int index, const char** name, size_t* offset_bytes, int* dims_nhw_ld);

/* re-layout of the TF-format weights for the kernels (K-padded, BN folded, transposed for dgrad).
 * pwc: once per checkpoint; trainable: after every optimizer apply (udet_train_step does it). */
int udet_pack_pwc(udet_plan* plan, const float* w_pwc, void* workspace, void* stream);
int udet_pack_trainable(udet_plan* plan, const float* w_gen, const float* w_rec, void* workspace, void* stream);

/* ModelPWCNet().predict_from_img_pairs(img1, img2): models/PWCNet/model_pwcnet.py:61-76 -> nn() :599-649.
 * img1,img2 [B,in_h,in_w,3] in [-0.5,0.5]; result in buffer "flow_full" [B,in_h,in_w,2]. */
int udet_pwc_forward(udet_plan* plan, const float* img1, const float* img2, void* workspace, void* stream);
/* adversarial_learner.py:83-204: PWC flow, legacy resize of image/flow, flow/normalizer, generator_net,
 * `ncalls` recover_net invocations (3 = train graph, 1 = test graph :509-513, 0 = aug-test graph :579),
 * and (ncalls==3) the 8 entries of losses{} (:196-204) into buffer "losses".
 * Results: buffers "image","flow","mask","pred" ([ncalls*B,...]). */
int udet_forward(udet_plan* plan, const float* img1, const float* img2, int ncalls, void* workspace, void* stream);
/* Cross-step pipelining.  PWC-Net is frozen (adversarial_learner.py:211-214), so the flow of the NEXT pair does not
 * depend on this step's optimizer update: udet_prefetch_flow enqueues PWC flow + the two resizes of (img1,img2) on the
 * plan's side streams, forked from `stream` at call time, into staging buffers ("image.next","flow.next"); it runs
 * concurrently with whatever is enqueued on `stream` afterwards (typically udet_backward of the current step).
 * udet_forward_prefetched joins it, moves the staging buffers into "image"/"flow" and continues like udet_forward.
 * img1/img2 must stay valid until that join.  One prefetch may be pending at a time. */
int udet_prefetch_flow(udet_plan* plan, const float* img1, const float* img2, void* workspace, void* stream);
int udet_forward_prefetched(udet_plan* plan, int ncalls, void* workspace, void* stream);
/* the first half of udet_forward_prefetched alone (join + staging -> "image"/"flow"); follow it with the NEXT
 * udet_prefetch_flow and then udet_forward_from_flow: the next pair's PWC flow then also overlaps this step's forward */
int udet_prefetch_consume(udet_plan* plan, void* workspace, void* stream);
/* same but starting from caller-filled "image" and "flow" buffers (generator_net/recover_net surface, nets.py:4,45) */
int udet_forward_from_flow(udet_plan* plan, int ncalls, void* workspace, void* stream);
/* generator_net(images, flows) alone (models/nets.py:4-42): reads "image","flow", writes "mask" (flow standardisation
 * of models/utils/flow_utils.py:5-12 included). */
int udet_generator_forward(udet_plan* plan, void* workspace, void* stream);
/* generator_net's own contract (models/nets.py:4-42): `flows` arrives ALREADY standardised (the call site passes
 * preprocess_flow_batch(flow), adversarial_learner.py:99-105).  Reads the caller-packed buffer "gen.in" ([.,.,.,8]: image 3,
 * standardised flow 2, zeros), writes "mask". */
int udet_generator_layers(udet_plan* plan, void* workspace, void* stream);
/* recover_net(img1, flow_masked, mask) alone (models/nets.py:45-110) on n*B samples whose inputs the caller packed into
 * "rec.imgin" ([.,.,.,4]: image, 0) and "rec.fin" ([.,.,.,4]: flow_masked(2), 1, 1-mask); writes "pred". */
int udet_recover_forward(udet_plan* plan, int n, void* workspace, void* stream);
/* optimizer.compute_gradients of losses['generator'] w.r.t. MaskNet and/or losses['recover'] w.r.t. FlownetS
 * (models/utils/loss_utils.py:18; adversarial_learner.py:211-234) into the flat gradient buffers. */
int udet_backward(udet_plan* plan, int which, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec,
                  void* workspace, void* stream);
/* Data-parallel overlap: makes `stream` (a communication stream) wait until the flat gradient buffer of `net` written by the
 * last udet_backward is final -- with which=3 the recover gradients are final while the longer generator-loss pass still
 * runs, so their all-reduce overlaps it. */
int udet_stream_wait_grads(udet_plan* plan, int net, void* stream);
/* the two passes alone: train_generator_op / train_recover_op's compute_gradients (adversarial_learner.py:224-234) */
int udet_generator_backward(udet_plan* plan, const float* w_gen, float* g_gen, void* workspace, void* stream);
int udet_recover_backward(udet_plan* plan, const float* w_rec, float* g_rec, void* workspace, void* stream);
/* train_op, first half (loss_utils.py:19-21): out2 (device) = {mean over the variables of mean|g_v|, that < 1e-5 ? 1 : 0};
 * g is the flat gradient buffer of `net` (1 or 2). */
int udet_grad_absmean(udet_plan* plan, int net, const float* g, float* out2, void* workspace, void* stream);
/* the rest of train_op (loss_utils.py:19-32): clip +-0.2 / escape noise (generator only), Adam apply with the
 * shared beta-power accumulators; g is overwritten with the clipped gradient. */
int udet_apply(udet_plan* plan, int net, float* w, float* g, float* m, float* v, void* workspace, void* stream);
/* conv_fp16 plans: synchronises the pending overflow reports and returns how many optimizer updates were dropped so far because
 * their gradients held non-finite values (0 for fp32 plans); never an error by itself */
long udet_fp16_overflow_count(udet_plan* plan);
/* Number of optimizer updates applied so far (ONE Adam object for both networks: shared beta powers, adversarial_learner.py:216).
 * conv_fp16 plans: an update dropped by the overflow guard does not count.  The drop is known on the host only when the apply has
 * run, so both calls first wait for the plan's pending overflow reports (a stream synchronisation in that mode only) and book them:
 * the count returned -- e.g. into a checkpoint -- equals the updates that really happened, and a count set here is not decremented
 * afterwards by the report of an apply that preceded the call.  Applies ENQUEUED between a dropped update and its report used the
 * advanced count for their bias correction (at most the other network's apply and this network's next): a transient of one step. */
long udet_get_adam_step(const udet_plan* plan);
void udet_set_adam_step(udet_plan* plan, long t);
/* pack + forward + backward + apply for `which` on one GPU (no gradient exchange) */
int udet_train_step(udet_plan* plan, int which, const float* img1, const float* img2, float* w_gen, float* w_rec,
                    float* g_gen, float* g_rec, float* m_gen, float* v_gen, float* m_rec, float* v_rec, void* workspace,
                    void* stream);

/* Autotuner (the MI355X analogue of TF's cuDNN autotune the reference relies on implicitly): runs one untimed
 * forward + both backward passes over random data; every distinct convolution problem of the plan times its candidate
 * kernel configurations on `stream` and the fastest is cached process-wide (keyed by problem shape) for all later
 * launches.  Needs packed weights (udet_pack_pwc / udet_pack_trainable); overwrites g_gen / g_rec and re-zeroes the
 * activation regions of the workspace.  Optional: without it the built-in heuristics choose the configurations.
 * Every winner's output on the tuning data is compared with the built-in configuration's before it is cached (a
 * configuration that differs is rejected and reported on stderr); the comparison buffers are the one temporary device
 * allocation the library makes, freed before udet_autotune returns. */
int udet_autotune(udet_plan* plan, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec, void* workspace,
                  void* stream);
int udet_tuned_shapes(void);
/* The tuned configurations as a text file (keys are hashes of the problem shapes): a later process loads them instead of
 * tuning again, e.g. a rocprofv3 trace that should contain timed steps only.  udet_tune_load returns the number of entries
 * read (>= 0) or an error code.  The header line carries the build's tuning ABI: a file of another build is rejected.  Loaded
 * entries are not trusted blindly: at launch time a cached configuration is used only if its tile / kernel family is
 * instantiated, its split count is clamped to the launch's capacity, and anything else falls back to the built-in choice. */
int udet_tune_save(const char* path);
int udet_tune_load(const char* path);
/* autotuner winners rejected because their output differed from the built-in configuration's (0 on a healthy build) */
int udet_tune_rejected(void);
/* Measurement aid (bench.py): between begin/end every convolution / warp-cost-volume launch group is timed on the launch
 * stream.  out[cat*5 + {0..4}] = {groups, kernel ms, algorithmic FLOPs, algorithmic bytes, bracket ms} for cat 0 conv fwd,
 * 1 conv dgrad, 2 conv wgrad, 3 (unused), 4 warp + cost volume.  "kernel ms" sums the kernels' own start -> stop times (event
 * pairs carried by the dispatch packets: what rocprofv3 --kernel-trace reports); "bracket ms" is hipEventRecord before / after
 * each group, which additionally contains the event packets and dispatch gaps. */
int udet_profile_begin(udet_plan* plan);
int udet_profile_end(udet_plan* plan, double* out, int ncat, void* stream);

/* Concurrency switch.  A plan runs independent chains of a step on its own side streams, forked from / joined to the caller's
 * stream with events.  on = 0 collapses every lane onto the caller's stream: the plain program order, results
 * bit-identical (tests/test_autotune_gpu.py), used for serial kernel traces.  Takes effect from the next call on the plan.
 *
 * Environment variables the library reads -- all diagnostic, none on a launch path, none changes a result:
 *   UDET_SERIAL=1      read once by udet_plan_create: the plan starts with udet_plan_set_concurrent(plan, 0);
 *   UDET_TUNE_LOG=1    read by the autotuner (udet_autotune): one stderr line per tuned problem shape;
 *   UDET_PROF_DUMP=F   read by udet_profile_end: appends one CSV line per launch group of the measurement pass to F.
 * Kernel-selection forcing for tests lives in a separate library (include/udet_debug.h, libudet_debug.so). */
int udet_plan_set_concurrent(udet_plan* plan, int on);

/* Lane placement.  ROCm maps every stream of a process onto one of GPU_MAX_HW_QUEUES (default 4) hardware queues when the stream is
 * created, and two streams on one queue execute in submission order -- which lanes of a plan share a queue changes the step time by up
 * to 20 % and depends on every other stream the process (PyTorch's pool, RCCL) created before.  A plan therefore owns candidate
 * streams (eight; up to 32 are drawn while fewer than three independent queues have been found) and, the first time it is driven from a given caller stream, probes (a 200 us spin kernel on one stream, an empty kernel on
 * the other) which candidates run concurrently with the caller's stream and with each other, then lays its six lanes out on
 * four independent queues: {0 = the caller's stream, 2} {1} {3} {4, 5}; with fewer independent queues lanes are merged.  That first
 * call synchronises the device once (~3 ms).  udet_plan_lane_queues places the lanes for `stream` if that has not happened yet and
 * writes the queue-group index of each of the six lanes (0 = the caller's queue) to queue[6]; returns the number of independent
 * queues in use (4 on a default runtime), < 0 on error.  Do not raise GPU_MAX_HW_QUEUES: above four queues a cross-stream dependency
 * costs 79 us instead of 12 (profiles/r03_hop_bench.txt). */
int udet_plan_lane_queues(udet_plan* plan, void* stream, int* queue);
/* The same layout without the timing probe, for a host that knows its streams: side_streams[0..n-1] (n = 0..3, created by the host
 * and kept alive as long as the plan is driven from `stream`) are taken to sit on n distinct hardware queues, none of them `stream`'s;
 * lanes {1} {3} {4, 5} go to them in that order (fewer streams: merged exactly as a probe that finds fewer queues would).  Replaces
 * any placement the plan holds for `stream`.  A probe is a ~200 us spin kernel against an empty kernel, repeated up to three times
 * when it sees no overlap (a positive result is proof, a negative one may be a slow host or a GPU shared with another process);
 * a deployment that cannot tolerate that heuristic pins. */
int udet_plan_pin_lanes(udet_plan* plan, void* stream, void* const* side_streams, int n);

#ifdef __cplusplus
}
#endif
#endif /* UDET_H */
