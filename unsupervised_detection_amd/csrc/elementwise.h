#pragma once
#include "common.h"
namespace udet {
int launch_resize_bilinear_fwd(const float* x, int ldx, int x_coff, int N, int H, int W, float* y, int ldy, int y_coff,
                               int OH, int OW, int C, float mul, float div, hipStream_t s);
int launch_resize_bilinear_bwd(const float* dy, int ldy, int y_coff, int N, int OH, int OW, float* dx, int ldx, int x_coff,
                               int H, int W, int C, int accumulate, hipStream_t s);
int launch_upb_ring(const float* x, int ld, int N, int H, int W, float* xh, hipStream_t s);
int launch_upb_ring_fold(const float* dxh, int ld, int N, int H, int W, float* dx, hipStream_t s);
int launch_share_samples(float* buf, long P, int ld, int coff, int C, int copies, hipStream_t s);
int launch_fold_samples(float* buf, long P, int ld, int coff, int C, int copies, hipStream_t s);
int launch_emit_du(const float* d, const float* a, float* u, long P, int ld, int coff, int C, int act, float alpha, hipStream_t s);
int launch_pool2x2_sum(const float* du, float* dx, int N, int H, int W, int C, hipStream_t s);
int launch_pack_pwc_input(const float* i1, const float* i2, float* x8, long P, hipStream_t s);
int launch_gen_input(const float* img, const float* f, double* part, float* gin, int B, long HW, hipStream_t s);
size_t flow_stats_doubles(int B);
int launch_mask_rec_inputs(const float* logits, const float* f, float* mask, float* fin, long P, int ncalls, hipStream_t s);
int launch_pack_imgin(const float* img, float* imgin, long P, int ncalls, hipStream_t s);
int launch_losses(const float* f, const float* mask, const float* pred, long HW, int B, float cbn, float eps,
                  float num_pixels, float* part, float* losses, float* coef, float* sums, hipStream_t s);
size_t loss_part_floats(int B);
int launch_rec_loss_bwd(const float* f, const float* mask, const float* pred, float* dpred, long BHW, float cbn,
                        float inv_np, hipStream_t s);
int launch_gen_loss_bwd(const float* f, const float* mask, const float* pred, const float* coef, float* dpred,
                        float* dmask, long HW, int B, float cbn, hipStream_t s);
int launch_mask_bwd(const float* dmask, const float* dfin, const float* f, const float* mask, float* dlogits, long P,
                    hipStream_t s);
int launch_grad_absmean(const float* g, const long* seg_off, const long* seg_len, int nvars, float* vmean, float thresh,
                        float* out, hipStream_t s);
int launch_adam(float* w, float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps, float clip,
                const float* flag, uint64_t seed, uint64_t step, hipStream_t s, int mode = 0, const int* skip = nullptr);
int launch_nonfinite_count(const float* g, long n, int* out, hipStream_t s);
int launch_flow_normalize(const float* f, double* part, float* out, int B, long HW, hipStream_t s);
int launch_charbonnier(const float* gt, const float* pred, const float* mask, int mc, int B, long HW, float cbn, float* part,
                       float* out, hipStream_t s);
int launch_crop_flip_resize(const void* src, int src_u8, int nearest, int N, int H, int W, int C, const int* prm, float* dst, int OH,
                            int OW, float div, float add, hipStream_t s);
int launch_mask_stats(const float* pred, const float* gt, int N, int H, int W, float threshold, float gt_threshold, double* out,
                      hipStream_t s);
int launch_fill_uniform(float* x, long n, uint64_t seed, float lo, float hi, hipStream_t s);
int launch_axpy(const float* x, float* y, long n, float a, int accumulate, hipStream_t s);
}  // namespace udet
