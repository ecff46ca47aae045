#pragma once
#include "common.h"
namespace udet {
void same_pad(int in, int k, int s, int d, int* before, int* out);
void conv_setup_fwd(ConvParams& p, int N, int H, int W, int kh, int kw, int s, int d);
int conv_dgrad_classes(int s, int H, int W);
void conv_force_config(int bm, int bn, int ks);
// direct kernels for the 2-channel heads (conv_thin.hip): eligibility of a launch / the launches
bool conv_thin_n_ok(const ConvParams& p);
bool conv_thin_k_ok(const ConvParams& p);
int launch_conv_thin_n(const ConvParams& p, hipStream_t stream);
int launch_conv_thin_k(const ConvParams& p, hipStream_t stream);
// Winograd F(2x2,3x3) family (conv_wino.hip): eligibility (3x3 taps of uniform dilation, stride 1, Kc % 8 == 0, p.wino_u set), the
// launch (variant bit 0: 0 = 64 tiles x 64 channels per workgroup, 1 = 128 tiles x 32 channels; bit 1: the eight-wave form, two waves
// per SIMD; ks K slices through the split-K slabs)
// and the weight transform (mode 7 / 8 of PackJob) as a one-off launch
bool conv_wino_geometry(const ConvParams& p, int* dil, int widx_at[9]);
bool conv_wino_ok(const ConvParams& p);
bool conv_wino_variant_ok(const ConvParams& p, int variant);
int conv_wino_np(int cout);  // padded N extent of the transformed weights
size_t conv_wino_floats(int Kc, int cout);
long conv_wino_workgroups(const ConvParams& p, int variant);
int conv_wino_max_ksplit(const ConvParams& p, int variant);
int launch_conv_wino(ConvParams& p, int variant, int ks, hipStream_t stream);
int launch_splitk_second_pass(const ConvParams& p, hipStream_t stream);  // conv_igemm.hip: sums p.partial's ksplit slabs + epilogue
int launch_wino_pack(const float* src, float* dst, int R, int C, int Kc, int np, int k_split, int k_gap, int transposed, hipStream_t stream);
// U straight from PACKED weights [tap][Kc][ldw] and the launch's own tap table (single-operator launches, tests)
int launch_wino_from_packed(const ConvParams& p, float* dst, int np, hipStream_t stream);
void conv_debug_f16(int on);  // fp16 multiplication in the single-operator launches (plans carry udet_config.conv_fp16)
int conv_debug_f16_on();
int conv_last_config();
void conv_set_tuning(int on);   // autotuner: while on, unseen problem shapes are timed and the best configuration cached
int conv_tuned_shapes();
void conv_clear_tuning();
int conv_tune_rejected();      // winners whose output differed from the reference configuration's (never cached)
void conv_tune_note_reject();
// tuning-time scratch (two device buffers of `floats` each, alive while tuning is on) and the max-abs comparison the tuners use
float* tune_scratch(size_t floats, int which);
bool tune_compare(const float* a, const float* b, size_t n, hipStream_t stream, float* diff_out, float* scale_out);
void wgrad_set_tuning(int on);
// Winograd-domain filter gradient (conv_wgrad_wino.hip): eligibility of a (plain-view) launch, the slice count for `wanted` (0: one
// workgroup per CU), the GEMM launch into conv_wgrad.hip's slab layout
bool wgrad_wino_ok(const WgradParams& p);
int wgrad_wino_slices(const WgradParams& p, int wanted);
int launch_wgrad_wino(const WgradParams& q, int slices, int ldn, hipStream_t stream);
int wgrad_last_config();  // split count | variant << 20 of the most recent launch_wgrad_T (variant 3: the Winograd-domain family)
void wgrad_force(int nsplit, int dma);  // debug hook: nsplit > 0 pins the split count (clamped to the capacity), dma 0 / 1 / 2 the staging variant, 3 the Winograd-domain family where eligible (-1: as tuned)
int wgrad_tuned_shapes();
void conv_tune_dump(FILE* f);
void conv_tune_put(unsigned long long key, int bm, int bn, int ks, int ws, int fold, int tail);
void wgrad_tune_dump(FILE* f);
void conv_pair_tune_dump(FILE* f);  // "p <pair key> bm bn ks ws" (ws < 0: the two problems stay apart)
void conv_pair_tune_put(unsigned long long key, int bm, int bn, int ks, int ws);
int conv_pair_tuned_shapes();
void conv_pair_clear_tuning();
void conv_force_pair(int on);  // test hook: 1 = pair every compatible couple, 0 = never, -1 = as tuned / heuristic
int conv_last_pair();          // 1: the most recent launch_conv_pair went out as ONE launch
// bumped whenever a kernel family, a tile set or a problem key changes: tuning files of another build are rejected (udet_tune_load)
#define UDET_TUNE_ABI 6
void wgrad_tune_put(unsigned long long key, int cfg);  // debugging / tuning hook: bm == 0 and ks < 0 restore the heuristics
bool conv_setup_dgrad(ConvParams& p, int cls, int N, int H, int W, int kh, int kw, int s, int d);
int launch_pack_weights(const float* src, float* dst, int T, int R, int C, int Kc, int ldw, int k_split, int k_gap,
                        int mode, const float* scale, hipStream_t stream);
int launch_pack_taps_into_n(const float* src, float* dst, int T, int R, int C, int Kc, int ldz, int k_split, int k_gap,
                            int transposed, hipStream_t stream);
int launch_pack_jobs(const PackJob* jobs_dev, int njobs, const float* wsrc, float* ws, float bn_c, hipStream_t stream);
int launch_fold_bn(const float* b, const float* gamma, const float* beta, float c, float* scale, float* bias_f, int n,
                   hipStream_t stream);
int launch_copy_channels(const float* src, int lds, int s_coff, float* dst, int ldd, int d_coff, long P, int C, float mul,
                         float add, hipStream_t stream);
int launch_warp(const float* img, const float* flow, int ldf, int f_coff, float flow_scale, float* out, int N, int H,
                int W, int C, int* dbg_idx, float* dbg_alpha, hipStream_t stream);
int launch_cost_volume(const float* c1, const float* wr, float* out, int ldo, int o_coff, int N, int H, int W, int C,
                       hipStream_t stream);
// fused warp -> cost volume -> slab (flow == null: no warp, level 6; c1_coff < 0: no c1 segment; warped_dbg: optional [N,H,W,C])
int launch_warp_cost_volume(const float* c1, const float* c2, const float* flow, int ldf, int f_coff, float flow_scale, float* out,
                            int ldo, int corr_coff, int c1_coff, float* warped_dbg, int N, int H, int W, int C, hipStream_t stream);
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace udet
