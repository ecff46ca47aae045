/* Test-only hooks (NOT part of the drop-in surface in udet.h, NOT exported by libudet.so): they live in libudet_debug.so, a
 * small library that links against libudet.so; tools/conv_bench.py and the kernel-family tests load it
 * (unsupervised_detection_amd/_devel.py) to pin one convolution kernel family / tile / split-K mode and to ask which one ran. */
#ifndef UDET_DEBUG_H
#define UDET_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* force (bm, bn, split-K) for every convolution launch (bm bit 16: non-specialised kernel, bit 17: LDS-DMA staging,
 * bit 18: tile-resident kernel with bm & 0xffff = tile height, bit 19: self-staging LDS-DMA kernel, bit 20: split-K summed by a
 * second launch, bit 21: split-K summed by the last-arriving workgroup, bit 22 / 23: LDS-DMA kernel with a 3 / 4 stage
 * ring, bit 24: the direct kernels for 2-channel inputs / outputs where the launch is eligible); (0,0,-1) restores.  A forced family a launch is not
 * eligible for falls back to the built-in choice.  Process-global state: never call it from product code. */
void udet_debug_force_conv(int bm, int bn, int ks);
/* fp16 multiplication (fp32 accumulation) in the single-operator convolution entry points; plans take it from
 * udet_config.conv_fp16.  Process-global state: tests only. */
void udet_debug_conv_fp16(int on);
/* while on, the first single-op launch of every distinct problem shape times its candidate configurations and caches the winner
 * (what udet_autotune does for a plan); tools/conv_bench.py / wgrad_bench.py use it to measure the tuned kernels stand-alone */
/* filter gradient: nsplit > 0 pins the number of pixel slices (clamped to the workspace capacity), dma = 0 / 1 / 2 the staging variant
 * (register-staged / LDS-DMA with a 2- / 3-stage ring / 3: the Winograd-domain family; -1: as tuned); nsplit = 0 restores the tuned / heuristic choice */
void udet_debug_force_wgrad(int nsplit, int dma);
/* what the most recent filter-gradient launch ran: K slices | variant << 20 (0 register-staged, 1 / 2 LDS-DMA, 3 the Winograd-domain family of
 * conv_wgrad_wino.hip; dma = 3 above forces it where a launch is eligible: 3x3 stride-1, whole 64-channel blocks) */
int udet_debug_last_wgrad(void);
/* plans created after this call: the recover decoder's backward-data pass takes the low-resolution ("up-conv algebra") form on every
 * level whose source has at least `v` pixels (batch included); v < 0 restores the default of 8192.  Tests use 0 on small plans. */
void udet_debug_upb_min_pixels(long v);
void udet_debug_set_tuning(int on);
/* pair launches (two convolutions of the same geometry in ONE launch: the recover net's two encoders, csrc/conv_igemm.hip launch_conv_pair):
 * on = 1 pairs every compatible couple whatever the tuner thinks, 0 never pairs, -1 restores (tuned / heuristic choice);
 * udet_debug_last_pair: 1 when the most recent pair call went out as one launch */
void udet_debug_force_pair(int on);
int udet_debug_last_pair(void);
/* two forward convolutions (same shape parameters; cin a multiple of 8; batches na / nb; HWIO weights) through the pair launcher;
 * workspace >= (2 * k*k*cin*roundup(cout,4) + 4 Mi + 8192) floats */
#include <stddef.h>
int udet_debug_conv2d_pair(const float* xa, const float* xb, const float* wa, const float* wb, const float* ba, const float* bb, float* ya, float* yb,
                           int na, int nb, int h, int w, int cin, int cout, int k, int stride, int dilation, int act, float alpha, void* workspace,
                           size_t workspace_bytes, void* stream);
/* (The experiment knobs of earlier rounds -- lane choices and the work-skipping ablation mask -- are no longer reachable from any library
 * that links against libudet.so: they are compiled out of it.  `make -C unsupervised_detection_amd/csrc exp` builds libudet_exp.so, the same
 * sources with -DUDET_EXPERIMENT, which exports udet_exp_knob(id, value); tools/knob_bench.py is its only user.  csrc/plan.h lists the ids.) */
/* what the most recent convolution launch actually ran: family (0 plain, 1 wave-specialised, 2 LDS-DMA, 3 tile-resident,
 * 6 self-staging LDS-DMA, 7 / 8 direct kernel for two input / two output channels) | tile rows << 8 | split count << 20 | folded split-K << 28 */
int udet_debug_last_conv(void);
#ifdef __cplusplus
}
#endif
#endif
