// Internal definitions shared by the HIP translation units of libudet.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/udet.h"

namespace udet {

// ---------------------------------------------------------------- errors ----
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
#define UDET_HIP(x)                                   \
  do {                                                \
    hipError_t _e = (x);                              \
    if (_e != hipSuccess) return udet::hip_fail(_e, #x); \
  } while (0)
#define UDET_TRY(x)            \
  do {                         \
    int _s = (x);              \
    if (_s != 0) return _s;    \
  } while (0)

// ------------------------------------------------------------- launches ----
// Every hot kernel is launched through UDET_LAUNCH.  While a measurement pass is active (udet_profile_begin) the launch
// carries a start / stop event pair on its own dispatch packet (hipExtLaunchKernelGGL): the elapsed time between them is
// the kernel's execution time on the device -- the figure `rocprofv3 --kernel-trace` reports -- without the event-record
// packets and dispatch gaps that a hipEventRecord bracket around the launch adds (several microseconds per bracket).
struct LaunchSink {
  hipEvent_t* ev;  // pairs
  int n, cap;
};
extern thread_local LaunchSink* g_launch_sink;
bool launch_sink_next(hipEvent_t* a, hipEvent_t* b);
#define UDET_LAUNCH(kernel, grid, block, shmem, stream, ...)                                             \
  do {                                                                                                   \
    hipEvent_t ea_, eb_;                                                                                 \
    if (udet::g_launch_sink && udet::launch_sink_next(&ea_, &eb_))                                       \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ea_, eb_, 0, __VA_ARGS__);               \
    else                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                               \
  } while (0)

// ----------------------------------------------------------- activations ----
enum Act { ACT_NONE = UDET_ACT_NONE, ACT_LEAKY = UDET_ACT_LEAKY, ACT_ELU = UDET_ACT_ELU };

// expm1(v) for v <= 0 (the negative branch of ELU, tf.nn.elu).  libdevice's expm1f is ~40 instructions and sat in every generator layer's
// store loop: +4.4 us on a 48 x 96 x 128 output, +9 us on 96 x 192 x 64 (tools/wino_fixed_cost.py, round 6).  Here: exp(v) - 1 where that
// does not cancel (v <= -0.25: exp(v) <= 0.78, error ~1e-7 of the result), the degree-6 Taylor polynomial above (remainder v^7 / 5040 <
// 1.3e-8 at |v| = 0.25).  Max error against float64 expm1: 2.5e-7 absolute on [-88, 0] (tests/test_ops_gpu.py::test_elu_accuracy).
__device__ __forceinline__ float elu_negative(float v) {
  const float e = __expf(v) - 1.0f;
  const float t = v * (1.f + v * (0.5f + v * (0.16666667f + v * (0.041666668f + v * (0.0083333338f + v * 0.0013888889f)))));
  return v > -0.25f ? t : e;
}
// (one uniform test per call -- ELU -- instead of a chain of `if (act == ...)`: inside unrolled store loops hipcc turned that chain into
// five scalar branches per element; none / leaky share one formula with slope = 1 for none.  Values unchanged, bit for bit.)
__device__ __forceinline__ float act_fwd(float v, int act, float alpha) {
  const float slope = act == ACT_LEAKY ? alpha : 1.f;
  float r = v > 0.f ? v : v * slope;
  if (act == ACT_ELU) r = v > 0.f ? v : elu_negative(v);
  return r;
}
// derivative expressed through the saved *output* a = act(u)
// (TF LeakyReluGrad: features>0 ? g : alpha*g; EluGrad: out<0 ? (out+1)*g : g): a > 0 ? 1 : fma(a, ue, us) with
// (ue, us) = (0, 1) none, (0, alpha) leaky, (1, 1) ELU -- branch-free
__device__ __forceinline__ float act_dfo(float a, int act, float alpha) {
  const float ue = act == ACT_ELU ? 1.f : 0.f, us = act == ACT_LEAKY ? alpha : 1.f;
  return a > 0.f ? 1.f : fmaf(a, ue, us);
}

// ------------------------------------------------ implicit-GEMM conv op ----
// One launch computes, for every output sub-grid pixel q=(n,qy,qx) and channel co,
//   y[n, qy*osy+ooy, qx*osx+oox, y_coff+co] (=|+=) epi( sum_t sum_k X(n, qy*isy+t.dy, qx*isx+t.dx, k) * Wp[t.widx][k][co] )
// X(..) = x[.., x_coff+k] (zero outside the grid; optional NN x2 upsample of x; optional
// multiplication by act'(xa[..]) for the backward-data pass).  This single form covers the
// forward convolutions (any k/stride/dilation, TF SAME padding), stride-1 dgrad, stride-2
// dgrad and conv2d_transpose (one launch per output parity class).
#define UDET_MAX_TAPS 49
struct ConvTap {
  int dy, dx, widx;
};

// n / d for 0 <= n < 2^31 with one multiply-high (d >= 1), built on the host
struct FastDiv {
  unsigned d, mul, sh;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d; f.mul = 0; f.sh = 0;
  if (d > 1) {
    unsigned l = 0;
    while ((1u << l) < d) ++l;  // ceil(log2 d)
    const unsigned long long pw = 1ull << (31 + l);
    f.mul = (unsigned)((pw + d - 1) / d);
    f.sh = l - 1;
  }
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh); }

// One segment of a segmented launch (ConvParams::nseg): a sub-grid of the output with its own tap list
struct ConvSeg {
  int oy, ox;   // output pixel of the segment's quotient pixel (0, 0)
  int h, w;     // quotient pixels; pixel (qy, qx) writes (oy + qy * osy, ox + qx * osx)
  int prow0;    // row of the segment's first pixel in the launch's flat row space (split-K slabs); filled by launch_conv
  FastDiv fd_hw, fd_w;  // filled by launch_conv
};
#define UDET_MAX_SEGS 16

struct ConvParams {
  // A operand (input activations / output-gradients), NHWC with channel stride ldx
  const float* x;
  int ldx, x_coff;
  int N, H, W;     // logical input grid (already x2 when up_shift==1)
  int up_shift;    // 1: x is stored at (H/2,W/2) and read through nearest-neighbour x2
  const float* xa; // optional saved activation (same layout as x) -> X *= act'(xa)
  int xact;
  float xalpha;
  // B operand: packed weights [ntaps_total][Kc][ldw]
  const float* wp;
  int Kc, ldw;
  const float* bias;  // [Cout] or null
  // output
  float* y;
  int ldy, y_coff, Cout;
  int OH, OW;            // full output grid
  int OHq, OWq;          // sub-grid handled by this launch
  int osy, osx, ooy, oox;
  int isy, isx;
  int ntaps;
  ConvTap taps[UDET_MAX_TAPS];
  // Output-parity classes merged into one launch (stride-2 dgrad / conv2d_transpose): class c = (py,px) = (c>>1, c&1)
  // owns taps [cls_tap[c], cls_tap[c+1]) and writes y[.., qy*2+py, qx*2+px, ..].  ncls == 1: one class, all taps,
  // (ooy,oox) as given.
  int ncls;
  int cls_tap[5];
  // Segmented launch (nseg > 0): instead of the parity classes of ONE quotient grid the launch covers nseg <= 16 output sub-grids, each
  // with its own taps [seg_tap[s], seg_tap[s + 1]) of the DEVICE table tap_tab (offsets relative to the segment's own quotient pixels;
  // widx may address several weight sets laid out back to back behind wp).  Inside the kernels a segment is handled exactly like a
  // class; isy / osy are shared.  Only the implicit-GEMM families take such launches (the recover decoder's up-conv algebra, plan_exec.hip:
  // interior / last row / last column / corner of the output in one launch).  ncls, OHq, OWq, ooy, oox, taps[] are unused.
  int nseg;
  ConvSeg seg[UDET_MAX_SEGS];
  int seg_tap[UDET_MAX_SEGS + 1];
  const ConvTap* tap_tab;
  int Mall;               // rows of the launch's flat row space: ncls * N * OHq * OWq, or the sum over the segments; filled by launch_conv
  FastDiv fd_ohw, fd_ow;  // filled by launch_conv
  int kfast;              // LDS-DMA kernels: stages never straddle taps (no up-sampled read; bit 0: Kc >= 32, 32-wide stages, bit 1: Kc >= 16,
                          // 16-wide stages): wave-uniform K cursor; filled by launch_conv
  const float* zero16;    // >= 16 bytes of zeros in device memory (source of halo / tail lanes of the LDS-DMA kernel); may be null
  // epilogue
  int act;
  float alpha;
  const float* res;  // residual added after the activation (forward skip / backward skip-gradient)
  int ldres, res_coff;
  float* y2;  // optional second output: activation before the residual add
  int ldy2, y2_coff;
  int accumulate;  // y += result
  // backward-data launches may also emit dU = (final y) * act'(saved activation) for output channels [u_c0,u_c1): the
  // operand the NEXT backward-data / backward-filter launches consume, so that they need no act' on load
  float* uo;
  int ldu, u_coff;
  const float* ua;
  int ldua, ua_coff, uact;
  float ualpha;
  int u_c0, u_c1;
  // split-K
  int ksplit;
  float* partial;  // [ksplit][Mtot][ldp]
  size_t partial_cap;  // capacity of `partial` in floats
  int ldp;
  // fold != 0: the workgroup that draws an output tile's last ticket sums the slabs and runs the epilogue (no second launch).
  // tickets: >= UDET_MAX_TICKETS zero-initialised ints private to the launch stream (self-resetting); null disables folding.
  int fold;
  int* tickets;
  // tail split (LDS-DMA kernel; set by the launcher): x-blocks [0, tail_full) run unsplit, every later x-block is cut into
  // tail_ks K slices -- the launch's last, partly filled round of workgroups becomes tail_ks times as many short ones.  Slabs hold
  // the tail rows only: [tail_ks][Mall - tail_prow0][ldp].  tail_ks <= 1: off.
  int tail_full, tail_ks, tail_prow0;
  // f16 != 0 (udet_config.conv_fp16): the LDS-DMA kernels convert their fragments to fp16 (round to nearest even) and multiply with
  // v_mfma_f32_32x32x8_f16, accumulating in fp32; tensors stay fp32 in memory.  The x operand is scaled by f16_xscale on conversion
  // and the accumulators by 1 / f16_xscale before the epilogue: gradients (backward-data) use 4096 against fp16 underflow.
  int f16;
  float f16_xscale;
  // Winograd F(2x2,3x3) family (conv_wino.hip): the launch's 3x3 tap set pre-transformed, U = G g G^T laid out
  // [Kc / 8][16 positions][2 lane halves][wino_np][4 channels]; null: the family is not available for this launch
  const float* wino_u;
  int wino_np;
  // real input channels when fewer than Kc (the rest are zero padding of the tensor and zero rows of the packed weights); 0: Kc.
  // Lets the direct kernel for 2-channel inputs (conv_thin.hip) skip the padding.
  int kreal;
};
#define UDET_MAX_TICKETS 4096

int launch_conv(ConvParams& p, hipStream_t stream);
// two independent problems of the same geometry (batch / operands / epilogue may differ) in ONE launch where the LDS-DMA family can take
// both and the tuner found that faster; otherwise launch_conv(a), launch_conv(b).  a.partial == b.partial: the lane's split-K scratch.
int launch_conv_pair(ConvParams& a, ConvParams& b, hipStream_t stream);
// y[n, oy, ox, y_coff+co] = bias[co] + sum_{taps t of the pixel's parity class} Z[n, qy+dy_t, qx+dx_t, widx_t*Cout+co]
// (zero outside Z's grid); g describes taps / classes / output lattice exactly like a convolution launch
int launch_tap_gather(const ConvParams& g, const float* z, int ldz, hipStream_t stream);

// One job of the table-driven weight re-layout (see pack_jobs_kernel in conv_host.hip): offsets are in floats,
// src/gamma/beta/bias relative to the network's flat TF-order weight buffer, dst relative to the workspace.
struct PackJob {
  long src_off, dst_off, gamma_off, beta_off;  // gamma_off < 0: no BN fold
  long total;
  int T, R, C, Kc, ldw, k_split, k_gap, mode;  // mode 0 forward, 1 transposed, 2 bias (dst[c] = b*gamma*c + beta | b),
                                               // 3/4 taps-into-N: dst[kmap(r)][t*C+c] = src[t][r][c] (3) | src[t][c][r] (4)
                                               // 5/6 NN x2 + 3x3 as four 2x2 convolutions (T = 16 = class*4 + tap), forward / transposed
                                               // 7/8 Winograd U = G g G^T of a 3x3 filter, forward / backward-data (taps mirrored, [k=co][n=ci]):
                                               //     dst [Kc / 8][16][2][ldw = wino_np][4] (ConvParams::wino_u)
                                               // 9/10 recover decoder as four 3x3 convolutions on the ringed low-resolution grid, forward /
                                               //     backward-data: dst [36][Kc][ldw]; beta_off = row variant * 3 + column variant (conv_host.hip)
};

// --------------------------------------------------------------- wgrad ----
// dW[t][ci][co] = sum_q X(q@t)[ci] * (dY[q][co] * act'(ya[q][co]))
struct WgradParams {
  const float* x;
  int ldx, x_coff;
  int N, H, W, up_shift;
  int Cin;
  const float* dy;  // gradient wrt the layer *output* (post-activation)
  const float* ya;  // saved activation (same layout as dy) or null
  int ldy, y_coff, Cout;
  int yact;
  float yalpha;
  int OH, OW, isy, isx;
  // ycls != 0 (NN x2 + 3x3 as four 2x2 convolutions, see plan_exec.hip): tap t belongs to output parity class t >> 2 and pairs X
  // at its offset with dU on that class's sub-lattice of the (OHf, OWf) = (2 OH, 2 OW) grid; an M tile never straddles classes
  int ycls, OHf, OWf;
  int ntaps;
  ConvTap taps[UDET_MAX_TAPS];  // widx = tap index in the HWIO weight
  float* dw;        // [ntaps_total][Cin][Cout]  (HWIO, written, not accumulated)
  float* db;        // [Cout] or null
  float* partial;   // workspace
  size_t partial_floats;
  const float* zero16;  // >= 16 bytes of device zeros (LDS-DMA variant: source of halo / tail lanes); may be null
  // filled by launch_wgrad_T
  int swapped, oCin, oCout, bias_m;  // operand-swapped GEMM view for <=4-channel outputs (see launch_wgrad_T)
  int Cin4, Mpad;
  float* pbias;
  FastDiv fd_ohw, fd_ow;
  // generator finalisation (BN folded): dW = G*gs[co]; dgamma = c*(sum W*G + b*S); dbeta = S; db = gs*S
  const float* w;       // HWIO weights (for dgamma)
  const float* b;       // bias
  const float* gamma;   // null -> plain conv
  float* dgamma;
  float* dbeta;
  float bn_c;
  // f16 != 0: the LDS-DMA kernel multiplies in fp16 (v_mfma_f32_32x32x8_f16, fp32 accumulation); the gradient operand dU is scaled by
  // f16_yscale on conversion (fp16 underflow) and the partial sums by its inverse before they are stored
  int f16;
  float f16_yscale;
};
int launch_wgrad_T(WgradParams& p, int T, hipStream_t stream);
int launch_bn_finalize(float* dw, int T, int Cin, int Cout, const float* w, const float* b, const float* gamma, float bn_c, float* pd,
                       float* db, float* dgamma, float* dbeta, hipStream_t stream);
int launch_wgrad_up_combine(const float* deff, float* dw, int Cin, int Cout, hipStream_t stream);
size_t wgrad_partial_floats_needed(int T, int Cin, int Cout);

}  // namespace udet
