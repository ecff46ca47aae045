// HBM-bound glue kernels of the hot path: legacy bilinear resize (fwd / adjoint), NN x2 adjoint,
// flow standardisation, input packers, mask / recover-input assembly, Charbonnier losses and their
// gradients, clipped Adam.  Compiled with -ffp-contract=off so that the resize index math rounds
// like the reference's float32 graph.
#include "common.h"
#include "elementwise.h"

namespace udet {

static inline int grid_for(long total, int cap = 4096) {
  long nb = (total + 255) / 256;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// ---------------------------------------------------------------------------
// block reductions (256 threads = 4 wave64)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* sm /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// ---------------------------------------------------------------------------
// TF-1.13 ResizeBilinear, align_corners=False, no half-pixel centres
// (tf.image.resize_images / resize_bilinear: models/adversarial_learner.py:87-90,
//  models/nets.py:108, models/utils/convolution_utils.py:88, models/PWCNet/model_pwcnet.py:646)
//   scale=in/out ; src=i*scale ; lo=(int)src ; hi=min(lo+1,in-1) ; t=src-lo
//   top=tl+(tr-tl)*tx ; bot=bl+(br-bl)*tx ; out=top+(bot-top)*ty   then * mul / div
// ---------------------------------------------------------------------------
__device__ __forceinline__ void legacy_coord(int i, float scale, int in, int& lo, int& hi, float& t) {
  const float src = (float)i * scale;
  lo = (int)src;
  hi = min(lo + 1, in - 1);
  t = src - (float)lo;
}

__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(const float* __restrict__ x, int ldx, int x_coff, int N,
                                                                  int H, int W, float* __restrict__ y, int ldy, int y_coff,
                                                                  int OH, int OW, int C, float mul, float div) {
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  const long total = (long)N * OH * OW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long pix = e / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    int y0, y1, x0, x1;
    float ty, tx;
    legacy_coord(oy, sy, H, y0, y1, ty);
    legacy_coord(ox, sx, W, x0, x1, tx);
    const float* b = x + x_coff + c;
    const float tl = b[(((long)n * H + y0) * W + x0) * ldx], tr = b[(((long)n * H + y0) * W + x1) * ldx];
    const float bl = b[(((long)n * H + y1) * W + x0) * ldx], br = b[(((long)n * H + y1) * W + x1) * ldx];
    const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
    float v = top + (bot - top) * ty;
    v = v * mul;
    if (div != 1.f) v = v / div;
    y[pix * ldy + y_coff + c] = v;
  }
}
// ---------------------------------------------------------------------------
// Input stage ("next" row N1): the readers' per-image pipeline fused into one pass --
//   [uint8 -> v/div + add]  (preprocess_image / preprocess_mask: data/davis2016_data_utils.py:86-99)
//   flip (left-right / top-down: data/aug_flips.py:3-16)  ->  crop window (tf.random_crop / tf.image.central_crop:
//   :101-133)  ->  legacy bilinear or nearest-neighbour resize to (OH, OW)  (tf.image.resize_images).
// prm[n] = {y0, x0, crop_h, crop_w, flip_lr, flip_td} per sample (null: whole image, no flip).  The conversion is
// applied to every tap BEFORE interpolation, as the reference converts before it resizes (same float32 op order).
// ---------------------------------------------------------------------------
template <typename T, int NEAREST>
__global__ __launch_bounds__(256) void crop_flip_resize_kernel(const T* __restrict__ src, int N, int H, int W, int C,
                                                               const int* __restrict__ prm, float* __restrict__ dst, int OH, int OW,
                                                               float div, float add) {
  const long total = (long)N * OH * OW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long pix = e / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    int y0 = 0, x0 = 0, ch = H, cw = W, flr = 0, ftd = 0;
    if (prm) { y0 = prm[n * 6]; x0 = prm[n * 6 + 1]; ch = prm[n * 6 + 2]; cw = prm[n * 6 + 3]; flr = prm[n * 6 + 4]; ftd = prm[n * 6 + 5]; }
    const float sy = (float)ch / (float)OH, sx = (float)cw / (float)OW;
    auto tap = [&](int cy, int cx) -> float {
      int yy = y0 + cy, xx = x0 + cx;
      if (ftd) yy = H - 1 - yy;
      if (flr) xx = W - 1 - xx;
      float v = (float)src[(((long)n * H + yy) * W + xx) * C + c];
      if (div != 1.f) v = v / div;
      return v + add;
    };
    float v;
    if (NEAREST) {  // ResizeNearestNeighbor, align_corners=False: min(floor(i*scale), in-1)
      const int cy = min((int)floorf((float)oy * sy), ch - 1), cx = min((int)floorf((float)ox * sx), cw - 1);
      v = tap(cy, cx);
    } else {
      int ly, hy, lx, hx;
      float ty, tx;
      legacy_coord(oy, sy, ch, ly, hy, ty);
      legacy_coord(ox, sx, cw, lx, hx, tx);
      const float tl = tap(ly, lx), tr = tap(ly, hx), bl = tap(hy, lx), br = tap(hy, hx);
      const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
      v = top + (bot - top) * ty;
    }
    dst[e] = v;
  }
}
int launch_crop_flip_resize(const void* src, int src_u8, int nearest, int N, int H, int W, int C, const int* prm, float* dst, int OH,
                            int OW, float div, float add, hipStream_t s) {
  const long total = (long)N * OH * OW * C;
  const dim3 g(grid_for(total, 8192)), b(256);
  if (src_u8 && nearest) hipLaunchKernelGGL((crop_flip_resize_kernel<unsigned char, 1>), g, b, 0, s, (const unsigned char*)src, N, H, W, C, prm, dst, OH, OW, div, add);
  else if (src_u8) hipLaunchKernelGGL((crop_flip_resize_kernel<unsigned char, 0>), g, b, 0, s, (const unsigned char*)src, N, H, W, C, prm, dst, OH, OW, div, add);
  else if (nearest) hipLaunchKernelGGL((crop_flip_resize_kernel<float, 1>), g, b, 0, s, (const float*)src, N, H, W, C, prm, dst, OH, OW, div, add);
  else hipLaunchKernelGGL((crop_flip_resize_kernel<float, 0>), g, b, 0, s, (const float*)src, N, H, W, C, prm, dst, OH, OW, div, add);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// the same resize, four channels per thread (channel windows that are float4-aligned: every slab of the decoder)
__global__ __launch_bounds__(256) void resize_bilinear_fwd4_kernel(const float* __restrict__ x, int ldx, int x_coff, int N, int H,
                                                                   int W, float* __restrict__ y, int ldy, int y_coff, int OH,
                                                                   int OW, int C4, float mul, float div) {
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  const long total = (long)N * OH * OW * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    const long pix = e / C4;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    int y0, y1, x0, x1;
    float ty, tx;
    legacy_coord(oy, sy, H, y0, y1, ty);
    legacy_coord(ox, sx, W, x0, x1, tx);
    const float* b = x + x_coff + c4 * 4;
    const float4 tl = *reinterpret_cast<const float4*>(b + (((long)n * H + y0) * W + x0) * ldx);
    const float4 tr = *reinterpret_cast<const float4*>(b + (((long)n * H + y0) * W + x1) * ldx);
    const float4 bl = *reinterpret_cast<const float4*>(b + (((long)n * H + y1) * W + x0) * ldx);
    const float4 br = *reinterpret_cast<const float4*>(b + (((long)n * H + y1) * W + x1) * ldx);
    auto lerp = [&](float a, float bq, float c, float d) {
      const float top = a + (bq - a) * tx, bot = c + (d - c) * tx;
      float v = (top + (bot - top) * ty) * mul;
      if (div != 1.f) v = v / div;
      return v;
    };
    float4 o;
    o.x = lerp(tl.x, tr.x, bl.x, br.x);
    o.y = lerp(tl.y, tr.y, bl.y, br.y);
    o.z = lerp(tl.z, tr.z, bl.z, br.z);
    o.w = lerp(tl.w, tr.w, bl.w, br.w);
    *reinterpret_cast<float4*>(y + pix * ldy + y_coff + c4 * 4) = o;
  }
}
int launch_resize_bilinear_fwd(const float* x, int ldx, int x_coff, int N, int H, int W, float* y, int ldy, int y_coff,
                               int OH, int OW, int C, float mul, float div, hipStream_t s) {
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && x_coff % 4 == 0 && y_coff % 4 == 0 &&
      !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)) {
    const long total4 = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(resize_bilinear_fwd4_kernel, dim3(grid_for(total4, 8192)), dim3(256), 0, s, x, ldx, x_coff, N, H, W, y, ldy,
                       y_coff, OH, OW, C / 4, mul, div);
    UDET_HIP(hipGetLastError());
    return UDET_OK;
  }
  const long total = (long)N * OH * OW * C;
  hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, x_coff, N, H, W, y, ldy,
                     y_coff, OH, OW, C, mul, div);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// adjoint (gather form, deterministic): dx[iy,ix] (+)= sum_{oy,ox} wy(oy,iy)*wx(ox,ix)*dy[oy,ox]
__device__ __forceinline__ float legacy_weight(int o, float scale, int in, int i) {
  int lo, hi;
  float t;
  legacy_coord(o, scale, in, lo, hi, t);
  float w = 0.f;
  if (lo == i) w += 1.f - t;
  if (hi == i) w += t;
  return w;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ dy, int ldy, int y_coff, int N,
                                                                  int OH, int OW, float* __restrict__ dx, int ldx,
                                                                  int x_coff, int H, int W, int C, int accumulate) {
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  const float isy = (float)OH / (float)H, isx = (float)OW / (float)W;
  const long total = (long)N * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long pix = e / C;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const int oy_lo = max(0, (int)floorf((float)(iy - 1) * isy) - 1), oy_hi = min(OH - 1, (int)ceilf((float)(iy + 1) * isy) + 1);
    const int ox_lo = max(0, (int)floorf((float)(ix - 1) * isx) - 1), ox_hi = min(OW - 1, (int)ceilf((float)(ix + 1) * isx) + 1);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float wy = legacy_weight(oy, sy, H, iy);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float wx = legacy_weight(ox, sx, W, ix);
        if (wx == 0.f) continue;
        acc += wy * wx * dy[(((long)n * OH + oy) * OW + ox) * ldy + y_coff + c];
      }
    }
    float* d = dx + pix * ldx + x_coff + c;
    *d = accumulate ? *d + acc : acc;
  }
}
// Adjoint of the exact 2x legacy up-resize (every resize of the recover decoder): out[2a] = in[a],
// out[2a+1] = in[a] + (in[min(a+1,n-1)] - in[a])/2, so din[a] = dout[2a] + w+ * dout[2a+1] + dout[2a-1]/2 with
// w+ = 1/2 (1 in the last row / column, where both taps are in[a]).  Four channels per thread.
__global__ __launch_bounds__(256) void resize2x_bwd4_kernel(const float* __restrict__ dy, int ldy, int y_coff, int N, int H, int W,
                                                            float* __restrict__ dx, int ldx, int x_coff, int C4, int accumulate) {
  const int OH = 2 * H, OW = 2 * W;
  const long total = (long)N * H * W * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    const long pix = e / C4;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    float wy[3], wx[3];
    wy[0] = iy > 0 ? 0.5f : 0.f; wy[1] = 1.f; wy[2] = iy == H - 1 ? 1.f : 0.5f;
    wx[0] = ix > 0 ? 0.5f : 0.f; wx[1] = 1.f; wx[2] = ix == W - 1 ? 1.f : 0.5f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int oy = 2 * iy - 1 + a;
      if (oy < 0) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ox = 2 * ix - 1 + b;
        if (ox < 0) continue;
        const float w = wy[a] * wx[b];
        const float4 g = *reinterpret_cast<const float4*>(dy + (((long)n * OH + oy) * OW + ox) * ldy + y_coff + c4 * 4);
        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
      }
    }
    float4* d = reinterpret_cast<float4*>(dx + pix * ldx + x_coff + c4 * 4);
    if (accumulate) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    *d = acc;
  }
}
int launch_resize_bilinear_bwd(const float* dy, int ldy, int y_coff, int N, int OH, int OW, float* dx, int ldx, int x_coff,
                               int H, int W, int C, int accumulate, hipStream_t s) {
  if (OH == 2 * H && OW == 2 * W && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && x_coff % 4 == 0 && y_coff % 4 == 0 &&
      !((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dy)) & 15)) {
    const long total4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(resize2x_bwd4_kernel, dim3(grid_for(total4, 8192)), dim3(256), 0, s, dy, ldy, y_coff, N, H, W, dx, ldx, x_coff,
                       C / 4, accumulate);
    UDET_HIP(hipGetLastError());
    return UDET_OK;
  }
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, ldy, y_coff, N, OH, OW, dx,
                     ldx, x_coff, H, W, C, accumulate);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// adjoint of resize_nearest_neighbor(x2, align_corners=True) == replicate: 2x2 sum
__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const float* __restrict__ du, float* __restrict__ dx, int N, int H,
                                                          int W, int C) {
  const int c4n = C >> 2;
  const long total = (long)N * H * W * c4n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % c4n);
    const long pix = e / c4n;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const float* b = du + ((((long)n * 2 * H + 2 * y) * 2 * W) + 2 * x) * C + c4 * 4;
    const float4 a0 = *reinterpret_cast<const float4*>(b), a1 = *reinterpret_cast<const float4*>(b + C);
    const float4 a2 = *reinterpret_cast<const float4*>(b + (long)2 * W * C), a3 = *reinterpret_cast<const float4*>(b + (long)2 * W * C + C);
    float4 o;
    o.x = (a0.x + a1.x) + (a2.x + a3.x);
    o.y = (a0.y + a1.y) + (a2.y + a3.y);
    o.z = (a0.z + a1.z) + (a2.z + a3.z);
    o.w = (a0.w + a1.w) + (a2.w + a3.w);
    *reinterpret_cast<float4*>(dx + pix * C + c4 * 4) = o;
  }
}
int launch_pool2x2_sum(const float* du, float* dx, int N, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(grid_for((long)N * H * W * (C / 4))), dim3(256), 0, s, du, dx, N, H, W, C);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// u[p][coff+c] = d[p][coff+c] * act'(a[p][coff+c]), c < C: dU of a region whose gradient was finalised by a
// non-convolution kernel (resize adjoint, 2x2 pooling)
__global__ __launch_bounds__(256) void emit_du_kernel(const float* __restrict__ d, const float* __restrict__ a,
                                                      float* __restrict__ u, long P, int ld, int coff, int C, int act,
                                                      float alpha) {
  const long total = P * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long pix = e / C;
    const long o = pix * ld + coff + (e - pix * C);
    u[o] = d[o] * act_dfo(a[o], act, alpha);
  }
}
int launch_emit_du(const float* d, const float* a, float* u, long P, int ld, int coff, int C, int act, float alpha, hipStream_t s) {
  hipLaunchKernelGGL(emit_du_kernel, dim3(grid_for(P * C)), dim3(256), 0, s, d, a, u, P, ld, coff, C, act, alpha);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// Shared image encoder of the batched recover calls (nets.py:57-65: every call of recover_net sees the same image, so
// encoder A is evaluated for the first P pixels (B samples) only).
//  share:  buf[(k*P + p)*ld + coff + c] = buf[p*ld + coff + c],  k = 1..copies-1      (forward: fan the skip tensors out)
//  fold :  buf[p*ld + coff + c] += buf[(P+p)*ld + coff + c] + buf[(2P+p)*ld + coff + c] ...   (backward: sum the calls'
//          output gradients; fixed order (x0 + x1) + x2)
// ---------------------------------------------------------------------------
template <int V, bool FOLD>
__global__ __launch_bounds__(256) void share_fold_kernel(float* __restrict__ buf, long P, int ld, int coff, int C, int copies) {
  const int CV = C / V;
  const long total = P * CV;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long pix = e / CV;
    const long o = pix * ld + coff + (e - pix * CV) * V;
    if (V == 4) {
      float4 v = *reinterpret_cast<const float4*>(buf + o);
      for (int k = 1; k < copies; ++k) {
        float4* q = reinterpret_cast<float4*>(buf + o + (long)k * P * ld);
        if (FOLD) {
          const float4 w = *q;
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        } else {
          *q = v;
        }
      }
      if (FOLD) *reinterpret_cast<float4*>(buf + o) = v;
    } else {
      float v = buf[o];
      for (int k = 1; k < copies; ++k) {
        if (FOLD) v += buf[o + (long)k * P * ld];
        else buf[o + (long)k * P * ld] = v;
      }
      if (FOLD) buf[o] = v;
    }
  }
}
static int launch_share_fold(float* buf, long P, int ld, int coff, int C, int copies, bool fold, hipStream_t s) {
  if (copies < 2 || C < 1) return UDET_OK;
  const bool v4 = (ld % 4 == 0) && (coff % 4 == 0) && (C % 4 == 0);
  const long n = P * (v4 ? C / 4 : C);
  if (v4) {
    if (fold) hipLaunchKernelGGL((share_fold_kernel<4, true>), dim3(grid_for(n)), dim3(256), 0, s, buf, P, ld, coff, C, copies);
    else hipLaunchKernelGGL((share_fold_kernel<4, false>), dim3(grid_for(n)), dim3(256), 0, s, buf, P, ld, coff, C, copies);
  } else {
    if (fold) hipLaunchKernelGGL((share_fold_kernel<1, true>), dim3(grid_for(n)), dim3(256), 0, s, buf, P, ld, coff, C, copies);
    else hipLaunchKernelGGL((share_fold_kernel<1, false>), dim3(grid_for(n)), dim3(256), 0, s, buf, P, ld, coff, C, copies);
  }
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int launch_share_samples(float* buf, long P, int ld, int coff, int C, int copies, hipStream_t s) {
  return launch_share_fold(buf, P, ld, coff, C, copies, false, s);
}
int launch_fold_samples(float* buf, long P, int ld, int coff, int C, int copies, hipStream_t s) {
  return launch_share_fold(buf, P, ld, coff, C, copies, true, s);
}

// ---------------------------------------------------------------------------
// PWC input: x8[2B,H,W,4] = [img+0.5 (3) | 0] for the 2B stacked images   (model_pwcnet.py:39-56 adapt_x)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_pwc_input_kernel(const float* __restrict__ i1, const float* __restrict__ i2,
                                                             float* __restrict__ x8, long P) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < 2 * P; e += (long)gridDim.x * 256) {
    const float* s = e < P ? i1 + e * 3 : i2 + (e - P) * 3;
    float4 a = make_float4(s[0] + 0.5f, s[1] + 0.5f, s[2] + 0.5f, 0.f);
    *reinterpret_cast<float4*>(x8 + e * 4) = a;
  }
}
int launch_pack_pwc_input(const float* i1, const float* i2, float* x8, long P, hipStream_t s) {
  hipLaunchKernelGGL(pack_pwc_input_kernel, dim3(grid_for(2 * P)), dim3(256), 0, s, i1, i2, x8, P);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// preprocess_flow_batch (models/utils/flow_utils.py:5-12): per sample & channel
// (f-mean)/sqrt(var), population variance, no epsilon.  Stage 1: double-precision
// partial sum / sum of squares; stage 2 (fused in the generator-input packer) finishes.
// ---------------------------------------------------------------------------
#define FS_BLOCKS 32
__global__ __launch_bounds__(256) void flow_stats_kernel(const float* __restrict__ f, long HW, double* __restrict__ part) {
  __shared__ double sm[4];
  const int n = blockIdx.y;
  double s0 = 0, s1 = 0, q0 = 0, q1 = 0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const float2 v = *reinterpret_cast<const float2*>(f + ((long)n * HW + p) * 2);
    s0 += v.x; s1 += v.y;
    q0 += (double)v.x * v.x; q1 += (double)v.y * v.y;
  }
  s0 = block_sum(s0, sm); s1 = block_sum(s1, sm); q0 = block_sum(q0, sm); q1 = block_sum(q1, sm);
  if (threadIdx.x == 0) {
    double* o = part + ((long)n * FS_BLOCKS + blockIdx.x) * 4;
    o[0] = s0; o[1] = s1; o[2] = q0; o[3] = q1;
  }
}
// gin[B,H,W,8] = [image(3), (flow-mean)/std (2), 0,0,0]   (models/nets.py:14, adversarial_learner.py:99-105)
__global__ __launch_bounds__(256) void pack_gen_input_kernel(const float* __restrict__ img, const float* __restrict__ f,
                                                             const double* __restrict__ part, float* __restrict__ gin,
                                                             long HW) {
  __shared__ float st[4];
  const int n = blockIdx.y;
  if (threadIdx.x < 2) {
    double s = 0, q = 0;
    for (int b = 0; b < FS_BLOCKS; ++b) {
      s += part[((long)n * FS_BLOCKS + b) * 4 + threadIdx.x];
      q += part[((long)n * FS_BLOCKS + b) * 4 + 2 + threadIdx.x];
    }
    const double mean = s / (double)HW, var = q / (double)HW - mean * mean;
    st[threadIdx.x] = (float)mean;
    st[2 + threadIdx.x] = sqrtf((float)var);
  }
  __syncthreads();
  const float m0 = st[0], m1 = st[1], d0 = st[2], d1 = st[3];
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const long q = (long)n * HW + p;
    const float2 v = *reinterpret_cast<const float2*>(f + q * 2);
    *reinterpret_cast<float4*>(gin + q * 8) = make_float4(img[q * 3], img[q * 3 + 1], img[q * 3 + 2], (v.x - m0) / d0);
    *reinterpret_cast<float4*>(gin + q * 8 + 4) = make_float4((v.y - m1) / d1, 0.f, 0.f, 0.f);
  }
}
int launch_gen_input(const float* img, const float* f, double* part, float* gin, int B, long HW, hipStream_t s) {
  hipLaunchKernelGGL(flow_stats_kernel, dim3(FS_BLOCKS, B), dim3(256), 0, s, f, HW, part);
  hipLaunchKernelGGL(pack_gen_input_kernel, dim3(grid_for(HW, 256), B), dim3(256), 0, s, img, f, part, gin, HW);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
size_t flow_stats_doubles(int B) { return (size_t)B * FS_BLOCKS * 4; }
// the standardised flow alone (the generator-input packer above fuses the same arithmetic): out[B,H,W,2]
__global__ __launch_bounds__(256) void flow_normalize_kernel(const float* __restrict__ f, const double* __restrict__ part,
                                                             float* __restrict__ out, long HW) {
  __shared__ float st[4];
  const int n = blockIdx.y;
  if (threadIdx.x < 2) {
    double s = 0, q = 0;
    for (int b = 0; b < FS_BLOCKS; ++b) {
      s += part[((long)n * FS_BLOCKS + b) * 4 + threadIdx.x];
      q += part[((long)n * FS_BLOCKS + b) * 4 + 2 + threadIdx.x];
    }
    const double mean = s / (double)HW, var = q / (double)HW - mean * mean;
    st[threadIdx.x] = (float)mean;
    st[2 + threadIdx.x] = sqrtf((float)var);
  }
  __syncthreads();
  const float m0 = st[0], m1 = st[1], d0 = st[2], d1 = st[3];
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const long q = (long)n * HW + p;
    const float2 v = *reinterpret_cast<const float2*>(f + q * 2);
    *reinterpret_cast<float2*>(out + q * 2) = make_float2((v.x - m0) / d0, (v.y - m1) / d1);
  }
}
int launch_flow_normalize(const float* f, double* part, float* out, int B, long HW, hipStream_t s) {
  hipLaunchKernelGGL(flow_stats_kernel, dim3(FS_BLOCKS, B), dim3(256), 0, s, f, HW, part);
  hipLaunchKernelGGL(flow_normalize_kernel, dim3(grid_for(HW, 256), B), dim3(256), 0, s, f, part, out, HW);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// mask + recover inputs (models/nets.py:38-41,50-52; adversarial_learner.py:107-131)
//   m = softmax(logits/10)[0] ; cm = 1-m
//   call 0: [f*(1-m),  1, 1-m ]   call 1: [f*(1-cm), 1, 1-cm]   call 2: [0,0,1,0]
//   imgin = image replicated for the 3 calls (8-channel padded)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_rec_inputs_kernel(const float* __restrict__ logits /*ld 8*/,
                                                              const float* __restrict__ f, float* __restrict__ mask,
                                                              float* __restrict__ fin, long P, int ncalls) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < P; q += (long)gridDim.x * 256) {
    const float l0 = logits[q * 8] / 10.f, l1 = logits[q * 8 + 1] / 10.f;
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float m = e0 / (e0 + e1);
    mask[q] = m;
    const float cm = 1.f - m;
    const float2 v = *reinterpret_cast<const float2*>(f + q * 2);
    const float a0 = 1.f - m, a1 = 1.f - cm;
    if (ncalls > 0) {
      *reinterpret_cast<float4*>(fin + q * 4) = make_float4(v.x * a0, v.y * a0, 1.f, a0);
    }
    if (ncalls > 1) {
      *reinterpret_cast<float4*>(fin + (P + q) * 4) = make_float4(v.x * a1, v.y * a1, 1.f, a1);
    }
    if (ncalls > 2) {
      *reinterpret_cast<float4*>(fin + (2 * P + q) * 4) = make_float4(0.f, 0.f, 1.f, 0.f);
    }
  }
}
int launch_mask_rec_inputs(const float* logits, const float* f, float* mask, float* fin, long P, int ncalls, hipStream_t s) {
  hipLaunchKernelGGL(mask_rec_inputs_kernel, dim3(grid_for(P)), dim3(256), 0, s, logits, f, mask, fin, P, ncalls);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// imgin[call][q] = [image(3) | 0] for the `ncalls` recover invocations (nets.py:57: the image encoder's input)
__global__ __launch_bounds__(256) void pack_imgin_kernel(const float* __restrict__ img, float* __restrict__ imgin, long P,
                                                         int ncalls) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < P; q += (long)gridDim.x * 256) {
    const float4 im = make_float4(img[q * 3], img[q * 3 + 1], img[q * 3 + 2], 0.f);
    for (int k = 0; k < ncalls; ++k) *reinterpret_cast<float4*>(imgin + (k * P + q) * 4) = im;
  }
}
int launch_pack_imgin(const float* img, float* imgin, long P, int ncalls, hipStream_t s) {
  hipLaunchKernelGGL(pack_imgin_kernel, dim3(grid_for(P)), dim3(256), 0, s, img, imgin, P, ncalls);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// Charbonnier sums (models/utils/loss_utils.py:34-51; adversarial_learner.py:144-189)
//  per sample b:  R=sum phi(f-p)*m  Rc=sum phi(f-pc)*cm  P=sum phi(f-q)  Dr=sum phi(f-q)*m  Dcr=sum phi(f-q)*cm
// ---------------------------------------------------------------------------
__device__ __forceinline__ float charb(float e, float cbn) {
  const float s = e * e + 0.001f * 0.001f;
  return cbn == 0.5f ? sqrtf(s) : powf(s, cbn);
}
// d phi / d e
__device__ __forceinline__ float charb_d(float e, float cbn) {
  const float s = e * e + 0.001f * 0.001f;
  return cbn == 0.5f ? e / sqrtf(s) : cbn * powf(s, cbn - 1.f) * 2.f * e;
}
#define LS_BLOCKS 64
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ f, const float* __restrict__ mask,
                                                        const float* __restrict__ pred /*[3B,HW,2]*/, long HW, int B,
                                                        float cbn, float* __restrict__ part /*[B][LS_BLOCKS][5]*/) {
  __shared__ float sm[4];
  const int n = blockIdx.y;
  float R = 0, Rc = 0, Pp = 0, Dr = 0, Dc = 0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const long q = (long)n * HW + p;
    const float2 fl = *reinterpret_cast<const float2*>(f + q * 2);
    const float2 a = *reinterpret_cast<const float2*>(pred + q * 2);
    const float2 b = *reinterpret_cast<const float2*>(pred + ((long)B * HW + q) * 2);
    const float2 c = *reinterpret_cast<const float2*>(pred + ((long)2 * B * HW + q) * 2);
    const float m = mask[q], cm = 1.f - m;
    const float pa = charb(fl.x - a.x, cbn) + charb(fl.y - a.y, cbn);
    const float pb = charb(fl.x - b.x, cbn) + charb(fl.y - b.y, cbn);
    const float pc = charb(fl.x - c.x, cbn) + charb(fl.y - c.y, cbn);
    R += pa * m; Rc += pb * cm; Pp += pc; Dr += pc * m; Dc += pc * cm;
  }
  R = block_sum(R, sm); Rc = block_sum(Rc, sm); Pp = block_sum(Pp, sm); Dr = block_sum(Dr, sm); Dc = block_sum(Dc, sm);
  if (threadIdx.x == 0) {
    float* o = part + ((long)n * LS_BLOCKS + blockIdx.x) * 5;
    o[0] = R; o[1] = Rc; o[2] = Pp; o[3] = Dr; o[4] = Dc;
  }
}
// losses[8] in the order of adversarial_learner.py:196-204: generator, recover, red_rate, red_rate_compl,
// reconstruction_loss, reconstruction_compl_loss, denominator_red_rate, denominator_red_rate_compl
// coef[b][4] = {dG/dR_b, dG/dD_b, dG/dRc_b, dG/dDc_b}
__global__ void loss_finish_kernel(const float* __restrict__ part, int B, float eps, float num_pixels,
                                   float* __restrict__ losses, float* __restrict__ coef, float* __restrict__ sums) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float red = 0.f, redc = 0.f, rs = 0.f;
  for (int b = 0; b < B; ++b) {
    float v[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < LS_BLOCKS; ++k)
      for (int j = 0; j < 5; ++j) v[j] += part[((long)b * LS_BLOCKS + k) * 5 + j];
    const float D = v[3] + eps, Dc = v[4] + eps;
    red += 1.f - v[0] / D;
    redc += 1.f - v[1] / Dc;
    rs += v[0] + v[1] + v[2];
    coef[b * 4 + 0] = -1.f / (B * D);
    coef[b * 4 + 1] = v[0] / (B * D * D);
    coef[b * 4 + 2] = -1.f / (B * Dc);
    coef[b * 4 + 3] = v[1] / (B * Dc * Dc);
    for (int j = 0; j < 5; ++j) sums[b * 5 + j] = v[j];
    if (b == 0) { losses[4] = v[0]; losses[5] = v[1]; losses[6] = D; losses[7] = Dc; }
  }
  red /= B; redc /= B;
  losses[0] = red + redc;
  losses[1] = rs / num_pixels;
  losses[2] = red;
  losses[3] = redc;
}
int launch_losses(const float* f, const float* mask, const float* pred, long HW, int B, float cbn, float eps,
                  float num_pixels, float* part, float* losses, float* coef, float* sums, hipStream_t s) {
  hipLaunchKernelGGL(loss_sums_kernel, dim3(LS_BLOCKS, B), dim3(256), 0, s, f, mask, pred, HW, B, cbn, part);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, part, B, eps, num_pixels, losses, coef, sums);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
size_t loss_part_floats(int B) { return (size_t)B * LS_BLOCKS * 5; }

// charbonnier_loss(gt_flows, pred_flows, masks, cbn) alone (loss_utils.py:34-51): out[b] = sum_{h,w,c} phi(gt - pred) * mask,
// mask [B,H,W,mc] with mc = 1 (broadcast over the two flow channels), 2, or null (ones).  Deterministic two-stage sum.
__global__ __launch_bounds__(256) void charbonnier_sum_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              const float* __restrict__ mask, int mc, long HW, float cbn,
                                                              float* __restrict__ part /*[B][LS_BLOCKS]*/) {
  __shared__ float sm[4];
  const int n = blockIdx.y;
  float acc = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const long q = (long)n * HW + p;
    const float2 a = *reinterpret_cast<const float2*>(gt + q * 2);
    const float2 b = *reinterpret_cast<const float2*>(pred + q * 2);
    const float m0 = mask ? mask[q * mc] : 1.f, m1 = mask ? mask[q * mc + (mc - 1)] : 1.f;
    acc += charb(a.x - b.x, cbn) * m0 + charb(a.y - b.y, cbn) * m1;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) part[(long)n * LS_BLOCKS + blockIdx.x] = acc;
}
__global__ void charbonnier_finish_kernel(const float* __restrict__ part, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float v = 0.f;
  for (int k = 0; k < LS_BLOCKS; ++k) v += part[(long)b * LS_BLOCKS + k];
  out[b] = v;
}
int launch_charbonnier(const float* gt, const float* pred, const float* mask, int mc, int B, long HW, float cbn, float* part,
                       float* out, hipStream_t s) {
  hipLaunchKernelGGL(charbonnier_sum_kernel, dim3(LS_BLOCKS, B), dim3(256), 0, s, gt, pred, mask, mc, HW, cbn, part);
  hipLaunchKernelGGL(charbonnier_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, B, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// d(recover_loss)/d pred for the 3 calls  (recover_loss = (sum R + sum Rc + sum P)/num_pixels)
__global__ __launch_bounds__(256) void rec_loss_bwd_kernel(const float* __restrict__ f, const float* __restrict__ mask,
                                                           const float* __restrict__ pred, float* __restrict__ dpred,
                                                           long BHW, float cbn, float inv_np) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < BHW; q += (long)gridDim.x * 256) {
    const float2 fl = *reinterpret_cast<const float2*>(f + q * 2);
    const float m = mask[q], cm = 1.f - m;
    const float2 a = *reinterpret_cast<const float2*>(pred + q * 2);
    const float2 b = *reinterpret_cast<const float2*>(pred + (BHW + q) * 2);
    const float2 c = *reinterpret_cast<const float2*>(pred + (2 * BHW + q) * 2);
    *reinterpret_cast<float2*>(dpred + q * 2) = make_float2(-charb_d(fl.x - a.x, cbn) * m * inv_np, -charb_d(fl.y - a.y, cbn) * m * inv_np);
    *reinterpret_cast<float2*>(dpred + (BHW + q) * 2) = make_float2(-charb_d(fl.x - b.x, cbn) * cm * inv_np, -charb_d(fl.y - b.y, cbn) * cm * inv_np);
    *reinterpret_cast<float2*>(dpred + (2 * BHW + q) * 2) = make_float2(-charb_d(fl.x - c.x, cbn) * inv_np, -charb_d(fl.y - c.y, cbn) * inv_np);
  }
}
int launch_rec_loss_bwd(const float* f, const float* mask, const float* pred, float* dpred, long BHW, float cbn,
                        float inv_np, hipStream_t s) {
  hipLaunchKernelGGL(rec_loss_bwd_kernel, dim3(grid_for(BHW)), dim3(256), 0, s, f, mask, pred, dpred, BHW, cbn, inv_np);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// d(generator_loss)/d pred (calls 0,1) and the direct mask terms
__global__ __launch_bounds__(256) void gen_loss_bwd_kernel(const float* __restrict__ f, const float* __restrict__ mask,
                                                           const float* __restrict__ pred, const float* __restrict__ coef,
                                                           float* __restrict__ dpred, float* __restrict__ dmask, long HW,
                                                           int B, float cbn) {
  const long BHW = (long)B * HW;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < BHW; q += (long)gridDim.x * 256) {
    const int n = (int)(q / HW);
    const float cR = coef[n * 4], cD = coef[n * 4 + 1], cRc = coef[n * 4 + 2], cDc = coef[n * 4 + 3];
    const float2 fl = *reinterpret_cast<const float2*>(f + q * 2);
    const float m = mask[q], cm = 1.f - m;
    const float2 a = *reinterpret_cast<const float2*>(pred + q * 2);
    const float2 b = *reinterpret_cast<const float2*>(pred + (BHW + q) * 2);
    const float2 c = *reinterpret_cast<const float2*>(pred + (2 * BHW + q) * 2);
    *reinterpret_cast<float2*>(dpred + q * 2) = make_float2(-cR * charb_d(fl.x - a.x, cbn) * m, -cR * charb_d(fl.y - a.y, cbn) * m);
    *reinterpret_cast<float2*>(dpred + (BHW + q) * 2) = make_float2(-cRc * charb_d(fl.x - b.x, cbn) * cm, -cRc * charb_d(fl.y - b.y, cbn) * cm);
    const float pa = charb(fl.x - a.x, cbn) + charb(fl.y - a.y, cbn);
    const float pb = charb(fl.x - b.x, cbn) + charb(fl.y - b.y, cbn);
    const float pc = charb(fl.x - c.x, cbn) + charb(fl.y - c.y, cbn);
    dmask[q] = cR * pa + cD * pc - cRc * pb - cDc * pc;
  }
}
int launch_gen_loss_bwd(const float* f, const float* mask, const float* pred, const float* coef, float* dpred,
                        float* dmask, long HW, int B, float cbn, hipStream_t s) {
  hipLaunchKernelGGL(gen_loss_bwd_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, s, f, mask, pred, coef, dpred, dmask,
                     HW, B, cbn);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// dlogits from dmask(direct) + the gradients that reached the recover inputs of calls 0 and 1
//   call 0 input = [f*(1-m),1,1-m] -> dm -= g0.f + g0[3] ; call 1 input = [f*(1-cm),1,1-cm] -> dm += g1.f + g1[3]
//   m = softmax(l/10)[0] -> dl0 = dm*m*(1-m)/10, dl1 = -dl0
__global__ __launch_bounds__(256) void mask_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ dfin,
                                                       const float* __restrict__ f, const float* __restrict__ mask,
                                                       float* __restrict__ dlogits /*ld 8*/, long P) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < P; q += (long)gridDim.x * 256) {
    const float4 g0 = *reinterpret_cast<const float4*>(dfin + q * 4);
    const float4 g1 = *reinterpret_cast<const float4*>(dfin + (P + q) * 4);
    const float2 fl = *reinterpret_cast<const float2*>(f + q * 2);
    float dm = dmask[q];
    dm -= g0.x * fl.x + g0.y * fl.y + g0.w;
    dm += g1.x * fl.x + g1.y * fl.y + g1.w;
    const float m = mask[q];
    const float d0 = dm * m * (1.f - m) / 10.f;
    dlogits[q * 8] = d0;
    dlogits[q * 8 + 1] = -d0;
  }
}
int launch_mask_bwd(const float* dmask, const float* dfin, const float* f, const float* mask, float* dlogits, long P,
                    hipStream_t s) {
  hipLaunchKernelGGL(mask_bwd_kernel, dim3(grid_for(P)), dim3(256), 0, s, dmask, dfin, f, mask, dlogits, P);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// train_op (models/utils/loss_utils.py:12-32) + tf.train.AdamOptimizer apply
// ---------------------------------------------------------------------------
// stage 1: per-variable sum|g|, each variable split over GA_SPLIT workgroups (deterministic two-stage reduction)
#define GA_SPLIT 8
__global__ __launch_bounds__(256) void grad_absmean_kernel(const float* __restrict__ g, const long* __restrict__ seg_off,
                                                           const long* __restrict__ seg_len, float* __restrict__ vpart) {
  __shared__ float sm[4];
  const long off = seg_off[blockIdx.x], len = seg_len[blockIdx.x];
  const long b = len * blockIdx.y / GA_SPLIT, e = len * (blockIdx.y + 1) / GA_SPLIT;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long i = b + threadIdx.x;
  for (; i + 768 < e; i += 1024) {
    s0 += fabsf(g[off + i]);
    s1 += fabsf(g[off + i + 256]);
    s2 += fabsf(g[off + i + 512]);
    s3 += fabsf(g[off + i + 768]);
  }
  for (; i < e; i += 256) s0 += fabsf(g[off + i]);
  const float s = block_sum((s0 + s1) + (s2 + s3), sm);
  if (threadIdx.x == 0) vpart[blockIdx.x * GA_SPLIT + blockIdx.y] = s;
}
// stage 2: mean over variables of mean|g| -> flag (1 = replace gradients by |U(-clip,clip)|)
__global__ void grad_flag_kernel(const float* __restrict__ vpart, const long* __restrict__ seg_len, int nvars, float thresh,
                                 float* __restrict__ out /*[2]: avg, flag*/) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < nvars; ++i) {
      float v = 0.f;
      for (int j = 0; j < GA_SPLIT; ++j) v += vpart[i * GA_SPLIT + j];
      s += v / (float)seg_len[i];
    }
    s /= (float)nvars;
    out[0] = s;
    out[1] = s < thresh ? 1.f : 0.f;
  }
}
int launch_grad_absmean(const float* g, const long* seg_off, const long* seg_len, int nvars, float* vmean, float thresh,
                        float* out, hipStream_t s) {
  hipLaunchKernelGGL(grad_absmean_kernel, dim3(nvars, GA_SPLIT), dim3(256), 0, s, g, seg_off, seg_len, vmean);
  hipLaunchKernelGGL(grad_flag_kernel, dim3(1), dim3(64), 0, s, vmean, seg_len, nvars, thresh, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// counter-based uniform in [0,1): splitmix64 of (seed, step, index) -- identical on every rank
__device__ __host__ __forceinline__ float udet_uniform01(uint64_t seed, uint64_t step, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + idx + 0x94D049BB133111EBull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// g <- flag ? |U(-clip,clip)| : clip(g) ; m,v,w <- Adam (TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t))
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                   float clip, const float* __restrict__ flag, uint64_t seed,
                                                   uint64_t step, int mode, const int* __restrict__ skip) {
  // mode 0: clip / noise + Adam (the step's fused form); 1: clip / noise only (train_op's clipped_grad_and_vars);
  // 2: Adam only on the gradient as it is (optimizer.apply_gradients)
  // skip (fp16 mode): number of non-finite gradient values counted by nonfinite_count_kernel -- the whole update is dropped
  if (skip && *skip != 0) return;
  const bool noise = flag && flag[1] != 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i];
    if (mode != 2) {
      if (noise) gi = fabsf((udet_uniform01(seed, step, (uint64_t)i) * 2.f - 1.f) * clip);
      else gi = fminf(fmaxf(gi, -clip), clip);
      g[i] = gi;  // clipped_grad_and_vars (loss_utils.py:31) -- what the reference logs as gradient histograms
      if (mode == 1) continue;
    }
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}
__global__ __launch_bounds__(256) void fill_uniform_kernel(float* __restrict__ x, long n, uint64_t seed, float lo, float hi) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    x[i] = lo + (hi - lo) * udet_uniform01(seed, 0, (uint64_t)i);
}
int launch_fill_uniform(float* x, long n, uint64_t seed, float lo, float hi, hipStream_t s) {
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, s, x, n, seed, lo, hi);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// ---------------------------------------------------------------------------
// evaluation tail (models/utils/general_utils.py:89-159, test_generator.py:19-40): everything both IoU variants and
// the MAE need, per sample, in one pass:  out[n][8] = {border sum, |pred|, |gt|, |pred & gt|,
//   sum pred*|gt-1|, sum (1-pred)*|gt|, sum (1-pred)*|gt-1|, sum pred*|gt|}   with pred = mask > threshold,
// gt = gt_mask > gt_threshold; the border sum adds the two top / bottom rows and the two left / right columns
// (corner pixels twice, like the reference's four overlapping strips).  Counts are exact (double accumulation).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_stats_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W,
                                                         float threshold, float gt_threshold, double* __restrict__ out) {
  __shared__ double sm[4];
  const int n = blockIdx.x;
  const long HW = (long)H * W;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long q = threadIdx.x; q < HW; q += 256) {
    const int y = (int)(q / W), x = (int)(q - (long)y * W);
    const float g = gt[n * HW + q];
    const bool pr = pred[n * HW + q] > threshold, gb = g > gt_threshold;
    const int strips = (y < 2) + (y >= H - 2) + (x < 2) + (x >= W - 2);
    const double a1 = fabs((double)g - 1.0), a0 = fabs((double)g);
    if (pr) {
      v[0] += strips; v[1] += 1.0; v[4] += a1; v[7] += a0;
      if (gb) v[3] += 1.0;
    } else {
      v[5] += a0; v[6] += a1;
    }
    if (gb) v[2] += 1.0;
  }
  for (int k = 0; k < 8; ++k) {
    const double s = block_sum(v[k], sm);
    if (threadIdx.x == 0) out[(long)n * 8 + k] = s;
    __syncthreads();
  }
}
int launch_mask_stats(const float* pred, const float* gt, int N, int H, int W, float threshold, float gt_threshold, double* out,
                      hipStream_t s) {
  hipLaunchKernelGGL(mask_stats_kernel, dim3(N), dim3(256), 0, s, pred, gt, H, W, threshold, gt_threshold, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int launch_adam(float* w, float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps, float clip,
                const float* flag, uint64_t seed, uint64_t step, hipStream_t s, int mode, const int* skip) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, w, g, m, v, n, lr_t, b1, b2, eps, clip, flag, seed,
                     step, mode, skip);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// fp16 mode: gradient operands are scaled by 4096 before the fp16 conversion, so |dU| > 16 becomes inf there and reaches the flat
// gradient buffer as inf / NaN (nothing on the way maps a non-finite value back to a finite one; the clip of the optimizer would).
// out[0] = number of non-finite values of g[0, n), out[1] += the same (running total of the plan)
__global__ __launch_bounds__(256) void nonfinite_count_kernel(const float* __restrict__ g, long n, int* __restrict__ out) {
  int c = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) c += !(fabsf(g[i]) <= 3.402823466e38f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) {
    atomicAdd(out, c);
    atomicAdd(out + 1, c);
  }
}
int launch_nonfinite_count(const float* g, long n, int* out, hipStream_t s) {
  UDET_HIP(hipMemsetAsync(out, 0, sizeof(int), s));
  hipLaunchKernelGGL(nonfinite_count_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, g, n, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// Recover decoder as per-parity-class 3x3 convolutions on the LOW-resolution grid (plan_exec.hip "up-conv algebra"): legacy bilinear x2 +
// 4x4 SAME convolution = four 3x3 convolutions with merged weights once the source carries a one-pixel ring:
//   x^[R][C] = sy(R) sx(C) x[clamp(R - 1)][clamp(C - 1)],  R in [0, H + 2), C in [0, W + 2),  s = -1 on ring row / column 0 (the zero
//   padding of the up-sampled grid), +1 elsewhere (the clamp of the legacy resize at the bottom / right edge).
// upb_ring: x -> x^.  upb_ring_fold: its adjoint, dx[j][i] = sum over the ring cells that read x[j][i] (written, not accumulated).
// float4 over channels.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upb_ring_kernel(const float* __restrict__ x, int ldx, int N, int H, int W, float* __restrict__ xh, int C4) {
  const int PH = H + 2, PW = W + 2;
  const long total = (long)N * PH * PW * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    const long pix = e / C4;
    const int C = (int)(pix % PW), R = (int)((pix / PW) % PH), n = (int)(pix / ((long)PW * PH));
    const int cy = R < 1 ? 0 : (R - 1 > H - 1 ? H - 1 : R - 1), cx = C < 1 ? 0 : (C - 1 > W - 1 ? W - 1 : C - 1);
    const float sg = ((R == 0) != (C == 0)) ? -1.f : 1.f;
    const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + cy) * W + cx) * ldx + c4 * 4);
    *reinterpret_cast<float4*>(xh + pix * ldx + c4 * 4) = make_float4(v.x * sg, v.y * sg, v.z * sg, v.w * sg);
  }
}
int launch_upb_ring(const float* x, int ld, int N, int H, int W, float* xh, hipStream_t s) {
  const long total = (long)N * (H + 2) * (W + 2) * (ld / 4);
  hipLaunchKernelGGL(upb_ring_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, x, ld, N, H, W, xh, ld / 4);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
__global__ __launch_bounds__(256) void upb_ring_fold_kernel(const float* __restrict__ dxh, int ld, int N, int H, int W, float* __restrict__ dx, int C4) {
  const int PW = W + 2, PH = H + 2;
  const long total = (long)N * H * W * C4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    const long pix = e / C4;
    const int i = (int)(pix % W), j = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    // ring rows that read x[j]: j + 1, and row 0 (negated) for j = 0, row H + 1 for j = H - 1; columns alike; fixed order
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
      int R;
      float sy = 1.f;
      if (ry == 0) R = j + 1;
      else if (j == 0) { R = 0; sy = -1.f; }
      else if (j == H - 1) R = H + 1;
      else continue;
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) {
        int C;
        float sx = 1.f;
        if (rx == 0) C = i + 1;
        else if (i == 0) { C = 0; sx = -1.f; }
        else if (i == W - 1) C = W + 1;
        else continue;
        const float4 v = *reinterpret_cast<const float4*>(dxh + (((long)n * PH + R) * PW + C) * ld + c4 * 4);
        const float sg = sy * sx;
        a.x += sg * v.x; a.y += sg * v.y; a.z += sg * v.z; a.w += sg * v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + pix * ld + c4 * 4) = a;
  }
}
int launch_upb_ring_fold(const float* dxh, int ld, int N, int H, int W, float* dx, hipStream_t s) {
  if (H < 2 || W < 2) { set_error("upb_ring_fold: source grid must be at least 2x2"); return UDET_ERR_SHAPE; }
  const long total = (long)N * H * W * (ld / 4);
  hipLaunchKernelGGL(upb_ring_fold_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, dxh, ld, N, H, W, dx, ld / 4);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// y = a*x (+ y)   small helper for skip-gradients that need no conv
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a,
                                                   int accumulate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = accumulate ? y[i] + a * x[i] : a * x[i];
}
int launch_axpy(const float* x, float* y, long n, float a, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, a, accumulate);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
