// Post-processing stage on the GPU: the ensemble buffers stay in HBM from the generator to the
// final mask instead of travelling through .mat files.
//   post_processing/generate_soft_score_from_buffer.py   sanity_check :116-125, rectify_pred_mask :98-114 (scipy.misc.imresize =
//        bytescale + Pillow's 8-bit bilinear resampler), score accumulation / min-max :38-93, propagate :127-231 (cv2.remap)
//   post_processing/crf_refine.py                         refine :110-138 (gaussian unary, dense CRF mean field)
// The frames are 192x384: every kernel here is a few microseconds of byte / integer / small-float work, written for exact
// agreement with the CPU restatement (oracle/oracle_post.py), not for a roofline.  Masks and scores are accumulated in double like
// the reference's numpy float64 arrays.  Compiled with -ffp-contract=off (Makefile).
#include <math.h>

#include "common.h"

namespace udet {

// ---- block reductions (one workgroup of 1024 threads handles a whole frame: deterministic, no second pass) -------------
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* sm, Op op) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) sm[t] = op(sm[t], sm[t + s]);
    __syncthreads();
  }
  const T r = sm[0];
  __syncthreads();
  return r;
}

// sanity_check: mean over the four two-pixel border strips (corners counted twice); one workgroup per frame
__global__ __launch_bounds__(1024) void post_border_mean_kernel(const float* __restrict__ s, int H, int W, double* __restrict__ out) {
  __shared__ double sm[1024];
  const float* p = s + (size_t)blockIdx.x * H * W;
  double acc = 0.0;
  for (int i = threadIdx.x; i < H * W; i += 1024) {
    const int y = i / W, x = i - y * W;
    const int cnt = (y < 2) + (y >= H - 2) + (x < 2) + (x >= W - 2);
    acc += (double)cnt * (double)p[i];
  }
  acc = block_reduce(acc, sm, [](double a, double b) { return a + b; });
  if (threadIdx.x == 0) out[blockIdx.x] = acc / (1.0 * (4.0 * W + 4.0 * H));
}

// scipy.misc.bytescale of the window [y0,y0+h) x [x0,x0+w) of a double image with row stride ld: min / max of the window, then
// uint8((x - min) * (255 / (max - min)) clipped + 0.5).  One workgroup.
__global__ __launch_bounds__(1024) void post_bytescale_kernel(const double* __restrict__ src, int ld, int y0, int x0, int h, int w,
                                                              unsigned char* __restrict__ dst) {
  __shared__ double sm[1024];
  double mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < h * w; i += 1024) {
    const double v = src[(size_t)(y0 + i / w) * ld + x0 + i % w];
    mn = fmin(mn, v);
    mx = fmax(mx, v);
  }
  mn = block_reduce(mn, sm, [](double a, double b) { return fmin(a, b); });
  mx = block_reduce(mx, sm, [](double a, double b) { return fmax(a, b); });
  double cscale = mx - mn;
  if (cscale == 0.0) cscale = 1.0;
  const double scale = 255.0 / cscale;
  for (int i = threadIdx.x; i < h * w; i += 1024) {
    double b = (src[(size_t)(y0 + i / w) * ld + x0 + i % w] - mn) * scale + 0.0;
    b = fmin(fmax(b, 0.0), 255.0) + 0.5;
    dst[i] = (unsigned char)b;
  }
}

// One pass of Pillow's 8-bit resampler (Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc): for output index o along
// `axis` (1: x, 0: y): ss = 1 << 21; ss += pixel[bounds[o][0] + k] * kk[o][k], k < bounds[o][1]; out = clip8(ss >> 22).
__global__ __launch_bounds__(256) void post_resample_u8_kernel(const unsigned char* __restrict__ src, int h, int w,
                                                               unsigned char* __restrict__ dst, int oh, int ow,
                                                               const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                                               int axis) {
  const int total = oh * ow;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / ow, x = i - y * ow;
    const int o = axis ? x : y;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    int ss = 1 << 21;
    for (int k = 0; k < n; ++k) {
      const int pix = axis ? src[(size_t)y * w + lo + k] : src[(size_t)(lo + k) * w + x];
      ss += pix * kk[o * ksize + k];
    }
    ss >>= 22;
    dst[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

// rectify_pred_mask's tail: the uint8 patch [hh,ww] is placed at (y0,x0) of an [H,W] canvas of zeros and the canvas divided by
// (max + 1e-6); `accumulate`: score (+)= canvas.  One workgroup.
__global__ __launch_bounds__(1024) void post_place_kernel(const unsigned char* __restrict__ patch, int hh, int ww, int y0, int x0, int H,
                                                          int W, double* __restrict__ canvas) {
  __shared__ double sm[1024];
  double mx = 0.0;  // the canvas holds zeros outside the patch and uint8 >= 0 inside
  for (int i = threadIdx.x; i < hh * ww; i += 1024) mx = fmax(mx, (double)patch[i]);
  mx = block_reduce(mx, sm, [](double a, double b) { return fmax(a, b); });
  const double den = mx + 1e-6;
  for (int i = threadIdx.x; i < H * W; i += 1024) {
    const int y = i / W - y0, x = i % W - x0;
    const double v = (y >= 0 && y < hh && x >= 0 && x < ww) ? (double)patch[(size_t)y * ww + x] : 0.0;
    canvas[i] = v / den;
  }
}

// pred_mask = (score - min) / (max - min + 1e-6)     (:88-90); one workgroup
__global__ __launch_bounds__(1024) void post_minmax_norm_kernel(const double* __restrict__ score, int n, double* __restrict__ out) {
  __shared__ double sm[1024];
  double mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 1024) {
    mn = fmin(mn, score[i]);
    mx = fmax(mx, score[i]);
  }
  mn = block_reduce(mn, sm, [](double a, double b) { return fmin(a, b); });
  mx = block_reduce(mx, sm, [](double a, double b) { return fmax(a, b); });
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = (score[i] - mn) / (mx - mn + 1e-6);
}

// cv2.remap(src, flow + grid, None, INTER_LINEAR), BORDER_CONSTANT 0 (OpenCV imgwarp.cpp remapBilinear): coordinates rounded
// to 1/32 pixel (round half to even), float weights (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx, taps outside the image read 0
__global__ __launch_bounds__(256) void post_remap_kernel(const float* __restrict__ src, const float* __restrict__ flow_uv,
                                                         float* __restrict__ dst, int H, int W) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    const float mxf = (float)((double)flow_uv[2 * i] + (double)x), myf = (float)((double)flow_uv[2 * i + 1] + (double)y);
    long sx = (long)rint((double)mxf * 32.0), sy = (long)rint((double)myf * 32.0);
    long ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) / 32.f, fy = (float)(sy & 31) / 32.f;
    ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    auto tap = [&](long yy, long xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[yy * W + xx] : 0.f; };
    float o = tap(iy, ix) * ((1.f - fy) * (1.f - fx));
    o = o + tap(iy, ix + 1) * ((1.f - fy) * fx);
    o = o + tap(iy + 1, ix) * (fy * (1.f - fx));
    o = o + tap(iy + 1, ix + 1) * (fy * fx);
    dst[i] = o;
  }
}

// y = a * (x / (max(x) + 1e-8)) (+ b * y) ; then optionally y /= (max(y) + 1e-8)      (propagate :178-184).  One workgroup.
__global__ __launch_bounds__(1024) void post_blend_kernel(const float* __restrict__ x, float a, float* __restrict__ y, float b, int n,
                                                          int renorm) {
  __shared__ float sm[1024];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, x[i]);
  mx = block_reduce(mx, sm, [](float p, float q) { return fmaxf(p, q); });
  const float den = (float)((double)mx + 1e-8);
  float my = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float v = x[i] / den;
    const float r = b == 0.f ? a * v : a * v + b * y[i];
    y[i] = r;
    my = fmaxf(my, r);
  }
  if (!renorm) return;
  my = block_reduce(my, sm, [](float p, float q) { return fmaxf(p, q); });
  const float den2 = (float)((double)my + 1e-8);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) y[i] = y[i] / den2;
}

// ---- dense CRF (Kraehenbuehl & Koltun 2011), one bilateral Potts term, exact Gaussian kernel inside a (2R+1)^2 window --------
// feat[i] = (r, g, b, g_i) with g_i the field being filtered.  out_i = sum_{j != i, |dy|,|dx| <= R} exp(-|dp|^2 / 2 sxy^2
// - |dI|^2 / 2 srgb^2) * g_j
__global__ __launch_bounds__(256) void crf_filter_kernel(const float4* __restrict__ feat, float* __restrict__ out, int H, int W, int R,
                                                         float inv_sxy2, float inv_srgb2) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const float4 c = feat[(size_t)y * W + x];
  float acc = 0.f;
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -R; dx <= R; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W || (dy == 0 && dx == 0)) continue;
      const float4 f = feat[(size_t)yy * W + xx];
      const float dr = c.x - f.x, dg = c.y - f.y, db = c.z - f.z;
      const float k = expf(-0.5f * ((float)(dy * dy + dx * dx) * inv_sxy2 + ((dr * dr + dg * dg) + db * db) * inv_srgb2));
      acc += k * f.w;
    }
  }
  out[(size_t)y * W + x] = acc;
}
// feat.w <- value * scale (per pixel)
__global__ __launch_bounds__(256) void crf_set_field_kernel(float4* __restrict__ feat, const float* __restrict__ v, const float* __restrict__ scale,
                                                            int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) feat[i].w = (v ? v[i] : 1.f) * (scale ? scale[i] : 1.f);
}
__global__ __launch_bounds__(256) void crf_pack_image_kernel(const unsigned char* __restrict__ img, float4* __restrict__ feat, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    feat[i] = make_float4((float)img[3 * i], (float)img[3 * i + 1], (float)img[3 * i + 2], 1.f);
}
// norm = 1 / sqrt(K 1 + 1e-20)
__global__ __launch_bounds__(256) void crf_norm_kernel(const float* __restrict__ k1, float* __restrict__ norm, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) norm[i] = 1.f / sqrtf(k1[i] + 1e-20f);
}
// Q <- softmax(-unary + compat * msg) with msg_1 = norm * K(norm Q1), msg_0 = norm * (K(norm) - K(norm Q1))   (Q0 + Q1 = 1);
// first: msg = 0.  q1 receives Q1.
__global__ __launch_bounds__(256) void crf_update_kernel(const float* __restrict__ unary, const float* __restrict__ norm,
                                                         const float* __restrict__ kn, const float* __restrict__ kq1, float compat,
                                                         float* __restrict__ q0, float* __restrict__ q1, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float e0 = -unary[i], e1 = -unary[n + i];
    if (kq1) {
      const float m1 = norm[i] * kq1[i], m0 = norm[i] * (kn[i] - kq1[i]);
      e0 += compat * m0;
      e1 += compat * m1;
    }
    const float mx = fmaxf(e0, e1);
    const float p0 = expf(e0 - mx), p1 = expf(e1 - mx), s = p0 + p1;
    q0[i] = p0 / s;
    q1[i] = p1 / s;
  }
}

// separable Gaussian, symmetric ('reflect') boundary, double; axis 0 / 1; weights k[2r+1] on device
__global__ __launch_bounds__(256) void post_gauss1d_kernel(const double* __restrict__ src, double* __restrict__ dst, int H, int W,
                                                           const double* __restrict__ k, int r, int axis) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W, n = axis ? W : H, c = axis ? x : y;
    double acc = 0.0;
    for (int j = -r; j <= r; ++j) {
      int p = c + j;
      while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
      acc += k[j + r] * (axis ? src[(size_t)y * W + p] : src[(size_t)p * W + x]);
    }
    dst[i] = acc;
  }
}

}  // namespace udet

using namespace udet;

extern "C" {

int udet_post_border_mean(const float* s, int n, int h, int w, double* out, void* stream) {
  if (!s || !out || n < 1 || h < 4 || w < 4) { set_error("post_border_mean: bad argument"); return UDET_ERR_ARG; }
  hipLaunchKernelGGL(post_border_mean_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, s, h, w, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_bytescale(const double* src, int ld, int y0, int x0, int h, int w, unsigned char* dst, void* stream) {
  if (!src || !dst || h < 1 || w < 1 || y0 < 0 || x0 < 0 || ld < x0 + w) { set_error("post_bytescale: bad argument"); return UDET_ERR_ARG; }
  hipLaunchKernelGGL(post_bytescale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, src, ld, y0, x0, h, w, dst);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_resample_u8(const unsigned char* src, int h, int w, unsigned char* dst, int oh, int ow, const int* kk, const int* bounds,
                          int ksize, int axis, void* stream) {
  if (!src || !dst || !kk || !bounds || ksize < 1 || (axis ? oh != h : ow != w)) { set_error("post_resample_u8: bad argument"); return UDET_ERR_ARG; }
  int nb = (oh * ow + 255) / 256;
  hipLaunchKernelGGL(post_resample_u8_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, (hipStream_t)stream, src, h, w, dst, oh, ow, kk,
                     bounds, ksize, axis);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_place(const unsigned char* patch, int hh, int ww, int y0, int x0, int h, int w, double* canvas, void* stream) {
  if (!patch || !canvas || hh < 1 || ww < 1 || y0 < 0 || x0 < 0 || y0 + hh > h || x0 + ww > w) { set_error("post_place: bad argument"); return UDET_ERR_ARG; }
  hipLaunchKernelGGL(post_place_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, patch, hh, ww, y0, x0, h, w, canvas);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_minmax_norm(const double* score, int n, double* out, void* stream) {
  if (!score || !out || n < 1) { set_error("post_minmax_norm: bad argument"); return UDET_ERR_ARG; }
  hipLaunchKernelGGL(post_minmax_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, n, out);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_remap(const float* src, const float* flow_uv, float* dst, int h, int w, void* stream) {
  if (!src || !flow_uv || !dst || src == dst || h < 1 || w < 1) { set_error("post_remap: bad argument (src and dst must differ)"); return UDET_ERR_ARG; }
  int nb = (h * w + 255) / 256;
  hipLaunchKernelGGL(post_remap_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, (hipStream_t)stream, src, flow_uv, dst, h, w);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_blend(const float* x, float a, float* y, float b, int n, int renorm, void* stream) {
  if (!x || !y || n < 1) { set_error("post_blend: bad argument"); return UDET_ERR_ARG; }
  hipLaunchKernelGGL(post_blend_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, a, y, b, n, renorm);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int udet_post_gauss1d(const double* src, double* dst, int h, int w, const double* k, int r, int axis, void* stream) {
  if (!src || !dst || !k || src == dst || r < 0) { set_error("post_gauss1d: bad argument"); return UDET_ERR_ARG; }
  int nb = (h * w + 255) / 256;
  hipLaunchKernelGGL(post_gauss1d_kernel, dim3(nb > 1024 ? 1024 : nb), dim3(256), 0, (hipStream_t)stream, src, dst, h, w, k, r, axis);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
size_t udet_post_crf_workspace_bytes(int h, int w) { return ((size_t)h * w * (16 + 4 * 4) + 1024); }
int udet_post_dense_crf(const float* unary, const unsigned char* image_rgb, int h, int w, float sxy, float srgb, float compat, int iters,
                        int radius, float* q, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const int n = h * w;
  if (!unary || !image_rgb || !q || h < 1 || w < 1 || iters < 0 || radius < 1 || sxy <= 0.f || srgb <= 0.f) { set_error("post_dense_crf: bad argument"); return UDET_ERR_ARG; }
  if (!workspace || workspace_bytes < udet_post_crf_workspace_bytes(h, w) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
    set_error("post_dense_crf: workspace needs %zu bytes, 16-byte aligned", udet_post_crf_workspace_bytes(h, w));
    return UDET_ERR_ARG;
  }
  float4* feat = (float4*)workspace;
  float* norm = (float*)(feat + n);
  float* kn = norm + n;
  float* kq = kn + n;
  float* tmp = kq + n;
  const int nb = (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256;
  const dim3 grid((w + 15) / 16, (h + 15) / 16);
  const float isxy = 1.f / (sxy * sxy), isrgb = 1.f / (srgb * srgb);
  hipLaunchKernelGGL(crf_pack_image_kernel, dim3(nb), dim3(256), 0, s, image_rgb, feat, n);
  hipLaunchKernelGGL(crf_filter_kernel, grid, dim3(256), 0, s, feat, tmp, h, w, radius, isxy, isrgb);           // K 1
  hipLaunchKernelGGL(crf_norm_kernel, dim3(nb), dim3(256), 0, s, tmp, norm, n);
  hipLaunchKernelGGL(crf_set_field_kernel, dim3(nb), dim3(256), 0, s, feat, (const float*)nullptr, norm, n);
  hipLaunchKernelGGL(crf_filter_kernel, grid, dim3(256), 0, s, feat, kn, h, w, radius, isxy, isrgb);            // K norm
  hipLaunchKernelGGL(crf_update_kernel, dim3(nb), dim3(256), 0, s, unary, norm, kn, (const float*)nullptr, compat, q, q + n, n);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(crf_set_field_kernel, dim3(nb), dim3(256), 0, s, feat, q + n, norm, n);
    hipLaunchKernelGGL(crf_filter_kernel, grid, dim3(256), 0, s, feat, kq, h, w, radius, isxy, isrgb);          // K (norm Q1)
    hipLaunchKernelGGL(crf_update_kernel, dim3(nb), dim3(256), 0, s, unary, norm, kn, kq, compat, q, q + n, n);
  }
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // extern "C"
