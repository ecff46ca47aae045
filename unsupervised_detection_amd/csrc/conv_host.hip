// Host-side helpers that turn a TF-style convolution description into ConvParams
// (tap lists, sub-grids, TF 'SAME' padding) + the weight packing kernels.
#include <string.h>

#include <stdint.h>

#include "common.h"
#include "conv_host.h"
#include "wino_pack.h"

namespace udet {

void same_pad(int in, int k, int s, int d, int* before, int* out) {
  const int o = (in + s - 1) / s;
  int total = (o - 1) * s + (k - 1) * d + 1 - in;
  if (total < 0) total = 0;
  *before = total / 2;
  *out = o;
}

static inline int floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

// drop taps that fall outside the input grid for every output position of the launch
static void push_tap(ConvParams& p, int dy, int dx, int widx) {
  const int ymax = (p.OHq - 1) * p.isy + dy, xmax = (p.OWq - 1) * p.isx + dx;
  if (dy >= p.H || ymax < 0 || dx >= p.W || xmax < 0) return;
  p.taps[p.ntaps].dy = dy;
  p.taps[p.ntaps].dx = dx;
  p.taps[p.ntaps].widx = widx;
  ++p.ntaps;
}

void conv_setup_fwd(ConvParams& p, int N, int H, int W, int kh, int kw, int s, int d) {
  int pt, pl, oh, ow;
  same_pad(H, kh, s, d, &pt, &oh);
  same_pad(W, kw, s, d, &pl, &ow);
  p.N = N; p.H = H; p.W = W;
  p.OH = oh; p.OW = ow; p.OHq = oh; p.OWq = ow;
  p.osy = p.osx = 1; p.ooy = p.oox = 0;
  p.isy = p.isx = s;
  p.ntaps = 0;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) push_tap(p, ky * d - pt, kx * d - pl, ky * kw + kx);
}

// Number of launches the backward-data pass of a stride-s SAME conv over an (H,W) input needs: stride 2 on an even
// grid runs its four output-parity classes as ONE launch (ConvParams::ncls = 4); odd grids fall back to one launch
// per class.
int conv_dgrad_classes(int s, int H, int W) { return s == 1 ? 1 : ((H % 2 == 0 && W % 2 == 0) ? 1 : s * s); }

// Backward-data of the SAME conv (N,H,W) --k,s,d--> (N,OHf,OWf); also the forward of
// tf.layers.conv2d_transpose (k=4,s=2,'same') when called with H=2h, W=2w.
// The A operand lives on the (OHf,OWf) grid, the result on the (H,W) grid.
// `cls` indexes the launches of conv_dgrad_classes(s,H,W).
bool conv_setup_dgrad(ConvParams& p, int cls, int N, int H, int W, int kh, int kw, int s, int d) {
  int pt, pl, ohf, owf;
  same_pad(H, kh, s, d, &pt, &ohf);
  same_pad(W, kw, s, d, &pl, &owf);
  p.N = N; p.H = ohf; p.W = owf;
  p.OH = H; p.OW = W;
  p.isy = p.isx = 1;
  p.ntaps = 0;
  p.ncls = 1;
  if (s == 1) {
    p.OHq = H; p.OWq = W; p.osy = p.osx = 1; p.ooy = p.oox = 0;
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) push_tap(p, pt - ky * d, pl - kx * d, ky * kw + kx);
    return p.ntaps > 0;
  }
  const bool merged = (H % 2 == 0 && W % 2 == 0);  // s == 2
  p.osy = p.osx = s;
  const int c_lo = merged ? 0 : cls, c_hi = merged ? 4 : cls + 1;
  for (int c = c_lo; c < c_hi; ++c) {
    const int py = c / s, px = c % s;
    p.ooy = py; p.oox = px;
    p.OHq = (H - py + s - 1) / s;
    p.OWq = (W - px + s - 1) / s;
    if (merged) p.cls_tap[c] = p.ntaps;
    if (p.OHq <= 0 || p.OWq <= 0) {
      if (!merged) return false;
      continue;
    }
    for (int ky = 0; ky < kh; ++ky) {
      const int vy = py + pt - ky * d;
      if (((vy % 2) + 2) % 2) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int vx = px + pl - kx * d;
        if (((vx % 2) + 2) % 2) continue;
        push_tap(p, floordiv2(vy), floordiv2(vx), ky * kw + kx);
      }
    }
  }
  if (merged) {
    p.ncls = 4;
    p.cls_tap[4] = p.ntaps;
    p.ooy = p.oox = 0;
    p.OHq = H / 2; p.OWq = W / 2;
  }
  return true;  // a class may have no taps: the kernel still writes zeros / bias for it
}

// ---------------------------------------------------------------------------
// weight packing: src [T][R][C] (HWIO: R=Cin, C=Cout) -> dst [T][Kc][ldw]
//   mode 0 (forward):           dst[t][kmap(r)][c] = src[t][r][c] * scale[c]
//   mode 1 (transposed/dgrad):  dst[t][c][r]       = src[t][r][c] * scale[c]
// kmap inserts a zero gap of k_gap rows at row k_split (slab padding channels).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int T,
                                                           int R, int C, int Kc, int ldw, int k_split, int k_gap,
                                                           int mode, const float* __restrict__ scale) {
  const long total = (long)T * Kc * ldw;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % ldw);
    const int k = (int)((e / ldw) % Kc);
    const int t = (int)(e / ((long)ldw * Kc));
    float v = 0.f;
    int ks = k;  // source index of packed row k (-1 inside the zero gap)
    if (k >= k_split) ks = (k < k_split + k_gap) ? -1 : k - k_gap;
    const int r = mode == 0 ? ks : n, c = mode == 0 ? n : ks;
    if (r >= 0 && r < R && c >= 0 && c < C) {
      v = src[((long)t * R + r) * C + c];
      if (scale) v *= scale[c];
    }
    dst[e] = v;
  }
}

int launch_pack_weights(const float* src, float* dst, int T, int R, int C, int Kc, int ldw, int k_split, int k_gap,
                        int mode, const float* scale, hipStream_t stream) {
  const long total = (long)T * Kc * ldw;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  UDET_LAUNCH(pack_weights_kernel, dim3(nb), dim3(256), 0, stream, src, dst, T, R, C, Kc, ldw, k_split, k_gap,
                     mode, scale);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// merge matrices of pack modes 9 / 10: [parity][variant: interior, last, last - interior][merged tap a][4x4 tap k]
__constant__ float UPB_MERGE[2][3][3][4] = {
    {{{.5f, 0.f, 0.f, 0.f}, {.5f, 1.f, .5f, 0.f}, {0.f, 0.f, .5f, 1.f}},
     {{.5f, 0.f, 0.f, 0.f}, {.5f, 1.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 0.f}},
     {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, .5f, 0.f}, {0.f, 0.f, -.5f, -1.f}}},
    {{{1.f, .5f, 0.f, 0.f}, {0.f, .5f, 1.f, .5f}, {0.f, 0.f, 0.f, .5f}},
     {{1.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}},
     {{0.f, .5f, 0.f, 0.f}, {0.f, -.5f, -1.f, -.5f}, {0.f, 0.f, 0.f, -.5f}}}};

// All re-layout jobs of one network in ONE launch (blockIdx.y = job): the per-step repack of the trainable weights
// was ~140 launches of ~3 us.  BN (inference, moving stats 0/1) is folded on the fly: scale = gamma*c,
// bias' = b*gamma*c + beta.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, const float* __restrict__ wsrc,
                                                        float* __restrict__ ws, float bn_c) {
  const PackJob j = jobs[blockIdx.y];
  const float* src = wsrc + j.src_off;
  float* dst = ws + j.dst_off;
  const float* gamma = j.gamma_off >= 0 ? wsrc + j.gamma_off : nullptr;
  if (j.mode == 2) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256)
      dst[e] = gamma ? src[e] * (gamma[e] * bn_c) + wsrc[j.beta_off + e] : src[e];
    return;
  }
  if (j.mode == 9 || j.mode == 10) {
    // Recover decoder (legacy bilinear x2 + 4x4 SAME convolution) as four 3x3 convolutions on the ringed low-resolution grid: merged
    // weights W~[cls][a][b] = sum_kl Ry[a][k] Rx[b][l] w[k][l], T = 36 = class * 9 + a * 3 + b, class (py, px).  Ry / Rx are the merge
    // matrices of the row / column parity, in the variant the job names (j.beta_off = row variant * 3 + column variant):
    //   0 interior  E = [1/2 0 0 0; 1/2 1 1/2 0; 0 0 1/2 1]   O = [1 1/2 0 0; 0 1/2 1 1/2; 0 0 0 1/2]
    //   1 last low-resolution row / column  E = [1/2 0 0 0; 1/2 1 1 0; 0]   O = [1 1 0 0; 0; 0]      2: last - interior
    // mode 9: dst [36][k = ci (gap map)][n = co]; mode 10 (backward-data): dst [36][k = co][n = ci]
    // One work item per (k, n): its sixteen 4x4 taps are read once and merged separably (rows, then columns) into the 36 outputs.
    const int rv = (int)(j.beta_off / 3), cv = (int)(j.beta_off % 3);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) {
      const int n = (int)(e % j.ldw), k = (int)(e / j.ldw);
      int ks = k;
      if (k >= j.k_split) ks = (k < j.k_split + j.k_gap) ? -1 : k - j.k_gap;
      const int ci = j.mode == 9 ? ks : n, co = j.mode == 9 ? n : ks;
      const bool real = ci >= 0 && ci < j.R && co >= 0 && co < j.C;
      float w[4][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ll = 0; ll < 4; ++ll) w[kk][ll] = real ? src[((long)(kk * 4 + ll) * j.R + ci) * j.C + co] : 0.f;
      const long plane = (long)j.Kc * j.ldw;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float t1[3][4];  // rows merged
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int ll = 0; ll < 4; ++ll) {
            float v = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v += UPB_MERGE[py][rv][a][kk] * w[kk][ll];
            t1[a][ll] = v;
          }
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
              float v = 0.f;
#pragma unroll
              for (int ll = 0; ll < 4; ++ll) v += UPB_MERGE[px][cv][bb][ll] * t1[a][ll];
              dst[(long)((py * 2 + px) * 9 + a * 3 + bb) * plane + e] = v;
            }
      }
    }
    return;
  }
  if (j.mode == 7 || j.mode == 8) {  // Winograd U = G g G^T (conv_wino.hip)
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) wino_pack_item(j, src, gamma, bn_c, dst, e);
    return;
  }
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % j.ldw);
    const int k = (int)((e / j.ldw) % j.Kc);
    int t = (int)(e / ((long)j.ldw * j.Kc));
    float v = 0.f;
    int ks = k;  // source index of packed row k (-1 inside the zero gap)
    if (k >= j.k_split) ks = (k < j.k_split + j.k_gap) ? -1 : k - j.k_gap;
    int r = j.mode == 0 ? ks : n, c = j.mode == 0 ? n : ks;
    if (j.mode == 3 || j.mode == 4) {  // taps folded into the N axis: column n = t*C + c
      t = n / j.C;
      c = n - t * j.C;
      r = ks;
      if (t >= j.T) r = -1;
    }
    if (j.mode >= 5) {
      // NN x2 + 3x3 as four 2x2 convolutions: t = class*4 + tap, class (py,px), tap (ty,tx); the taps of the 3x3 filter that
      // read the same low-resolution row: py=0: {0}, {1,2}; py=1: {0,1}, {2} (columns alike).  mode 5: [t][k=ci][n=co],
      // mode 6 (backward-data): [t][k=co][n=ci]
      const int cls = t >> 2, ty = (t >> 1) & 1, tx = t & 1, py = cls >> 1, px = cls & 1;
      const int y0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), y1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
      const int x0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), x1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
      const int ci = j.mode == 5 ? ks : n, co = j.mode == 5 ? n : ks;
      if (ci >= 0 && ci < j.R && co >= 0 && co < j.C) {
        for (int ky = y0; ky <= y1; ++ky)
          for (int kx = x0; kx <= x1; ++kx) v += src[((long)(ky * 3 + kx) * j.R + ci) * j.C + co];
        if (gamma) v *= gamma[co] * bn_c;
      }
      dst[e] = v;
      continue;
    }
    if (r >= 0 && r < j.R && c >= 0 && c < j.C) {
      v = j.mode == 4 ? src[((long)t * j.C + c) * j.R + r] : src[((long)t * j.R + r) * j.C + c];
      if (gamma) v *= gamma[c] * bn_c;
    }
    dst[e] = v;
  }
}
int launch_pack_jobs(const PackJob* jobs_dev, int njobs, const float* wsrc, float* ws, float bn_c, hipStream_t stream) {
  if (njobs < 1) return UDET_OK;
  UDET_LAUNCH(pack_jobs_kernel, dim3(48, njobs), dim3(256), 0, stream, jobs_dev, wsrc, ws, bn_c);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// one-off variant of the same re-layout for a single job passed by value (PWC-Net packing, single-op entry points)
__global__ __launch_bounds__(256) void pack_job_kernel(const PackJob j, const float* __restrict__ wsrc, float* __restrict__ ws) {
  const float* src = wsrc + j.src_off;
  float* dst = ws + j.dst_off;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % j.ldw);
    const int k = (int)((e / j.ldw) % j.Kc);
    int ks = k;
    if (k >= j.k_split) ks = (k < j.k_split + j.k_gap) ? -1 : k - j.k_gap;
    const int t = n / j.C, c = n - t * j.C;
    float v = 0.f;
    if (ks >= 0 && ks < j.R && t < j.T) v = j.mode == 4 ? src[((long)t * j.C + c) * j.R + ks] : src[((long)t * j.R + ks) * j.C + c];
    dst[e] = v;
  }
}
int launch_pack_taps_into_n(const float* src, float* dst, int T, int R, int C, int Kc, int ldz, int k_split, int k_gap,
                            int transposed, hipStream_t stream) {
  PackJob j;
  memset(&j, 0, sizeof(j));
  j.T = T; j.R = R; j.C = C; j.Kc = Kc; j.ldw = ldz; j.k_split = k_split; j.k_gap = k_gap;
  j.mode = transposed ? 4 : 3; j.total = (long)Kc * ldz; j.gamma_off = -1;
  int nb = (int)((j.total + 255) / 256);
  if (nb > 1024) nb = 1024;
  UDET_LAUNCH(pack_job_kernel, dim3(nb), dim3(256), 0, stream, j, src, dst);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// gather-sum over the taps of a GEMM + gather head (see common.h)
__global__ __launch_bounds__(256) void tap_gather_kernel(const ConvParams g, const float* __restrict__ z, int ldz) {
  const long total = (long)g.N * g.OH * g.OW * g.Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int co = (int)(e % g.Cout);
    const long pix = e / g.Cout;
    const int ox = (int)(pix % g.OW), oy = (int)((pix / g.OW) % g.OH), n = (int)(pix / ((long)g.OW * g.OH));
    int cls = 0, qy = oy, qx = ox;
    if (g.ncls > 1) {
      cls = (oy & 1) * 2 + (ox & 1);
      qy = oy >> 1;
      qx = ox >> 1;
    }
    float v = g.bias ? g.bias[co] : 0.f;
    // eight taps in flight per trip (a branchy one-load-per-trip loop pays one L2 latency per tap: 25 of them for the
    // 5x5 head); taps outside the image / past the class contribute +0, added in tap order
    const int t1 = g.cls_tap[cls + 1];
    for (int t0 = g.cls_tap[cls]; t0 < t1; t0 += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u < t1 ? t0 + u : t1 - 1;
        const int iy = qy + g.taps[t].dy, ix = qx + g.taps[t].dx;
        const bool ok = t0 + u < t1 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        a[u] = ok ? z[((size_t)(n * g.H + iy) * g.W + ix) * ldz + g.taps[t].widx * g.Cout + co] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v += a[u];
    }
    g.y[(size_t)pix * g.ldy + g.y_coff + co] = v;
  }
}
// the 2-channel heads (every flow / up_feat / upflow layer): one thread per pixel, float2 per tap
__global__ __launch_bounds__(256) void tap_gather2_kernel(const ConvParams g, const float* __restrict__ z, int ldz) {
  const long total = (long)g.N * g.OH * g.OW;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int ox = (int)(pix % g.OW), oy = (int)((pix / g.OW) % g.OH), n = (int)(pix / ((long)g.OW * g.OH));
    int cls = 0, qy = oy, qx = ox;
    if (g.ncls > 1) {
      cls = (oy & 1) * 2 + (ox & 1);
      qy = oy >> 1;
      qx = ox >> 1;
    }
    float2 v = g.bias ? make_float2(g.bias[0], g.bias[1]) : make_float2(0.f, 0.f);
    const int t1 = g.cls_tap[cls + 1];
    for (int t0 = g.cls_tap[cls]; t0 < t1; t0 += 8) {
      float2 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u < t1 ? t0 + u : t1 - 1;
        const int iy = qy + g.taps[t].dy, ix = qx + g.taps[t].dx;
        const bool ok = t0 + u < t1 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        a[u] = ok ? *reinterpret_cast<const float2*>(z + ((size_t)(n * g.H + iy) * g.W + ix) * ldz + g.taps[t].widx * 2)
                  : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v.x += a[u].x;
        v.y += a[u].y;
      }
    }
    *reinterpret_cast<float2*>(g.y + (size_t)pix * g.ldy + g.y_coff) = v;
  }
}
int launch_tap_gather(const ConvParams& g, const float* z, int ldz, hipStream_t stream) {
  ConvParams q = g;
  if (q.ncls != 4) { q.ncls = 1; q.cls_tap[0] = 0; q.cls_tap[1] = q.ntaps; }
  if (q.Cout == 2 && ldz % 2 == 0 && q.ldy % 2 == 0 && q.y_coff % 2 == 0 && !((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(q.y)) & 7)) {
    const long pixels = (long)q.N * q.OH * q.OW;
    int nb2 = (int)((pixels + 255) / 256);
    if (nb2 > 4096) nb2 = 4096;
    UDET_LAUNCH(tap_gather2_kernel, dim3(nb2), dim3(256), 0, stream, q, z, ldz);
    UDET_HIP(hipGetLastError());
    return UDET_OK;
  }
  const long total = (long)q.N * q.OH * q.OW * q.Cout;
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  UDET_LAUNCH(tap_gather_kernel, dim3(nb), dim3(256), 0, stream, q, z, ldz);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// BN (inference, moving stats 0/1) folded into the conv:  scale = gamma*c, bias' = b*gamma*c + beta
__global__ void fold_bn_kernel(const float* __restrict__ b, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float c, float* __restrict__ scale, float* __restrict__ bias_f, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = gamma[i] * c;
    scale[i] = s;
    bias_f[i] = b[i] * s + beta[i];
  }
}
int launch_fold_bn(const float* b, const float* gamma, const float* beta, float c, float* scale, float* bias_f, int n,
                   hipStream_t stream) {
  UDET_LAUNCH(fold_bn_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, b, gamma, beta, c, scale, bias_f, n);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// strided channel copy: dst[p][d_coff + c] = src[p][s_coff + c] * mul, c < C
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int lds, int s_coff,
                                                            float* __restrict__ dst, int ldd, int d_coff, long P, int C,
                                                            float mul, float add) {
  const long total = P * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long pix = e / C;
    const int c = (int)(e - pix * C);
    dst[pix * ldd + d_coff + c] = src[pix * lds + s_coff + c] * mul + add;
  }
}
int launch_copy_channels(const float* src, int lds, int s_coff, float* dst, int ldd, int d_coff, long P, int C, float mul,
                         float add, hipStream_t stream) {
  const long total = P * C;
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  UDET_LAUNCH(copy_channels_kernel, dim3(nb), dim3(256), 0, stream, src, lds, s_coff, dst, ldd, d_coff, P, C, mul,
                     add);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
