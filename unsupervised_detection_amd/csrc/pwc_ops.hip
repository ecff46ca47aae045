// Bandwidth-bound PWC-Net kernels: dense backward warp and the 81-channel cost volume.
#include <stdlib.h>

#include "common.h"

namespace udet {

// ---------------------------------------------------------------------------
// dense_image_warp  (models/PWCNet/core_warp.py:153-202, _interpolate_bilinear :42-150)
// One thread per (pixel, 4 channels).  The grid-index math is done with explicitly
// rounded fp32 operations (no FMA contraction) so that floor/ceil indices and alphas
// are bit-identical to the reference's float32 graph:
//   q = grid - flow*scale ; floor = min(max(0,floor(q)), size-2) ; alpha = clamp(q-floor,0,1)
//   top = ax*(tr-tl)+tl ; bot = ax*(br-bl)+bl ; out = ay*(bot-top)+top
// ---------------------------------------------------------------------------
// This translation unit is compiled with -ffp-contract=off (see Makefile): every * + - below
// rounds separately, exactly like the reference's float32 TF graph.
__device__ __forceinline__ float lerp_rn(float a, float lo, float hi) {
#pragma clang fp contract(off)
  const float d = hi - lo;
  const float m = a * d;
  return m + lo;
}

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ img, const float* __restrict__ flow, int ldf,
                                                   int f_coff, float flow_scale, float* __restrict__ out, int N, int H,
                                                   int W, int C, int* __restrict__ dbg_idx, float* __restrict__ dbg_alpha) {
  const int c4n = C >> 2;
  const long total = (long)N * H * W * c4n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % c4n);
    const long pix = e / c4n;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int n = (int)(pix / ((long)W * H));
    const float* f = flow + pix * ldf + f_coff;
    const float fy = f[0] * flow_scale, fx = f[1] * flow_scale;
    const float qy = (float)y - fy, qx = (float)x - fx;
    const float flo_y = fminf(fmaxf(0.f, floorf(qy)), (float)(H - 2));
    const float flo_x = fminf(fmaxf(0.f, floorf(qx)), (float)(W - 2));
    const int iy = (int)flo_y, ix = (int)flo_x;
    const float ay = fminf(fmaxf(0.f, (qy - flo_y)), 1.f);
    const float ax = fminf(fmaxf(0.f, (qx - flo_x)), 1.f);
    if (dbg_idx && c4 == 0) {
      dbg_idx[pix * 2 + 0] = iy;
      dbg_idx[pix * 2 + 1] = ix;
      dbg_alpha[pix * 2 + 0] = ay;
      dbg_alpha[pix * 2 + 1] = ax;
    }
    const float* base = img + (((long)n * H + iy) * W + ix) * C + c4 * 4;
    const float4 tl = *reinterpret_cast<const float4*>(base);
    const float4 tr = *reinterpret_cast<const float4*>(base + C);
    const float4 bl = *reinterpret_cast<const float4*>(base + (long)W * C);
    const float4 br = *reinterpret_cast<const float4*>(base + (long)W * C + C);
    float4 o;
    o.x = lerp_rn(ay, lerp_rn(ax, tl.x, tr.x), lerp_rn(ax, bl.x, br.x));
    o.y = lerp_rn(ay, lerp_rn(ax, tl.y, tr.y), lerp_rn(ax, bl.y, br.y));
    o.z = lerp_rn(ay, lerp_rn(ax, tl.z, tr.z), lerp_rn(ax, bl.z, br.z));
    o.w = lerp_rn(ay, lerp_rn(ax, tl.w, tr.w), lerp_rn(ax, bl.w, br.w));
    *reinterpret_cast<float4*>(out + pix * C + c4 * 4) = o;
  }
}

int launch_warp(const float* img, const float* flow, int ldf, int f_coff, float flow_scale, float* out, int N, int H,
                int W, int C, int* dbg_idx, float* dbg_alpha, hipStream_t stream) {
  if (C % 4 != 0 || H < 2 || W < 2) {
    set_error("warp: C=%d must be a multiple of 4 and H,W >= 2 (got %dx%d)", C, H, W);
    return UDET_ERR_SHAPE;
  }
  const long total = (long)N * H * W * (C / 4);
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  UDET_LAUNCH(warp_kernel, dim3(nb), dim3(256), 0, stream, img, flow, ldf, f_coff, flow_scale, out, N, H, W, C,
                     dbg_idx, dbg_alpha);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// cost_volume  (models/PWCNet/core_costvol.py:20-40): out[.., dy*9+dx] =
//   leaky0.1( mean_c( c1[y,x,c] * warp[y+dy-4, x+dx-4, c] ) ), zero outside.
// One workgroup per 8x8 pixel tile: the 16x16 halo of `warp` and the 8x8 tile of c1
// are staged through LDS in 32-channel slices (each input byte is read from HBM once,
// the 81x re-use happens in LDS); thread (pixel, g) accumulates displacements g, g+4, ...
// ---------------------------------------------------------------------------
#define CV_T 8
#define CV_R 4
#define CV_HALO (CV_T + 2 * CV_R)
#define CV_CS 36  // 32-channel slice + 4 pad floats: 144-B pixel stride spreads ds_read_b128 over all 16 slots
#define CV_ND 21

__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ c1, const float* __restrict__ wr,
                                                          float* __restrict__ out, int ldo, int o_coff, int N, int H,
                                                          int W, int C) {
  __shared__ __attribute__((aligned(16))) float sw[CV_HALO * CV_HALO * CV_CS];
  __shared__ __attribute__((aligned(16))) float s1[CV_T * CV_T * CV_CS];
  const int t = threadIdx.x;
  const int tiles_x = (W + CV_T - 1) / CV_T, tiles_y = (H + CV_T - 1) / CV_T;
  // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs, each with its own L2): consecutive tiles -- whose
  // 16x16 halos overlap -- go to the same XCD, so the overlap is fetched once per XCD instead of once per tile
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int y0 = by * CV_T, x0 = bx * CV_T;
  const int p = t & 63, g = t >> 6;
  const int py = p >> 3, px = p & 7;
  float acc[CV_ND];
#pragma unroll
  for (int j = 0; j < CV_ND; ++j) acc[j] = 0.f;

  // Global -> registers -> LDS: all loads of a 32-channel slice are issued back to back (one memory latency per slice, not
  // one per loop trip), and the next slice is fetched before this slice's arithmetic so that its latency hides behind it.
  constexpr int HL = CV_HALO * CV_HALO * 8 / 256;  // float4 halo loads per thread (8)
  constexpr int TL = CV_T * CV_T * 8 / 256;        // float4 c1 loads per thread (2)
  int hoff[HL], toff[TL];                           // global element offsets (-1: outside the image); N*H*W*C < 2^31
#pragma unroll
  for (int u = 0; u < HL; ++u) {
    const int hp = (t + u * 256) >> 3;
    const int hy = hp / CV_HALO, hx = hp - hy * CV_HALO;
    const int yy = y0 + hy - CV_R, xx = x0 + hx - CV_R;
    hoff[u] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? ((n * H + yy) * W + xx) * C : -1;
  }
#pragma unroll
  for (int u = 0; u < TL; ++u) {
    const int tp = (t + u * 256) >> 3;
    const int yy = y0 + (tp >> 3), xx = x0 + (tp & 7);
    toff[u] = (yy < H && xx < W) ? ((n * H + yy) * W + xx) * C : -1;
  }
  const int c4 = t & 7;  // (t + u*256) & 7
  float4 hv[HL], tv[TL];
  auto fetch = [&](int cb) {
    const bool cok = c4 * 4 < min(32, C - cb);
#pragma unroll
    for (int u = 0; u < HL; ++u) {
      hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok && hoff[u] >= 0) hv[u] = *reinterpret_cast<const float4*>(wr + hoff[u] + cb + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < TL; ++u) {
      tv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok && toff[u] >= 0) tv[u] = *reinterpret_cast<const float4*>(c1 + toff[u] + cb + c4 * 4);
    }
  };
  fetch(0);
  for (int cb = 0; cb < C; cb += 32) {
#pragma unroll
    for (int u = 0; u < HL; ++u) *reinterpret_cast<float4*>(&sw[((t + u * 256) >> 3) * CV_CS + c4 * 4]) = hv[u];
#pragma unroll
    for (int u = 0; u < TL; ++u) *reinterpret_cast<float4*>(&s1[((t + u * 256) >> 3) * CV_CS + c4 * 4]) = tv[u];
    __syncthreads();
    if (cb + 32 < C) fetch(cb + 32);
    float4 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const float4*>(&s1[p * CV_CS + q * 4]);
#pragma unroll
    for (int j = 0; j < CV_ND; ++j) {
      const int d = g + 4 * j;
      if (d < 81) {
        const int dy = d / 9, dx = d - dy * 9;
        const float* wp = &sw[((py + dy) * CV_HALO + px + dx) * CV_CS];
        float s = acc[j];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(wp + q * 4);
          s = fmaf(a[q].x, b.x, s);
          s = fmaf(a[q].y, b.y, s);
          s = fmaf(a[q].z, b.z, s);
          s = fmaf(a[q].w, b.w, s);
        }
        acc[j] = s;
      }
    }
    __syncthreads();
  }
  // results -> LDS [64][81] -> coalesced 81-float rows
  float* so = sw;  // reuse (64*81 floats < halo buffer)
#pragma unroll
  for (int j = 0; j < CV_ND; ++j) {
    const int d = g + 4 * j;
    if (d < 81) {
      float v = acc[j] / (float)C;
      so[p * 81 + d] = v > 0.f ? v : 0.1f * v;
    }
  }
  __syncthreads();
  for (int e = t; e < 64 * 81; e += 256) {
    const int tp = e / 81, d = e - tp * 81;
    const int yy = y0 + (tp >> 3), xx = x0 + (tp & 7);
    if (yy < H && xx < W) out[(((long)n * H + yy) * W + xx) * ldo + o_coff + d] = so[e];
  }
}

int launch_cost_volume(const float* c1, const float* wr, float* out, int ldo, int o_coff, int N, int H, int W, int C,
                       hipStream_t stream) {
  if (C % 4 != 0) {
    set_error("cost_volume: C=%d must be a multiple of 4", C);
    return UDET_ERR_SHAPE;
  }
  if ((long)N * H * W * C >= (1L << 31)) {
    set_error("cost_volume: tensor too large for 32-bit element offsets");
    return UDET_ERR_SHAPE;
  }
  const int tiles = ((W + CV_T - 1) / CV_T) * ((H + CV_T - 1) / CV_T) * N;
  UDET_LAUNCH(cost_volume_kernel, dim3(tiles), dim3(256), 0, stream, c1, wr, out, ldo, o_coff, N, H, W, C);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---------------------------------------------------------------------------
// Fused warp -> cost volume (+ the c1 segment of the level slab): model_pwcnet.py:616-623 in ONE launch per pyramid level.
//   warped = dense_image_warp(c2, flow * scale)            (core_warp.py:153-202; skipped when flow == null: level 6)
//   out[.., corr_coff + d] = leaky0.1(mean_c c1 * warped@d)  (core_costvol.py:20-40)
//   out[.., c1_coff + c]   = c1                              (the tf.concat([corr, c1, up_flow, up_feat]) of :622)
// One workgroup per TxT pixel tile.  Thread hp computes the grid-index math of halo pixel hp once (floor indices and alphas,
// the same explicitly rounded operations as warp_kernel -> bit-identical); per 32-channel slice the (T+8)^2 halo of the
// WARPED features is produced straight into LDS from four corner gathers of c2 (the warped tensor never exists in HBM), the
// c1 tile goes to LDS and to the slab, and thread (pixel, g) accumulates displacements g, g+NG, ... exactly like
// cost_volume_kernel (same FMA order: results are bit-identical to udet_warp followed by udet_cost_volume).
// xcds < 8: only workgroups whose hardware slot falls on the first `xcds` XCDs work (the others exit): the small pyramid
// levels then stay inside one or two L2s instead of every XCD fetching the whole level for a handful of tiles.
// ---------------------------------------------------------------------------
template <int T, int QS>
__global__ __launch_bounds__(256, QS == 4 ? 4 : 2) void warp_cost_volume_kernel(const float* __restrict__ c1, const float* __restrict__ c2,
                                                               const float* __restrict__ flow, int ldf, int f_coff, float flow_scale,
                                                               float* __restrict__ out, int ldo, int corr_coff, int c1_coff,
                                                               float* __restrict__ warped_dbg, int N, int H, int W, int C, int xcds) {
  constexpr int HALO = T + 2 * CV_R, NHP = HALO * HALO, NP = T * T, NG = 256 / NP, ND = (81 + NG - 1) / NG;
  constexpr int CS = QS * 4 + 4;             // LDS pixel stride: a slice of QS channel quads + 4 pad floats (conflict-free ds_read_b128)
  constexpr int QSH = QS == 8 ? 3 : 2;       // log2(QS)
  constexpr int HL = (NHP * QS + 255) / 256;  // float4 halo items per thread and slice
  constexpr int TL = (NP * QS + 255) / 256;   // float4 c1 items per thread and slice
  constexpr int GB = 2;                      // gather batch: items per thread whose four corner loads are in flight together
  __shared__ __attribute__((aligned(16))) float sw[(NHP * CS > NP * 81 ? NHP * CS : NP * 81)];
  __shared__ __attribute__((aligned(16))) float s1[NP * CS];
  __shared__ int h_off[NHP];       // element offset of the top-left corner in c2 (-1: halo pixel outside the image)
  __shared__ float2 h_a[NHP];      // (alpha_y, alpha_x)
  const int t = threadIdx.x;
  const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
  int bid = blockIdx.x;
  {
    const int slot = bid & 7;
    if (slot >= xcds) return;
    const int nwg = (gridDim.x >> 3) * xcds, idx = bid >> 3;  // gridDim.x is a multiple of 8
    const int q = nwg / xcds;                                  // workgroups per active XCD (nwg is a multiple of xcds)
    bid = slot * q + idx;                                      // consecutive tiles (overlapping halos) on the same XCD
    if (bid >= tiles_x * tiles_y * N) return;
  }
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int y0 = by * T, x0 = bx * T;
  const int p = t % NP, g = t / NP;
  const int py = p / T, px = p % T;

  // ---- grid-index math of this thread's halo pixel (core_warp.py:99-115), bit-identical to warp_kernel ----
  if (t < NHP) {
    const int hy = t / HALO, hx = t - hy * HALO;
    const int yy = y0 + hy - CV_R, xx = x0 + hx - CV_R;
    int off = -1;
    float ay = 0.f, ax = 0.f;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      if (flow) {
        const float* f = flow + (((long)n * H + yy) * W + xx) * ldf + f_coff;
        const float fy = f[0] * flow_scale, fx = f[1] * flow_scale;
        const float qy = (float)yy - fy, qx = (float)xx - fx;
        const float flo_y = fminf(fmaxf(0.f, floorf(qy)), (float)(H - 2));
        const float flo_x = fminf(fmaxf(0.f, floorf(qx)), (float)(W - 2));
        ay = fminf(fmaxf(0.f, (qy - flo_y)), 1.f);
        ax = fminf(fmaxf(0.f, (qx - flo_x)), 1.f);
        off = ((n * H + (int)flo_y) * W + (int)flo_x) * C;
      } else {
        off = ((n * H + yy) * W + xx) * C;
      }
    }
    h_off[t] = off;
    h_a[t] = make_float2(ay, ax);
  }
  int toff[TL];  // c1 tile pixel offsets (-1 outside)
#pragma unroll
  for (int u = 0; u < TL; ++u) {
    const int e = t + u * 256, tp = e >> QSH;
    const int yy = y0 + tp / T, xx = x0 + tp % T;
    toff[u] = (tp < NP && yy < H && xx < W) ? ((n * H + yy) * W + xx) : -1;
  }
  float acc[ND];
  int woff[ND];  // LDS offset of displacement g + NG*j's halo pixel for this thread's pixel
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    acc[j] = 0.f;
    const int d = g + NG * j, dy = d / 9, dx = d - dy * 9;
    woff[j] = ((py + dy) * HALO + px + dx) * CS;
  }
  const int c4 = t & (QS - 1);
  const long rowC = (long)W * C;
  __syncthreads();

  for (int cb = 0; cb < C; cb += QS * 4) {
    const bool cok = c4 * 4 < min(QS * 4, C - cb);
    // c1 tile: LDS + the slab's c1 segment
#pragma unroll
    for (int u = 0; u < TL; ++u) {
      const int e = t + u * 256;
      if (e < NP * QS) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok && toff[u] >= 0) {
          v = *reinterpret_cast<const float4*>(c1 + (long)toff[u] * C + cb + c4 * 4);
          if (c1_coff >= 0) *reinterpret_cast<float4*>(out + (long)toff[u] * ldo + c1_coff + cb + c4 * 4) = v;
        }
        *reinterpret_cast<float4*>(&s1[(e >> QSH) * CS + c4 * 4]) = v;
      }
    }
    // warped halo: four corner gathers per (halo pixel, channel quad), in batches of GB items (4*GB loads in flight)
#pragma unroll 1  // one batch of 16 gathers in flight (unrolling both batches doubled the VGPR count)
    for (int ub = 0; ub < HL; ub += GB) {
      float4 tl[GB], tr[GB], bl[GB], br[GB];
      int hp[GB];
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int e = t + (ub + k) * 256;
        hp[k] = e >> QSH;
        tl[k] = tr[k] = bl[k] = br[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ub + k < HL && hp[k] < NHP) {
          const int o = h_off[hp[k]];
          if (cok && o >= 0) {
            const float* base = c2 + o + cb + c4 * 4;
            tl[k] = *reinterpret_cast<const float4*>(base);
            if (flow) {
              tr[k] = *reinterpret_cast<const float4*>(base + C);
              bl[k] = *reinterpret_cast<const float4*>(base + rowC);
              br[k] = *reinterpret_cast<const float4*>(base + rowC + C);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        if (ub + k < HL && hp[k] < NHP) {
          float4 o = tl[k];
          if (flow) {
            const float2 a = h_a[hp[k]];
            o.x = lerp_rn(a.x, lerp_rn(a.y, tl[k].x, tr[k].x), lerp_rn(a.y, bl[k].x, br[k].x));
            o.y = lerp_rn(a.x, lerp_rn(a.y, tl[k].y, tr[k].y), lerp_rn(a.y, bl[k].y, br[k].y));
            o.z = lerp_rn(a.x, lerp_rn(a.y, tl[k].z, tr[k].z), lerp_rn(a.y, bl[k].z, br[k].z));
            o.w = lerp_rn(a.x, lerp_rn(a.y, tl[k].w, tr[k].w), lerp_rn(a.y, bl[k].w, br[k].w));
          }
          *reinterpret_cast<float4*>(&sw[hp[k] * CS + c4 * 4]) = o;
          if (warped_dbg && cok) {  // test hook: the tile's own (centre) pixels of the warped tensor
            const int hy = hp[k] / HALO, hx = hp[k] - hy * HALO;
            const int yy = y0 + hy - CV_R, xx = x0 + hx - CV_R;
            if (hy >= CV_R && hy < CV_R + T && hx >= CV_R && hx < CV_R + T && yy < H && xx < W)
              *reinterpret_cast<float4*>(warped_dbg + (((long)n * H + yy) * W + xx) * C + cb + c4 * 4) = o;
          }
        }
      }
    }
    __syncthreads();
    // channel quad outer, displacement inner: one a quad and a handful of b quads live at a time (the displacement-outer form
    // kept 8 a quads + all unrolled b loads alive: 196 VGPRs, 2 waves per SIMD).  Every accumulator still sees its channels in
    // the order q = 0..7, (x, y, z, w): bit-identical to cost_volume_kernel.
#pragma unroll 1
    for (int q = 0; q < QS; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(&s1[p * CS + q * 4]);
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        if (g + NG * j < 81) {
          const float4 b = *reinterpret_cast<const float4*>(&sw[woff[j] + q * 4]);
          float s = acc[j];
          s = fmaf(a.x, b.x, s);
          s = fmaf(a.y, b.y, s);
          s = fmaf(a.z, b.z, s);
          s = fmaf(a.w, b.w, s);
          acc[j] = s;
        }
      }
    }
    __syncthreads();
  }
  float* so = sw;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int d = g + NG * j;
    if (d < 81) {
      const float v = acc[j] / (float)C;
      so[p * 81 + d] = v > 0.f ? v : 0.1f * v;
    }
  }
  __syncthreads();
  for (int e = t; e < NP * 81; e += 256) {
    const int tp = e / 81, d = e - tp * 81;
    const int yy = y0 + tp / T, xx = x0 + tp % T;
    if (yy < H && xx < W) out[(((long)n * H + yy) * W + xx) * ldo + corr_coff + d] = so[e];
  }
}

int launch_warp_cost_volume(const float* c1, const float* c2, const float* flow, int ldf, int f_coff, float flow_scale, float* out,
                            int ldo, int corr_coff, int c1_coff, float* warped_dbg, int N, int H, int W, int C, hipStream_t stream) {
  if (C % 4 != 0 || (flow && (H < 2 || W < 2))) {
    set_error("warp_cost_volume: C=%d must be a multiple of 4 and H,W >= 2 (got %dx%d)", C, H, W);
    return UDET_ERR_SHAPE;
  }
  if ((long)N * H * W * (C > ldo ? C : ldo) >= (1L << 31)) {
    set_error("warp_cost_volume: tensor too large for 32-bit element offsets");
    return UDET_ERR_SHAPE;
  }
  if (c1_coff >= 0 && ((c1_coff | ldo) & 3)) {
    set_error("warp_cost_volume: c1 segment offset %d / row stride %d must be multiples of 4", c1_coff, ldo);
    return UDET_ERR_ALIGN;
  }
  // small levels: 4x4 tiles (4x the workgroups) on as few XCDs as still give every tile its own CU
  const long pixels = (long)N * H * W;
  const int T = pixels <= 4096 ? 4 : 8;
  const int tiles = ((W + T - 1) / T) * ((H + T - 1) / T) * N;
  const int xcds = tiles <= 32 ? 1 : (tiles <= 64 ? 2 : (tiles <= 128 ? 4 : 8));
  const int per = (tiles + xcds - 1) / xcds;
  const dim3 grid(per * 8);
  // 8x8 tiles stage 16-channel slices (29 KB of LDS: four workgroups per CU -- the 960 tiles of level 2 are one round, with
  // 32-channel slices they were 1.25 rounds of three: 33 -> 28 us); the small levels' 4x4 tiles keep 32 channels per barrier pair
  if (T == 4)
    UDET_LAUNCH((warp_cost_volume_kernel<4, 8>), grid, dim3(256), 0, stream, c1, c2, flow, ldf, f_coff, flow_scale, out, ldo, corr_coff,
                       c1_coff, warped_dbg, N, H, W, C, xcds);
  else
    UDET_LAUNCH((warp_cost_volume_kernel<8, 4>), grid, dim3(256), 0, stream, c1, c2, flow, ldf, f_coff, flow_scale, out, ldo, corr_coff,
                       c1_coff, warped_dbg, N, H, W, C, xcds);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
