// Direct (non-MFMA) convolution kernels for the two degenerate GEMM shapes of the path: the 2-channel heads.
//
// Every flow / up_feat / upflow head of PWC-Net and of the recover decoder (models/PWCNet/model_pwcnet.py:283-286,503-506;
// models/nets.py:81-107) has TWO output channels over a deep input (50 .. 664 channels), and its backward-data pass has two
// INPUT channels and a wide output.  On the matrix cores both are 2 useful columns (rows) of a 32-wide tile; the step ran them as
// a 1x1 GEMM over (tap, channel) columns + a gather pass (two or three launches, the 5x5 head 92 us) and as K = 36 .. 100 implicit
// GEMMs at 1 - 15 TFLOP/s.  Here:
//   conv_thin_n_kernel  (N <= 2 outputs): a workgroup stages the input halo of its 8x32 (4x64) output pixels once per block of 32
//                       channels as [channel][pixel] in LDS; a thread owns one output pixel and walks taps x channels with one
//                       conflict-free ds_read_b32 and two FMAs whose weight operands are SCALAR loads (the weights of a
//                       (tap, channel) are the same for every pixel).  The four output-parity classes of a transposed head share the
//                       staged tile.  Bound by the LDS read rate: 25 taps x 50 channels on 221 k pixels in ~20 us.
//   conv_thin_k_kernel  (K <= 2 real input channels, wide output): a lane owns one OUTPUT channel and keeps its (tap, k) weights in
//                       registers; a wave walks pixels, whose 2-channel input window comes from a small LDS tile by broadcast
//                       reads.  Stores are 256-byte rows.  Bound by the output traffic.
// Same ConvParams contract and epilogue as the implicit-GEMM kernels; selected by launch_conv (families 7 / 8), verified against the
// implicit-GEMM result by the autotuner like every other family.
#include "common.h"

namespace udet {

__device__ __forceinline__ void thin_epilogue(const ConvParams& p, int off, int n, float v) {
  if (p.bias) v += p.bias[n];
  v = act_fwd(v, p.act, p.alpha);
  if (p.y2) p.y2[(size_t)off * p.ldy2 + p.y2_coff + n] = v;
  if (p.res) v += p.res[(size_t)off * p.ldres + p.res_coff + n];
  float* dst = p.y + (size_t)off * p.ldy + p.y_coff + n;
  if (p.accumulate) v += *dst;
  *dst = v;
  if (p.uo && n >= p.u_c0 && n < p.u_c1)
    p.uo[(size_t)off * p.ldu + p.u_coff + n] = v * act_dfo(p.ua[(size_t)off * p.ldua + p.ua_coff + n], p.uact, p.ualpha);
}

// The same epilogue for a row of W pixels of one output channel (pixel u at off0 + u * osx, the first `nvalid` exist), in two phases:
// every load of the row (accumulate / residual / saved activation) is issued before the first store.  Pixel by pixel, each load waits
// for the previous pixel's store -- the compiler must assume they alias -- and a launch whose arithmetic takes 20 us spends 75 in
// its epilogue (the 5x5 flow head's backward-data pass).
// (the operands a row's epilogue reads -- accumulate / residual / saved activation -- can be fetched ahead of the row's multiply-adds)
template <int W>
__device__ __forceinline__ void thin_row_fetch(const ConvParams& p, int off0, int nvalid, int n, float (&accv)[W], float (&resv)[W], float (&uav)[W]) {
  const bool emit = p.uo && n >= p.u_c0 && n < p.u_c1;
#pragma unroll
  for (int u = 0; u < W; ++u) {
    const size_t off = (size_t)(off0 + u * p.osx);
    const bool on = u < nvalid;
    accv[u] = (p.accumulate && on) ? p.y[off * p.ldy + p.y_coff + n] : 0.f;
    resv[u] = (p.res && on) ? p.res[off * p.ldres + p.res_coff + n] : 0.f;
    uav[u] = (emit && on) ? p.ua[off * p.ldua + p.ua_coff + n] : 0.f;
  }
}
template <int W>
__device__ __forceinline__ void thin_row_apply(const ConvParams& p, int off0, int nvalid, int n, const float (&val)[W], const float (&accv)[W],
                                               const float (&resv)[W], const float (&uav)[W]) {
  const bool emit = p.uo && n >= p.u_c0 && n < p.u_c1;
  const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int u = 0; u < W; ++u) {
    if (u >= nvalid) break;
    const size_t off = (size_t)(off0 + u * p.osx);
    float v = act_fwd(val[u] + bias, p.act, p.alpha);
    if (p.y2) p.y2[off * p.ldy2 + p.y2_coff + n] = v;
    if (p.res) v += resv[u];
    if (p.accumulate) v += accv[u];
    p.y[off * p.ldy + p.y_coff + n] = v;
    if (emit) p.uo[off * p.ldu + p.u_coff + n] = v * act_dfo(uav[u], p.uact, p.ualpha);
  }
}
template <int W>
__device__ __forceinline__ void thin_epilogue_row(const ConvParams& p, int off0, int nvalid, int n, const float (&val)[W]) {
  float accv[W], resv[W], uav[W];
  thin_row_fetch<W>(p, off0, nvalid, n, accv, resv, uav);
  thin_row_apply<W>(p, off0, nvalid, n, val, accv, resv, uav);
}

struct ThinGeom {
  int min_dy, min_dx, PH, PW;  // halo tile of one workgroup on the (logical) input grid
};
struct ThinWin { int widx_at[25]; };  // window position r * KW + c -> weight matrix of the tap there (resolved on the host: a
                                       // kernarg table indexed through another one is 25 dependent scalar loads per workgroup)
static void thin_geom(const ConvParams& p, int th, int tw, ThinGeom* g) {
  int mn_y = 0, mx_y = 0, mn_x = 0, mx_x = 0;
  for (int t = 0; t < p.ntaps; ++t) {
    if (t == 0 || p.taps[t].dy < mn_y) mn_y = p.taps[t].dy;
    if (t == 0 || p.taps[t].dy > mx_y) mx_y = p.taps[t].dy;
    if (t == 0 || p.taps[t].dx < mn_x) mn_x = p.taps[t].dx;
    if (t == 0 || p.taps[t].dx > mx_x) mx_x = p.taps[t].dx;
  }
  g->min_dy = mn_y; g->min_dx = mn_x;
  g->PH = (th - 1) * p.isy + (mx_y - mn_y) + 1;
  g->PW = (tw - 1) * p.isx + (mx_x - mn_x) + 1;
}

// ------------------------------------------------------------------------------------------------------------ thin N ----
#define THIN_CB 32  // channels staged per pass
template <int TH>
__global__ __launch_bounds__(256) void conv_thin_n_kernel(const ConvParams p, const ThinGeom g) {
  constexpr int TW = 256 / TH;
  extern __shared__ __attribute__((aligned(16))) float xs_[];  // [THIN_CB][PIXP] input tile | [ntaps][THIN_CB] float2 weights
  const int PIX = g.PH * g.PW, PIXP = PIX | 1;
  float2* ws_ = reinterpret_cast<float2*>(xs_ + (((size_t)THIN_CB * PIXP + 1) & ~(size_t)1));
  const int t = threadIdx.x, ty = t / TW, tx = t - ty * TW;
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = by * TH, ox0 = bx * TW;
  const int iy0 = oy0 * p.isy + g.min_dy, ix0 = ox0 * p.isx + g.min_dx;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const float* xb = p.x + (size_t)n * Hs * Ws * p.ldx + p.x_coff;
  const int ncls = p.ncls > 1 ? 4 : 1;
  const int base = (ty * p.isy) * g.PW + tx * p.isx;
  float acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = 0.f;

  for (int c0 = 0; c0 < p.Kc; c0 += THIN_CB) {
    const int cw = p.Kc - c0 < THIN_CB ? p.Kc - c0 : THIN_CB, cq = cw >> 2;
    __syncthreads();  // the previous pass's reads are done
    for (int e = t; e < PIX * cq; e += 256) {
      const int pix = e / cq, c4 = e - pix * cq;
      const int py = pix / g.PW, px = pix - py * g.PW;
      const int iy = iy0 + py, ix = ix0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const float4*>(xb + (size_t)((iy >> p.up_shift) * Ws + (ix >> p.up_shift)) * p.ldx + c0 + c4 * 4);
      float* d = xs_ + (size_t)(c4 * 4) * PIXP + pix;
      d[0] = v.x; d[PIXP] = v.y; d[2 * PIXP] = v.z; d[3 * PIXP] = v.w;
    }
    // this pass's weights: every lane reads the same (tap, channel) pair at a time -- LDS broadcast reads, not 2 global loads per
    // multiply-add (the compiler does not turn the uniform global loads into scalar loads)
    for (int e = t; e < p.ntaps * cw; e += 256) {
      const int tap = e / cw, ci = e - tap * cw;
      const float* wr = p.wp + ((size_t)p.taps[tap].widx * p.Kc + c0 + ci) * p.ldw;
      ws_[tap * THIN_CB + ci] = make_float2(wr[0], p.Cout > 1 ? wr[1] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      if (cls >= ncls) break;
      const int t0 = p.cls_tap[cls], t1 = p.cls_tap[cls + 1];
      float a0 = acc[cls][0], a1 = acc[cls][1];
      for (int tap = t0; tap < t1; ++tap) {
        const float* xr = xs_ + base + (p.taps[tap].dy - g.min_dy) * g.PW + (p.taps[tap].dx - g.min_dx);
        const float2* wr = ws_ + tap * THIN_CB;
#pragma unroll 8
        for (int ci = 0; ci < cw; ++ci) {
          const float a = xr[(size_t)ci * PIXP];
          const float2 w = wr[ci];
          a0 = fmaf(a, w.x, a0);
          a1 = fmaf(a, w.y, a1);
        }
      }
      acc[cls][0] = a0; acc[cls][1] = a1;
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= p.OHq || ox >= p.OWq) return;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    if (cls >= ncls) break;
    const int ooy = p.ncls > 1 ? (cls >> 1) : p.ooy, oox = p.ncls > 1 ? (cls & 1) : p.oox;
    const int off = (n * p.OH + oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
    thin_epilogue(p, off, 0, acc[cls][0]);
    if (p.Cout > 1) thin_epilogue(p, off, 1, acc[cls][1]);
  }
}

// The same for a DENSE KH x KW window at unit stride and spacing (the recover net's 5x5 flow head, the generator's conv17).  What the
// generic kernel above lost on the 5x5 head (PMC, profiles/r05_pmc_flow1_before.txt: 88 us; 4826 VALU instructions per wave for 2800 FMAs --
// the per-read address additions of a run-time tile stride --, a 32-channel pass = 62 KB of LDS = two workgroups per CU, and with two waves
// per SIMD the waves sit in s_waitcnt for 46 % of their cycles while VALU and LDS are 43 % / 38 % busy):
//  * every LDS offset of the multiply-add loop is a compile-time constant (immediate offsets, no address arithmetic; both outputs in one
//    v_pk_fma_f32);
//  * a thread owns TWO vertically adjacent pixels: the (KH + 1) x KW input column feeds both (0.6 input reads per product instead of 1)
//    and a weight read serves two products.  First version, one pixel per thread: 70 us with the LDS 51 % busy -- the broadcast weight reads
//    (a ds_read of 64 lanes costs its full 4-8 LDS cycles whether or not the lanes share an address) were two thirds of the LDS time;
//  * a pass stages CB = 8 channels of a 16 x 32 pixel tile: 25 KB of LDS, six workgroups per CU.
template <int KH, int KW, int CB>
__global__ __launch_bounds__(256) void conv_thin_nw_kernel(const ConvParams p, const ThinWin win) {
  constexpr int TH = 16, TW = 32, PH = TH + KH - 1, PW = TW + KW - 1, PIX = PH * PW, PIXP = PIX | 1, T = KH * KW;
  __shared__ __attribute__((aligned(16))) float xs_[CB * PIXP + 1];  // [CB][PIXP] input tile
  __shared__ __attribute__((aligned(16))) float2 ws_[T * CB];         // [tap][CB] weights of the two outputs
  const int t = threadIdx.x, ty = t / TW, tx = t - ty * TW;
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = by * TH, ox0 = bx * TW;
  const int iy0 = oy0 - (KH - 1) / 2, ix0 = ox0 - (KW - 1) / 2;
  const float* xb = p.x + (size_t)n * p.H * p.W * p.ldx + p.x_coff;
  const float* xt = xs_ + (2 * ty) * PW + tx;  // window origin of this thread's upper pixel
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;  // (a: pixel row 2 ty, b: row 2 ty + 1)
  // A pass's tile and weights travel global -> registers -> LDS, and the NEXT pass's loads are in flight under this pass's multiply-adds:
  // the head's 432 workgroups are fewer than two per CU, so nothing else hides a pass's memory latency (PMC of the version that loaded
  // and multiplied in turn: 55 us, the waves waiting for 58 % of their cycles, LDS 34 % / VALU 23 % busy).
  constexpr int NX = (PIX * (CB / 4) + 255) / 256, NW = (T * CB + 255) / 256;
  float4 rx[NX];
  float2 rw[NW];
  auto fetch = [&](int c0) {
    const int cw = p.Kc - c0 < CB ? p.Kc - c0 : CB, cq = cw >> 2;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = t + j * 256;
      const int pix = e / (CB / 4), c4 = e - pix * (CB / 4);
      const int py = pix / PW, px = pix - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < PIX * (CB / 4) && c4 < cq && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const float4*>(xb + (size_t)(iy * p.W + ix) * p.ldx + c0 + c4 * 4);
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int e = t + j * 256;
      const int tap = e / CB, ci = e - tap * CB;
      float2 w = make_float2(0.f, 0.f);  // (channels beyond the last block: zero weights against the zero-filled tile rows)
      if (e < T * CB && ci < cw) {
        const float* wr = p.wp + ((size_t)win.widx_at[tap] * p.Kc + c0 + ci) * p.ldw;
        w = make_float2(wr[0], p.Cout > 1 ? wr[1] : 0.f);
      }
      rw[j] = w;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = t + j * 256;
      if (e >= PIX * (CB / 4)) break;
      const int pix = e / (CB / 4), c4 = e - pix * (CB / 4);
      float* d = xs_ + (c4 * 4) * PIXP + pix;
      d[0] = rx[j].x; d[PIXP] = rx[j].y; d[2 * PIXP] = rx[j].z; d[3 * PIXP] = rx[j].w;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int e = t + j * 256;
      if (e < T * CB) ws_[e] = rw[j];
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < p.Kc; c0 += CB) {
    __syncthreads();  // the previous pass's reads are done
    stash();
    __syncthreads();
    if (c0 + CB < p.Kc) fetch(c0 + CB);
#pragma unroll 2
    for (int ci = 0; ci < CB; ++ci) {
      const float* xc = xt + ci * PIXP;
      const float2* wc = ws_ + ci;
      float col[KH + 1][KW];
#pragma unroll
      for (int r = 0; r < KH + 1; ++r)
#pragma unroll
        for (int c = 0; c < KW; ++c) col[r][c] = xc[r * PW + c];
#pragma unroll
      for (int r = 0; r < KH; ++r)
#pragma unroll
        for (int c = 0; c < KW; ++c) {
          const float2 w = wc[(r * KW + c) * CB];
          a0 = fmaf(col[r][c], w.x, a0);
          a1 = fmaf(col[r][c], w.y, a1);
          b0 = fmaf(col[r + 1][c], w.x, b0);
          b1 = fmaf(col[r + 1][c], w.y, b1);
        }
    }
  }
  const int ox = ox0 + tx;
  if (ox >= p.OWq) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int oy = oy0 + 2 * ty + h;
    if (oy >= p.OHq) break;
    const int off = (n * p.OH + oy * p.osy + p.ooy) * p.OW + ox * p.osx + p.oox;
    thin_epilogue(p, off, 0, h ? b0 : a0);
    if (p.Cout > 1) thin_epilogue(p, off, 1, h ? b1 : a1);
  }
}
// dense odd window at unit stride?  fills the position -> weight matrix table
static bool thin_n_window(const ConvParams& p, int* kh, ThinWin* win) {
  if (p.ncls > 1 || p.isy != 1 || p.isx != 1 || p.up_shift != 0 || (p.ntaps != 9 && p.ntaps != 25) || p.Kc % 4) return false;
  const int K = p.ntaps == 9 ? 3 : 5, h = (K - 1) / 2;
  for (int i = 0; i < 25; ++i) win->widx_at[i] = -1;
  for (int t = 0; t < p.ntaps; ++t) {
    const int r = p.taps[t].dy + h, c = p.taps[t].dx + h;
    if (r < 0 || r >= K || c < 0 || c >= K || win->widx_at[r * K + c] >= 0 || p.taps[t].widx < 0) return false;
    win->widx_at[r * K + c] = p.taps[t].widx;
  }
  *kh = K;
  return true;
}

static size_t thin_n_lds(const ConvParams& p, const ThinGeom& g) {
  const size_t pixp = (size_t)(g.PH * g.PW) | 1;
  return (((size_t)THIN_CB * pixp + 1) & ~(size_t)1) * sizeof(float) + (size_t)p.ntaps * THIN_CB * sizeof(float2);
}
// tile height: 4 (256 threads = 4 x 64) once 8 x 32 tiles would leave CUs idle, provided that wider halo tile still fits the LDS
static bool thin_n_pick(const ConvParams& p, int* th, ThinGeom* g) {
  const long t8 = (long)p.N * ((p.OHq + 7) / 8) * ((p.OWq + 31) / 32);
  for (int cand : {(t8 < 512 && p.OWq > 32) ? 4 : 8, 8}) {
    thin_geom(p, cand, 256 / cand, g);
    if (thin_n_lds(p, *g) <= 64 * 1024) { *th = cand; return true; }
  }
  return false;
}
bool conv_thin_n_ok(const ConvParams& p) {
  if (p.nseg) return false;  // segmented launches: implicit-GEMM families only
  if (p.Cout > 2 || p.Cout < 1 || p.xa != nullptr || p.Kc % 4 || p.ldw < 2 || p.ntaps < 1 || p.isy != p.isx || p.isy < 1 || p.isy > 2) return false;
  ThinGeom g;
  int th;
  return thin_n_pick(p, &th, &g);
}
int launch_conv_thin_n(const ConvParams& p, hipStream_t stream) {
  ThinGeom g;
  int th = 0;
  if (!conv_thin_n_ok(p) || !thin_n_pick(p, &th, &g)) { set_error("conv_thin_n: launch not eligible"); return UDET_ERR_UNSUPPORTED; }
  {  // dense 3x3 / 5x5 window at unit stride: the constant-stride kernel
    int K = 0;
    ThinWin win;
    if (thin_n_window(p, &K, &win)) {
      const int tiles = p.N * ((p.OHq + 15) / 16) * ((p.OWq + 31) / 32);
      // (16 channels per pass -- half the barriers, 49 KB of LDS -- measured: 51.8 us against 47.3 on the 5x5 head)
      if (K == 3) UDET_LAUNCH((conv_thin_nw_kernel<3, 3, 8>), dim3(tiles), dim3(256), 0, stream, p, win);
      else UDET_LAUNCH((conv_thin_nw_kernel<5, 5, 8>), dim3(tiles), dim3(256), 0, stream, p, win);
      UDET_HIP(hipGetLastError());
      return UDET_OK;
    }
  }
  const int tw = 256 / th;
  const size_t lds = thin_n_lds(p, g);
  const int tiles = p.N * ((p.OHq + th - 1) / th) * ((p.OWq + tw - 1) / tw);
  if (th == 8) UDET_LAUNCH(conv_thin_n_kernel<8>, dim3(tiles), dim3(256), lds, stream, p, g);
  else UDET_LAUNCH(conv_thin_n_kernel<4>, dim3(tiles), dim3(256), lds, stream, p, g);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ------------------------------------------------------------------------------------------------------------ thin K ----
// T: tap capacity (the launch has ntaps <= T), two real input channels.  Tile: 8 x 16 output pixels per workgroup and one block of
// 64 output channels; wave w walks tile rows 2w, 2w+1.
template <int T>
__global__ __launch_bounds__(256) void conv_thin_k_kernel(const ConvParams p, const ThinGeom g) {
  constexpr int TH = 8, TW = 16;
  extern __shared__ __attribute__((aligned(16))) float xs_[];  // [PIX] float2
  float2* xs = reinterpret_cast<float2*>(xs_);
  const int PIX = g.PH * g.PW;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = by * TH, ox0 = bx * TW;
  const int iy0 = oy0 + g.min_dy, ix0 = ox0 + g.min_dx;
  const float* xb = p.x + (size_t)n * p.H * p.W * p.ldx + p.x_coff;
  const int co = blockIdx.y * 64 + lane;
  // this lane's weights: w[t][k] = Wp[widx_t][k][co]
  float w[T][2];
  int toff[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const bool on = i < p.ntaps && co < p.ldw;
    const float* src = p.wp + ((size_t)(i < p.ntaps ? p.taps[i].widx : 0) * p.Kc) * p.ldw + (co < p.ldw ? co : 0);
    w[i][0] = on ? src[0] : 0.f;
    w[i][1] = on ? src[p.ldw] : 0.f;
    toff[i] = i < p.ntaps ? (p.taps[i].dy - g.min_dy) * g.PW + (p.taps[i].dx - g.min_dx) : 0;
  }
  for (int pix = t; pix < PIX; pix += 256) {
    const int py = pix / g.PW, px = pix - py * g.PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float2 v = make_float2(0.f, 0.f);
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = *reinterpret_cast<const float2*>(xb + (size_t)(iy * p.W + ix) * p.ldx);
    xs[pix] = v;
  }
  __syncthreads();
  if (co >= p.Cout) return;
  for (int r = 0; r < 2; ++r) {
    const int ty = wave * 2 + r, oy = oy0 + ty;
    if (oy >= p.OHq) break;
    // four neighbouring pixels at a time: eight independent accumulation chains hide the multiply-add and LDS latencies
    float out[TW];
#pragma unroll
    for (int tx = 0; tx < TW; tx += 4) {
      const float2* xr = xs + ty * g.PW + tx;
      float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
      if (ox0 + tx < p.OWq) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const float2* q = xr + toff[i];  // same address in every lane: broadcast reads
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float2 v = q[u];
            a0[u] = fmaf(v.x, w[i][0], a0[u]);
            a1[u] = fmaf(v.y, w[i][1], a1[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) out[tx + u] = a0[u] + a1[u];
    }
    thin_epilogue_row<TW>(p, (n * p.OH + oy * p.osy + p.ooy) * p.OW + ox0 * p.osx + p.oox, p.OWq - ox0, co, out);
  }
}

// The same for a DENSE KH x KW tap window at unit spacing (every stride-1, undilated head): the 2-channel input window of four
// neighbouring pixels -- KH x (KW + 3) float2 -- is read into registers in one burst and the multiply-adds index it statically.  The
// generic kernel above waits for the LDS after every tap (168 s_waitcnt for 200 reads in the 25-tap instantiation: 75 us for a launch
// whose multiply-adds take 18).
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_thin_kw_kernel(const ConvParams p, const ThinGeom g, const ThinWin win) {
  constexpr int TH = 8, TW = 16, T = KH * KW;
  extern __shared__ __attribute__((aligned(16))) float xs_[];  // [PIX] float2
  float2* xs = reinterpret_cast<float2*>(xs_);
  const int PIX = g.PH * g.PW;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int bx = bid % tiles_x, by = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = by * TH, ox0 = bx * TW;
  const int iy0 = oy0 + g.min_dy, ix0 = ox0 + g.min_dx;
  const float* xb = p.x + (size_t)n * p.H * p.W * p.ldx + p.x_coff;
  const int co = blockIdx.y * 64 + lane;
  float w[T][2];  // this lane's weights by window position
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const float* src = p.wp + ((size_t)win.widx_at[i] * p.Kc) * p.ldw + (co < p.ldw ? co : 0);
    w[i][0] = co < p.ldw ? src[0] : 0.f;
    w[i][1] = co < p.ldw ? src[p.ldw] : 0.f;
  }
  for (int pix = t; pix < PIX; pix += 256) {
    const int py = pix / g.PW, px = pix - py * g.PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float2 v = make_float2(0.f, 0.f);
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = *reinterpret_cast<const float2*>(xb + (size_t)(iy * p.W + ix) * p.ldx);
    xs[pix] = v;
  }
  __syncthreads();
  if (co >= p.Cout) return;
  for (int r = 0; r < 2; ++r) {
    const int ty = wave * 2 + r, oy = oy0 + ty;
    if (oy >= p.OHq) break;
    // the row's epilogue operands (accumulate / residual / saved activation of the dU emission) are requested BEFORE its multiply-adds:
    // behind them they were a second exposed memory latency per row (the 5x5 head's backward-data pass 69.8 / 57.6 -> 67.5 / 52.7 us).
    // (Measured and dropped, round 5: ACC / EMIT as template flags with wave-uniform row pointers -- hipcc then hoists the window reads of
    // all four pixel groups and runs out of registers: 282-320 VGPRs with weights parked in AGPRs, or 600-700 bytes of scratch under a
    // three-workgroup launch bound.  PMC of this form, profiles/r05_pmc_thin_k.txt: 2683 VALU + 1640 SALU instructions per wave for 800
    // packed FMAs, the waves waiting 39 % of their cycles at two per SIMD.)
    const int off0 = (n * p.OH + oy * p.osy + p.ooy) * p.OW + ox0 * p.osx + p.oox;
    float accv[TW], resv[TW], uav[TW];
    thin_row_fetch<TW>(p, off0, p.OWq - ox0, co, accv, resv, uav);
    float out[TW];
#pragma unroll
    for (int tx = 0; tx < TW; tx += 4) {
      float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
      if (ox0 + tx < p.OWq) {
        const float2* xr = xs + ty * g.PW + tx;  // same address in every lane: broadcast reads
#pragma unroll
        for (int wr = 0; wr < KH; ++wr) {
          float2 v[KW + 3];
#pragma unroll
          for (int wc = 0; wc < KW + 3; ++wc) v[wc] = xr[wr * g.PW + wc];
#pragma unroll
          for (int wc = 0; wc < KW; ++wc)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              a0[u] = fmaf(v[wc + u].x, w[wr * KW + wc][0], a0[u]);
              a1[u] = fmaf(v[wc + u].y, w[wr * KW + wc][1], a1[u]);
            }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) out[tx + u] = a0[u] + a1[u];
    }
    thin_row_apply<TW>(p, off0, p.OWq - ox0, co, out, accv, resv, uav);
  }
}
// dense window?  fills the position -> tap table
static bool thin_window(const ConvParams& p, const ThinGeom& g, int* kh, int* kw, ThinWin* win) {
  const int KH = g.PH - 7, KW = g.PW - 15;  // (thin_geom of an 8 x 16 tile at unit stride)
  if (!((KH == 3 && KW == 3) || (KH == 5 && KW == 5)) || p.ntaps != KH * KW) return false;
  for (int i = 0; i < 25; ++i) win->widx_at[i] = -1;
  for (int t = 0; t < p.ntaps; ++t) {
    const int pos = (p.taps[t].dy - g.min_dy) * KW + (p.taps[t].dx - g.min_dx);
    if (win->widx_at[pos] >= 0 || p.taps[t].widx < 0) return false;
    win->widx_at[pos] = p.taps[t].widx;
  }
  *kh = KH; *kw = KW;
  return true;
}

bool conv_thin_k_ok(const ConvParams& p) {
  if (p.nseg) return false;  // segmented launches: implicit-GEMM families only
  const int kr = p.kreal > 0 ? p.kreal : p.Kc;
  if (kr != 2 || p.Kc < 2 || p.ldx % 2 || p.x_coff % 2 || (reinterpret_cast<uintptr_t>(p.x) & 7)) return false;
  if (p.ncls == 4 || p.isy != 1 || p.isx != 1 || p.up_shift || p.xa != nullptr || p.ntaps < 1 || p.ntaps > 25 || p.Cout < 32) return false;
  ThinGeom g;
  thin_geom(p, 8, 16, &g);
  return (size_t)g.PH * g.PW * sizeof(float2) <= 32 * 1024;
}
int launch_conv_thin_k(const ConvParams& p, hipStream_t stream) {
  if (!conv_thin_k_ok(p)) { set_error("conv_thin_k: launch not eligible"); return UDET_ERR_UNSUPPORTED; }
  ThinGeom g;
  thin_geom(p, 8, 16, &g);
  const size_t lds = (size_t)g.PH * g.PW * sizeof(float2);
  const dim3 grid(p.N * ((p.OHq + 7) / 8) * ((p.OWq + 15) / 16), (p.Cout + 63) / 64);
  int kh = 0, kw = 0;
  ThinWin win;
  if (thin_window(p, g, &kh, &kw, &win)) {
    if (kh == 3) UDET_LAUNCH((conv_thin_kw_kernel<3, 3>), grid, dim3(256), lds, stream, p, g, win);
    else UDET_LAUNCH((conv_thin_kw_kernel<5, 5>), grid, dim3(256), lds, stream, p, g, win);
    UDET_HIP(hipGetLastError());
    return UDET_OK;
  }
  if (p.ntaps <= 9) UDET_LAUNCH(conv_thin_k_kernel<9>, grid, dim3(256), lds, stream, p, g);
  else UDET_LAUNCH(conv_thin_k_kernel<25>, grid, dim3(256), lds, stream, p, g);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
