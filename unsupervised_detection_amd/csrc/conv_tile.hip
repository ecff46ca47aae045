// Tile-resident direct convolution for the thin layers (few input channels, large spatial extent: PWC pyramid levels 1-2,
// generator conv1/16/17, the 7x7 / 5x5 stride-2 first layers of the recover encoders).  These launches are HBM/latency-bound
// in the implicit-GEMM kernel: their K axis is 4..9 LDS stages long, so every workgroup pays one global->LDS round trip
// per stage for a handful of MFMAs.  Here a workgroup
//   1. loads the input halo tile of its TH x 32 output pixels ONCE (every input byte read 1.1-1.3x instead of once per tap),
//      channel-major into LDS ([c][tile pixel]: the K-major image the MFMA A fragment wants, conflict-free for stride 1),
//   2. loads the launch's packed weights for its 32 output channels ([tap][c][32]) into LDS,
//   3. walks taps x channels with v_mfma_f32_32x32x2_f32 reading the A fragment at tap-shifted tile addresses
//      (im2col never materialised), 4 waves x (TH/4) tile rows each,
//      (layers deeper than 32 channels repeat 1-3 per block of 32 channels, accumulating in registers),
//   4. runs the common epilogue (bias / activation / residual / second output / dU emission).
// Same ConvParams contract as conv_igemm (tap list, NN x2 read, TF SAME padding through the tap offsets).  The four
// output-parity classes of a stride-2 backward-data pass / transposed convolution are blockIdx.z: each class is a stride-1
// convolution over the dY grid with its own taps (and its own halo geometry) whose outputs land on one sub-lattice.
#include "common.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void tile_epilogue(const ConvParams& p, int off, int n, float v) {
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

struct TileGeom {
  int min_dy, min_dx, PH, PW;  // tile origin offset and extent on the (logical, post-upsample) input grid
};
struct TileGeoms {
  TileGeom g[4];  // per output-parity class
};

template <int TH>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvParams p, const TileGeoms gs) {
  constexpr int TW = 32;
  constexpr int TM = TH / 4;  // tile rows (MFMA M blocks) per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cls = blockIdx.z;
  const TileGeom g = gs.g[cls];
  const int tap0 = p.cls_tap[cls], ntc = p.cls_tap[cls + 1] - tap0;
  const int ooy = p.ncls > 1 ? (cls >> 1) : p.ooy, oox = p.ncls > 1 ? (cls & 1) : p.oox;
  const int PIX = g.PH * g.PW;
  const int PIXP = PIX | 1;                       // odd row stride: the 4 transposing stores of a float4 spread over banks
  const int CB = p.Kc < 32 ? p.Kc : 32;           // channels resident per pass (deep layers walk Kc in blocks of 32)
  float* T = smem;                                // [CB][PIXP]
  float* Wl = smem + (size_t)CB * PIXP;           // [ntaps*CB][32]
  int* tapoff = reinterpret_cast<int*>(Wl + (size_t)ntc * CB * 32);  // [ntc]
  int* pixoff = tapoff + ntc;                 // [PIX]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * 32;
  const int iy0 = oy0 * p.isy + g.min_dy, ix0 = ox0 * p.isx + g.min_dx;  // logical input coords of tile pixel (0,0)
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const float* xb = p.x + (size_t)n * Hs * Ws * p.ldx + p.x_coff;

  // tap -> tile offset; tile pixel -> global element offset (-1: zero padding), once per workgroup
  for (int i = t; i < ntc; i += 256) tapoff[i] = (p.taps[tap0 + i].dy - g.min_dy) * g.PW + (p.taps[tap0 + i].dx - g.min_dx);
  for (int pix = t; pix < PIX; pix += 256) {
    const int py = pix / g.PW, px = pix - py * g.PW;
    const int iy = iy0 + py, ix = ix0 + px;
    int o = -1;
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) o = ((iy >> p.up_shift) * Ws + (ix >> p.up_shift)) * p.ldx;
    pixoff[pix] = o;
  }

  floatx16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) abase[i] = ((wave * TM + i) * p.isy) * g.PW + li * p.isx + lh * PIXP;  // + lh: channel 2kk+lh

  for (int c0 = 0; c0 < p.Kc; c0 += CB) {
    const int cw = p.Kc - c0 < CB ? p.Kc - c0 : CB;
    __syncthreads();  // offset tables written / previous pass's fragments read
    // ---- 1. input halo tile of channels [c0, c0+cw), transposed to channel-major ---------------------------------
    const int KQ = cw >> 2;
    const int kq_shift = (KQ & (KQ - 1)) == 0 ? __builtin_ctz(KQ) : -1;
    for (int e = t; e < PIX * KQ; e += 256) {
      const int pix = kq_shift >= 0 ? e >> kq_shift : e / KQ, c4 = e - pix * KQ;
      const int o = pixoff[pix];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o >= 0) v = *reinterpret_cast<const float4*>(xb + (size_t)o + c0 + c4 * 4);
      float* d = T + (size_t)(c4 * 4) * PIXP + pix;
      d[0] = v.x; d[PIXP] = v.y; d[2 * PIXP] = v.z; d[3 * PIXP] = v.w;
    }
    // ---- 2. weights [tap][c][32 columns of this N tile] ---------------------------------------------------------
    const int cw_shift = (cw & (cw - 1)) == 0 ? __builtin_ctz(cw) : -1;
    for (int e = t; e < ntc * cw * 8; e += 256) {
      const int c4 = e & 7, row = e >> 3;                 // row = tap*cw + c
      const int tap = cw_shift >= 0 ? row >> cw_shift : row / cw, c = row - tap * cw;
      const int nn = n0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nn < p.ldw) v = *reinterpret_cast<const float4*>(p.wp + ((size_t)p.taps[tap0 + tap].widx * p.Kc + c0 + c) * p.ldw + nn);
      *reinterpret_cast<float4*>(Wl + (size_t)row * 32 + c4 * 4) = v;
    }
    __syncthreads();

    // ---- 3. taps x channels -------------------------------------------------------------------------------------
    const int kpairs = cw >> 1;
    for (int tap = 0; tap < ntc; ++tap) {
      const int off = tapoff[tap];
      const float* wrow = Wl + (size_t)(tap * cw + lh) * 32 + li;
      for (int kk = 0; kk < kpairs; ++kk) {
        const float b = wrow[(size_t)kk * 64];
        const float* ta = T + (size_t)kk * 2 * PIXP + off;
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[abase[i]], b, acc[i], 0, 0, 0);
      }
    }
  }

  // ---- 4. epilogue --------------------------------------------------------------------------------------------------
  const int nn = n0 + li;
  if (nn >= p.Cout) return;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int oy = oy0 + wave * TM + i;
    if (oy >= p.OHq) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ox >= p.OWq) continue;
      const int off = (n * p.OH + oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
      tile_epilogue(p, off, nn, acc[i][r]);
    }
  }
}

// <= 16 output channels: the same kernel on v_mfma_f32_16x16x4_f32 (16-wide N: no padded MFMA columns; same FLOP rate).
// A fragment: lane -> pixel (l&15) of a 16-pixel half row, channel 4*kk + (l>>4); B: column l&15, same channel.
// The tile's channel stride is == 16 (mod 32) so the four channel groups of a wave read disjoint banks.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int TH>
__global__ __launch_bounds__(256) void conv_tile16_kernel(const ConvParams p, const TileGeoms gs) {
  constexpr int TW = 32;
  constexpr int TM = TH / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cls = blockIdx.z;
  const TileGeom g = gs.g[cls];
  const int tap0 = p.cls_tap[cls], ntc = p.cls_tap[cls + 1] - tap0;
  const int ooy = p.ncls > 1 ? (cls >> 1) : p.ooy, oox = p.ncls > 1 ? (cls & 1) : p.oox;
  const int PIX = g.PH * g.PW;
  const int PIXP = ((PIX + 15) & ~31) + 16;       // >= PIX, == 16 (mod 32)
  const int CB = p.Kc < 32 ? p.Kc : 32;
  float* T = smem;                                // [CB][PIXP]
  float* Wl = smem + (size_t)CB * PIXP;           // [ntaps*CB][16]
  int* tapoff = reinterpret_cast<int*>(Wl + (size_t)ntc * CB * 16);
  int* pixoff = tapoff + ntc;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lp = lane & 15, lg = lane >> 4;
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * 16;
  const int iy0 = oy0 * p.isy + g.min_dy, ix0 = ox0 * p.isx + g.min_dx;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const float* xb = p.x + (size_t)n * Hs * Ws * p.ldx + p.x_coff;

  for (int i = t; i < ntc; i += 256) tapoff[i] = (p.taps[tap0 + i].dy - g.min_dy) * g.PW + (p.taps[tap0 + i].dx - g.min_dx);
  for (int pix = t; pix < PIX; pix += 256) {
    const int py = pix / g.PW, px = pix - py * g.PW;
    const int iy = iy0 + py, ix = ix0 + px;
    int o = -1;
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) o = ((iy >> p.up_shift) * Ws + (ix >> p.up_shift)) * p.ldx;
    pixoff[pix] = o;
  }

  floatx4 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][h][r] = 0.f;
  int abase[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) abase[i][h] = ((wave * TM + i) * p.isy) * g.PW + (h * 16 + lp) * p.isx + lg * PIXP;

  for (int c0 = 0; c0 < p.Kc; c0 += CB) {
    const int cw = p.Kc - c0 < CB ? p.Kc - c0 : CB;
    __syncthreads();
    const int KQ = cw >> 2;
    const int kq_shift = (KQ & (KQ - 1)) == 0 ? __builtin_ctz(KQ) : -1;
    for (int e = t; e < PIX * KQ; e += 256) {
      const int pix = kq_shift >= 0 ? e >> kq_shift : e / KQ, c4 = e - pix * KQ;
      const int o = pixoff[pix];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o >= 0) v = *reinterpret_cast<const float4*>(xb + (size_t)o + c0 + c4 * 4);
      float* d = T + (size_t)(c4 * 4) * PIXP + pix;
      d[0] = v.x; d[PIXP] = v.y; d[2 * PIXP] = v.z; d[3 * PIXP] = v.w;
    }
    const int cw_shift = (cw & (cw - 1)) == 0 ? __builtin_ctz(cw) : -1;
    for (int e = t; e < ntc * cw * 4; e += 256) {
      const int c4 = e & 3, row = e >> 2;
      const int tap = cw_shift >= 0 ? row >> cw_shift : row / cw, c = row - tap * cw;
      const int nn = n0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nn < p.ldw) v = *reinterpret_cast<const float4*>(p.wp + ((size_t)p.taps[tap0 + tap].widx * p.Kc + c0 + c) * p.ldw + nn);
      *reinterpret_cast<float4*>(Wl + (size_t)row * 16 + c4 * 4) = v;
    }
    __syncthreads();

    const int kquads = cw >> 2;
    for (int tap = 0; tap < ntc; ++tap) {
      const int off = tapoff[tap];
      const float* wrow = Wl + (size_t)(tap * cw + lg) * 16 + lp;
      for (int kk = 0; kk < kquads; ++kk) {
        const float b = wrow[(size_t)kk * 64];
        const float* ta = T + (size_t)kk * 4 * PIXP + off;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[abase[i][h]], b, acc[i][h], 0, 0, 0);
      }
    }
  }

  const int nn = n0 + lp;
  if (nn >= p.Cout) return;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int oy = oy0 + wave * TM + i;
    if (oy >= p.OHq) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = ox0 + h * 16 + lg * 4 + r;
        if (ox >= p.OWq) continue;
        const int off = (n * p.OH + oy * p.osy + ooy) * p.OW + ox * p.osx + oox;
        tile_epilogue(p, off, nn, acc[i][h][r]);
      }
  }
}

// LDS bytes of the tile kernel for this launch at tile height th (0: not eligible); the maximum over the parity classes
size_t conv_tile_lds_bytes(const ConvParams& p, int th, TileGeoms* gout) {
  if (p.ntaps < 1 || p.xa != nullptr || p.Kc % 4 != 0 || p.isy != p.isx || p.isy < 1 || p.isy > 2) return 0;
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  if (ncls > 4) return 0;
  const size_t cb = p.Kc < 32 ? p.Kc : 32;  // channels resident per pass
  size_t worst = 0;
  for (int c = 0; c < ncls; ++c) {
    const int t0 = ncls > 1 ? p.cls_tap[c] : 0, t1 = ncls > 1 ? p.cls_tap[c + 1] : p.ntaps;
    int mn_y = 0, mx_y = 0, mn_x = 0, mx_x = 0;
    for (int t = t0; t < t1; ++t) {
      if (t == t0 || p.taps[t].dy < mn_y) mn_y = p.taps[t].dy;
      if (t == t0 || p.taps[t].dy > mx_y) mx_y = p.taps[t].dy;
      if (t == t0 || p.taps[t].dx < mn_x) mn_x = p.taps[t].dx;
      if (t == t0 || p.taps[t].dx > mx_x) mx_x = p.taps[t].dx;
    }
    TileGeom g;
    g.min_dy = mn_y; g.min_dx = mn_x;
    g.PH = (th - 1) * p.isy + (mx_y - mn_y) + 1;
    g.PW = 31 * p.isx + (mx_x - mn_x) + 1;
    if (gout) gout->g[c] = g;
    const size_t pix = (size_t)g.PH * g.PW, ntc = (size_t)(t1 - t0);
    const size_t pixp = p.Cout <= 16 ? (size_t)(((pix + 15) & ~(size_t)31) + 16) : (pix | 1);
    const size_t bytes = (cb * pixp + ntc * cb * (p.Cout <= 16 ? 16 : 32) + ntc + pix + 8) * sizeof(float);
    worst = bytes > worst ? bytes : worst;
  }
  return worst;
}

int launch_conv_tile(const ConvParams& p, int th, hipStream_t stream) {
  TileGeoms g;
  const size_t lds = conv_tile_lds_bytes(p, th, &g);
  if (lds == 0 || lds > 96 * 1024 || (th != 8 && th != 4)) {
    set_error("conv_tile: launch not eligible");
    return UDET_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile16_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  const int tiles = ((p.OWq + 31) / 32) * ((p.OHq + th - 1) / th) * p.N;
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  if (p.Cout <= 16) {
    dim3 grid16(tiles, 1, ncls);
    if (th == 8) hipLaunchKernelGGL(conv_tile16_kernel<8>, grid16, dim3(256), lds, stream, p, g);
    else hipLaunchKernelGGL(conv_tile16_kernel<4>, grid16, dim3(256), lds, stream, p, g);
    UDET_HIP(hipGetLastError());
    return UDET_OK;
  }
  dim3 grid(tiles, (p.Cout + 31) / 32, ncls);
  if (th == 8) hipLaunchKernelGGL(conv_tile_kernel<8>, grid, dim3(256), lds, stream, p, g);
  else hipLaunchKernelGGL(conv_tile_kernel<4>, grid, dim3(256), lds, stream, p, g);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
