// Tile-resident direct convolution for the thin layers (few input channels, large spatial extent: PWC pyramid levels 1-2,
// generator conv1/16/17, the 7x7 / 5x5 stride-2 first layers of the recover encoders).  These launches are HBM/latency-bound
// in the implicit-GEMM kernel: their K axis is 4..9 LDS stages long, so every workgroup pays one global->LDS round trip
// per stage for a handful of MFMAs.  Here a workgroup
//   1. loads the input halo tile of its TH x 32 output pixels ONCE (every input byte read 1.1-1.3x instead of once per tap)
//      into LDS as [channel quad][tile pixel] float4s,
//   2. loads the launch's packed weights for its 32 (16) output channels into LDS as [(tap, channel quad)][column] float4s,
//   3. walks (tap, channel quad) with MFMAs whose A fragment is read at tap-shifted tile addresses (im2col never
//      materialised): one ds_read_b128 per operand feeds four MFMAs; 4 waves x (TH/4) tile rows each,
//      (layers deeper than 32 channels repeat 1-3 per block of 32 channels, accumulating in registers),
//   4. runs the common epilogue (bias / activation / residual / second output / dU emission).
// Same ConvParams contract as conv_igemm (tap list, NN x2 read, TF SAME padding through the tap offsets).  The four
// output-parity classes of a stride-2 backward-data pass / transposed convolution are blockIdx.z: each class is a stride-1
// convolution over the dY grid with its own taps (and its own halo geometry) whose outputs land on one sub-lattice.
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

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
  int cb;         // channels resident per pass: 32, or 16 (half the LDS: one or two more workgroups per CU, twice the passes)
};

// NW = 32: v_mfma_f32_32x32x2_f32 (two lane halves share a K step); NW = 16: v_mfma_f32_16x16x4_f32 for <= 16 output channels
// (four lane groups share a K step, no padded MFMA columns; same FLOP rate).
//
// LDS layout: the K axis of one channel pass is cut into QUADS of 4 consecutive channels of one tap,
//   T4[c4][tile pixel]  (float4: channels 4*c4 .. 4*c4+3 of that pixel; odd row stride PIXP)
//   W4[quad][column]    (float4: the quad's 4 weights of that output channel)     quad = tap * (cw/4) + c4
// so a global float4 (4 channels of a pixel) is ONE ds_write_b128, and a lane fetches the A (and B) operands of FOUR MFMAs with
// one ds_read_b128.  Lane group `grp` of a wave walks quads grp, grp+NG, ...; any assignment of K indices to (MFMA, lane group)
// is valid as long as A and B agree, and a group that runs past the last quad multiplies by a zero quad.
typedef float floatx4 __attribute__((ext_vector_type(4)));
// register staging capacity per thread and channel pass: TILE_MT float4 halo loads, TILE_MW weight items (4 float4 each).
// The small variant keeps the VGPR count low enough for 3-4 workgroups per CU on the single-pass thin layers.
#define TILE_MT_BIG 16
#define TILE_MW_BIG 3
#define TILE_MT_SMALL 8
#define TILE_MW_SMALL 2
template <int NW> struct TileAcc;
template <> struct TileAcc<32> { floatx16 v; };
template <> struct TileAcc<16> { floatx4 v[2]; };

template <int TH, int NW, int TILE_MT, int TILE_MW, bool F16 = false>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvParams p, const TileGeoms gs) {
  constexpr int TW = 32;
  constexpr int TM = TH / 4;     // tile rows per wave
  constexpr int NG = 64 / NW;    // lane groups sharing one MFMA K step
  constexpr int NQ4 = NW / 4;    // float4s per weight row
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  const int cls = blockIdx.z;
  const TileGeom g = gs.g[cls];
  const int tap0 = p.cls_tap[cls], ntc = p.cls_tap[cls + 1] - tap0;
  const int ooy = p.ncls > 1 ? (cls >> 1) : p.ooy, oox = p.ncls > 1 ? (cls & 1) : p.oox;
  const int PIX = g.PH * g.PW;
  const int PIXP = PIX | 1;                       // odd: the b128 stores of one pixel's channel quads spread over all banks
  const int CB = p.Kc < gs.cb ? p.Kc : gs.cb;     // channels resident per pass (deep layers walk Kc in blocks of 32 or 16)
  const int CQB = CB >> 2;
  float4* T4 = smem4;                             // [CQB][PIXP]
  float4* W4 = smem4 + (size_t)CQB * PIXP;        // [ntc*CQB + NG][NW]   (rows past the last quad: zeros)
  int* tapoff = reinterpret_cast<int*>(W4 + ((size_t)ntc * CQB + NG) * NW);  // [ntc]
  int* pixoff = tapoff + ntc;                     // [PIX]
  int* qoff = pixoff + PIX;                       // [ntc*CQB + 3*NG]  quad -> float4 offset of its tile row (padding: 0)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = lane & (NW - 1), grp = lane / NW;
  const int tiles_x = (p.OWq + TW - 1) / TW, tiles_y = (p.OHq + TH - 1) / TH;
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * NW;
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

  TileAcc<NW> acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if constexpr (NW == 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i].v[r] = 0.f;
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i].v[h][r] = 0.f;
    }
  }
  // tile pixel (in float4 units) of this lane's A rows: 32-wide: pixel `col` of tile row i; 16-wide: pixels col, 16+col
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) abase[i] = ((wave * TM + i) * p.isy) * g.PW + col * p.isx;
  const int ahalf = 16 * p.isx;
  const float xscale = F16 ? p.f16_xscale : 1.f;

  // Global -> registers -> LDS in two steps: `fetch` issues every load of a channel pass back to back (one memory latency per
  // pass instead of one per loop trip) and `commit` writes them to LDS.  The next pass is fetched right before this pass's
  // MFMA loop, so its latency hides behind the arithmetic.
  float4 tv[TILE_MT];
  float4 wv[TILE_MW][4];
  auto fetch = [&](int c0, int cw) {
    const int cq = cw >> 2, NQ = ntc * cq;
    const int cq_shift = (cq & (cq - 1)) == 0 ? __builtin_ctz(cq) : -1;
#pragma unroll
    for (int u = 0; u < TILE_MT; ++u) {
      const int e = t + u * 256;
      tv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < PIX * cq) {
        const int pix = cq_shift >= 0 ? e >> cq_shift : e / cq, c4 = e - pix * cq;
        const int o = pixoff[pix];
        if (o >= 0) tv[u] = *reinterpret_cast<const float4*>(xb + (size_t)o + c0 + c4 * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < TILE_MW; ++u) {
      const int e = t + u * 256;
      wv[u][0] = wv[u][1] = wv[u][2] = wv[u][3] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NQ * NQ4) {
        const int n4 = e % NQ4, q = e / NQ4;
        const int tap = cq_shift >= 0 ? q >> cq_shift : q / cq, c4 = q - tap * cq;
        const int nn = n0 + n4 * 4;
        if (nn < p.ldw) {
          const float* src = p.wp + ((size_t)p.taps[tap0 + tap].widx * p.Kc + c0 + c4 * 4) * p.ldw + nn;
          wv[u][0] = *reinterpret_cast<const float4*>(src);
          wv[u][1] = *reinterpret_cast<const float4*>(src + p.ldw);
          wv[u][2] = *reinterpret_cast<const float4*>(src + 2 * (size_t)p.ldw);
          wv[u][3] = *reinterpret_cast<const float4*>(src + 3 * (size_t)p.ldw);
        }
      }
    }
  };
  auto commit = [&](int cw) {
    const int cq = cw >> 2, NQ = ntc * cq;
    const int cq_shift = (cq & (cq - 1)) == 0 ? __builtin_ctz(cq) : -1;
    // input halo tile: one b128 store per global float4
#pragma unroll
    for (int u = 0; u < TILE_MT; ++u) {
      const int e = t + u * 256;
      if (e < PIX * cq) {
        const int pix = cq_shift >= 0 ? e >> cq_shift : e / cq, c4 = e - pix * cq;
        T4[(size_t)c4 * PIXP + pix] = tv[u];
      }
    }
    // weights: rows 4*c4 .. 4*c4+3 of a tap, 4 columns each, transposed to [quad][column][4 channels]
#pragma unroll
    for (int u = 0; u < TILE_MW; ++u) {
      const int e = t + u * 256;
      if (e < NQ * NQ4) {
        const int n4 = e % NQ4, q = e / NQ4;
        float4* d = W4 + (size_t)q * NW + n4 * 4;
        d[0] = make_float4(wv[u][0].x, wv[u][1].x, wv[u][2].x, wv[u][3].x);
        d[1] = make_float4(wv[u][0].y, wv[u][1].y, wv[u][2].y, wv[u][3].y);
        d[2] = make_float4(wv[u][0].z, wv[u][1].z, wv[u][2].z, wv[u][3].z);
        d[3] = make_float4(wv[u][0].w, wv[u][1].w, wv[u][2].w, wv[u][3].w);
      }
    }
    for (int e = t; e < NG * NW; e += 256) W4[(size_t)NQ * NW + e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = t; q < NQ + 3 * NG; q += 256) {
      int o = 0;
      if (q < NQ) {
        const int tap = cq_shift >= 0 ? q >> cq_shift : q / cq, c4 = q - tap * cq;
        o = c4 * PIXP + tapoff[tap];
      }
      qoff[q] = o;
    }
  };

  // One call site each for fetch / commit / the MFMA loop (they are large once unrolled): iteration c0 = -CB only fetches
  // the first pass; afterwards pass c0 is committed, pass c0 + CB is fetched (in flight during the MFMA loop), pass c0 runs.
  for (int c0 = -CB; c0 < p.Kc; c0 += CB) {
    const int cw = c0 < 0 ? CB : (p.Kc - c0 < CB ? p.Kc - c0 : CB);
    const int cq = cw >> 2, NQ = ntc * cq;
    __syncthreads();  // offset tables written / previous pass's fragments read
    if (c0 >= 0) {
      commit(cw);
      __syncthreads();
    }
    if (c0 + CB < p.Kc) fetch(c0 + CB, p.Kc - (c0 + CB) < CB ? p.Kc - (c0 + CB) : CB);
    if (c0 < 0) continue;

    // ---- 3. quads: one b128 per operand feeds four MFMAs ---------------------------------------------------------
    // Branch-free and software-pipelined: quad -> tile offset comes from the qoff table (padding quads point at offset 0 and
    // multiply by zero weight rows); step j+1's fragments and step j+2's offset are in flight while step j's MFMAs issue.
    constexpr int NA = NW == 32 ? TM : 2 * TM;  // A fragments (float4) per step
    const int steps = (NQ + NG - 1) / NG;
    int q = grp;
    int off1 = qoff[q + NG];
    float4 b = W4[(size_t)q * NW + col];
    float4 a[NA];
    {
      const float4* ta = T4 + qoff[q];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (NW == 32) {
          a[i] = ta[abase[i]];
        } else {
          a[2 * i] = ta[abase[i]];
          a[2 * i + 1] = ta[abase[i] + ahalf];
        }
      }
    }
    for (int j = 0; j < steps; ++j) {
      float4 bn = b, an[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) an[i] = a[i];
      int off2 = 0;
      if (j + 1 < steps) {  // uniform
        q += NG;
        off2 = qoff[q + NG];
        bn = W4[(size_t)q * NW + col];
        const float4* ta = T4 + off1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (NW == 32) {
            an[i] = ta[abase[i]];
          } else {
            an[2 * i] = ta[abase[i]];
            an[2 * i + 1] = ta[abase[i] + ahalf];
          }
        }
      }
      if constexpr (F16) {  // a lane group's quad (4 channels of a tap) is one fp16 operand of the K = 8 / K = 16 MFMA
        const halfx4 bh = halfx4{(_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (NW == 32) {
            const halfx4 ah = halfx4{(_Float16)(a[i].x * xscale), (_Float16)(a[i].y * xscale), (_Float16)(a[i].z * xscale), (_Float16)(a[i].w * xscale)};
            acc[i].v = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc[i].v, 0, 0, 0);
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 av = a[2 * i + h];
              const halfx4 ah = halfx4{(_Float16)(av.x * xscale), (_Float16)(av.y * xscale), (_Float16)(av.z * xscale), (_Float16)(av.w * xscale)};
              acc[i].v[h] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[i].v[h], 0, 0, 0);
            }
          }
        }
      } else
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (NW == 32) {
          acc[i].v = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b.x, acc[i].v, 0, 0, 0);
          acc[i].v = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b.y, acc[i].v, 0, 0, 0);
          acc[i].v = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b.z, acc[i].v, 0, 0, 0);
          acc[i].v = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b.w, acc[i].v, 0, 0, 0);
        } else {
          // v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent one (same accumulator) only after 40: the two halves'
          // accumulators alternate, so consecutive MFMAs never depend on each other
          acc[i].v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i].x, b.x, acc[i].v[0], 0, 0, 0);
          acc[i].v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i + 1].x, b.x, acc[i].v[1], 0, 0, 0);
          acc[i].v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i].y, b.y, acc[i].v[0], 0, 0, 0);
          acc[i].v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i + 1].y, b.y, acc[i].v[1], 0, 0, 0);
          acc[i].v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i].z, b.z, acc[i].v[0], 0, 0, 0);
          acc[i].v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i + 1].z, b.z, acc[i].v[1], 0, 0, 0);
          acc[i].v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i].w, b.w, acc[i].v[0], 0, 0, 0);
          acc[i].v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * i + 1].w, b.w, acc[i].v[1], 0, 0, 0);
        }
      }
      // issue order pinned: this step's LDS reads first (they complete under the MFMAs), then the MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, NA + 2, 0);
      if constexpr (!F16) __builtin_amdgcn_sched_group_barrier(0x008, 4 * NA, 0);
      b = bn;
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = an[i];
      off1 = off2;
    }
  }

  if (F16 && xscale != 1.f) {
    const float inv = 1.f / xscale;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (NW == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i].v[r] *= inv;
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i].v[h][r] *= inv;
      }
    }
  }
  // ---- 4. epilogue --------------------------------------------------------------------------------------------------
  const int nn = n0 + col;
  if (nn >= p.Cout) return;
  // the common forward case (bias + activation, plain store) takes a short path: the generic epilogue's per-element
  // branches cost more than the MFMA loop of a thin layer
  const bool plain = !p.y2 && !p.res && !p.accumulate && !p.uo;
  const float bias = p.bias ? p.bias[nn] : 0.f;
  auto rows = [&](auto&& emit) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int oy = oy0 + wave * TM + i;
      if (oy >= p.OHq) continue;
      const int row = (n * p.OH + oy * p.osy + ooy) * p.OW + oox;  // output pixel index of ox = 0
      if constexpr (NW == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * grp;
          if (ox < p.OWq) emit(row, ox, acc[i].v[r]);
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ox = ox0 + h * 16 + grp * 4 + r;
            if (ox < p.OWq) emit(row, ox, acc[i].v[h][r]);
          }
      }
    }
  };
  const EpiAct ea = epi_act(p);
  if (plain) {
    float* ycol = p.y + p.y_coff + nn;
    const int xstep = p.osx * p.ldy;
    // (the activation is selected once, not per element: conv_epilogue.h, EpiAct)
    if (p.act == ACT_ELU) rows([&](int row, int ox, float v) { ycol[(size_t)row * p.ldy + (size_t)ox * xstep] = act_fwd_c<true>(v + bias, ea.slope); });
    else rows([&](int row, int ox, float v) { ycol[(size_t)row * p.ldy + (size_t)ox * xstep] = act_fwd_c<false>(v + bias, ea.slope); });
    return;
  }
  // Residual / accumulate / second output / dU emission (the stride-2 backward-data launches): ONE ROW of the thread's outputs at a time,
  // every operand of the row requested before the row's first store.  Element by element (round 5) each load waited behind the previous
  // element's stores -- loads and stores share vmcnt on gfx950, a wait for the one drains the other: 16 drains per row instead of one.
  const bool emit_u = p.uo && nn >= p.u_c0 && nn < p.u_c1;
  auto generic = [&](auto ELU) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int oy = oy0 + wave * TM + i;
      if (oy >= p.OHq) continue;
      const int row = (n * p.OH + oy * p.osy + ooy) * p.OW + oox;
      constexpr int NE = NW == 32 ? 16 : 8;
      float rs[NE], ac[NE], ua[NE];
      int off[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int ox = NW == 32 ? ox0 + (e & 3) + 8 * (e >> 2) + 4 * grp : ox0 + (e >> 2) * 16 + grp * 4 + (e & 3);
        off[e] = ox < p.OWq ? row + ox * p.osx : -1;
        rs[e] = ac[e] = ua[e] = 0.f;
        if (off[e] >= 0) {
          if (p.res) rs[e] = p.res[(size_t)off[e] * p.ldres + p.res_coff + nn];
          if (p.accumulate) ac[e] = p.y[(size_t)off[e] * p.ldy + p.y_coff + nn];
          if (emit_u) ua[e] = p.ua[(size_t)off[e] * p.ldua + p.ua_coff + nn];
        }
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        if (off[e] < 0) continue;
        float v;
        if constexpr (NW == 32) v = acc[i].v[e];
        else v = acc[i].v[e >> 2][e & 3];
        v = act_fwd_c<decltype(ELU)::value>(v + bias, ea.slope);
        if (p.y2) p.y2[(size_t)off[e] * p.ldy2 + p.y2_coff + nn] = v;
        if (p.res) v += rs[e];
        if (p.accumulate) v += ac[e];
        p.y[(size_t)off[e] * p.ldy + p.y_coff + nn] = v;
        if (emit_u) p.uo[(size_t)off[e] * p.ldu + p.u_coff + nn] = v * act_dfo_c(ua[e], ea);
      }
    }
  };
  if (p.act == ACT_ELU) generic(std::true_type());
  else generic(std::false_type());
}

// LDS bytes of the tile kernel for this launch at tile height th (0: not eligible); the maximum over the parity classes
size_t conv_tile_lds_bytes(const ConvParams& p, int th, int cbmax, TileGeoms* gout, bool* big) {
  if (big) *big = false;
  if (p.ntaps < 1 || p.xa != nullptr || p.Kc % 4 != 0 || p.isy != p.isx || p.isy < 1 || p.isy > 2) return 0;
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  if (ncls > 4) return 0;
  if (cbmax != 32 && cbmax != 16) return 0;
  if (gout) gout->cb = cbmax;
  const size_t cb = p.Kc < cbmax ? p.Kc : cbmax;  // channels resident per pass
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
    const size_t pix = (size_t)g.PH * g.PW, ntc = (size_t)(t1 - t0), cqb = cb / 4, nw = p.Cout <= 16 ? 16 : 32;
    const size_t ng = 64 / nw;
    const size_t bytes = (cqb * (pix | 1) + (ntc * cqb + ng) * nw) * 16 + (ntc + pix + ntc * cqb + 3 * ng + 8) * sizeof(float);
    if (pix * cqb > 256 * TILE_MT_BIG || ntc * cqb * (nw / 4) > 256 * TILE_MW_BIG) return 0;  // per-thread register staging capacity
    if (big && (pix * cqb > 256 * TILE_MT_SMALL || ntc * cqb * (nw / 4) > 256 * TILE_MW_SMALL)) *big = true;
    worst = bytes > worst ? bytes : worst;
  }
  return worst;
}

int launch_conv_tile(const ConvParams& p, int th, int cbmax, hipStream_t stream) {
  TileGeoms g;
  bool big = false;
  const size_t lds = conv_tile_lds_bytes(p, th, cbmax, &g, &big);
  if (lds == 0 || lds > 96 * 1024 || (th != 8 && th != 4)) {
    set_error("conv_tile: launch not eligible");
    return UDET_ERR_UNSUPPORTED;
  }
  typedef void (*Kern)(const ConvParams, const TileGeoms);
  static const Kern K[2][2][2] = {  // [th == 4][16-wide][big staging]
      {{conv_tile_kernel<8, 32, TILE_MT_SMALL, TILE_MW_SMALL>, conv_tile_kernel<8, 32, TILE_MT_BIG, TILE_MW_BIG>},
       {conv_tile_kernel<8, 16, TILE_MT_SMALL, TILE_MW_SMALL>, conv_tile_kernel<8, 16, TILE_MT_BIG, TILE_MW_BIG>}},
      {{conv_tile_kernel<4, 32, TILE_MT_SMALL, TILE_MW_SMALL>, conv_tile_kernel<4, 32, TILE_MT_BIG, TILE_MW_BIG>},
       {conv_tile_kernel<4, 16, TILE_MT_SMALL, TILE_MW_SMALL>, conv_tile_kernel<4, 16, TILE_MT_BIG, TILE_MW_BIG>}}};
  static bool attr_set = false;
  if (!attr_set) {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 2; ++c)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K[a][b][c]), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  const int tiles = ((p.OWq + 31) / 32) * ((p.OHq + th - 1) / th) * p.N;
  const int ncls = p.ncls > 1 ? p.ncls : 1;
  const bool n16 = p.Cout <= 16;
  dim3 grid(tiles, n16 ? 1 : (p.Cout + 31) / 32, ncls);
  if (p.f16) {
    static const Kern K16[2][2][2] = {
        {{conv_tile_kernel<8, 32, TILE_MT_SMALL, TILE_MW_SMALL, true>, conv_tile_kernel<8, 32, TILE_MT_BIG, TILE_MW_BIG, true>},
         {conv_tile_kernel<8, 16, TILE_MT_SMALL, TILE_MW_SMALL, true>, conv_tile_kernel<8, 16, TILE_MT_BIG, TILE_MW_BIG, true>}},
        {{conv_tile_kernel<4, 32, TILE_MT_SMALL, TILE_MW_SMALL, true>, conv_tile_kernel<4, 32, TILE_MT_BIG, TILE_MW_BIG, true>},
         {conv_tile_kernel<4, 16, TILE_MT_SMALL, TILE_MW_SMALL, true>, conv_tile_kernel<4, 16, TILE_MT_BIG, TILE_MW_BIG, true>}}};
    static bool attr16 = false;
    if (!attr16) {
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int c = 0; c < 2; ++c)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K16[a][b][c]), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr16 = true;
    }
    UDET_LAUNCH(K16[th == 4][n16][big], grid, dim3(256), lds, stream, p, g);
  } else {
    UDET_LAUNCH(K[th == 4][n16][big], grid, dim3(256), lds, stream, p, g);
  }
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
