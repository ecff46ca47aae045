// Winograd-domain FILTER GRADIENT of the 3x3 stride-1 convolutions on the fp32 matrix cores of gfx950 (round 5).
//
// Transposing F(2x2,3x3),  Y = A^T [ (G g G^T) .* (B^T d B) ] A,  gives
//     dg = G^T [ sum_tiles (B^T d B) .* (A dU A^T) ] G
// i.e. 16 position GEMMs  M[p][q][ci][co] = sum_tiles V[p][q][ci] E[p][q][co]  whose K axis is the 2x2 output tiles: 16 multiplications per
// tile and channel pair instead of the 36 of  dW[t][ci][co] = sum_q X(q@t)[ci] dU[q][co]  (conv_wgrad.hip).  Dilation d is d*d independent
// d = 1 problems on the output sub-lattices, all summed into the same nine taps.
//
// One workgroup = 8 waves = TWO per SIMD, one 64 (ci) x 64 (co) block pair and one K slice (a run of "strips" of eight tiles).  A wave tile is a
// 32 x 32 block; waves 0-3 ("role 0") own position rows {1, 2} of their block, waves 4-7 ("role 1") rows {0, 3}: 8 accumulators = 128
// registers per wave, so one wave multiplies while the other reads and transforms (one wave per SIMD with all 16 positions: 55 us for
// the position GEMMs of a 128 -> 128 layer, the parts adding up serially; this form: 40 us; conv_wgrad_dma_kernel: 57 us --
// tools/wino_wgrad_bench.hip, profiles/r05_wino_wgrad_proto.txt).  Lane = channel, lane half = tile parity of a tile pair (K = 2 per
// MFMA): a lane reads the raw 4x4 input patch (role 0: rows 1, 2 only) and the 2x2 dU tile of ITS channel from LDS with ds_read_b32 and
// forms B^T d B / A dU A^T in registers (row p of B^T d needs patch rows {0,2} {1,2} {2,1} {1,3}; row p of A dU is e0, e0 + e1, e0 - e1,
// -e1).  A strip's operands -- 4 rows x 18 pixels x 64 channels of X, 2 x 16 x 64 of dU -- travel global -> registers -> LDS (two buffers),
// the next strip in flight under this one's MFMAs; everything outside a sub-lattice's grid is zero (borders, ragged sizes).
// The output transform G^T M G runs in registers; the roles' row sums meet through LDS once.  The workgroup leaves its 9 x 64 x 64 block in
// the slab of its K slice in conv_wgrad.hip's own layout ([slice][tap * Cin + ci][co] + bias partials), so the slab reductions, the BN
// finalisation and the tuner's verification are shared.
//
// Replaces TF-1.13 Conv2DBackpropFilter / BiasAddGrad behind optimizer.compute_gradients (models/utils/loss_utils.py:18) for the
// generator's 64- and 128-channel 3x3 layers (models/nets.py:21-33).
#include <type_traits>

#include <mutex>

#include "common.h"
#include "conv_host.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WwgGeom {
  int d;          // dilation
  int tap_at[9];  // position (a, b) of the 3x3 grid -> row block of the slab (index into WgradParams::taps)
  int rows;       // tile rows per sub-lattice (of the largest sub-lattice)
  int per_row;    // strips (eight tiles) per tile row
  int strips;     // N * d * d * rows * per_row
  int slices, ldn;
};

constexpr int WWG_XP = 18, WWG_XS = 4 * WWG_XP * 64, WWG_US = 2 * 16 * 64;  // floats per stage: X halo rows [4][18][64], dU rows [2][16][64]
constexpr int WWG_LDS = 6 * 16 * 64 * 4 * 4;                                // the roles' exchange (96 KB) >= the two staging buffers (52 KB)

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_wino8_kernel(const WgradParams p, const WwgGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem8[];
  float (*xs)[WWG_XS] = reinterpret_cast<float (*)[WWG_XS]>(smem8);
  float (*us)[WWG_US] = reinterpret_cast<float (*)[WWG_US]>(smem8 + 2 * WWG_XS);
  float (*xch)[6][16][64] = reinterpret_cast<float (*)[6][16][64]>(smem8);  // (aliases the staging buffers: used behind the K loop)
  constexpr int XP = WWG_XP, XS = WWG_XS, US = WWG_US;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int role = wave >> 2, sub = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int cib = sub & 1, cob = sub >> 1;
  const int nci = p.Cin / 64, nblk = nci * (p.Cout / 64);
  // the channel-block pairs of one K slice are consecutive workgroups of ONE XCD (they read the same strips): logical id = slice * blocks + block
  int bid;
  {
    const int nwg = gridDim.x, h = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = h & 7, idx = h >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int blk = bid % nblk, slice = bid / nblk;
  const int ci0 = (blk % nci) * 64, co0 = (blk / nci) * 64;
  const int s_begin = (int)((long)g.strips * slice / g.slices), s_end = (int)((long)g.strips * (slice + 1) / g.slices);
  const int d = g.d;

  floatx16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum = 0.f;  // bias gradient: column sums of dU (role 0, ci block 0 of the first ci block pair)

  constexpr int NXV = (XS / 4 + 511) / 512, NUV = US / 4 / 512;
  float4 rx[NXV], ru[NUV];
  auto fetch = [&](int s) {
    // strip s -> (image n, sub-lattice (sy, sx), tile row ty, first tile tx0)
    int r = s / g.per_row;
    const int tx0 = (s - r * g.per_row) * 8;
    const int ty = r % g.rows;
    r /= g.rows;
    const int sx = r % d;
    r /= d;
    const int sy = r % d, n = r / d;
    const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;  // this sub-lattice's grid
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int e = t + j * 512;  // float4 index: [row 4][px 18][c4 16]
      const int c4 = e & 15, px = (e >> 4) % XP, rr = (e >> 4) / XP;
      const int yy = 2 * ty - 1 + rr, xx = 2 * tx0 - 1 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < XS / 4 && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws)
        v = *reinterpret_cast<const float4*>(p.x + ((size_t)(n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + ci0 + c4 * 4);
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NUV; ++j) {
      const int e = t + j * 512;  // [row 2][px 16][c4 16]
      const int c4 = e & 15, px = (e >> 4) & 15, rr = e >> 8;
      const int yy = 2 * ty + rr, xx = 2 * tx0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy < Hs && xx < Ws)
        v = *reinterpret_cast<const float4*>(p.dy + ((size_t)(n * p.OH + sy + d * yy) * p.OW + sx + d * xx) * p.ldy + p.y_coff + co0 + c4 * 4);
      ru[j] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int e = t + j * 512;
      if (e < XS / 4) *reinterpret_cast<float4*>(&xs[buf][e * 4]) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < NUV; ++j) *reinterpret_cast<float4*>(&us[buf][(t + j * 512) * 4]) = ru[j];
  };

  // BARRIER CONTRACT of the two role instantiations below: both are the SAME lambda, instantiated for R = 0 / 1, and every
  // __syncthreads() in it is reached unconditionally with a trip count that depends on workgroup-uniform values only (s_begin, s_end: the
  // K slice of blockIdx.x) -- never on R, never on a per-wave or per-lane value.  The waves of the two roles therefore execute the same
  // NUMBER of s_barrier instructions, though at different program counters; gfx950's s_barrier counts arrivals of the workgroup's waves
  // whatever their PC, which is what makes this legal on this target (the library is gfx950-only).  Any edit that makes a role skip or
  // add a loop trip (an early `continue`, a role-specific strip range) deadlocks the workgroup: keep role-specific code strictly between
  // the barriers.
  auto body = [&](auto ROLE) {
    constexpr int R = decltype(ROLE)::value;
    if (s_begin < s_end) {
      fetch(s_begin);
      stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const bool more = s + 1 < s_end;
      if (more) fetch(s + 1);
      const float* xb = &xs[buf][cib * 32 + li];
      const float* ub = &us[buf][cob * 32 + li];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // tile pair k: this lane half's tile j = 2 k + lh
        const int j = 2 * k + lh;
        float e[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) e[r][c] = ub[(r * 16 + 2 * j + c) * 64];
        float ta[4], tb[4];  // the two rows of B^T d this role owns,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        if (R == 0) {
          float d1[4], d2[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { d1[c] = xb[(1 * XP + 2 * j + c) * 64]; d2[c] = xb[(2 * XP + 2 * j + c) * 64]; }
#pragma unroll
          for (int c = 0; c < 4; ++c) { ta[c] = d1[c] + d2[c]; tb[c] = d2[c] - d1[c]; }  // rows 1, 2
          bsum += (e[0][0] + e[0][1]) + (e[1][0] + e[1][1]);
        } else {
          float d0[4], d1[4], d2[4], d3[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            d0[c] = xb[(0 * XP + 2 * j + c) * 64]; d1[c] = xb[(1 * XP + 2 * j + c) * 64];
            d2[c] = xb[(2 * XP + 2 * j + c) * 64]; d3[c] = xb[(3 * XP + 2 * j + c) * 64];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { ta[c] = d0[c] - d2[c]; tb[c] = d1[c] - d3[c]; }  // rows 0, 3
        }
        const float Va[4] = {ta[0] - ta[2], ta[1] + ta[2], ta[2] - ta[1], ta[1] - ta[3]};
        const float Vb[4] = {tb[0] - tb[2], tb[1] + tb[2], tb[2] - tb[1], tb[1] - tb[3]};
        float fa[2], fb[2];  // the two rows of A e,  A = [1 0; 1 1; 1 -1; 0 -1]
        if (R == 0) { fa[0] = e[0][0] + e[1][0]; fa[1] = e[0][1] + e[1][1]; fb[0] = e[0][0] - e[1][0]; fb[1] = e[0][1] - e[1][1]; }
        else { fa[0] = e[0][0]; fa[1] = e[0][1]; fb[0] = -e[1][0]; fb[1] = -e[1][1]; }
        const float Ea[4] = {fa[0], fa[0] + fa[1], fa[0] - fa[1], -fa[1]};
        const float Eb[4] = {fb[0], fb[0] + fb[1], fb[0] - fb[1], -fb[1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Va[c], Ea[c], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[4 + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vb[c], Eb[c], acc[4 + c], 0, 0, 0);
      }
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  };
  if (role == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());

  // column (q) transform of the own two position rows: z[b] = sum_q G[q][b] M[q],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
  float za[3][16], zb[3][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    za[0][r] = acc[0][r] + 0.5f * (acc[1][r] + acc[2][r]);
    za[1][r] = 0.5f * (acc[1][r] - acc[2][r]);
    za[2][r] = 0.5f * (acc[1][r] + acc[2][r]) + acc[3][r];
    zb[0][r] = acc[4][r] + 0.5f * (acc[5][r] + acc[6][r]);
    zb[1][r] = 0.5f * (acc[5][r] - acc[6][r]);
    zb[2][r] = 0.5f * (acc[5][r] + acc[6][r]) + acc[7][r];
  }
  if (role == 1) {  // position rows 0 (za) and 3 (zb) travel to the role-0 wave of the same tile
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { xch[sub][b][r][lane] = za[b][r]; xch[sub][3 + b][r][lane] = zb[b][r]; }
  }
  __syncthreads();
  if (role == 1) return;
  // slab of this K slice: [tap * Cin + ci][ldn] (conv_wgrad.hip); row transform: h0 = M0 + (M1 + M2) / 2, h1 = (M1 - M2) / 2, h2 = (M1 + M2) / 2 + M3
  float* dst = p.partial + (size_t)slice * p.Mpad * g.ldn;
  const int co = co0 + cob * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ci = ci0 + cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float m0 = xch[sub][b][r][lane], m3 = xch[sub][3 + b][r][lane];
      const float sm = 0.5f * (za[b][r] + zb[b][r]), df = 0.5f * (za[b][r] - zb[b][r]);
      dst[((size_t)g.tap_at[0 * 3 + b] * p.Cin4 + ci) * g.ldn + co] = m0 + sm;
      dst[((size_t)g.tap_at[1 * 3 + b] * p.Cin4 + ci) * g.ldn + co] = df;
      dst[((size_t)g.tap_at[2 * 3 + b] * p.Cin4 + ci) * g.ldn + co] = sm + m3;
    }
  }
  // bias partial of this slice: the two lane halves (tile parities) of a column, from the waves of the first ci block
  bsum += __shfl_xor(bsum, 32);
  if (ci0 == 0 && cib == 0 && lh == 0) p.pbias[(size_t)slice * g.ldn + co] = bsum;
}

// the launch's taps are the full 3x3 grid {-d, 0, d}^2 of a stride-1 convolution (no culled tap), channels in whole 64-blocks
static bool wwg_geometry(const WgradParams& p, WwgGeom* g) {
  if (p.ntaps != 9 || p.isy != 1 || p.isx != 1 || p.up_shift != 0 || p.ycls || p.swapped || p.OH != p.H || p.OW != p.W) return false;
  int d = 0;
  for (int t = 0; t < 9; ++t) {
    const int a = p.taps[t].dy < 0 ? -p.taps[t].dy : p.taps[t].dy;
    if (a > d) d = a;
  }
  if (d < 1) return false;
  for (int i = 0; i < 9; ++i) g->tap_at[i] = -1;
  for (int t = 0; t < 9; ++t) {
    const int dy = p.taps[t].dy, dx = p.taps[t].dx;
    if (dy % d != 0 || dx % d != 0) return false;
    const int a = dy / d + 1, b = dx / d + 1;
    if (a < 0 || a > 2 || b < 0 || b > 2 || g->tap_at[a * 3 + b] >= 0) return false;
    g->tap_at[a * 3 + b] = t;
  }
  g->d = d;
  const int Hs = (p.H + d - 1) / d, Ws = (p.W + d - 1) / d;  // the largest sub-lattice
  g->rows = (Hs + 1) / 2;
  g->per_row = ((Ws + 1) / 2 + 7) / 8;
  const long strips = (long)p.N * d * d * g->rows * g->per_row;
  if (strips > (1L << 30)) return false;
  g->strips = (int)strips;
  return true;
}
bool wgrad_wino_ok(const WgradParams& p) {
  WwgGeom g;
  if (p.f16 || p.ya != nullptr || p.Cin % 64 || p.Cout % 64 || p.Cin < 64 || p.Cout < 64) return false;
  if (p.ldx % 4 || p.x_coff % 4 || p.ldy % 4 || p.y_coff % 4) return false;
  if ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.dy)) & 15) return false;
  if (!wwg_geometry(p, &g)) return false;
  // padding of the tile grid: sub-lattices whose tiles mostly multiply zeros (dilation 16 on a 48 x 96 grid: 3 x 6 pixels in 2 x 8 tiles)
  const long useful = (long)p.N * p.H * p.W, padded = (long)g.strips * 8 * 4;
  return padded <= 3 * useful;
}
// K slices that give every CU one workgroup (two waves per SIMD)
int wgrad_wino_slices(const WgradParams& p, int wanted) {
  WwgGeom g;
  if (!wwg_geometry(p, &g)) return 1;
  const int blocks = (p.Cin / 64) * (p.Cout / 64);
  int s = wanted > 0 ? wanted : (256 + blocks - 1) / blocks;
  if (s > g.strips) s = g.strips;
  return s < 1 ? 1 : s;
}
// q.partial / q.pbias / q.Mpad / q.Cin4 as launch_wgrad_T sets them for the plain view; ldn: slab row stride
int launch_wgrad_wino(const WgradParams& q, int slices, int ldn, hipStream_t stream) {
  WwgGeom g;
  if (!wgrad_wino_ok(q) || !wwg_geometry(q, &g) || slices < 1) {
    set_error("wgrad_wino: launch not eligible");
    return UDET_ERR_UNSUPPORTED;
  }
  g.slices = slices > g.strips ? g.strips : slices;
  g.ldn = ldn;
  // 96 KB of dynamic LDS needs the attribute once per DEVICE (a code object is loaded per device) -- and exactly once under concurrent
  // first launches: one once_flag per device ordinal
  {
    constexpr int MAXDEV = 64;
    static std::once_flag once[MAXDEV];
    static hipError_t attr[MAXDEV];
    int dev = 0;
    UDET_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAXDEV) { set_error("wgrad_wino: device ordinal %d out of range", dev); return UDET_ERR_ARG; }
    std::call_once(once[dev], [&]() {
      attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_wino8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WWG_LDS);
    });
    UDET_HIP(attr[dev]);
  }
  const int blocks = (q.Cin / 64) * (q.Cout / 64);
  UDET_LAUNCH(conv_wgrad_wino8_kernel, dim3(blocks * g.slices), dim3(512), WWG_LDS, stream, q, g);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

}  // namespace udet
