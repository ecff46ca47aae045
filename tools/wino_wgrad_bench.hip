// Prototype: Winograd-domain FILTER GRADIENT of a 3x3 stride-1 convolution on the fp32 matrix cores of gfx950 -- round 5's answer to "a
// sub-9-products filter gradient" (models/nets.py:19-36 through loss_utils.py:18).  Transposing F(2x2,3x3):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        =>        dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G
// 16 position GEMMs  M[p][q][ci][co] = sum_tiles V[p][q][ci] E[p][q][co]  with K = tiles (2x2 output pixels): 16 multiplications per tile
// and channel pair instead of 36.  A wave owns all 16 positions of a 32 (ci) x 32 (co) block: 16 accumulators = 256 registers, one
// wave per SIMD; both transforms sit on the LDS -> VGPR path (lane = channel, lane half = tile parity).  K slices over strips of eight
// tiles; every slice leaves its 9 x 32 x 32 block (output transform applied in registers) in a slab, a second kernel sums the slabs --
// the structure of conv_wgrad.hip.  What this file measures is the GEMM part against conv_wgrad_dma_kernel's 56 us on the generator's
// 128 -> 128 layers (4 x 48 x 96 pixels), and the price of the slabs: 16 accumulator sets per channel-block pair mean 64 K slices to
// fill 1024 SIMDs at one wave each -- 37.7 MB of partial sums per layer whatever its shape, against 16.5 MB for the direct form.
//   hipcc --offload-arch=gfx950 -O3 -o wino_wgrad_bench tools/wino_wgrad_bench.hip && ./wino_wgrad_bench [N H W Cin Cout slices]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Prob {
  const float* x;   // [N][H][W][Cin]
  const float* du;  // [N][H][W][Cout]
  float* partial;   // [slices][9][Cin][Cout]
  int N, H, W, Cin, Cout, slices, strips;  // strips of 8 tiles; H, W even, W % 16 == 0
  int abl;  // ablation bits (timing only, results wrong): 1 no slab stores, 2 no global fetch after the first strip, 4 no MFMAs, 8 no LDS operand reads
};

// strip s -> (n, ty, tx0): TX = W / 2 tiles per tile row, 8 tiles per strip
__device__ __forceinline__ void strip_pos(const Prob& p, int s, int* n, int* ty, int* tx0) {
  const int per_row = p.W / 16, rows = p.H / 2;
  const int r = s / per_row;
  *tx0 = (s - r * per_row) * 8;
  *n = r / rows;
  *ty = r - *n * rows;
}

constexpr int XP = 18, XS = 4 * XP * 64, US = 2 * 16 * 64;  // floats per stage: X halo rows [4][18][64], dU rows [2][16][64]

// VER 1: hipcc's own schedule.  VER 2: software-pipelined by hand as conv_wino.hip is -- the raw reads of tile pair k + 1 first, then the
// transform additions of pair k, then its 16 MFMAs back to back (order pinned with sched_barrier).
template <int VER, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wwg_kernel(const Prob p) {
  __shared__ __attribute__((aligned(16))) float xs[2][XS];
  __shared__ __attribute__((aligned(16))) float us[2][US];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int cib = wave & 1, cob = wave >> 1;
  const int nci = p.Cin / 64, nco = p.Cout / 64;
  const int blk = blockIdx.x % (nci * nco), slice = blockIdx.x / (nci * nco);
  const int ci0 = (blk % nci) * 64, co0 = (blk / nci) * 64;
  const int s_begin = (int)((long)p.strips * slice / p.slices), s_end = (int)((long)p.strips * (slice + 1) / p.slices);

  floatx16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // staging: global -> registers -> LDS (prototype: no LDS-DMA), next strip in flight under this strip's MFMAs
  constexpr int NXV = (XS / 4 + 255) / 256, NUV = US / 4 / 256;
  float4 rx[NXV], ru[NUV];
  auto fetch = [&](int s) {
    int n, ty, tx0;
    strip_pos(p, s, &n, &ty, &tx0);
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int e = t + j * 256;  // float4 index: [row 4][px 18][c4 16]
      const int c4 = e & 15, px = (e >> 4) % XP, r = (e >> 4) / XP;
      const int y = 2 * ty - 1 + r, x = 2 * tx0 - 1 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < XS / 4 && y >= 0 && y < p.H && x >= 0 && x < p.W)
        v = *reinterpret_cast<const float4*>(p.x + ((size_t)(n * p.H + y) * p.W + x) * p.Cin + ci0 + c4 * 4);
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NUV; ++j) {
      const int e = t + j * 256;  // [row 2][px 16][c4 16]
      const int c4 = e & 15, px = (e >> 4) & 15, r = e >> 8;
      ru[j] = *reinterpret_cast<const float4*>(p.du + ((size_t)(n * p.H + 2 * ty + r) * p.W + 2 * tx0 + px) * p.Cout + co0 + c4 * 4);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int e = t + j * 256;
      if (e < XS / 4) *reinterpret_cast<float4*>(&xs[buf][e * 4]) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < NUV; ++j) *reinterpret_cast<float4*>(&us[buf][(t + j * 256) * 4]) = ru[j];
  };

  if (s_begin < s_end) {
    fetch(s_begin);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    if (more && !(ABL & 2)) fetch(s + 1);
    const float* xb = &xs[buf][cib * 32 + li];
    const float* ub = &us[buf][cob * 32 + li];
    auto rd = [&](int k, float (&d)[4][4], float (&e)[2][2]) {
      const int j = 2 * k + lh;  // this lane half's tile of pair k
      if (ABL & 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) d[r][c] = (float)(r + c + k);
        e[0][0] = e[0][1] = e[1][0] = e[1][1] = (float)k;
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = xb[(r * XP + 2 * j + c) * 64];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) e[r][c] = ub[(r * 16 + 2 * j + c) * 64];
    };
    auto mul = [&](const float (&d)[4][4], const float (&e)[2][2]) {
      // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
      float tt[4][4], V[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tt[0][c] = d[0][c] - d[2][c];
        tt[1][c] = d[1][c] + d[2][c];
        tt[2][c] = d[2][c] - d[1][c];
        tt[3][c] = d[1][c] - d[3][c];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        V[r][0] = tt[r][0] - tt[r][2];
        V[r][1] = tt[r][1] + tt[r][2];
        V[r][2] = tt[r][2] - tt[r][1];
        V[r][3] = tt[r][1] - tt[r][3];
      }
      // E = A e A^T,  A = [1 0; 1 1; 1 -1; 0 -1]
      float f[4][2], E[4][4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f[0][c] = e[0][c];
        f[1][c] = e[0][c] + e[1][c];
        f[2][c] = e[0][c] - e[1][c];
        f[3][c] = -e[1][c];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        E[r][0] = f[r][0];
        E[r][1] = f[r][0] + f[r][1];
        E[r][2] = f[r][0] - f[r][1];
        E[r][3] = -f[r][1];
      }
      if (VER == 2) __builtin_amdgcn_sched_barrier(0);
      if (ABL & 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r * 4 + c][0] += V[r][c] * E[r][c];
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r * 4 + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[r][c], E[r][c], acc[r * 4 + c], 0, 0, 0);
      if (VER == 2) __builtin_amdgcn_sched_barrier(0);
    };
    if (VER == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d[4][4], e[2][2];
        rd(k, d, e);
        mul(d, e);
      }
    } else {
      float d0[4][4], e0[2][2], d1[4][4], e1[2][2];
      rd(0, d0, e0);
      rd(1, d1, e1);
      __builtin_amdgcn_sched_barrier(0);
      mul(d0, e0);
      rd(2, d0, e0);
      __builtin_amdgcn_sched_barrier(0);
      mul(d1, e1);
      rd(3, d1, e1);
      __builtin_amdgcn_sched_barrier(0);
      mul(d0, e0);
      mul(d1, e1);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // output transform dg = G^T M G,  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], per accumulator element; slab [slice][tap][ci][co]
  float* dst = p.partial + (size_t)slice * 9 * p.Cin * p.Cout;
  if (ABL & 1) {  // one store per lane keeps the accumulators alive
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += acc[q][r];
    dst[t] = sacc;
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ci = ci0 + cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, co = co0 + cob * 32 + li;
    float h[3][4];  // h[a][q] = sum_p G[p][a] M[p][q]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float m0 = acc[0 * 4 + q][r], m1 = acc[1 * 4 + q][r], m2 = acc[2 * 4 + q][r], m3 = acc[3 * 4 + q][r];
      h[0][q] = m0 + 0.5f * (m1 + m2);
      h[1][q] = 0.5f * (m1 - m2);
      h[2][q] = 0.5f * (m1 + m2) + m3;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float g0 = h[a][0] + 0.5f * (h[a][1] + h[a][2]), g1 = 0.5f * (h[a][1] - h[a][2]), g2 = 0.5f * (h[a][1] + h[a][2]) + h[a][3];
      dst[((size_t)(a * 3 + 0) * p.Cin + ci) * p.Cout + co] = g0;
      dst[((size_t)(a * 3 + 1) * p.Cin + ci) * p.Cout + co] = g1;
      dst[((size_t)(a * 3 + 2) * p.Cin + ci) * p.Cout + co] = g2;
    }
  }
}

// VER 3: EIGHT waves, two per SIMD (conv_wino8_kernel's idea): waves 0-3 ("role 0") own position rows {1, 2} of their 32 x 32 block, waves
// 4-7 ("role 1") rows {0, 3} -- 8 accumulators = 128 registers per wave, so that one wave multiplies while the other reads and transforms.
// Row p of B^T d needs patch rows {0,2} {1,2} {2,1} {1,3}: role 0 reads rows 1, 2 only; row p of A dY: e0, e0 + e1, e0 - e1, -e1.
// The output transform's row sums meet through LDS once, in the epilogue: h0 = M0 + (M1 + M2) / 2, h1 = (M1 - M2) / 2, h2 = (M1 + M2) / 2 + M3.
template <int ABL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wwg8_kernel(const Prob p) {
  extern __shared__ __attribute__((aligned(16))) float smem8[];  // 96 KB: the two staging buffers, then (aliased) the roles' exchange
  float (*xs)[XS] = reinterpret_cast<float (*)[XS]>(smem8);
  float (*us)[US] = reinterpret_cast<float (*)[US]>(smem8 + 2 * XS);
  float (*xch)[6][16][64] = reinterpret_cast<float (*)[6][16][64]>(smem8);  // role 1 -> role 0: [wave tile][value][accumulator register][lane]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int role = wave >> 2, sub = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int cib = sub & 1, cob = sub >> 1;
  const int nci = p.Cin / 64, nco = p.Cout / 64;
  const int blk = blockIdx.x % (nci * nco), slice = blockIdx.x / (nci * nco);
  const int ci0 = (blk % nci) * 64, co0 = (blk / nci) * 64;
  const int s_begin = (int)((long)p.strips * slice / p.slices), s_end = (int)((long)p.strips * (slice + 1) / p.slices);

  floatx16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  constexpr int NXV = (XS / 4 + 511) / 512, NUV = US / 4 / 512;
  float4 rx[NXV], ru[NUV];
  auto fetch = [&](int s) {
    int n, ty, tx0;
    strip_pos(p, s, &n, &ty, &tx0);
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
      const int e = t + j * 512;
      const int c4 = e & 15, px = (e >> 4) % XP, r = (e >> 4) / XP;
      const int y = 2 * ty - 1 + r, x = 2 * tx0 - 1 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < XS / 4 && y >= 0 && y < p.H && x >= 0 && x < p.W)
        v = *reinterpret_cast<const float4*>(p.x + ((size_t)(n * p.H + y) * p.W + x) * p.Cin + ci0 + c4 * 4);
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NUV; ++j) {
      const int e = t + j * 512;
      const int c4 = e & 15, px = (e >> 4) & 15, r = e >> 8;
      ru[j] = *reinterpret_cast<const float4*>(p.du + ((size_t)(n * p.H + 2 * ty + r) * p.W + 2 * tx0 + px) * p.Cout + co0 + c4 * 4);
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
      if (more && !(ABL & 2)) fetch(s + 1);
      const float* xb = &xs[buf][cib * 32 + li];
      const float* ub = &us[buf][cob * 32 + li];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = 2 * k + lh;
        float e[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) e[r][c] = ub[(r * 16 + 2 * j + c) * 64];
        float ta[4], tb[4];  // the two rows of B^T d this role owns
        if (R == 0) {
          float d1[4], d2[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { d1[c] = xb[(1 * XP + 2 * j + c) * 64]; d2[c] = xb[(2 * XP + 2 * j + c) * 64]; }
#pragma unroll
          for (int c = 0; c < 4; ++c) { ta[c] = d1[c] + d2[c]; tb[c] = d2[c] - d1[c]; }  // rows 1, 2
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
        float Va[4] = {ta[0] - ta[2], ta[1] + ta[2], ta[2] - ta[1], ta[1] - ta[3]};
        float Vb[4] = {tb[0] - tb[2], tb[1] + tb[2], tb[2] - tb[1], tb[1] - tb[3]};
        float fa[2], fb[2];  // the two rows of A e
        if (R == 0) { fa[0] = e[0][0] + e[1][0]; fa[1] = e[0][1] + e[1][1]; fb[0] = e[0][0] - e[1][0]; fb[1] = e[0][1] - e[1][1]; }
        else { fa[0] = e[0][0]; fa[1] = e[0][1]; fb[0] = -e[1][0]; fb[1] = -e[1][1]; }
        float Ea[4] = {fa[0], fa[0] + fa[1], fa[0] - fa[1], -fa[1]};
        float Eb[4] = {fb[0], fb[0] + fb[1], fb[0] - fb[1], -fb[1]};
        if (!(ABL & 4)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Va[c], Ea[c], acc[c], 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[4 + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vb[c], Eb[c], acc[4 + c], 0, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) { acc[c][0] += Va[c] * Ea[c]; acc[4 + c][0] += Vb[c] * Eb[c]; }
        }
      }
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  };
  if (role == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());

  // column (q) transform of the own two rows: z[b] = sum_q G[q][b] M[q]
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
  if (role == 1) {  // rows 0 (za) and 3 (zb)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { xch[sub][b][r][lane] = za[b][r]; xch[sub][3 + b][r][lane] = zb[b][r]; }
  }
  __syncthreads();
  if (role == 1) return;
  float* dst = p.partial + (size_t)slice * 9 * p.Cin * p.Cout;
  if (ABL & 1) {
    float sacc = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += za[b][r] + zb[b][r] + xch[sub][b][r][lane];
    dst[t] = sacc;
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ci = ci0 + cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, co = co0 + cob * 32 + li;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float m0 = xch[sub][b][r][lane], m3 = xch[sub][3 + b][r][lane];  // rows 0, 3 (role 1); za / zb: rows 1, 2
      const float sm = 0.5f * (za[b][r] + zb[b][r]), df = 0.5f * (za[b][r] - zb[b][r]);
      dst[((size_t)(0 * 3 + b) * p.Cin + ci) * p.Cout + co] = m0 + sm;
      dst[((size_t)(1 * 3 + b) * p.Cin + ci) * p.Cout + co] = df;
      dst[((size_t)(2 * 3 + b) * p.Cin + ci) * p.Cout + co] = sm + m3;
    }
  }
}

__global__ __launch_bounds__(256) void wwg_reduce(const float* __restrict__ partial, float* __restrict__ dw, long n, int slices) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float s = 0.f;
    int k = 0;
    for (; k + 7 < slices; k += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = partial[(size_t)(k + u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; k < slices; ++k) s += partial[(size_t)k * n + e];
    dw[e] = s;
  }
}

int main(int argc, char** argv) {
  int N = 4, H = 48, W = 96, Cin = 128, Cout = 128, slices = 0, ver = 1;
  if (argc >= 6) { N = atoi(argv[1]); H = atoi(argv[2]); W = atoi(argv[3]); Cin = atoi(argv[4]); Cout = atoi(argv[5]); }
  if (argc >= 7) slices = atoi(argv[6]);
  if (argc >= 8) ver = atoi(argv[7]);
  const int abl = argc >= 9 ? atoi(argv[8]) : 0;
  if (H % 2 || W % 16 || Cin % 64 || Cout % 64) { fprintf(stderr, "H even, W %% 16 == 0, channels %% 64 == 0\n"); return 1; }
  const int strips = N * (H / 2) * (W / 16), blocks = (Cin / 64) * (Cout / 64);
  if (slices <= 0) slices = 256 / blocks;
  if (slices > strips) slices = strips;
  const size_t nx = (size_t)N * H * W * Cin, nu = (size_t)N * H * W * Cout, nw = (size_t)9 * Cin * Cout;
  std::vector<float> hx(nx), hu(nu);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hu) v = rnd();
  float *dx, *du, *dp, *dw;
  CK(hipMalloc(&dx, nx * 4));
  CK(hipMalloc(&du, nu * 4));
  CK(hipMalloc(&dp, nw * slices * 4));
  CK(hipMalloc(&dw, nw * 4));
  CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, hu.data(), nu * 4, hipMemcpyHostToDevice));
  Prob p{dx, du, dp, N, H, W, Cin, Cout, slices, strips, abl};
  const int grid = blocks * slices;
  auto gemm = [&]() {
#define WWG(V_, A_) hipLaunchKernelGGL((wwg_kernel<V_, A_>), dim3(grid), dim3(256), 0, 0, p)
#define WWG8(A_) do { static bool once_ = false; if (!once_) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wwg8_kernel<A_>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); once_ = true; } hipLaunchKernelGGL((wwg8_kernel<A_>), dim3(grid), dim3(512), 98304, 0, p); } while (0)
    if (ver == 3) {
      if (abl == 0) WWG8(0); else if (abl == 1) WWG8(1); else if (abl == 2) WWG8(2); else if (abl == 3) WWG8(3); else if (abl == 4) WWG8(4); else WWG8(7);
      return;
    }
    if (ver == 1) WWG(1, 0);
    else if (abl == 0) WWG(2, 0);
    else if (abl == 1) WWG(2, 1);
    else if (abl == 2) WWG(2, 2);
    else if (abl == 3) WWG(2, 3);
    else if (abl == 4) WWG(2, 4);
    else if (abl == 8) WWG(2, 8);
    else if (abl == 9) WWG(2, 9);
    else if (abl == 11) WWG(2, 11);
    else if (abl == 12) WWG(2, 12);
    else WWG(2, 15);
#undef WWG
  };
  auto run = [&]() {
    gemm();
    hipLaunchKernelGGL(wwg_reduce, dim3((int)((nw + 255) / 256)), dim3(256), 0, 0, dp, dw, (long)nw, slices);
  };
  run();
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  std::vector<float> hw(nw);
  CK(hipMemcpy(hw.data(), dw, nw * 4, hipMemcpyDeviceToHost));
  // reference on sampled channel pairs (double)
  const int cis[6] = {0, 5, 37, 63, Cin / 2, Cin - 1}, cos_[6] = {0, 9, 31, 32, Cout / 2 + 1, Cout - 1};
  double maxerr = 0, maxref = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int ic = 0; ic < 6; ++ic)
        for (int oc = 0; oc < 6; ++oc) {
          const int ci = cis[ic], co = cos_[oc];
          double s = 0;
          for (int n = 0; n < N; ++n)
            for (int y = 0; y < H; ++y) {
              const int yy = y + a - 1;
              if (yy < 0 || yy >= H) continue;
              for (int x = 0; x < W; ++x) {
                const int xx = x + b - 1;
                if (xx < 0 || xx >= W) continue;
                s += (double)hx[((size_t)(n * H + yy) * W + xx) * Cin + ci] * hu[((size_t)(n * H + y) * W + x) * Cout + co];
              }
            }
          const double g = hw[((size_t)(a * 3 + b) * Cin + ci) * Cout + co];
          if (fabs(g - s) > maxerr) maxerr = fabs(g - s);
          if (fabs(s) > maxref) maxref = fabs(s);
        }
  if (abl) printf("ABLATION %d (results are wrong on purpose)\n", abl);
  printf("v%d N=%d %dx%d %d->%d: %d strips, %d blocks x %d slices = %d workgroups; slabs %.1f MB\n", ver, N, H, W, Cin, Cout, strips, blocks, slices,
         grid, nw * slices * 4 / 1e6);
  printf("max |err| %.3e against max |ref| %.3e (324 sampled entries, float64 reference): %s\n", maxerr, maxref,
         maxerr < 2e-4 * (maxref > 1 ? maxref : 1) ? "OK" : "MISMATCH");
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  for (int i = 0; i < 10; ++i) run();
  const int reps = 30;
  float t_gemm = 0, t_all = 0;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) gemm();
  CK(hipEventRecord(e1, 0));
  for (int i = 0; i < reps; ++i) run();
  CK(hipEventRecord(e2, 0));
  CK(hipEventSynchronize(e2));
  CK(hipEventElapsedTime(&t_gemm, e0, e1));
  CK(hipEventElapsedTime(&t_all, e1, e2));
  const double gf = 2.0 * N * H * W * 9.0 * Cin * Cout * 1e-9;
  printf("position GEMMs alone %.1f us (%.1f TFLOP/s direct-equivalent, matrix pipe at %.2f of peak for 16/36 of the products); with the slab "
         "reduction %.1f us\n", t_gemm / reps * 1e3, gf / (t_gemm / reps), gf * 16 / 36 / (t_gemm / reps) / 157.3, t_all / reps * 1e3);
  return 0;
}
