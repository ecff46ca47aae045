// Backward-filter (wgrad) of the convolutions on the fp32 matrix cores, plus bias / BN
// parameter gradients:
//     dW[t][ci][co] = sum_q X(q@t)[ci] * dU[q][co],   dU = dY * act'(saved output),   db[co] = sum_q dU[q][co]
// Replaces TF-1.13 Conv2DBackpropFilter / BiasAddGrad / FusedBatchNormGrad(inference) reached
// through optimizer.compute_gradients (models/utils/loss_utils.py:18).
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "common.h"
#include <algorithm>
#include <vector>
#include "conv_host.h"
#include "plan.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

// One GEMM per layer:  M = (tap, input channel) flattened (m = tap*Cin4 + ci, Cin4 = Cin rounded up to 4),
// N = output channels, K = output pixels, split across workgroups (blockIdx.y).  A 128-row M tile of a 7x7 conv
// over 4 channels therefore holds 32 taps instead of one tap padded 16x, and every tap of a layer shares one pass
// over dU.  Stage = 32 pixels; both operands are pixel-major in memory == the K-major LDS image the MFMA
// fragments want, so staging is plain float4 copies (the A float4 of a thread always comes from the same tap).
// The bias gradient (column sums of dU) is accumulated from the B stages already in LDS by the m-tile-0 workgroups.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p, int co_tiles, int nsplit) {
  constexpr int BKP = 32;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WAVES_M * WAVES_N == 4 && TM * 32 == WTM && TN * 32 == WTN, "tile");
  constexpr int A_F4_ROW = BM / 4, B_F4_ROW = BN / 4;
  constexpr int A_PIX = 256 / A_F4_ROW, B_PIX = 256 / B_F4_ROW;  // pixels staged per pass
  constexpr int A_LD = BKP / A_PIX, B_LD = BKP / B_PIX;
  static_assert(A_LD >= 1 && B_LD >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float As[2][BKP][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BKP][BN];
  __shared__ int2 tap_yx[UDET_MAX_TAPS];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the (M tile, N tile) workgroups of ONE pixel slice read the same dU rows and the same (tap-shifted) X rows; as
  // consecutive hardware ids they were dealt round-robin to the eight XCDs and every L2 fetched the slice for itself (PMC, round 4:
  // 5.4 GB per step for 1.1 GB of operands).  Each XCD now walks whole slices: logical id = slice * tiles + tile.
  int bx, by;
  {
    const int nx = gridDim.x, nwg = nx * gridDim.y, h = blockIdx.x + nx * blockIdx.y;
    const int q = nwg >> 3, r = nwg & 7, xcd = h & 7, idx = h >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = gridDim.y == 1 ? 0 : l / nx;  // (one pixel slice: no run-time division)
    bx = l - by * nx;
  }
  const int mt = co_tiles == 1 ? bx : bx / co_tiles;
  const int m0 = mt * BM, co0 = (bx - mt * co_tiles) * BN;
  const int OHW = p.OH * p.OW;
  const int Q = p.N * OHW;
  const int nchunks = (Q + BKP - 1) / BKP;
  // (nchunks * nsplit < 2^31: 32-bit divisions -- a 64-bit one is ~150 instructions in front of every workgroup's first load)
  const int c_begin = (int)((unsigned)(nchunks * by) / (unsigned)nsplit), c_end = (int)((unsigned)(nchunks * (by + 1)) / (unsigned)nsplit);
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const int cout4 = (p.Cout + 3) & ~3;

  for (int i = t; i < p.ntaps; i += 256) tap_yx[i] = make_int2(p.taps[i].dy, p.taps[i].dx);
  __syncthreads();
  // this thread's A column group: 4 consecutive input channels of one tap
  const int a_m4 = t % A_F4_ROW;
  const int a_m = m0 + a_m4 * 4;
  const int a_tap = a_m / p.Cin4, a_ci = a_m - a_tap * p.Cin4;
  const bool a_on = a_tap < p.ntaps;
  int a_dy = 0, a_dx = 0;
  if (a_on) { a_dy = tap_yx[a_tap].x; a_dx = tap_yx[a_tap].y; }
  const float* a_src = p.x + p.x_coff + a_ci;
  const int b_c4 = t % B_F4_ROW;
  const int b_co = co0 + b_c4 * 4;
  const bool b_on = b_co < cout4;
  const int ycl = p.ycls ? (m0 / p.Cin4) >> 2 : 0;
  // bias gradient = column sums of dU: by the first M tile (plain), by the first M tile of every parity class (class-structured
  // form: each class sees its own quarter of the pixels; the reduction adds the four)
  const bool bias_wg = p.ycls ? (m0 % (4 * p.Cin4) == 0) : (mt == 0);
  const int bias_groups = p.ycls ? 4 : 1;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;

  float4 ra[A_LD], rb[B_LD];
  auto load_chunk = [&](int c) {
    const int q0 = c * BKP;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int q = q0 + t / A_F4_ROW + j * A_PIX;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_on && q < Q) {
        const int n = (int)fdiv(q, p.fd_ohw), rem = q - n * OHW;
        const int oy = (int)fdiv(rem, p.fd_ow), ox = rem - oy * p.OW;
        int iy = oy * p.isy + a_dy, ix = ox * p.isx + a_dx;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          iy >>= p.up_shift;
          ix >>= p.up_shift;
          v = *reinterpret_cast<const float4*>(a_src + (size_t)((n * Hs + iy) * Ws + ix) * p.ldx);
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int q = q0 + t / B_F4_ROW + j * B_PIX;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_on && q < Q) {
        size_t pix = (size_t)q;
        if (p.ycls) {  // this M tile's parity class: dU on its sub-lattice of the full-resolution grid
          const int n = (int)fdiv(q, p.fd_ohw), rem = q - n * OHW;
          const int oy = (int)fdiv(rem, p.fd_ow), ox = rem - oy * p.OW;
          pix = ((size_t)n * p.OHf + 2 * oy + (ycl >> 1)) * p.OWf + 2 * ox + (ycl & 1);
        }
        const size_t off = pix * p.ldy + p.y_coff + b_co;
        v = *reinterpret_cast<const float4*>(p.dy + off);
        if (p.ya) {
          const float4 a = *reinterpret_cast<const float4*>(p.ya + off);
          v.x *= act_dfo(a.x, p.yact, p.yalpha);
          v.y *= act_dfo(a.y, p.yact, p.yalpha);
          v.z *= act_dfo(a.z, p.yact, p.yalpha);
          v.w *= act_dfo(a.w, p.yact, p.yalpha);
        }
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) *reinterpret_cast<float4*>(&As[buf][t / A_F4_ROW + j * A_PIX][a_m4 * 4]) = ra[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) *reinterpret_cast<float4*>(&Bs[buf][t / B_F4_ROW + j * B_PIX][b_c4 * 4]) = rb[j];
  };

  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk(0);
  }
  __syncthreads();
  int buf = 0;
  for (int c = c_begin; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
    {
      // register double-buffered fragments, order pinned (see conv_igemm.hip)
      float a[2][TM], b[2][TN];
      auto frag = [&](int s, int kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[s][i] = As[buf][kk * 2 + lh][wm * WTM + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s][j] = Bs[buf][kk * 2 + lh][wn * WTN + j * 32 + li];
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BKP / 2; ++kk) {
        if (kk + 1 < BKP / 2) frag((kk + 1) & 1, kk + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
        if (kk + 1 < BKP / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    }
    if (p.swapped) {  // bias = column sums of the centre tap's rows of the (gathered) A stage
      if (co0 == 0 && t < p.Cin4 && p.bias_m >= m0 && p.bias_m < m0 + BM) {
#pragma unroll
        for (int k = 0; k < BKP; ++k) bsum += As[buf][k][p.bias_m - m0 + t];
      }
    } else if (bias_wg && t < BN) {
#pragma unroll
      for (int k = 0; k < BKP; ++k) bsum += Bs[buf][k][t];
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // partial[split][Mpad][ldn] (+ bias partials pb[split][ldn]); padding rows / columns are written too (zeros)
  const int ldn = co_tiles * BN;
  float* dst = p.partial + (size_t)by * p.Mpad * ldn;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < TN; ++j) dst[(size_t)m * ldn + co0 + wn * WTN + j * 32 + li] = acc[i][j][r];
    }
  if (p.swapped) {
    if (co0 == 0 && t < p.Cin4 && p.bias_m >= m0 && p.bias_m < m0 + BM) p.pbias[(size_t)by * ldn + t] = bsum;
  } else if (bias_wg && t < BN) {
    p.pbias[((size_t)by * bias_groups + ycl) * ldn + co0 + t] = bsum;
  }
}

// LDS-DMA variant (dU operand, i.e. no act' on load): 512-thread workgroups, waves 4-7 stage both pixel-major operands
// with global_load_lds_dwordx4 (they ARE the K-major LDS image: no swizzle needed, fragments stay conflict-free
// ds_read_b32), halo / tail lanes read a zero block; waves 0-3 run the MFMAs and the bias column sums.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, bool F16 = false>
__global__ __launch_bounds__(512, NS == 2 ? 4 : 2) void conv_wgrad_dma_kernel(const WgradParams p, int co_tiles, int nsplit) {
  static_assert(NS == 2 || NS == 3, "stages");
  constexpr int BKP = 32;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WAVES_M * WAVES_N == 4 && TM * 32 == WTM && TN * 32 == WTN, "tile");
  constexpr int A_F4_ROW = BM / 4, B_F4_ROW = BN / 4;
  constexpr int A_LD = BKP * A_F4_ROW / 256, B_F4 = BKP * B_F4_ROW;
  constexpr int B_LD = (B_F4 + 255) / 256;
  static_assert(A_LD * 256 == BKP * A_F4_ROW, "tile");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  __shared__ __attribute__((aligned(16))) float As[NS][BKP][BM];
  __shared__ __attribute__((aligned(16))) float Bs[NS][BKP][BN];
  __shared__ int2 tap_yx[UDET_MAX_TAPS];

  const int tid = threadIdx.x;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int t = tid & 255, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the (M tile, N tile) workgroups of ONE pixel slice read the same dU rows and the same (tap-shifted) X rows; as
  // consecutive hardware ids they were dealt round-robin to the eight XCDs and every L2 fetched the slice for itself (PMC, round 4:
  // 5.4 GB per step for 1.1 GB of operands).  Each XCD now walks whole slices: logical id = slice * tiles + tile.
  int bx, by;
  {
    const int nx = gridDim.x, nwg = nx * gridDim.y, h = blockIdx.x + nx * blockIdx.y;
    const int q = nwg >> 3, r = nwg & 7, xcd = h & 7, idx = h >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = gridDim.y == 1 ? 0 : l / nx;  // (one pixel slice: no run-time division)
    bx = l - by * nx;
  }
  const int mt = co_tiles == 1 ? bx : bx / co_tiles;
  const int m0 = mt * BM, co0 = (bx - mt * co_tiles) * BN;
  const int OHW = p.OH * p.OW;
  const int Q = p.N * OHW;
  const int nchunks = (Q + BKP - 1) / BKP;
  // (nchunks * nsplit < 2^31: 32-bit divisions -- a 64-bit one is ~150 instructions in front of every workgroup's first load)
  const int c_begin = (int)((unsigned)(nchunks * by) / (unsigned)nsplit), c_end = (int)((unsigned)(nchunks * (by + 1)) / (unsigned)nsplit);
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const int cout4 = (p.Cout + 3) & ~3;

  for (int i = tid; i < p.ntaps; i += 512) tap_yx[i] = make_int2(p.taps[i].dy, p.taps[i].dx);
  __syncthreads();

  if (role == 1) {
    __builtin_amdgcn_s_setprio(3);  // staging waves issue ahead of the MFMA waves (see conv_igemm_dma_kernel)
    const int ycl = p.ycls ? (m0 / p.Cin4) >> 2 : 0;
    const int a_m4 = t % A_F4_ROW;
    const int a_m = m0 + a_m4 * 4;
    const int a_tap = a_m / p.Cin4, a_ci = a_m - a_tap * p.Cin4;
    const bool a_on = a_tap < p.ntaps;
    int a_dy = 0, a_dx = 0;
    if (a_on) { a_dy = tap_yx[a_tap].x; a_dx = tap_yx[a_tap].y; }
    const float* a_src = p.x + p.x_coff + a_ci;
    const float* zero = p.zero16;
    auto issue = [&](int buf, int c) {
      const int q0 = c * BKP;
#pragma unroll
      for (int j = 0; j < A_LD; ++j) {
        const int q = q0 + (t + j * 256) / A_F4_ROW;
        const float* src = zero;
        if (a_on && q < Q) {
          const int n = (int)fdiv(q, p.fd_ohw), rem = q - n * OHW;
          const int oy = (int)fdiv(rem, p.fd_ow), ox = rem - oy * p.OW;
          int iy = oy * p.isy + a_dy, ix = ox * p.isx + a_dx;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
            iy >>= p.up_shift;
            ix >>= p.up_shift;
            src = a_src + (size_t)((n * Hs + iy) * Ws + ix) * p.ldx;
          }
        }
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&As[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int idx = t + j * 256;
        if (B_F4 % 256 != 0 && idx - lane + 63 >= B_F4 && idx - lane >= B_F4) continue;  // whole wave beyond the tile
        const int kp = idx / B_F4_ROW, c4 = idx - kp * B_F4_ROW;
        const int q = q0 + kp, co = co0 + c4 * 4;
        size_t pix = (size_t)q;
        if (p.ycls && q < Q) {
          const int n = (int)fdiv(q, p.fd_ohw), rem = q - n * OHW;
          const int oy = (int)fdiv(rem, p.fd_ow), ox = rem - oy * p.OW;
          pix = ((size_t)n * p.OHf + 2 * oy + (ycl >> 1)) * p.OWf + 2 * ox + (ycl & 1);
        }
        const float* src = (idx < B_F4 && q < Q && co < cout4) ? p.dy + (pix * p.ldy + p.y_coff + co) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
    };
    // NS-deep ring (conv_igemm_dma_kernel): NS - 1 stages in flight; every lane issues A_LD + B_LD DMA instructions per stage
    static_assert(NS == 2 || B_F4 % 256 == 0, "a ring deeper than 2 counts the DMA instructions per stage");
    constexpr int L = A_LD + B_LD;
    auto landed = [&](int newer) {
      if (NS > 2 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    int issued = c_begin, ibuf = 0;
    for (int s = 0; s < NS - 1 && issued < c_end; ++s) {
      issue(ibuf, issued);
      ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
      ++issued;
    }
    landed(issued - c_begin - 1);
    for (int c = c_begin; c < c_end; ++c) {
      if (issued < c_end) {
        issue(ibuf, issued);
        ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
        ++issued;
      }
      landed(issued - c - 2);
    }
    return;
  }

  const int ycl = p.ycls ? (m0 / p.Cin4) >> 2 : 0;
  const bool bias_wg = p.ycls ? (m0 % (4 * p.Cin4) == 0) : (mt == 0);
  const int bias_groups = p.ycls ? 4 : 1;
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;
  const float yscale = F16 ? p.f16_yscale : 1.f;  // on the gradient operand: B, or A in the swapped view
  const float ascale = p.swapped ? yscale : 1.f, bscale = p.swapped ? 1.f : yscale;
  auto handover = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  handover();
  int buf = 0;
  for (int c = c_begin; c < c_end; ++c) {
    if constexpr (F16) {  // K = 8 pixels per MFMA: this lane half's four consecutive pixels of a column are one fp16 operand
      halfx4 a[2][TM], b[2][TN];
      auto frag = [&](int s, int kk) {
        const int k0 = kk * 8 + lh * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = wm * WTM + i * 32 + li;
          a[s][i] = halfx4{(_Float16)(As[buf][k0][m] * ascale), (_Float16)(As[buf][k0 + 1][m] * ascale), (_Float16)(As[buf][k0 + 2][m] * ascale),
                           (_Float16)(As[buf][k0 + 3][m] * ascale)};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = wn * WTN + j * 32 + li;
          b[s][j] = halfx4{(_Float16)(Bs[buf][k0][n] * bscale), (_Float16)(Bs[buf][k0 + 1][n] * bscale), (_Float16)(Bs[buf][k0 + 2][n] * bscale),
                           (_Float16)(Bs[buf][k0 + 3][n] * bscale)};
        }
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BKP / 8; ++kk) {
        if (kk + 1 < BKP / 8) frag((kk + 1) & 1, kk + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
      }
    } else {
      float a[2][TM], b[2][TN];
      auto frag = [&](int s, int kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[s][i] = As[buf][kk * 2 + lh][wm * WTM + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s][j] = Bs[buf][kk * 2 + lh][wn * WTN + j * 32 + li];
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BKP / 2; ++kk) {
        if (kk + 1 < BKP / 2) frag((kk + 1) & 1, kk + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
        if (kk + 1 < BKP / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    }
    if (p.swapped) {  // bias = column sums of the centre tap's rows of the (gathered) A stage
      if (co0 == 0 && t < p.Cin4 && p.bias_m >= m0 && p.bias_m < m0 + BM) {
#pragma unroll
        for (int k = 0; k < BKP; ++k) bsum += As[buf][k][p.bias_m - m0 + t];
      }
    } else if (bias_wg && t < BN) {
#pragma unroll
      for (int k = 0; k < BKP; ++k) bsum += Bs[buf][k][t];
    }
    handover();
    buf = buf + 1 == NS ? 0 : buf + 1;
  }
  if (F16 && yscale != 1.f) {
    const float inv = 1.f / yscale;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  const int ldn = co_tiles * BN;
  float* dst = p.partial + (size_t)by * p.Mpad * ldn;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < TN; ++j) dst[(size_t)m * ldn + co0 + wn * WTN + j * 32 + li] = acc[i][j][r];
    }
  if (p.swapped) {
    if (co0 == 0 && t < p.Cin4 && p.bias_m >= m0 && p.bias_m < m0 + BM) p.pbias[(size_t)by * ldn + t] = bsum;
  } else if (bias_wg && t < BN) {
    p.pbias[((size_t)by * bias_groups + ycl) * ldn + co0 + t] = bsum;
  }
}

// dw[widx][ci][co] = sum_s partial[s][tap*Cin4+ci][co] ; db[co] = sum_s pbias[s][co].  SL lanes share one element
// (each sums every SL-th split, then a fixed-order shuffle tree): deterministic, and parallel for tiny filters.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradParams p, int ldn, int nsplit) {
  const int Mreal = p.ntaps * p.Cin4;
  const int ncol = p.Cout;  // GEMM-view columns
  const long total = (long)(Mreal + 1) * ncol;
  const size_t slab = (size_t)p.Mpad * ldn;
  const int sl = threadIdx.x % SL;
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) / SL; e < total; e += (long)gridDim.x * (256 / SL)) {
    const int m = (int)(e / ncol), co = (int)(e - (long)m * ncol);
    const bool is_bias = m == Mreal;
    if (is_bias && p.swapped && co >= p.oCout) continue;  // (uniform over the SL lanes of an element)
    const float* src = is_bias ? p.pbias + co : p.partial + (size_t)m * ldn + co;
    const size_t stride = is_bias ? (size_t)ldn : slab;
    const int nsum = is_bias && p.ycls ? 4 * nsplit : nsplit;  // (class-structured form: four class partials per split)
    float s = 0.f;  // eight partials in flight per trip, added in split order
    int k = sl;
    for (; k + 7 * SL < nsum; k += 8 * SL) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(k + u * SL) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += a[u];
    }
    for (; k < nsum; k += SL) s += src[k * stride];
#pragma unroll
    for (int d = SL / 2; d > 0; d >>= 1) s += __shfl_xor(s, d, SL);
    if (sl != 0) continue;
    if (is_bias) {
      if (p.db) p.db[co] = s;
      continue;
    }
    const int tap = m / p.Cin4, ci = m - tap * p.Cin4;
    if (p.swapped) {  // rows are (tap, output channel), columns input channels
      if (ci < p.oCout) p.dw[((size_t)p.taps[tap].widx * p.oCin + co) * p.oCout + ci] = s;
    } else if (ci < p.Cin) {
      p.dw[((size_t)p.taps[tap].widx * p.Cin + ci) * p.Cout + co] = s;
    }
  }
}

// ---- BN-folded generator layers in TWO launches behind the GEMM instead of three (reduce 5 us, bn_dot 9 us, bn_finish 4 us on a 57 us
// GEMM, on the one queue the generator's filter gradients share: profiles/r04_step_ablation.txt): the slab reduction also scales
// (G * gamma*c) and forms the per-channel dot sum W*G that dgamma needs.  A block owns RB consecutive (tap, ci) rows and all columns:
// thread (rg = t / CW, co = t % CW) sums the splits of its rows in split order (the numbers wgrad_reduce_kernel<1> produces), stores
// G * gamma*c, keeps W*G; the blocks' partial dots go to pd[block][co], and a one-block launch adds them in block order.
// (Measured and dropped: the same in ONE launch, the last-arriving block finalising behind a ticket -- 55 / 36 us instead of 18:
// device-scope stores / loads of the partial dots cost more than the launch they save.)
#define BND_SPLIT 16
#define BNR_MAXBLK 1024
// SL lanes share one element (round 5): the split loop of an element is `nsplit` dependent-latency loads deep -- 256 slabs for the
// generator's first layer, whose 6400 elements are 25 blocks: 12.6 us for 8 MB.  Lane sl sums every SL-th split (eight loads in flight),
// the SL partials meet in LDS and are added in lane order: still one fixed summation order per split count.
template <int CW, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_bn_kernel(const WgradParams p, int ldn, int nsplit, int RB, float* __restrict__ pd) {
  constexpr int RG = 256 / CW;   // thread rows of a block
  constexpr int RR = RG / SL;    // element rows served at a time
  static_assert(RR >= 1 && RR * SL == RG, "split lanes");
  __shared__ float red[256];
  __shared__ float part[256];
  const int t = threadIdx.x, co = t % CW, rg = t / CW, rr = rg / SL, sl = rg % SL;
  const int Mreal = p.ntaps * p.Cin4;
  const size_t slab = (size_t)p.Mpad * ldn;
  const int r0 = blockIdx.x * RB;
  float dot = 0.f;
  const float gs = co < p.Cout ? p.gamma[co] * p.bn_c : 0.f;
  for (int mb = r0; mb < r0 + RB && mb < Mreal; mb += RR) {  // (uniform trip count: the barriers below are reached by every thread)
    const int m = mb + rr;
    const int tap = m / p.Cin4, ci = m - tap * p.Cin4;
    const bool on = co < p.Cout && m < r0 + RB && m < Mreal && ci < p.Cin;
    float s = 0.f;
    if (on) {
      const float* src = p.partial + (size_t)m * ldn + co;
      int k = sl;
      for (; k + 7 * SL < nsplit; k += 8 * SL) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = src[(size_t)(k + u * SL) * slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
      }
      for (; k < nsplit; k += SL) s += src[(size_t)k * slab];
    }
    if (SL > 1) {
      part[t] = s;
      __syncthreads();
      if (sl == 0) {
#pragma unroll
        for (int u = 1; u < SL; ++u) s += part[t + u * CW];
      }
      __syncthreads();
    }
    if (on && sl == 0) {
      const size_t e = ((size_t)p.taps[tap].widx * p.Cin + ci) * p.Cout + co;
      dot = fmaf(p.w[e], s, dot);
      p.dw[e] = s * gs;
    }
  }
  red[t] = dot;  // (zero in the lanes with sl != 0)
  __syncthreads();
  if (rg == 0 && co < p.Cout) {
    float d = red[co];
#pragma unroll
    for (int g = 1; g < RG; ++g) d += red[g * CW + co];
    pd[(size_t)blockIdx.x * p.Cout + co] = d;
  }
}
// second (last) launch of the BN-folded layers: per channel, the blocks' dots in block order and the bias partials in split order -- both
// spread over the eight thread groups of a block (a group sums a contiguous run, the runs are added in group order)
__global__ __launch_bounds__(256) void wgrad_bn_finish2_kernel(const WgradParams p, int ldn, int nsplit, int nb, const float* __restrict__ pd) {
  __shared__ float red[256];
  __shared__ float redS[256];
  const int t = threadIdx.x, cl = t & 31, rg = t >> 5, co = blockIdx.x * 32 + cl;  // 8 thread groups x 32 channels per block
  const int per = (nb + 7) / 8, b0 = rg * per, b1 = b0 + per < nb ? b0 + per : nb;
  const int pers = (nsplit + 7) / 8, k0 = rg * pers, k1 = k0 + pers < nsplit ? k0 + pers : nsplit;
  float d = 0.f, S = 0.f;
  if (co < p.Cout) {
    int b = b0;
    for (; b + 11 < b1; b += 12) {
      float a[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) a[u] = pd[(size_t)(b + u) * p.Cout + co];
#pragma unroll
      for (int u = 0; u < 12; ++u) d += a[u];
    }
    for (; b < b1; ++b) d += pd[(size_t)b * p.Cout + co];
    int k = k0;
    for (; k + 7 < k1; k += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = p.pbias[(size_t)(k + u) * ldn + co];
#pragma unroll
      for (int u = 0; u < 8; ++u) S += a[u];
    }
    for (; k < k1; ++k) S += p.pbias[(size_t)k * ldn + co];
  }
  red[t] = d;
  redS[t] = S;
  __syncthreads();
  if (rg == 0 && co < p.Cout) {
    float dsum = red[cl], Ssum = redS[cl];
#pragma unroll
    for (int g = 1; g < 8; ++g) { dsum += red[g * 32 + cl]; Ssum += redS[g * 32 + cl]; }
    p.dgamma[co] = p.bn_c * (dsum + p.b[co] * Ssum);
    p.dbeta[co] = Ssum;
    p.db[co] = p.gamma[co] * p.bn_c * Ssum;
  }
}
// the separate form (single-operator launches with the class-structured / operand-swapped views): dgamma's dot sum W*G per output channel
__global__ __launch_bounds__(256) void bn_dot_kernel(const float* __restrict__ w, const float* __restrict__ g, int R, int C,
                                                     float* __restrict__ pd) {
  __shared__ float red[256];
  const int t = threadIdx.x, cx = t & 63, ry = t >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r_begin = (int)((long)R * blockIdx.y / BND_SPLIT), r_end = (int)((long)R * (blockIdx.y + 1) / BND_SPLIT);
  float s = 0.f;
  if (c < C)
    for (int r = r_begin + ry; r < r_end; r += 4) s = fmaf(w[(size_t)r * C + c], g[(size_t)r * C + c], s);
  red[t] = s;
  __syncthreads();
  if (ry == 0 && c < C) pd[(size_t)blockIdx.y * C + c] = red[cx] + red[64 + cx] + red[128 + cx] + red[192 + cx];
}
// dw (holding G) *= gamma*c ; dgamma = c*(dot + b*S) ; dbeta = S ; db = gamma*c*S   (S arrives in db)
__global__ __launch_bounds__(256) void bn_finish_kernel(float* __restrict__ dw, long n, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ b, float bn_c, const float* __restrict__ pd,
                                                        float* __restrict__ db, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    dw[e] *= gamma[c] * bn_c;
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      float dot = 0.f;
      for (int j = 0; j < BND_SPLIT; ++j) dot += pd[(size_t)j * C + c];
      const float S = db[c];
      dgamma[c] = bn_c * (dot + b[c] * S);
      dbeta[c] = S;
      db[c] = gamma[c] * bn_c * S;
    }
  }
}

// BN-folded generator layers: dw (holding G) *= gamma*c ; dgamma = c*(sum W*G + b*S) ; dbeta = S ; db = gamma*c*S  (S arrives in db)
int launch_bn_finalize(float* dw, int T, int Cin, int Cout, const float* w, const float* b, const float* gamma, float bn_c, float* pd,
                       float* db, float* dgamma, float* dbeta, hipStream_t stream) {
  const size_t wsz = (size_t)T * Cin * Cout;
  int nbw = (int)((wsz + 255) / 256);
  if (nbw > 2048) nbw = 2048;
  UDET_LAUNCH(bn_dot_kernel, dim3((Cout + 63) / 64, BND_SPLIT), dim3(256), 0, stream, w, dw, T * Cin, Cout, pd);
  UDET_LAUNCH(bn_finish_kernel, dim3(nbw), dim3(256), 0, stream, dw, (long)wsz, Cout, gamma, b, bn_c, pd, db, dgamma, dbeta);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// dW[ky][kx] = sum over the four parity classes of dWeff[class][tap of that class that contains (ky,kx)]   (NN x2 + 3x3)
__global__ __launch_bounds__(256) void wgrad_up_combine_kernel(const float* __restrict__ deff, float* __restrict__ dw, int CC) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 9 * CC; e += gridDim.x * 256) {
    const int k = e / CC, r = e - k * CC, ky = k / 3, kx = k - ky * 3;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int py = c >> 1, px = c & 1;
      const int ty = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0), tx = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
      v += deff[(size_t)(4 * c + 2 * ty + tx) * CC + r];
    }
    dw[e] = v;
  }
}
int launch_wgrad_up_combine(const float* deff, float* dw, int Cin, int Cout, hipStream_t stream) {
  const int CC = Cin * Cout;
  UDET_LAUNCH(wgrad_up_combine_kernel, dim3((9 * CC + 255) / 256 > 1024 ? 1024 : (9 * CC + 255) / 256), dim3(256), 0, stream, deff, dw, CC);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
size_t wgrad_partial_floats_needed(int T, int Cin, int Cout) {
  // BN dot partials + one split of (bias, filter) partials at the padded tile sizes
  const size_t cin4 = (size_t)((Cin + 3) & ~3), mpad = (T * cin4 + 127) / 128 * 128, ldn = (size_t)(Cout + 127) / 128 * 128;
  return (size_t)BNR_MAXBLK * Cout + ldn + mpad * ldn + 64;
}

static int g_force_wsplit = 0, g_force_wdma = -1;  // test / tool hook (libudet_debug.so): pin the split count / staging variant
static int g_wlast = 0;                             // configuration of the most recent launch_wgrad_T: split count | variant << 20 (3: Winograd family)
int wgrad_last_config() { return g_wlast; }
void wgrad_force(int nsplit, int dma) { g_force_wsplit = nsplit > 0 ? nsplit : 0; g_force_wdma = dma; }
static std::unordered_map<uint64_t, int> g_wcache;  // problem shape -> split count | (LDS-DMA variant: 1 / 2 for a 2- / 3-stage ring) << 20
static std::mutex g_wcache_mu;
static int g_wtuning = 0;
void wgrad_set_tuning(int on) { g_wtuning = on; }
int wgrad_tuned_shapes() { std::lock_guard<std::mutex> l(g_wcache_mu); return (int)g_wcache.size(); }
void wgrad_tune_dump(FILE* f) {
  std::lock_guard<std::mutex> l(g_wcache_mu);
  for (auto& kv : g_wcache) fprintf(f, "w %llu %d\n", (unsigned long long)kv.first, kv.second);
}
void wgrad_tune_put(unsigned long long key, int cfg) {
  std::lock_guard<std::mutex> l(g_wcache_mu);
  g_wcache[(uint64_t)key] = cfg;
}

template <int BM, int BN, int WM_, int WN_>
static void wgrad_launch(const WgradParams& p, int m_tiles, int co_tiles, int nsplit, int dma, hipStream_t stream) {  // dma: 0 off, 1 / 2: 2- / 3-stage ring
  dim3 grid(m_tiles * co_tiles, nsplit);
  if (dma && p.f16) UDET_LAUNCH((conv_wgrad_dma_kernel<BM, BN, WM_, WN_, 2, true>), grid, dim3(512), 0, stream, p, co_tiles, nsplit);
  else if (dma == 2) UDET_LAUNCH((conv_wgrad_dma_kernel<BM, BN, WM_, WN_, 3>), grid, dim3(512), 0, stream, p, co_tiles, nsplit);
  else if (dma) UDET_LAUNCH((conv_wgrad_dma_kernel<BM, BN, WM_, WN_, 2>), grid, dim3(512), 0, stream, p, co_tiles, nsplit);
  else UDET_LAUNCH((conv_wgrad_kernel<BM, BN, WM_, WN_>), grid, dim3(256), 0, stream, p, co_tiles, nsplit);
}

// p.taps must list the (non-culled) taps with widx = ky*kw+kx; T = kh*kw.
int launch_wgrad_T(WgradParams& p, int T, hipStream_t stream) {
  if (conv_debug_f16_on()) p.f16 = 1;
  if (p.f16 && !(p.f16_yscale > 0.f)) p.f16_yscale = 1.f;
  if (p.f16 && g_wtuning) p.f16_yscale = 1.f;  // (tuning data: see launch_conv)
  if (p.ldx % 4 || p.x_coff % 4 || p.ldy % 4 || p.y_coff % 4) {
    set_error("wgrad: ldx=%d x_coff=%d ldy=%d y_coff=%d must be multiples of 4", p.ldx, p.x_coff, p.ldy, p.y_coff);
    return UDET_ERR_ALIGN;
  }
  // <=4 output channels over a deep input: the plain view pads the 2 MFMA output columns to 32 for every (tap, ci) row.
  // Swap the operands instead: rows = (tap, output channel) gathered from dU at the mirrored tap offsets, columns =
  // input channels read straight from X, K = input pixels:  G[(t,co)][ci] = sum_q dU[q - d_t][co] * X[q][ci] = dW[t][ci][co].
  // The bias gradient is the column sum of the centre tap's rows.
  WgradParams g = p;
  g.swapped = 0; g.bias_m = -1; g.oCin = p.Cin; g.oCout = p.Cout;
  {
    int centre = -1;
    for (int t = 0; t < p.ntaps; ++t)
      if (p.taps[t].dy == 0 && p.taps[t].dx == 0) centre = t;
    // padded MFMA work of the two views: ceil(rows/128)*128 x the N tile
    auto tile_cost = [](long rows, int cols) {
      const int bn_ = cols > 64 ? 128 : (cols > 32 ? 64 : 32);
      return ((rows + 127) / 128 * 128) * (long)((cols + bn_ - 1) / bn_ * bn_);
    };
    const int co4 = (p.Cout + 3) & ~3, ci4 = (p.Cin + 3) & ~3;
    const bool cheaper = tile_cost((long)p.ntaps * co4, p.Cin) < tile_cost((long)p.ntaps * ci4, p.Cout);
    // <= 4 channels: always (as before); 5..16 channels (deconv1, conv16): when the swapped view pads less.  The bias rows
    // (centre tap x output channels) must not straddle an M tile: 128 % co4 == 0.
    const bool narrow = p.Cout <= 4 ? p.Cin >= 16 : (p.Cout <= 16 && 128 % co4 == 0 && cheaper);
    const bool swap = narrow && p.isy == 1 && p.isx == 1 && p.up_shift == 0 && p.H == p.OH && p.W == p.OW &&
                      p.ya == nullptr && centre >= 0;
    if (swap) {
      g.x = p.dy; g.ldx = p.ldy; g.x_coff = p.y_coff; g.Cin = p.Cout;
      g.dy = p.x; g.ldy = p.ldx; g.y_coff = p.x_coff; g.Cout = p.Cin;
      for (int t = 0; t < p.ntaps; ++t) { g.taps[t].dy = -p.taps[t].dy; g.taps[t].dx = -p.taps[t].dx; }
      g.swapped = 1;
      g.bias_m = centre * co4;
    }
  }
  const int BM = 128;
  const int bn = g.Cout > 64 ? 128 : (g.Cout > 32 ? 64 : 32);
  g.Cin4 = (g.Cin + 3) & ~3;
  if (p.ycls && (g.swapped || (4 * g.Cin4) % BM != 0 || p.ntaps != 16)) {
    set_error("wgrad: class-structured form needs 16 taps and 4*Cin a multiple of %d (Cin=%d)", BM, p.Cin);
    return UDET_ERR_SHAPE;
  }
  const int Mreal = g.ntaps * g.Cin4;
  const int m_tiles = Mreal > 0 ? (Mreal + BM - 1) / BM : 1, co_tiles = (g.Cout + bn - 1) / bn;
  g.Mpad = m_tiles * BM;
  const int ldn = co_tiles * bn;
  const long Q = (long)g.N * g.OH * g.OW;
  const int nchunks = (int)((Q + 31) / 32);
  g.fd_ohw = make_fastdiv((unsigned)(g.OH * g.OW));
  g.fd_ow = make_fastdiv((unsigned)g.OW);
  float* pd = p.partial;                                 // [BNR_MAXBLK][Cout]: per-block dot partials of the fused finaliser ([BND_SPLIT][Cout] of bn_dot)
  float* base = pd + (size_t)BNR_MAXBLK * p.Cout;
  base += (16 - ((uintptr_t)base / sizeof(float)) % 16) % 16;  // keep the slabs 64-byte aligned
  const size_t fixed = (size_t)(base - p.partial);
  const size_t bgroups = p.ycls ? 4 : 1;
  const size_t per_split = bgroups * (size_t)ldn + (size_t)g.Mpad * ldn;
  if (p.partial_floats < fixed + per_split) {
    set_error("wgrad: workspace too small (%zu < %zu floats)", p.partial_floats, fixed + per_split);
    return UDET_ERR_ARG;
  }
  const long tiles = (long)m_tiles * co_tiles;
  const size_t maxs = (p.partial_floats - fixed) / per_split;
  int cap = nchunks / 2 > 0 ? nchunks / 2 : 1;
  if ((size_t)cap > maxs) cap = (int)maxs;
  int nsplit = (int)((768 + tiles - 1) / tiles);
  if (nsplit > cap) nsplit = cap;
  if (nsplit < 1) nsplit = 1;
  if (p.f16 && p.ya == nullptr && p.zero16 != nullptr && !(reinterpret_cast<uintptr_t>(p.zero16) & 15)) nsplit |= 1 << 20;  // fp16 lives in the LDS-DMA kernel
  const size_t wsz = (size_t)T * p.Cin * p.Cout;
  if (p.ntaps < T) UDET_HIP(hipMemsetAsync(p.dw, 0, wsz * sizeof(float), stream));  // culled taps have zero gradient
  const long total = (long)(Mreal + 1) * g.Cout;
  const bool dma_ok = p.ya == nullptr && p.zero16 != nullptr && !(reinterpret_cast<uintptr_t>(p.zero16) & 15);
  // BN-folded layers (generator): the slab reduction also scales and forms dgamma's dot partials, a one-block launch finishes (two launches
  // behind the GEMM; wgrad_reduce_bn_kernel above)
  const bool fused_bn = p.gamma && !g.swapped && !p.ycls && p.ntaps == T && p.Cout <= 128 && p.db && p.dgamma && p.dbeta && p.w && p.b;
  // Winograd-domain family (conv_wgrad_wino.hip; variant 3 of the configuration word): 3x3 stride-1 layers with whole 64-channel blocks; its K
  // slices write the same slabs, so everything behind the GEMM is shared
  const bool wino_ok = !g.swapped && !p.ycls && !plan_knob(UDET_KNOB_NO_WGRAD_WINO) && wgrad_wino_ok(g);
  // returns UDET_OK, or the Winograd-domain family's error code when its GEMM could not be launched AND the direct form that replaces
  // it here was not wanted by the caller either (never: the fallback below always launches) -- the slabs are never reduced stale
  auto run = [&](int cfg) -> int {
    int ns = cfg & 0xfffff;
    int dma = cfg >> 20;
    if (dma == 3 && !wino_ok) dma = 1;
    if (dma != 3 && !dma_ok) dma = 0;
    if (dma == 3) ns = wgrad_wino_slices(g, ns > (int)maxs ? (int)maxs : ns);
    WgradParams q = g;
    q.pbias = base;                                        // [ns][bias groups][ldn]
    q.partial = base + (size_t)ns * bgroups * ldn;        // [ns][Mpad][ldn]
    if (dma == 3 && launch_wgrad_wino(q, ns, ldn, stream) != UDET_OK) {
      // (eligibility re-check or attribute failure: no GEMM went out.  The direct variant writes the same slab layout -- with the
      // heuristic slice count inside the capacity -- so the reduction behind it sums fresh partials, never stale workspace)
      dma = dma_ok ? 1 : 0;
      ns = nsplit & 0xfffff;
      if (ns > cap) ns = cap;
      if (ns < 1) ns = 1;
      q.partial = base + (size_t)ns * bgroups * ldn;
    }
    if (dma == 3) {}
    else if (bn == 128) wgrad_launch<128, 128, 2, 2>(q, m_tiles, co_tiles, ns, dma, stream);
    else if (bn == 64) wgrad_launch<128, 64, 2, 2>(q, m_tiles, co_tiles, ns, dma, stream);
    else wgrad_launch<128, 32, 4, 1>(q, m_tiles, co_tiles, ns, dma, stream);
    if (fused_bn) {  // reduction + scaling + dot partials, then the one-block finish
      const int cw = g.Cout > 64 ? 128 : (g.Cout > 32 ? 64 : (g.Cout > 16 ? 32 : 16));
      const int rg = 256 / cw;
      // lanes per element: the split loop is latency-bound, but only filters small enough to leave CUs idle gain from more, smaller
      // blocks (measured: the first layer's 25 blocks 12.6 -> 5.6 us with 8 lanes; the 128-channel layers' 576 blocks 6.4 -> 10.3 us with 2)
      int sl = ns >= 64 ? 8 : (ns >= 32 ? 4 : (ns >= 12 ? 2 : 1));
      if (sl > rg) sl = rg;
      while (sl > 1 && (long)((Mreal + rg - 1) / rg) * sl > 512) sl >>= 1;
      int rb = rg / sl;  // rows per block: ONE element per SL threads (the reduction needs every CU: eight rows per thread on 72 blocks
                         // took 30 us instead of 5)
      while ((Mreal + rb - 1) / rb > BNR_MAXBLK) rb *= 2;
      const int nb = (Mreal + rb - 1) / rb;
#define UDET_RBN(CW_, SL_) UDET_LAUNCH((wgrad_reduce_bn_kernel<CW_, SL_>), dim3(nb), dim3(256), 0, stream, q, ldn, ns, rb, pd)
      if (cw == 128) { if (sl == 2) UDET_RBN(128, 2); else UDET_RBN(128, 1); }
      else if (cw == 64) { if (sl == 4) UDET_RBN(64, 4); else if (sl == 2) UDET_RBN(64, 2); else UDET_RBN(64, 1); }
      else if (cw == 32) { if (sl == 8) UDET_RBN(32, 8); else if (sl == 4) UDET_RBN(32, 4); else if (sl == 2) UDET_RBN(32, 2); else UDET_RBN(32, 1); }
      else { if (sl == 8) UDET_RBN(16, 8); else if (sl == 4) UDET_RBN(16, 4); else if (sl == 2) UDET_RBN(16, 2); else UDET_RBN(16, 1); }
#undef UDET_RBN
      UDET_LAUNCH(wgrad_bn_finish2_kernel, dim3((g.Cout + 31) / 32), dim3(256), 0, stream, q, ldn, ns, nb, pd);
      return UDET_OK;
    }
    const int sl = (ns >= 64 && total * 64 <= 262144) ? 64 : ((ns >= 8 && total * 8 <= 262144) ? 8 : 1);
    const long nbl = (total * sl + 255) / 256;
    const int nb = (int)(nbl > 4096 ? 4096 : nbl);
    if (sl == 64) UDET_LAUNCH(wgrad_reduce_kernel<64>, dim3(nb), dim3(256), 0, stream, q, ldn, ns);
    else if (sl == 8) UDET_LAUNCH(wgrad_reduce_kernel<8>, dim3(nb), dim3(256), 0, stream, q, ldn, ns);
    else UDET_LAUNCH(wgrad_reduce_kernel<1>, dim3(nb), dim3(256), 0, stream, q, ldn, ns);
    return UDET_OK;
  };
  // autotuned split count (see conv_igemm.hip): kernel + reduction timed together
  {
    const int f[] = {p.N, p.H, p.W, p.up_shift, p.Cin, p.Cout, p.ntaps, p.OH, p.OW, p.isy, p.ya ? 1 : 0, p.ldx, p.ldy, cap, dma_ok ? 1 : 0, g.swapped, p.ycls, p.f16 ? 1 : 0,
                     p.ntaps > 0 ? p.taps[0].dy : 0, p.ntaps > 0 ? p.taps[0].dx : 0};  // (the first tap's offsets: the dilation, which the Winograd family's tile grid depends on)
    uint64_t key = 1469598103934665603ull;
    for (int v : f) { key ^= (uint64_t)(uint32_t)v; key *= 1099511628211ull; }
    bool have = false;
    {
      std::lock_guard<std::mutex> l(g_wcache_mu);
      auto it = g_wcache.find(key);
      if (it != g_wcache.end()) {
        // a cached entry may come from a file (udet_tune_load): only known variants, the slice count inside this launch's capacity
        const int hv = nsplit, v = it->second >> 20, ns = it->second & 0xfffff;
        if (v < 0 || v > 3 || ns < 1) nsplit = hv;
        else nsplit = (v != 3 && ns > cap ? cap : ns) | (v << 20);  // (variant 3 is clamped to its strips / the workspace below)
        have = true;
      }
    }
    if (!have && g_wtuning) {
      static hipEvent_t e0 = nullptr, e1 = nullptr;
      if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
      const int h = nsplit;
      float best_ms = 1e30f;
      int best = h;
      // candidates: powers of two around the heuristic + the split counts that fill whole rounds of the 256 CUs
      const int hc = h & 0xfffff;  // (the heuristic carries the staging variant in bit 20 in fp16 mode)
      std::vector<int> nss = {hc / 8, hc / 4, hc / 2, hc, hc * 2, hc * 4};
      for (int k : {1, 2, 3, 4, 6, 8}) {
        const int ns = (int)(256L * k / tiles);
        if (ns >= 1 && std::find(nss.begin(), nss.end(), ns) == nss.end()) nss.push_back(ns);
      }
      for (int dma = (p.f16 && dma_ok) ? 1 : 0; dma <= (dma_ok ? (p.f16 ? 1 : 2) : 0); ++dma)
        for (int ns : nss) {
          if (ns < 1 || ns > cap) continue;
          const int cfg = ns | (dma << 20);
          run(cfg);
          (void)hipEventRecord(e0, stream);
          for (int r = 0; r < 3; ++r) run(cfg);
          (void)hipEventRecord(e1, stream);
          if (hipEventSynchronize(e1) != hipSuccess) continue;
          float ms = 0.f;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms < best_ms) { best_ms = ms; best = cfg; }
        }
      if (wino_ok) {  // the Winograd-domain family: one workgroup per CU and channel-block pair, or a few more / fewer slices
        const int blocks = (g.Cin / 64) * (g.Cout / 64);
        for (int wg : {256, 192, 384}) {
          const int ns = wgrad_wino_slices(g, (wg + blocks - 1) / blocks);
          if (ns < 1 || (size_t)ns > maxs) continue;
          const int cfg = ns | (3 << 20);
          run(cfg);
          (void)hipEventRecord(e0, stream);
          for (int r = 0; r < 3; ++r) run(cfg);
          (void)hipEventRecord(e1, stream);
          if (hipEventSynchronize(e1) != hipSuccess) continue;
          float ms = 0.f;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (getenv("UDET_TUNE_LOG") && atoi(getenv("UDET_TUNE_LOG")) > 1)
            fprintf(stderr, "[udet tune]   wgrad winograd %d slices: %.1f us against %.1f (N=%d %dx%d Cin=%d Cout=%d)\n", ns, ms / 3 * 1e3f, best_ms / 3 * 1e3f, p.N, p.OH,
                    p.OW, p.Cin, p.Cout);
          if (ms < best_ms * 0.97f) { best_ms = ms; best = cfg; }
        }
      }
      nsplit = best;
      // the winner's filter / bias gradient must equal the heuristic configuration's (see conv_igemm.hip: candidate verification)
      if (best != h) {
        float* r0 = tune_scratch(wsz + (size_t)p.Cout, 0);
        bool ok = false;
        float diff = 0.f, scale = 0.f;
        if (r0) {
          run(h);
          (void)hipMemcpyAsync(r0, p.dw, wsz * sizeof(float), hipMemcpyDeviceToDevice, stream);
          if (p.db) (void)hipMemcpyAsync(r0 + wsz, p.db, (size_t)p.Cout * sizeof(float), hipMemcpyDeviceToDevice, stream);
          run(best);
          ok = tune_compare(r0, p.dw, wsz, stream, &diff, &scale);
          if (ok && p.db) ok = tune_compare(r0 + wsz, p.db, (size_t)p.Cout, stream, &diff, &scale);
        }
        if (!ok) {
          fprintf(stderr, "[udet tune] REJECTED wgrad N=%d %dx%d Cin=%d Cout=%d taps=%d: nsplit=%d dma=%d differs from the heuristic "
                  "configuration (max|diff| %.3e, scale %.3e)\n", p.N, p.OH, p.OW, p.Cin, p.Cout, p.ntaps, best & 0xfffff, best >> 20, diff, scale);
          conv_tune_note_reject();
          nsplit = h;
        }
      }
      if (getenv("UDET_TUNE_LOG"))
        fprintf(stderr, "[udet tune] wgrad N=%d %dx%d Cin=%d Cout=%d taps=%d -> nsplit=%d dma=%d (heuristic %d) %.1f us\n", p.N, p.OH,
                p.OW, p.Cin, p.Cout, p.ntaps, nsplit & 0xfffff, nsplit >> 20, h & 0xfffff, best_ms / 3 * 1e3f);
      std::lock_guard<std::mutex> l(g_wcache_mu);
      g_wcache[key] = nsplit;
    }
  }
  if (g_force_wsplit > 0) {
    const int ns = g_force_wsplit > cap ? cap : g_force_wsplit;
    int v = g_force_wdma >= 0 ? g_force_wdma : (nsplit >> 20);
    if (v == 3 && !wino_ok) v = dma_ok ? 1 : 0;  // (a launch the Winograd family cannot take keeps the direct form)
    if (v != 3 && !dma_ok) v = 0;
    nsplit = ns | (v << 20);
  }
  if ((nsplit >> 20) == 3 && !wino_ok) nsplit = (nsplit & 0xfffff) | ((dma_ok ? 1 : 0) << 20);  // (a cached entry of another build's rules)
  if ((nsplit >> 20) == 3) nsplit = wgrad_wino_slices(g, (nsplit & 0xfffff) > (int)maxs ? (int)maxs : (nsplit & 0xfffff)) | (3 << 20);
  g_wlast = nsplit;
  UDET_TRY(run(nsplit));
  UDET_HIP(hipGetLastError());
  int nbw = (int)((wsz + 255) / 256);
  if (nbw > 2048) nbw = 2048;
  if (p.gamma && !fused_bn) {
    if (!p.db || !p.dgamma || !p.dbeta || !p.w || !p.b) {
      set_error("wgrad: BN finalisation needs db, dgamma, dbeta, w and b");
      return UDET_ERR_ARG;
    }
    UDET_LAUNCH(bn_dot_kernel, dim3((p.Cout + 63) / 64, BND_SPLIT), dim3(256), 0, stream, p.w, p.dw, T * p.Cin,
                       p.Cout, pd);
    UDET_LAUNCH(bn_finish_kernel, dim3(nbw), dim3(256), 0, stream, p.dw, (long)wsz, p.Cout, p.gamma, p.b, p.bn_c, pd,
                       p.db, p.dgamma, p.dbeta);
    UDET_HIP(hipGetLastError());
  }
  return UDET_OK;
}

}  // namespace udet
