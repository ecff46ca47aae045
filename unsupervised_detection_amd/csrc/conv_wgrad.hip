// Backward-filter (wgrad) of the convolutions on the fp32 matrix cores, plus bias / BN
// parameter gradients.  GEMM view per filter tap t:
//     dW[t][ci][co] = sum_q X(q@t)[ci] * dU[q][co],   dU = dY * act'(saved output)
//   M = input channels, N = output channels, K = output pixels (split across workgroups).
// Both operands are pixel-major in memory ([pixel][channel]) which is exactly the K-major
// LDS image the MFMA fragments want, so staging is plain float4 copies.
// Replaces TF-1.13 Conv2DBackpropFilter / BiasAddGrad / FusedBatchNormGrad(inference) reached
// through optimizer.compute_gradients (models/utils/loss_utils.py:18).
#include "common.h"
#include "conv_host.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BKP>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p, int co_tiles, int nsplit, int T) {
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_F4_PER_ROW = BM / 4, B_F4_PER_ROW = BN / 4;
  constexpr int A_F4 = BKP * A_F4_PER_ROW, B_F4 = BKP * B_F4_PER_ROW;
  constexpr int A_LD = (A_F4 + 255) / 256, B_LD = (B_F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BKP][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BKP][BN];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ci0 = (blockIdx.x / co_tiles) * BM, co0 = (blockIdx.x % co_tiles) * BN;
  const ConvTap tap = p.taps[blockIdx.y];
  const int OHW = p.OH * p.OW;
  const int Q = p.N * OHW;
  const int nchunks = (Q + BKP - 1) / BKP;
  const int c_begin = (int)((long)nchunks * blockIdx.z / nsplit), c_end = (int)((long)nchunks * (blockIdx.z + 1) / nsplit);
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  const int cin4 = (p.Cin + 3) & ~3, cout4 = (p.Cout + 3) & ~3;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_LD], rb[B_LD];
  auto load_chunk = [&](int c) {
    const int q0 = c * BKP;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int idx = t + j * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < A_F4) {
        const int kp = idx / A_F4_PER_ROW, c4 = idx - kp * A_F4_PER_ROW;
        const int q = q0 + kp, ci = ci0 + c4 * 4;
        if (q < Q && ci < cin4) {
          const int n = q / OHW, rem = q - n * OHW;
          const int oy = rem / p.OW, ox = rem - oy * p.OW;
          int iy = oy * p.isy + tap.dy, ix = ox * p.isx + tap.dx;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
            iy >>= p.up_shift;
            ix >>= p.up_shift;
            v = *reinterpret_cast<const float4*>(p.x + (size_t)((n * Hs + iy) * Ws + ix) * p.ldx + p.x_coff + ci);
          }
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        const int kp = idx / B_F4_PER_ROW, c4 = idx - kp * B_F4_PER_ROW;
        const int q = q0 + kp, co = co0 + c4 * 4;
        if (q < Q && co < cout4) {
          const size_t off = (size_t)q * p.ldy + p.y_coff + co;
          v = *reinterpret_cast<const float4*>(p.dy + off);
          if (p.ya) {
            const float4 a = *reinterpret_cast<const float4*>(p.ya + off);
            v.x *= act_dfo(a.x, p.yact, p.yalpha);
            v.y *= act_dfo(a.y, p.yact, p.yalpha);
            v.z *= act_dfo(a.z, p.yact, p.yalpha);
            v.w *= act_dfo(a.w, p.yact, p.yalpha);
          }
        }
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int idx = t + j * 256;
      if (idx < A_F4) {
        const int kp = idx / A_F4_PER_ROW, c4 = idx - kp * A_F4_PER_ROW;
        *reinterpret_cast<float4*>(&As[buf][kp][c4 * 4]) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256;
      if (idx < B_F4) {
        const int kp = idx / B_F4_PER_ROW, c4 = idx - kp * B_F4_PER_ROW;
        *reinterpret_cast<float4*>(&Bs[buf][kp][c4 * 4]) = rb[j];
      }
    }
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
#pragma unroll
    for (int kk = 0; kk < BKP / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk * 2 + lh][wm * WTM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk * 2 + lh][wn * WTN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  float* dst = p.partial + ((size_t)blockIdx.z * T + tap.widx) * p.Cin * p.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ci >= p.Cin) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = co0 + wn * WTN + j * 32 + li;
        if (co < p.Cout) dst[(size_t)ci * p.Cout + co] = acc[i][j][r];
      }
    }
}

// dw[e] = sum_s partial[s][e]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           long n, int nsplit) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * n + e];
    dw[e] = s;
  }
}

// ---- bias gradient: S[co] = sum_q dY[q][co]*act'(ya[q][co]) ; two deterministic stages ----
#define BG_BLOCKS 128
__global__ __launch_bounds__(256) void bias_grad_stage1(const float* __restrict__ dy, const float* __restrict__ ya, int ld,
                                                        int coff, int C, int act, float alpha, long Q,
                                                        float* __restrict__ pb) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  const int cw = C < 256 ? C : 256;           // channels handled per pass
  const int rows = 256 / cw > 0 ? 256 / cw : 1;  // pixel rows handled in parallel
  const long q_begin = Q * blockIdx.x / gridDim.x, q_end = Q * (blockIdx.x + 1) / gridDim.x;
  for (int cbase = 0; cbase < C; cbase += cw) {
    const int cx = t % cw, ry = t / cw;
    const int c = cbase + cx;
    float s = 0.f;
    if (ry < rows && c < C) {
      for (long q = q_begin + ry; q < q_end; q += rows) {
        const size_t off = (size_t)q * ld + coff + c;
        float v = dy[off];
        if (ya) v *= act_dfo(ya[off], act, alpha);
        s += v;
      }
    }
    red[t] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
      float tot = 0.f;
      for (int r = 0; r < rows; ++r) tot += red[r * cw + cx];
      pb[(size_t)blockIdx.x * C + c] = tot;
    }
    __syncthreads();
  }
}
__global__ void bias_grad_stage2(const float* __restrict__ pb, float* __restrict__ db, int C, int nblk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += pb[(size_t)b * C + c];
    db[c] = s;
  }
}

// ---- BN-folded generator layers: dgamma needs sum_{t,ci} W*G per output channel ----
#define BND_SPLIT 16
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

size_t wgrad_partial_floats_needed(int T, int Cin, int Cout) {
  // one split of filter partials + bias-grad stage-1 partials + BN dot partials
  return (size_t)BG_BLOCKS * Cout + (size_t)BND_SPLIT * Cout + (size_t)T * Cin * Cout;
}

template <int BM, int BN>
static void wgrad_launch(const WgradParams& p, int ci_tiles, int co_tiles, int nsplit, int T, hipStream_t stream) {
  dim3 grid(ci_tiles * co_tiles, p.ntaps, nsplit);
  hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, 16>), grid, dim3(256), 0, stream, p, co_tiles, nsplit, T);
}

// p.taps must list the (non-culled) taps with widx = ky*kw+kx; T = kh*kw.
int launch_wgrad_T(WgradParams& p, int T, hipStream_t stream) {
  if (p.ldx % 4 || p.x_coff % 4 || p.ldy % 4 || p.y_coff % 4) {
    set_error("wgrad: ldx=%d x_coff=%d ldy=%d y_coff=%d must be multiples of 4", p.ldx, p.x_coff, p.ldy, p.y_coff);
    return UDET_ERR_ALIGN;
  }
  const int bm = p.Cin > 64 ? 128 : 64, bn = p.Cout > 64 ? 128 : 64;
  const int ci_tiles = (p.Cin + bm - 1) / bm, co_tiles = (p.Cout + bn - 1) / bn;
  const long Q = (long)p.N * p.OH * p.OW;
  const int nchunks = (int)((Q + 15) / 16);
  const size_t wsz = (size_t)T * p.Cin * p.Cout;
  float* pb = p.partial;                                   // [BG_BLOCKS][Cout]
  float* pd = pb + (size_t)BG_BLOCKS * p.Cout;             // [BND_SPLIT][Cout]
  float* pw = pd + (size_t)BND_SPLIT * p.Cout;             // [nsplit][T][Cin][Cout]
  const size_t fixed = (size_t)(pw - p.partial);
  if (p.partial_floats < fixed + wsz) {
    set_error("wgrad: workspace too small (%zu < %zu floats)", p.partial_floats, fixed + wsz);
    return UDET_ERR_ARG;
  }
  const long tiles = (long)ci_tiles * co_tiles * (p.ntaps > 0 ? p.ntaps : 1);
  int nsplit = (int)((768 + tiles - 1) / tiles);
  if (nsplit > nchunks / 4) nsplit = nchunks / 4;
  const size_t maxs = (p.partial_floats - fixed) / wsz;
  if ((size_t)nsplit > maxs) nsplit = (int)maxs;
  if (nsplit < 1) nsplit = 1;
  if (p.ntaps < T) UDET_HIP(hipMemsetAsync(pw, 0, (size_t)nsplit * wsz * sizeof(float), stream));
  WgradParams q = p;
  q.partial = pw;
  if (p.ntaps > 0) {
    if (bm == 128 && bn == 128) wgrad_launch<128, 128>(q, ci_tiles, co_tiles, nsplit, T, stream);
    else if (bm == 128) wgrad_launch<128, 64>(q, ci_tiles, co_tiles, nsplit, T, stream);
    else if (bn == 128) wgrad_launch<64, 128>(q, ci_tiles, co_tiles, nsplit, T, stream);
    else wgrad_launch<64, 64>(q, ci_tiles, co_tiles, nsplit, T, stream);
    UDET_HIP(hipGetLastError());
  }
  int nb = (int)((wsz + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, stream, pw, p.dw, (long)wsz, nsplit);
  UDET_HIP(hipGetLastError());
  if (p.db) {
    hipLaunchKernelGGL(bias_grad_stage1, dim3(BG_BLOCKS), dim3(256), 0, stream, p.dy, p.ya, p.ldy, p.y_coff, p.Cout,
                       p.yact, p.yalpha, Q, pb);
    hipLaunchKernelGGL(bias_grad_stage2, dim3((p.Cout + 127) / 128), dim3(128), 0, stream, pb, p.db, p.Cout, BG_BLOCKS);
    UDET_HIP(hipGetLastError());
  }
  if (p.gamma) {
    if (!p.db || !p.dgamma || !p.dbeta || !p.w || !p.b) {
      set_error("wgrad: BN finalisation needs db, dgamma, dbeta, w and b");
      return UDET_ERR_ARG;
    }
    hipLaunchKernelGGL(bn_dot_kernel, dim3((p.Cout + 63) / 64, BND_SPLIT), dim3(256), 0, stream, p.w, p.dw, T * p.Cin,
                       p.Cout, pd);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(nb), dim3(256), 0, stream, p.dw, (long)wsz, p.Cout, p.gamma, p.b, p.bn_c, pd,
                       p.db, p.dgamma, p.dbeta);
    UDET_HIP(hipGetLastError());
  }
  return UDET_OK;
}

}  // namespace udet
