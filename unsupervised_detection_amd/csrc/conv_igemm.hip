// Implicit-GEMM convolution on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   M = output pixels (n,qy,qx)   N = output channels   K = taps x input channels
//
// 256-thread workgroups (4 wave64), BM x BN output tile, BK input channels of one
// tap per LDS stage, double-buffered LDS, register-staged global->LDS copies.
// A tile is stored K-major in LDS ([BK][BM+pad]) so that the MFMA A fragment
// (lane l -> row l&31, k = l>>5) is a conflict-free ds_read_b32; the B tile
// ([BK][BN]) is the packed-weight layout itself.
//
// Replaces the TF-1.13 Conv2D / Conv2DBackpropInput kernels the reference calls through
// tf.layers.conv2d / tf.nn.conv2d / tf.layers.conv2d_transpose
// (models/utils/convolution_utils.py:46,81; models/PWCNet/model_pwcnet.py:161-165,286,484-504,562-574).
#include "common.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM * 32 == WTM && TN * 32 == WTN, "wave tile must be a multiple of 32");
  constexpr int APAD = (BK == 8) ? 4 : 2;
  constexpr int LDA = BM + APAD;
  constexpr int LDB = BN;
  constexpr int A_F4_PER_ROW = BK / 4;
  constexpr int A_ROWS_PER_PASS = 256 / A_F4_PER_ROW;
  constexpr int A_LD = (BM + A_ROWS_PER_PASS - 1) / A_ROWS_PER_PASS;
  constexpr int B_F4_PER_ROW = BN / 4;
  constexpr int B_F4 = BK * B_F4_PER_ROW;
  constexpr int B_LD = (B_F4 + 255) / 256;

  __shared__ float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  __shared__ int rowoff[BM];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: consecutive M tiles (which share input halos) stay on one XCD's L2.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int OHWq = p.OHq * p.OWq;
  const int Mtot = p.N * OHWq;
  const int m0 = bid * BM;
  const int n0 = blockIdx.y * BN;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;

  // ---- per-row bookkeeping ------------------------------------------------
  for (int r = t; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < Mtot) {
      const int n = m / OHWq, rem = m - n * OHWq;
      const int qy = rem / p.OWq, qx = rem - qy * p.OWq;
      off = p.ksplit > 1 ? m : (n * p.OH + qy * p.osy + p.ooy) * p.OW + qx * p.osx + p.oox;
    }
    rowoff[r] = off;
  }
  const int a_kq = t % A_F4_PER_ROW;
  int a_base[A_LD], a_iy0[A_LD], a_ix0[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int r = t / A_F4_PER_ROW + j * A_ROWS_PER_PASS;
    const int m = m0 + r;
    if (r < BM && m < Mtot) {
      const int n = m / OHWq, rem = m - n * OHWq;
      const int qy = rem / p.OWq, qx = rem - qy * p.OWq;
      a_base[j] = n * Hs * Ws;
      a_iy0[j] = qy * p.isy;
      a_ix0[j] = qx * p.isx;
    } else {
      a_base[j] = 0;
      a_iy0[j] = -(1 << 28);
      a_ix0[j] = 0;
    }
  }

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kchunks = p.Kc / BK;
  const int nchunks = p.ntaps * kchunks;
  int c_begin = 0, c_end = nchunks;
  if (p.ksplit > 1) {
    c_begin = (int)((long)nchunks * blockIdx.z / p.ksplit);
    c_end = (int)((long)nchunks * (blockIdx.z + 1) / p.ksplit);
  }

  float4 ra[A_LD], rb[B_LD];
  auto load_chunk = [&](int c) {
    const int tp = c / kchunks;
    const int kc = (c - tp * kchunks) * BK;
    const int dy = p.taps[tp].dy, dx = p.taps[tp].dx, widx = p.taps[tp].widx;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        iy >>= p.up_shift;
        ix >>= p.up_shift;
        const size_t off = (size_t)(a_base[j] + iy * Ws + ix) * p.ldx + p.x_coff + kc + a_kq * 4;
        v = *reinterpret_cast<const float4*>(p.x + off);
        if (p.xa) {
          const float4 a = *reinterpret_cast<const float4*>(p.xa + off);
          v.x *= act_dfo(a.x, p.xact, p.xalpha);
          v.y *= act_dfo(a.y, p.xact, p.xalpha);
          v.z *= act_dfo(a.z, p.xact, p.xalpha);
          v.w *= act_dfo(a.w, p.xact, p.xalpha);
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        const int krow = idx / B_F4_PER_ROW, c4 = idx - krow * B_F4_PER_ROW;
        const int n = n0 + c4 * 4;
        if (n < p.ldw) v = *reinterpret_cast<const float4*>(p.wp + ((size_t)widx * p.Kc + kc + krow) * p.ldw + n);
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int r = t / A_F4_PER_ROW + j * A_ROWS_PER_PASS;
      if (r < BM) {
        As[buf][a_kq * 4 + 0][r] = ra[j].x;
        As[buf][a_kq * 4 + 1][r] = ra[j].y;
        As[buf][a_kq * 4 + 2][r] = ra[j].z;
        As[buf][a_kq * 4 + 3][r] = ra[j].w;
      }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256;
      if (idx < B_F4) {
        const int krow = idx / B_F4_PER_ROW, c4 = idx - krow * B_F4_PER_ROW;
        *reinterpret_cast<float4*>(&Bs[buf][krow][c4 * 4]) = rb[j];
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
    for (int kk = 0; kk < BK / 2; ++kk) {
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

  // ---- epilogue -------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int off = rowoff[row];
      if (off < 0) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + li;
        float v = acc[i][j][r];
        if (p.ksplit > 1) {
          if (n < p.ldp) p.partial[((size_t)blockIdx.z * Mtot + off) * p.ldp + n] = v;
          continue;
        }
        if (n >= p.Cout) continue;
        if (p.bias) v += p.bias[n];
        v = act_fwd(v, p.act, p.alpha);
        if (p.y2) p.y2[(size_t)off * p.ldy2 + p.y2_coff + n] = v;
        if (p.res) v += p.res[(size_t)off * p.ldres + p.res_coff + n];
        float* dst = p.y + (size_t)off * p.ldy + p.y_coff + n;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

// second pass of a split-K launch: sum the partial slabs and run the epilogue
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvParams p) {
  const int OHWq = p.OHq * p.OWq;
  const int Mtot = p.N * OHWq;
  const long total = (long)Mtot * p.Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int m = (int)(e / p.Cout), n = (int)(e - (long)m * p.Cout);
    float v = 0.f;
    for (int s = 0; s < p.ksplit; ++s) v += p.partial[((size_t)s * Mtot + m) * p.ldp + n];
    const int nb = m / OHWq, rem = m - nb * OHWq;
    const int qy = rem / p.OWq, qx = rem - qy * p.OWq;
    const int off = (nb * p.OH + qy * p.osy + p.ooy) * p.OW + qx * p.osx + p.oox;
    if (p.bias) v += p.bias[n];
    v = act_fwd(v, p.act, p.alpha);
    if (p.y2) p.y2[(size_t)off * p.ldy2 + p.y2_coff + n] = v;
    if (p.res) v += p.res[(size_t)off * p.ldres + p.res_coff + n];
    float* dst = p.y + (size_t)off * p.ldy + p.y_coff + n;
    if (p.accumulate) v += *dst;
    *dst = v;
  }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_cfg(const ConvParams& p, hipStream_t stream) {
  const int Mtot = p.N * p.OHq * p.OWq;
  dim3 grid((Mtot + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.ksplit > 1 ? p.ksplit : 1);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N>), grid, dim3(256), 0, stream, p);
  UDET_HIP(hipGetLastError());
  if (p.ksplit > 1) {
    const long total = (long)Mtot * p.Cout;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(nb), dim3(256), 0, stream, p);
    UDET_HIP(hipGetLastError());
  }
  return UDET_OK;
}

int conv_pick_ksplit(const ConvParams& p, int bm, int bn, int bk) {
  const int Mtot = p.N * p.OHq * p.OWq;
  const long tiles = (long)((Mtot + bm - 1) / bm) * ((p.Cout + bn - 1) / bn);
  const int nchunks = p.ntaps * (p.Kc / bk);
  if (tiles >= 384 || !p.partial) return 1;
  int ks = (int)((512 + tiles - 1) / tiles);
  const int maxks = nchunks / 8 > 0 ? nchunks / 8 : 1;  // keep >= 8 chunks per split
  if (ks > maxks) ks = maxks;
  if (ks > 64) ks = 64;
  const size_t per_split = (size_t)Mtot * ((p.Cout + 3) & ~3);
  while (ks > 1 && per_split * ks > p.partial_cap) --ks;
  return ks < 2 ? 1 : ks;
}

int launch_conv(ConvParams& p, hipStream_t stream) {
  if (p.Kc % 8 != 0 || p.ldx % 4 != 0 || p.x_coff % 4 != 0 || p.ldw % 4 != 0) {
    set_error("conv: Kc=%d ldx=%d x_coff=%d ldw=%d violate the 8/4/4/4 alignment contract", p.Kc, p.ldx, p.x_coff, p.ldw);
    return UDET_ERR_ALIGN;
  }
  if (p.ntaps < 0 || p.ntaps > UDET_MAX_TAPS) {
    set_error("conv: ntaps=%d out of range", p.ntaps);
    return UDET_ERR_SHAPE;
  }
  if ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.wp)) & 15) {
    set_error("conv: x / packed weights must be 16-byte aligned");
    return UDET_ERR_ALIGN;
  }
  const bool k16 = (p.Kc % 16 == 0);
  int bm, bn;
  if (p.Cout <= 32) { bm = 256; bn = 32; }
  else if (p.Cout <= 64) { bm = 128; bn = 64; }
  else if (p.Cout <= 96) { bm = 128; bn = 96; }
  else { bm = 128; bn = 128; }
  p.ksplit = conv_pick_ksplit(p, bm, bn, k16 ? 16 : 8);
  if (p.ksplit > 1) p.ldp = (p.Cout + 3) & ~3;
#define UDET_CFG(BM_, BN_, WM_, WN_) \
  return k16 ? launch_cfg<BM_, BN_, 16, WM_, WN_>(p, stream) : launch_cfg<BM_, BN_, 8, WM_, WN_>(p, stream)
  if (bn == 32) { UDET_CFG(256, 32, 4, 1); }
  if (bn == 64) { UDET_CFG(128, 64, 2, 2); }
  if (bn == 96) { UDET_CFG(128, 96, 4, 1); }
  UDET_CFG(128, 128, 2, 2);
#undef UDET_CFG
}

}  // namespace udet
