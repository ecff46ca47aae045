// Fused Winograd F(2x2,3x3) convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          16 multiplications per 2x2 output tile and channel pair instead of 36
//
// for the 3x3 stride-1 convolutions of the path (PWC-Net's dense estimators and context networks, the generator's 64 / 128-channel
// layers) and their backward-data passes (a 3x3 stride-1 convolution with mirrored taps and transposed weights).  Dilation d is
// handled as d*d independent d = 1 problems on the output sub-lattices.
//
// One workgroup = 4 waves = ONE wave per SIMD with the 512-register budget.  A wave owns ALL 16 Winograd positions of a
// 32-tile x 32-channel block: 16 accumulators = 256 registers, so the output transform A^T M A needs no cross-wave traffic.
// Workgroup tile = (WTY x WTX x WN waves): (4 WTY) x (8 WTX) tiles of 2x2 pixels, 32 WN output channels.
// K runs in stages of 8 input channels over a 3-buffer LDS ring:
//   input stage : the RAW (2 TH + 2) x (2 TW + 2) pixel halo, [channel quad][row][column parity][column / 2][4 channels]
//                 (16-byte slots; row stride == 4 (mod 8) slots makes the ds_read_b128 of a lane's 4x4 patch conflict-free);
//                 the input transform B^T d B (32 additions per channel) sits on the LDS -> VGPR path
//   weight stage: the pre-transformed U = G g G^T as the lane layout of the MFMA B operand, [position][lane half][n][4 channels]
//                 (ConvParams::wino_u, built by pack mode 7 / 8 whenever the weights are re-laid-out)
// both land by global_load_lds_dwordx4 issued by the MFMA waves themselves.  Lanes 0-31 hold channels 0-3 of the stage, lanes
// 32-63 channels 4-7: MFMA j of a position consumes component j of both lane halves (K = 2 per MFMA).
// The loop is software-pipelined by hand for one wave per SIMD: a stage is four phases (phase i = position row i: 16 MFMAs on 4
// accumulators), the LDS reads of phase i + 1 are issued before the multiplications of phase i, and the wait + barrier that hands
// stage k + 1 over sits between phases 2 and 3 of stage k -- phase 3 prefetches phase 0 of the next stage, so no phase starts by
// waiting out the LDS latency behind a barrier.  The DMA is inline assembly on purpose: behind the builtin, hipcc orders every later
// LDS read after the DMA with s_waitcnt vmcnt(0) (it models the DMA as a store to LDS), which serialises fill and multiplication in
// a kernel whose waves do both.
//
// Replaces the TF-1.13 Conv2D / Conv2DBackpropInput kernels behind the 3x3 stride-1 layers of
// models/PWCNet/model_pwcnet.py:476-506,559-576 and models/nets.py:19-36 (same call sites as conv_igemm.hip).
// Measured (tools/wino_bench.hip, profiles/r04_wino_proto_v2.txt): pwcnet/ctxt/dc_conv21 439 us against 701 us for the
// implicit GEMM (182 TFLOP/s direct-equivalent); error against a double-precision direct convolution 1.3e-6 of the output scale.
#include <string.h>

#include <mutex>
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_host.h"
#include "wino_pack.h"

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// libudet_exp.so only (tools/wino_stamps.py): per-workgroup time stamps of the Winograd kernels -- where a launch's fixed cost goes.
// [workgroup < 512][8]: 0 entry, 1 halo zero-fill + tables done, 2 first stage landed, 3 K loop done, 4 tile in LDS (output transform
// done), 5 stores issued, 6 stores acknowledged (shader cycles)
#ifdef UDET_EXPERIMENT
__device__ long long g_wino_ts[512 * 8];
#define WINO_STAMP(i)                                                                                  \
  do {                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 512 && blockIdx.z == 0) g_wino_ts[blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
extern "C" int udet_exp_wino_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_ts), (size_t)(n < 512 * 8 ? n : 512 * 8) * sizeof(long long));
}
#else
#define WINO_STAMP(i) do {} while (0)
#endif

static constexpr int wino_row_slots(int pw) { return (pw + 3) / 8 * 8 + 4; }  // smallest S >= pw with S % 8 == 4

template <int WTY, int WTX, int WN>
struct WinoGeom {
  static constexpr int NS = 3;
  static constexpr int TH = 4 * WTY, TW = 8 * WTX;        // tiles of a workgroup
  static constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;  // halo pixels
  static constexpr int CS = PW / 2;                       // used slots per column parity
  static constexpr int S = wino_row_slots(PW);            // slots per halo row
  static constexpr int HP = S / 2;                        // slot offset of the odd columns
  static constexpr int IN_SLOTS = 2 * PH * S;
  static constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;  // DMA instructions per wave and stage (input)
  static constexpr int IN_BYTES = IN_INSTR * 256 * 16;
  static constexpr int BN = 32 * WN;
  static constexpr int W_BYTES = 16 * 2 * BN * 16;
  static constexpr int W_INSTR = W_BYTES / 4096;  // (weights)
  static constexpr int STAGE = IN_BYTES + W_BYTES;
  static constexpr int L = IN_INSTR + W_INSTR;
  static constexpr int LA = (L + 1) / 2;  // part A of a stage's DMA (issued in phase 3), the rest in the next phase 0
  static constexpr int LDS_BYTES = NS * STAGE;
  static_assert(S % 8 == 4 && S >= PW && HP >= CS, "row stride");
  static_assert(LDS_BYTES >= 4 * 128 * 32 * 4 && LDS_BYTES <= 160 * 1024, "the epilogue transposes 16 KB per wave through the stage buffers");
};

// the bias quad of a lane's four output channels, requested in front of the K loop (at the head of the store loop the load is a cold miss
// with nothing to hide behind); zero where the float4 store path will not use it (K slices, ragged channel counts, columns beyond the layer)
__device__ __forceinline__ float4 wino_bias_prefetch(const ConvParams& p, int n4) {
  return (p.ksplit <= 1 && (p.Cout & 3) == 0 && n4 < p.Cout) ? epi4_bias(p, n4) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// A wave's 4 x 8 tile block -- 128 output pixels x 32 channels in `xp` ([tile pixel][32 channels], the LDS transpose of the output
// transform) -- through the epilogue into memory, or into its K slice's slab.  Plain launches keep FOUR quads in flight per lane (bias
// loaded once, per-pixel operands requested before the first store: conv_epilogue.h, epi4_*): one quad at a time, each iteration waited
// out a global-load latency behind the previous iteration's store -- ~10 us of the 55 us a 16-stage generator layer took.
__device__ __forceinline__ void wino_store_block(const ConvParams& p, const float* xp, int lane, int n4, int n, int d, int sy, int sx, int Hs, int Ws,
                                                 int Yb, int Xb, const float4& bias_pre) {
  const bool slab = p.ksplit > 1;
  const bool vec = slab ? ((reinterpret_cast<uintptr_t>(p.partial) & 15) == 0 && (p.ldp & 3) == 0) : epilogue4_out_ok(p);
  const int c4 = (lane & 7) * 4;
  if (!slab && vec) {
    if (n4 >= p.Cout) return;
    const float4 bias = bias_pre;  // (requested in front of the K loop: wino_bias_prefetch)
    const EpiAct ea = epi_act(p);
    // pixel of quad P of the block (-1: outside the grid)
    auto pix = [&](int P) {
      const int m = P >> 2, a = (P >> 1) & 1, b = P & 1;
      const int oy = Yb + 2 * (m >> 3) + a, ox = Xb + 2 * (m & 7) + b;
      return (oy >= Hs || ox >= Ws) ? -1 : (n * p.H + sy + d * oy) * p.W + sx + d * ox;
    };
    auto loop = [&](auto ELU) {  // (the activation is selected HERE, once -- not per element inside the loop: conv_epilogue.h, EpiAct)
      if (epi4_plain(p)) {  // store-only launches: nothing in the loop waits for memory
#pragma unroll 1
        for (int k0 = 0; k0 < 16; k0 += 4) {  // four LDS reads in flight, then four stores
          float4 v[4];
          int off[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int P = (k0 + u) * 8 + (lane >> 3);
            v[u] = *reinterpret_cast<const float4*>(&xp[P * 32 + c4]);
            off[u] = pix(P);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (off[u] >= 0) epi4_finish_plain32<decltype(ELU)::value>(p, off[u], n4, v[u], bias, ea.slope);  // (conv_wino_ok: 32-bit offsets)
        }
        return;
      }
      // residual / accumulate / dU emission: the operands of EIGHT quads are requested before the first of them is stored (every wait
      // for a load also drains the stores issued before it: twice per tile here instead of once per quad)
#pragma unroll 1
      for (int k0 = 0; k0 < 16; k0 += 8) {
        float4 v[8];
        int off[8];
        Epi4Req rq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int P = (k0 + u) * 8 + (lane >> 3);
          v[u] = *reinterpret_cast<const float4*>(&xp[P * 32 + c4]);
          off[u] = pix(P);
          if (off[u] >= 0) epi4_request(p, off[u], n4, rq[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (off[u] >= 0) epi4_finish<decltype(ELU)::value>(p, off[u], n4, v[u], bias, rq[u], ea);
      }
    };
    if (p.act == ACT_ELU) loop(std::true_type());
    else loop(std::false_type());
    return;
  }
  const long slab_off = (long)blockIdx.z * p.N * p.H * p.W * p.ldp;
#pragma unroll 1
  for (int k = 0; k < 16; ++k) {
    const int P = k * 8 + (lane >> 3);
    const int m = P >> 2, a = (P >> 1) & 1, b = P & 1;
    const int oy = Yb + 2 * (m >> 3) + a, ox = Xb + 2 * (m & 7) + b;
    const float4 v = *reinterpret_cast<const float4*>(&xp[P * 32 + c4]);
    if (oy >= Hs || ox >= Ws) continue;
    const int off = (n * p.H + sy + d * oy) * p.W + sx + d * ox;
    if (slab) {
      float* dst = p.partial + (slab_off + (long)off * p.ldp + n4);
      if (vec) {
        if (n4 < p.ldp) *reinterpret_cast<float4*>(dst) = v;
      } else {
        if (n4 < p.ldp) dst[0] = v.x;
        if (n4 + 1 < p.ldp) dst[1] = v.y;
        if (n4 + 2 < p.ldp) dst[2] = v.z;
        if (n4 + 3 < p.ldp) dst[3] = v.w;
      }
      continue;
    }
    if (n4 < p.Cout) conv_epilogue(p, off, n4, v.x);
    if (n4 + 1 < p.Cout) conv_epilogue(p, off, n4 + 1, v.y);
    if (n4 + 2 < p.Cout) conv_epilogue(p, off, n4 + 2, v.z);
    if (n4 + 3 < p.Cout) conv_epilogue(p, off, n4 + 3, v.w);
  }
}

// The same for ONE pixel row of the wave tile's 2x2 tiles (the role-split kernels: role 0 stores the upper row a = 0 of every tile, role 1
// the lower one): `tp` holds [q = tile * 2 + b][32 channels], 64 pixels -- eight quads per lane.
__device__ __forceinline__ void wino_store_rows(const ConvParams& p, const float* tp, int lane, int n4, int n, int d, int sy, int sx, int Hs, int Ws, int Yb,
                                                int Xb, int a, const float4& bias_pre) {
  const bool slab = p.ksplit > 1;
  const bool vec = slab ? ((reinterpret_cast<uintptr_t>(p.partial) & 15) == 0 && (p.ldp & 3) == 0) : epilogue4_out_ok(p);
  const int c4 = (lane & 7) * 4;
  auto pix = [&](int q) {  // pixel of quad q (-1: outside the grid)
    const int m = q >> 1, b = q & 1;
    const int oy = Yb + 2 * (m >> 3) + a, ox = Xb + 2 * (m & 7) + b;
    return (oy >= Hs || ox >= Ws) ? -1 : (n * p.H + sy + d * oy) * p.W + sx + d * ox;
  };
  if (!slab && vec) {
    if (n4 >= p.Cout) return;
    const float4 bias = bias_pre;  // (requested in front of the K loop: wino_bias_prefetch)
    const EpiAct ea = epi_act(p);
    auto loop = [&](auto ELU) {
      if (epi4_plain(p)) {
#pragma unroll 1
        for (int k0 = 0; k0 < 8; k0 += 4) {
          float4 v[4];
          int off[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = (k0 + u) * 8 + (lane >> 3);
            v[u] = *reinterpret_cast<const float4*>(&tp[q * 32 + c4]);
            off[u] = pix(q);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (off[u] >= 0) epi4_finish_plain32<decltype(ELU)::value>(p, off[u], n4, v[u], bias, ea.slope);
        }
        return;
      }
      float4 v[8];  // residual / accumulate / dU emission: every operand of the wave's eight quads is requested before the first store
      int off[8];
      Epi4Req rq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = u * 8 + (lane >> 3);
        v[u] = *reinterpret_cast<const float4*>(&tp[q * 32 + c4]);
        off[u] = pix(q);
        if (off[u] >= 0) epi4_request(p, off[u], n4, rq[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (off[u] >= 0) epi4_finish<decltype(ELU)::value>(p, off[u], n4, v[u], bias, rq[u], ea);
    };
    if (p.act == ACT_ELU) loop(std::true_type());
    else loop(std::false_type());
    return;
  }
  const long slab_off = (long)blockIdx.z * p.N * p.H * p.W * p.ldp;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const int q = k * 8 + (lane >> 3), off = pix(q);
    const float4 v = *reinterpret_cast<const float4*>(&tp[q * 32 + c4]);
    if (off < 0) continue;
    if (slab) {
      float* dst = p.partial + (slab_off + (long)off * p.ldp + n4);
      if (vec) {
        if (n4 < p.ldp) *reinterpret_cast<float4*>(dst) = v;
      } else {
        if (n4 < p.ldp) dst[0] = v.x;
        if (n4 + 1 < p.ldp) dst[1] = v.y;
        if (n4 + 2 < p.ldp) dst[2] = v.z;
        if (n4 + 3 < p.ldp) dst[3] = v.w;
      }
      continue;
    }
    if (n4 < p.Cout) conv_epilogue(p, off, n4, v.x);
    if (n4 + 1 < p.Cout) conv_epilogue(p, off, n4 + 1, v.y);
    if (n4 + 2 < p.Cout) conv_epilogue(p, off, n4 + 2, v.z);
    if (n4 + 3 < p.Cout) conv_epilogue(p, off, n4 + 3, v.w);
  }
}
// Output transform + store of the role-split kernels (eight-wave and half-size forms).  acc[0..3] / acc[4..7] = the wave's two position rows
// ({1, 2} role 0, {0, 3} role 1) of a 32-tile x 32-channel wave tile; `area` = 16 KB of LDS shared by the tile's two role waves.  Round 6:
// BOTH roles store -- role 0 computes the upper pixel row of every 2x2 tile, Y0 = s0 + (s1 + s2), role 1 the lower one, Y1 = (s1 - s2) - s3
// (s_i: column sums of position row i): role 0 sends s1 - s2 and receives s0, role 1 the reverse -- half the exchange of the earlier form,
// in which role 1 sent both of its rows and left, and twice the waves on the transposition and the store loop (same values, bit for bit).
// Every wave of the workgroup must call it (two workgroup barriers inside).
__device__ __forceinline__ void wino_roles_epilogue(const ConvParams& p, floatx16 (&acc)[8], float* area, int role, int lane, int n4, int n, int d, int sy, int sx,
                                                    int Hs, int Ws, int Yb, int Xb, const float4& bias_pre) {
  const int li = lane & 31, lh = lane >> 5;
  float* const mine = area + role * 2048;          // [r][b][lane]: what this role sends
  const float* const theirs = area + (1 - role) * 2048;
  float keep[16][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s[2][2];  // [own row 0 / 1][b]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s[h][0] = acc[h * 4 + 0][r] + acc[h * 4 + 1][r] + acc[h * 4 + 2][r];
      s[h][1] = acc[h * 4 + 1][r] - acc[h * 4 + 2][r] - acc[h * 4 + 3][r];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      // role 0 (rows 1, 2): keeps s1 + s2, sends s1 - s2; role 1 (rows 0, 3): sends s0, keeps s3
      keep[r][b] = role == 0 ? s[0][b] + s[1][b] : s[1][b];
      mine[(r * 2 + b) * 64 + lane] = role == 0 ? s[0][b] - s[1][b] : s[0][b];
    }
  }
  __syncthreads();
  float y[16][2];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float o = theirs[(r * 2 + b) * 64 + lane];
      y[r][b] = role == 0 ? o + keep[r][b] : o - keep[r][b];
    }
  __syncthreads();  // (the transposition below overwrites the exchange area)
  float* const tp = area + role * 2048;  // [q = tile * 2 + b][32 channels]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;  // tile of the wave's 4 x 8 block held by accumulator register r
#pragma unroll
    for (int b = 0; b < 2; ++b) tp[(m * 2 + b) * 32 + li] = y[r][b];
  }
  __builtin_amdgcn_wave_barrier();  // same wave: LDS serves its instructions in order, only the compiler must not reorder
  wino_store_rows(p, tp, lane, n4, n, d, sy, sx, Hs, Ws, Yb, Xb, role, bias_pre);
}

template <int WTY, int WTX, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_kernel(const ConvParams p, const int dil,
                                                                                                  const int BY, const int BX) {
  static_assert(WTY * WTX * WN == 4, "4 waves");
  typedef WinoGeom<WTY, WTX, WN> G;
  constexpr int NS = G::NS, TH = G::TH, TW = G::TW, PH = G::PH, S = G::S, HP = G::HP, CS = G::CS;
  constexpr int IN_INSTR = G::IN_INSTR, IN_BYTES = G::IN_BYTES, BN = G::BN, STAGE = G::STAGE, L = G::L, LA = G::LA;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wt = wave / WN, wty = wt / WTX, wtx = wt % WTX;

  // XCD-aware block order: consecutive blocks (which share input halos) stay on one XCD's L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // the N blocks of one tile block are consecutive workgroups of ONE XCD: they read the same input halo out of that XCD's L2
  // (as blockIdx.y they ran a whole grid apart and the halo came from HBM twice: dc_conv21 338 MB fetched for a 140 MB input)
  const int NB = (p.Cout + BN - 1) / BN;
  const int nb = bid % NB;
  bid /= NB;
  const int d = dil;
  const int bx = bid % BX;
  int rem = bid / BX;
  const int by = rem % BY;
  rem /= BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;  // this output sub-lattice's grid
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;                          // first output pixel (sub-lattice coordinates)
  // K slice of this workgroup (stages of 8 channels)
  const int nkg_all = p.Kc >> 3;
  int kg0 = 0, kg1 = nkg_all;
  if (p.ksplit > 1) {
    kg0 = (int)((long)nkg_all * blockIdx.z / p.ksplit);
    kg1 = (int)((long)nkg_all * (blockIdx.z + 1) / p.ksplit);
  }

  // ---- per-lane DMA sources (constant over the stages up to the channel offset): byte offset from p.x + an EXEC mask of the lanes
  // inside the image.  The slots of the other lanes (halo beyond the border, row padding) are never written: zeroed once, in
  // every ring buffer -- no address select per DMA, the stage's channel offset rides in the scalar base
  unsigned in_voff[IN_INSTR];
  unsigned long long in_mask[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int Lx = (i * 4 + wave) * 64 + lane;
    const int quad = Lx / (PH * S), r2 = Lx - quad * (PH * S);
    const int row = r2 / S, s = r2 - row * S;
    const int par = s / HP, cs = s - par * HP;
    const int col = 2 * cs + par;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool ok = quad < 2 && cs < CS && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    in_voff[i] = ok ? (unsigned)(((n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + quad * 4) * 4u : 0u;
    in_mask[i] = __ballot(ok);
    if (!ok) {
#pragma unroll
      for (int b = 0; b < NS; ++b) *reinterpret_cast<float4*>(smem + b * STAGE + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // weights: (position, lane half) pair `pr` is a run of BN * 16 bytes at U + ((kg * 32 + pr) * np + nb * BN) * 4 floats
  const int np = p.wino_np;
  const float4 bias_pre = wino_bias_prefetch(p, nb * BN + wn * 32 + (lane & 7) * 4);
  const float* ubase = p.wino_u + (size_t)nb * BN * 4;
  const size_t ustride = (size_t)32 * np * 4;  // floats per stage
  unsigned w_voff;                             // this lane's byte offset inside a weight DMA instruction
  if (BN == 64) w_voff = lane * 16;
  else if (BN == 32) w_voff = (unsigned)((lane >> 5) * np + (lane & 31)) * 16;
  else w_voff = lane * 16;

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // DMA instructions [i0, i1) of stage kg into ring buffer `buf` (instruction index: input first, then weights)
  auto issue = [&](int kg, int buf, int i0, int i1) {
    const unsigned sb = lds0 + buf * STAGE;
    const int c0 = kg * 8;
    const float* us = ubase + (size_t)kg * ustride;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (i < i0 || i >= i1) continue;
      if (i < IN_INSTR) {
        asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(p.x + c0),
                     "s"(sb + (i * 4 + wave) * 1024), "s"(in_mask[i])
                     : "m0");
      } else {
        const int w = (i - IN_INSTR) * 4 + wave;  // wave-instruction index of the weight stage (1 KB each)
        const float* src;
        if (BN == 64) src = us + (size_t)w * np * 4;
        else if (BN == 32) src = us + (size_t)(2 * w) * np * 4;
        else src = us + (size_t)(w >> 1) * np * 4 + (w & 1) * 256;
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_voff), "s"(src), "s"(sb + IN_BYTES + w * 1024) : "m0");
      }
    }
  };

  floatx16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  const int a_base = ((lh * PH + 2 * (wty * 4 + ty)) * S + (wtx * 8 + tx)) * 16;  // this lane's patch origin inside an input stage
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;

  float4 row[4][4];
  float4 bfr[2][4];
  auto ld_row = [&](const char* sb, int r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) row[r][c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf = [&](const char* sb, int i, int s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bfr[s][q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };

  const int nkg = kg1 - kg0;
  if (nkg > 0) {
    issue(kg0, 0, 0, L);
    if (nkg > 1) issue(kg0 + 1, 1, 0, L);
    if (nkg > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ld_row(smem, 0);
    ld_row(smem, 2);
    ld_bf(smem, 0, 0);
  }

  int buf = 0;
  for (int k = 0; k < nkg; ++k) {
    const char* sb = smem + buf * STAGE;
    const int b1 = buf + 1 == NS ? 0 : buf + 1, b2 = b1 + 1 == NS ? 0 : b1 + 1;
    const char* sbn = smem + b1 * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // ---- LDS reads of the next phase, DMA parts ----
      if (i == 0) {
        ld_row(sb, 1);
        ld_bf(sb, 1, 1);
        if (k > 0 && k + 1 < nkg) issue(kg0 + k + 1, b1, LA, L);  // part B of the stage whose part A went out in the previous phase 3
      } else if (i == 1) {
        ld_bf(sb, 2, 0);
      } else if (i == 2) {
        ld_row(sb, 3);
        ld_bf(sb, 3, 1);
      } else {
        ld_row(sbn, 0);
        ld_row(sbn, 2);
        ld_bf(sbn, 0, 0);
        if (k + 2 < nkg) issue(kg0 + k + 2, b2, 0, LA);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase i: position row i.  t = B^T d row i: d0 - d2, d1 + d2, d2 - d1, d1 - d3 ----
      constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};
      // (the phase's 32 transform additions first, then its 16 MFMAs back to back: measured 7 % faster than letting hipcc interleave
      // them -- tools/wino_bench.hip, profiles/r04_wino_proto_sched.txt)
      float vv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = comp(row[RA[i]][c], j), b = comp(row[RB[i]][c], j);
          t[c] = i == 1 ? a + b : a - b;
        }
        vv[j][0] = t[0] - t[2];
        vv[j][1] = t[1] + t[2];
        vv[j][2] = t[2] - t[1];
        vv[j][3] = t[1] - t[3];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[4 * i + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j][q], comp(bfr[i & 1][q], j), acc[4 * i + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i == 2) {  // stage k + 1 is in LDS for every wave; the buffer of stage k - 1 is free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    buf = b1;
  }

  // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1], through LDS so that a lane leaves with four channels ----
  __syncthreads();  // (every wave has read its last fragments: the stage buffers are free)
  float* xp = reinterpret_cast<float*>(smem) + wave * (128 * 32);  // [tile pixel 128][channel 32] of this wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;  // tile of the wave's 4 x 8 block held by accumulator register r
    float s[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i][0] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
      s[i][1] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      xp[(m * 4 + b) * 32 + li] = s[0][b] + s[1][b] + s[2][b];      // pixel (a = 0, b)
      xp[(m * 4 + 2 + b) * 32 + li] = s[1][b] - s[2][b] - s[3][b];  // pixel (a = 1, b)
    }
  }
  __builtin_amdgcn_wave_barrier();  // same wave: LDS serves its instructions in order, only the compiler must not reorder
  wino_store_block(p, xp, lane, nb * BN + wn * 32 + (lane & 7) * 4, n, d, sy, sx, Hs, Ws, Y0 + 8 * wty, X0 + 16 * wtx, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Eight-wave form: TWO waves per SIMD.  What one wave per SIMD pays for (tools/wino_bench.hip, profiles/r04_wino_proto_*.txt): every
// ds_read_b128 and every DMA instruction the wave issues costs the matrix pipe tens of idle cycles that its own MFMAs cannot cover,
// however the instructions are placed.  Here waves 0-3 ("role 0") own position rows 1 and 2 of their 32-tile x 32-channel block and
// waves 4-7 ("role 1") rows 0 and 3 -- 8 accumulators = 128 registers per wave, so two waves share a SIMD and one multiplies while the
// other reads.  Row i of B^T d needs patch rows {0,2} {1,2} {2,1} {1,3}: role 0 reads rows 1 and 2 only (its second phase re-uses
// the first one's registers: d2 - d1 is formed beside d1 + d2), role 1 all four.  A stage is two phases of 16 MFMAs per wave, the
// hand-over barrier between them.  The output transform adds the roles' column sums through LDS once, in the epilogue:
// Y0 = s0 + (s1 + s2), Y1 = (s1 - s2) - s3.  Measured against the four-wave form: dc_conv21 388 vs 404 us, conv2_3 164 vs 177,
// conv2_4 168 vs 188, dc_conv31 184 vs 202 (prototype, same tiles).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int WTY, int WTX, int WN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino8_kernel(const ConvParams p, const int dil,
                                                                                                   const int BY, const int BX) {
  static_assert(WTY * WTX * WN == 4, "4 wave tiles");
  typedef WinoGeom<WTY, WTX, WN> G;
  constexpr int NS = G::NS, TH = G::TH, TW = G::TW, PH = G::PH, S = G::S, HP = G::HP, CS = G::CS;
  constexpr int IN_INSTR = G::IN_INSTR, IN_BYTES = G::IN_BYTES, BN = G::BN, STAGE = G::STAGE;
  constexpr int NW = G::W_BYTES / 1024;                     // weight DMA wave-instructions per stage
  constexpr int PER = (IN_INSTR * 4 + NW + 7) / 8;          // DMA instructions per wave aimed at
  constexpr int W0 = PER > IN_INSTR ? PER - IN_INSTR : 0;   // weight instructions of a role-0 wave (beside its IN_INSTR input ones)
  constexpr int W1 = (NW - 4 * W0) / 4;                     // ... of a role-1 wave
  static_assert(W1 >= 0 && 4 * W0 + 4 * W1 == NW, "weight split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WINO_STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, sub = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int wn = sub % WN, wt = sub / WN, wty = wt / WTX, wtx = wt % WTX;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // the N blocks of one tile block are consecutive workgroups of ONE XCD: they read the same input halo out of that XCD's L2
  // (as blockIdx.y they ran a whole grid apart and the halo came from HBM twice: dc_conv21 338 MB fetched for a 140 MB input)
  const int NB = (p.Cout + BN - 1) / BN;
  const int nb = bid % NB;
  bid /= NB;
  const int d = dil;
  const int bx = bid % BX;
  int rem = bid / BX;
  const int by = rem % BY;
  rem /= BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;
  const int nkg_all = p.Kc >> 3;
  int kg0 = 0, kg1 = nkg_all;
  if (p.ksplit > 1) {
    kg0 = (int)((long)nkg_all * blockIdx.z / p.ksplit);
    kg1 = (int)((long)nkg_all * (blockIdx.z + 1) / p.ksplit);
  }
  const int nkg = kg1 - kg0;

  // input DMA of the role-0 waves: per-lane byte offsets + EXEC masks, halo / pad slots zeroed once (see conv_wino_kernel)
  unsigned in_voff[IN_INSTR];
  unsigned long long in_mask[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int Lx = (i * 4 + sub) * 64 + lane;
    const int quad = Lx / (PH * S), r2 = Lx - quad * (PH * S);
    const int row = r2 / S, s = r2 - row * S;
    const int par = s / HP, cs = s - par * HP;
    const int col = 2 * cs + par;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool ok = quad < 2 && cs < CS && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    in_voff[i] = ok ? (unsigned)(((n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + quad * 4) * 4u : 0u;
    in_mask[i] = __ballot(ok);
    if (!ok && role == 0) {
#pragma unroll
      for (int b = 0; b < NS; ++b) *reinterpret_cast<float4*>(smem + b * STAGE + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  WINO_STAMP(1);
  const int np = p.wino_np;
  const float4 bias_pre = wino_bias_prefetch(p, nb * BN + wn * 32 + (lane & 7) * 4);
  const float* ubase = p.wino_u + (size_t)nb * BN * 4;
  const size_t ustride = (size_t)32 * np * 4;  // floats per stage
  const unsigned w_voff = BN == 32 ? (unsigned)((lane >> 5) * np + (lane & 31)) * 16 : (unsigned)lane * 16;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_in = [&](int kg, int buf, int i) {
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(p.x + kg * 8),
                 "s"(lds0 + buf * STAGE + (i * 4 + sub) * 1024), "s"(in_mask[i])
                 : "m0");
  };
  auto dma_w = [&](int kg, int buf, int w) {  // w: wave-instruction index of the weight stage (1 KB each)
    const float* us = ubase + (size_t)kg * ustride + (BN == 64 ? (size_t)w * np * 4 : (size_t)(2 * w) * np * 4);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_voff), "s"(us), "s"(lds0 + buf * STAGE + IN_BYTES + w * 1024) : "m0");
  };
  // this wave's DMA instructions [i0, i1) of a stage (role 0: input first, then its weights; role 1: weights)
  auto dma_part = [&](auto ROLE, int kg, int buf, int i0, int i1) {
    constexpr int R = decltype(ROLE)::value;
    constexpr int LR = R == 0 ? IN_INSTR + W0 : W1;
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      if (i < i0 || i >= i1) continue;
      if (R == 0) {
        if (i < IN_INSTR) dma_in(kg, buf, i);
        else dma_w(kg, buf, (i - IN_INSTR) * 4 + sub);
      } else {
        dma_w(kg, buf, 4 * W0 + i * 4 + sub);
      }
    }
  };

  floatx16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  const int a_base = ((lh * PH + 2 * (wty * 4 + ty)) * S + (wtx * 8 + tx)) * 16;
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;
  auto ld_row = [&](const char* sb, int r, float4 (&dst)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf = [&](const char* sb, int i, float4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };

  auto body = [&](auto ROLE) {
    constexpr int R = decltype(ROLE)::value;
    constexpr int LR = R == 0 ? IN_INSTR + W0 : W1;
    constexpr int LA = (LR + 1) / 2;
    constexpr int I0 = R == 0 ? 1 : 0, I1 = R == 0 ? 2 : 3;  // position rows of phase 0 / phase 1
    float4 ra[4], rb[4];  // role 0: patch rows 1, 2; role 1: rows 0, 2 (phase 0)
    float4 rc[4], rd[4];  // role 1: rows 1, 3 (phase 1)
    float4 bf0[4], bf1[4];
    float v1[4][4];       // role 0: phase 1's operands, formed during phase 0 from the same two rows
#pragma unroll
    for (int c = 0; c < 4; ++c) ra[c] = rb[c] = rc[c] = rd[c] = bf0[c] = bf1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nkg > 0) dma_part(ROLE, kg0, 0, 0, LR);
    if (nkg > 1) dma_part(ROLE, kg0 + 1, 1, 0, LR);
    if (nkg > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WINO_STAMP(2);
    ld_row(smem, R == 0 ? 1 : 0, ra);
    ld_row(smem, 2, rb);
    ld_bf(smem, I0, bf0);
    int buf = 0;
    for (int k = 0; k < nkg; ++k) {
      const char* sb = smem + buf * STAGE;
      const int b1 = buf + 1 == NS ? 0 : buf + 1, b2 = b1 + 1 == NS ? 0 : b1 + 1;
      const char* sbn = smem + b1 * STAGE;
      // ---------------- phase 0 ----------------
      if (R == 1) { ld_row(sb, 1, rc); ld_row(sb, 3, rd); }
      ld_bf(sb, I1, bf1);
      if (k > 0 && k + 1 < nkg) dma_part(ROLE, kg0 + k + 1, b1, LA, LR);  // (part A of that stage went out in the previous phase 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t[4], v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = comp(ra[c], j), b = comp(rb[c], j);
          t[c] = R == 0 ? a + b : a - b;  // row 1: d1 + d2; row 0: d0 - d2
        }
        v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        if (R == 0) {  // row 2: d2 - d1, kept for phase 1
          float u[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) u[c] = comp(rb[c], j) - comp(ra[c], j);
          v1[j][0] = u[0] - u[2]; v1[j][1] = u[1] + u[2]; v1[j][2] = u[2] - u[1]; v1[j][3] = u[1] - u[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], comp(bf0[q], j), acc[q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage k + 1 is in LDS for every wave; the buffer of stage k - 1 is free
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // ---------------- phase 1 ----------------
      ld_row(sbn, R == 0 ? 1 : 0, ra);
      ld_row(sbn, 2, rb);
      ld_bf(sbn, I0, bf0);
      if (k + 2 < nkg) dma_part(ROLE, kg0 + k + 2, b2, 0, LA);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
        if (R == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v1[j][q];
        } else {
          float t[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) t[c] = comp(rc[c], j) - comp(rd[c], j);  // row 3: d1 - d3
          v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], comp(bf1[q], j), acc[4 + q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      buf = b1;
    }
  };
  if (role == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());

  // ---- output transform + store: both roles (wino_roles_epilogue) ----
  WINO_STAMP(3);
  __syncthreads();  // (every wave has read its last fragments: the stage buffers are free)
  wino_roles_epilogue(p, acc, reinterpret_cast<float*>(smem) + sub * 4096, role, lane, nb * BN + wn * 32 + (lane & 7) * 4, n, d, sy, sx, Hs, Ws, Y0 + 8 * wty,
                      X0 + 16 * wtx, bias_pre);
#ifdef UDET_EXPERIMENT
  WINO_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WINO_STAMP(6);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Half-size form (round 6): 32 tiles x 64 channels per workgroup, FOUR waves = the two roles of the eight-wave form x two channel
// halves, a TWO-buffer LDS ring of 39 KB stages -- 78 KB per workgroup, so that TWO workgroups share a CU (two waves per SIMD again,
// now of different workgroups at different points of their stages).  Written for the launches the other forms leave half the chip
// idle on: the generator's 128-channel layers at 48 x 96 (models/nets.py:23-31) are 72 blocks of 64 tiles x 2 channel blocks = 144
// workgroups on 256 CUs, each alone on its CU at ~0.6 of the CU's matrix rate; as 288 half-size workgroups every CU has work and
// the 32 CUs that hold two run them interleaved.  One barrier per stage: stage k + 1's DMA goes out when the barrier that ended
// stage k - 1 has released its buffer, and has the whole of stage k (32 MFMAs per wave) to land.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WinoHalfGeom {
  static constexpr int NS = 2, TH = 4, TW = 8;
  static constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;  // 10 x 18 halo pixels
  static constexpr int CS = PW / 2, S = wino_row_slots(PW), HP = S / 2;
  static constexpr int IN_SLOTS = 2 * PH * S;              // 400
  static constexpr int IN_WINSTR = (IN_SLOTS + 63) / 64;   // 7 wave-instructions of 64 slots
  static constexpr int IN_INSTR = (IN_WINSTR + 3) / 4;     // per wave (the last round is partly filled)
  static constexpr int IN_BYTES = IN_WINSTR * 1024;
  static constexpr int BN = 64;
  static constexpr int W_BYTES = 16 * 2 * BN * 16;         // 32 KB
  static constexpr int NW = W_BYTES / 1024;                // 32 weight wave-instructions per stage, 8 per wave
  static constexpr int STAGE = IN_BYTES + W_BYTES;         // 39 KB
  static constexpr int LDS_BYTES = NS * STAGE;             // 78 KB: two workgroups per CU
  static_assert(LDS_BYTES >= 4 * 128 * 32 * 4 && 2 * LDS_BYTES <= 160 * 1024, "epilogue scratch / two workgroups per CU");
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino_half_kernel(const ConvParams p, const int dil, const int BY,
                                                                                                       const int BX) {
  typedef WinoHalfGeom G;
  constexpr int TH = G::TH, TW = G::TW, PH = G::PH, S = G::S, HP = G::HP, CS = G::CS;
  constexpr int IN_INSTR = G::IN_INSTR, IN_BYTES = G::IN_BYTES, BN = G::BN, STAGE = G::STAGE, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int NB = (p.Cout + BN - 1) / BN;
  const int nb = bid % NB;
  bid /= NB;
  const int d = dil;
  const int bx = bid % BX;
  int rem = bid / BX;
  const int by = rem % BY;
  rem /= BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;
  const int nkg_all = p.Kc >> 3;
  int kg0 = 0, kg1 = nkg_all;
  if (p.ksplit > 1) {
    kg0 = (int)((long)nkg_all * blockIdx.z / p.ksplit);
    kg1 = (int)((long)nkg_all * (blockIdx.z + 1) / p.ksplit);
  }
  const int nkg = kg1 - kg0;

  // input DMA: wave-instruction j = i * 4 + wave covers slots [64 j, 64 j + 64); byte offsets + EXEC masks as in the other forms
  unsigned in_voff[IN_INSTR];
  unsigned long long in_mask[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int j = i * 4 + wave;
    const int Lx = j * 64 + lane;
    const int quad = Lx / (PH * S), r2 = Lx - quad * (PH * S);
    const int row = r2 / S, s = r2 - row * S;
    const int par = s / HP, cs = s - par * HP;
    const int col = 2 * cs + par;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool in_stage = j < G::IN_WINSTR;
    const bool ok = in_stage && quad < 2 && cs < CS && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    in_voff[i] = ok ? (unsigned)(((n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + quad * 4) * 4u : 0u;
    in_mask[i] = __ballot(ok);
    if (!ok && in_stage) {
#pragma unroll
      for (int b = 0; b < G::NS; ++b) *reinterpret_cast<float4*>(smem + b * STAGE + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const int np = p.wino_np;
  const float4 bias_pre = wino_bias_prefetch(p, nb * BN + wn * 32 + (lane & 7) * 4);
  const float* ubase = p.wino_u + (size_t)nb * BN * 4;
  const size_t ustride = (size_t)32 * np * 4;  // floats per stage
  const unsigned w_voff = (unsigned)lane * 16;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_stage = [&](int kg, int buf) {  // this wave's share of a stage: its input rounds, then 8 of the 32 weight instructions
#pragma unroll
    for (int i = 0; i < IN_INSTR; ++i) {
      if (i * 4 + wave < G::IN_WINSTR)
        asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(p.x + kg * 8),
                     "s"(lds0 + buf * STAGE + (i * 4 + wave) * 1024), "s"(in_mask[i])
                     : "m0");
    }
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
      const int w = i * 4 + wave;
      const float* us = ubase + (size_t)kg * ustride + (size_t)w * np * 4;
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_voff), "s"(us), "s"(lds0 + buf * STAGE + IN_BYTES + w * 1024) : "m0");
    }
  };

  floatx16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  const int a_base = ((lh * PH + 2 * ty) * S + tx) * 16;
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;
  auto ld_row = [&](const char* sb, int r, float4 (&dst)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf = [&](const char* sb, int i, float4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };

  // (both role instantiations execute the same number of barriers: the trip count nkg is workgroup-uniform -- see the barrier contract
  // note in conv_wgrad_wino.hip)
  auto body = [&](auto ROLE) {
    constexpr int R = decltype(ROLE)::value;
    constexpr int I0 = R == 0 ? 1 : 0, I1 = R == 0 ? 2 : 3;  // position rows of phase 0 / phase 1
    float4 ra[4], rb[4], rc[4], rd[4], bf0[4], bf1[4];
    float v1[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ra[c] = rb[c] = rc[c] = rd[c] = bf0[c] = bf1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nkg > 0) dma_stage(kg0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int k = 0; k < nkg; ++k) {
      const char* sb = smem + (k & 1) * STAGE;
      if (k + 1 < nkg) dma_stage(kg0 + k + 1, (k + 1) & 1);  // (that buffer held stage k - 1: released by the barrier below)
      ld_row(sb, R == 0 ? 1 : 0, ra);
      ld_row(sb, 2, rb);
      ld_bf(sb, I0, bf0);
      if (R == 1) { ld_row(sb, 1, rc); ld_row(sb, 3, rd); }
      ld_bf(sb, I1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 0 ----------------
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t[4], v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = comp(ra[c], j), b = comp(rb[c], j);
          t[c] = R == 0 ? a + b : a - b;  // row 1: d1 + d2; row 0: d0 - d2
        }
        v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        if (R == 0) {  // row 2: d2 - d1, kept for phase 1
          float u[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) u[c] = comp(rb[c], j) - comp(ra[c], j);
          v1[j][0] = u[0] - u[2]; v1[j][1] = u[1] + u[2]; v1[j][2] = u[2] - u[1]; v1[j][3] = u[1] - u[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], comp(bf0[q], j), acc[q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 1 ----------------
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
        if (R == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v1[j][q];
        } else {
          float t[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) t[c] = comp(rc[c], j) - comp(rd[c], j);  // row 3: d1 - d3
          v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], comp(bf1[q], j), acc[4 + q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage k + 1 has landed (this wave's share); every wave is done reading stage k
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  if (role == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());

  // ---- output transform + store: both roles (wino_roles_epilogue), one 16 KB area per channel half ----
  wino_roles_epilogue(p, acc, reinterpret_cast<float*>(smem) + wn * 4096, role, lane, nb * BN + wn * 32 + (lane & 7) * 4, n, d, sy, sx, Hs, Ws, Y0, X0, bias_pre);
}

// ---- weight transform: U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] -------------------------------------------------------
__global__ __launch_bounds__(256) void wino_pack_kernel(const PackJob j, const float* __restrict__ src, float* __restrict__ dst) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < j.total; e += (long)gridDim.x * 256) wino_pack_item(j, src, nullptr, 1.f, dst, e);
}
int launch_wino_pack(const float* src, float* dst, int R, int C, int Kc, int np, int k_split, int k_gap, int transposed, hipStream_t stream) {
  PackJob j;
  memset(&j, 0, sizeof(j));
  j.T = 9; j.R = R; j.C = C; j.Kc = Kc; j.ldw = np; j.k_split = k_split; j.k_gap = k_gap;
  j.mode = transposed ? 8 : 7; j.total = (long)(Kc / 8) * 2 * np; j.gamma_off = -1;  // work items: (channel group, lane half, column)
  int nb = (int)((j.total + 255) / 256);
  if (nb > 2048) nb = 2048;
  UDET_LAUNCH(wino_pack_kernel, dim3(nb), dim3(256), 0, stream, j, src, dst);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
// from PACKED weights [tap][Kc][ldw] + the launch's tap table: widx_at[a * 3 + b] is the packed tap at offset ((a-1) d, (b-1) d)
struct WinoTapMap { int w[9]; };
__global__ __launch_bounds__(256) void wino_from_packed_kernel(const float* __restrict__ wp, int Kc, int ldw, WinoTapMap tm, float* __restrict__ dst,
                                                               int np, long total) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int jj = (int)(e & 3);
    const long r0 = e >> 2;
    const int n = (int)(r0 % np);
    const long r1 = r0 / np;
    const int kh = (int)(r1 & 1), pos = (int)((r1 >> 1) & 15), kg = (int)(r1 >> 5);
    const int k = kg * 8 + kh * 4 + jj;
    float val = 0.f;
    if (n < ldw && k < Kc) {
      const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
      const int pi = pos >> 2, pj = pos & 3;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) val += Gm[pi][a] * Gm[pj][b] * wp[((long)tm.w[a * 3 + b] * Kc + k) * ldw + n];
    }
    dst[e] = val;
  }
}
int launch_wino_from_packed(const ConvParams& p, float* dst, int np, hipStream_t stream) {
  int d = 0;
  WinoTapMap tm;
  if (!conv_wino_geometry(p, &d, tm.w)) {
    set_error("wino: the launch is not a 3x3 stride-1 convolution");
    return UDET_ERR_UNSUPPORTED;
  }
  const long total = (long)(p.Kc / 8) * 16 * 2 * np * 4;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  UDET_LAUNCH(wino_from_packed_kernel, dim3(nb), dim3(256), 0, stream, p.wp, p.Kc, p.ldw, tm, dst, np, total);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
int conv_wino_np(int cout) { return cout > 64 ? round_up(cout, 128) : (cout > 32 ? 64 : 32); }
size_t conv_wino_floats(int Kc, int cout) { return (size_t)(Kc / 8) * 16 * 2 * conv_wino_np(cout) * 4; }

// the launch's taps are the 3x3 grid {-d, 0, d}^2 of a stride-1 convolution onto the input grid
bool conv_wino_geometry(const ConvParams& p, int* dil, int widx_at[9]) {
  if (p.ntaps != 9 || p.ncls == 4 || p.up_shift != 0) return false;
  if (p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0) return false;
  if (p.OH != p.H || p.OW != p.W || p.OHq != p.OH || p.OWq != p.OW) return false;
  int d = 0;
  for (int t = 0; t < 9; ++t) {
    const int a = p.taps[t].dy < 0 ? -p.taps[t].dy : p.taps[t].dy;
    if (a > d) d = a;
  }
  if (d < 1) return false;
  for (int i = 0; i < 9; ++i) widx_at[i] = -1;
  for (int t = 0; t < 9; ++t) {
    const int dy = p.taps[t].dy, dx = p.taps[t].dx;
    if (dy % d != 0 || dx % d != 0) return false;
    const int a = dy / d + 1, b = dx / d + 1;
    if (a < 0 || a > 2 || b < 0 || b > 2 || widx_at[a * 3 + b] >= 0) return false;
    widx_at[a * 3 + b] = p.taps[t].widx;
  }
  *dil = d;
  return true;
}
bool conv_wino_ok(const ConvParams& p) {
  if (p.nseg) return false;  // segmented launches: implicit-GEMM families only
  int d, w[9];
  if (p.f16 || p.xa != nullptr || p.wino_u == nullptr || p.zero16 == nullptr) return false;
  if ((reinterpret_cast<uintptr_t>(p.zero16) | reinterpret_cast<uintptr_t>(p.wino_u) | reinterpret_cast<uintptr_t>(p.x)) & 15) return false;
  if (p.Kc < 8 || p.Kc % 8 != 0 || p.ldx % 4 != 0 || p.x_coff % 4 != 0) return false;
  if (p.wino_np < 32 || p.wino_np % 32 != 0 || p.wino_np < p.Cout) return false;
  // 32-bit element offsets on the output side, 32-bit BYTE offsets (the saddr form of the LDS-DMA) on the input side
  if ((long)p.N * p.H * p.W * (long)p.ldy >= (1L << 31) || (long)p.N * p.H * p.W * (long)p.ldx >= (1L << 30)) return false;
  return conv_wino_geometry(p, &d, w);
}
static int variant_bn(int v) { return v == 4 ? 64 : ((v & 1) == 0 ? 64 : 32); }  // bit 0: tile shape, bit 1: the eight-wave form; 4: the half-size form
bool conv_wino_variant_ok(const ConvParams& p, int v) {
  if (v < 0 || v > 4) return false;
  const int bn = variant_bn(v);
  if (p.wino_np % bn != 0) return false;
  return p.Cout > bn / 2 || bn == 32;  // (a block twice as wide as the layer only multiplies zeros)
}
static void variant_blocks(const ConvParams& p, int v, int d, int* BY, int* BX) {
  const int th = v == 4 ? 4 : 8, tw = (v & 1) ? 16 : 8;  // tiles per workgroup
  const int Hs = (p.H + d - 1) / d, Ws = (p.W + d - 1) / d;
  *BY = (Hs + 2 * th - 1) / (2 * th);
  *BX = (Ws + 2 * tw - 1) / (2 * tw);
}
long conv_wino_workgroups(const ConvParams& p, int v) {
  int d, w[9], BY, BX;
  if (!conv_wino_geometry(p, &d, w)) return 0;
  variant_blocks(p, v, d, &BY, &BX);
  return (long)p.N * d * d * BY * BX * ((p.Cout + variant_bn(v) - 1) / variant_bn(v));
}
int conv_wino_max_ksplit(const ConvParams& p, int v) {
  (void)v;
  if (!p.partial) return 1;
  int ks = (p.Kc / 8) / 6;  // >= 6 stages per slice
  if (ks > 16) ks = 16;
  const size_t per_split = (size_t)p.N * p.H * p.W * ((p.Cout + 3) & ~3);
  while (ks > 1 && per_split * ks > p.partial_cap) --ks;
  return ks < 1 ? 1 : ks;
}
int launch_splitk_second_pass(const ConvParams& p, hipStream_t stream);

template <int WTY, int WTX, int WN, bool EIGHT>
static int launch_variant(const ConvParams& p, int d, int BY, int BX, hipStream_t stream) {
  typedef WinoGeom<WTY, WTX, WN> G;
  auto kern = EIGHT ? conv_wino8_kernel<WTY, WTX, WN> : conv_wino_kernel<WTY, WTX, WN>;
  static std::once_flag once;
  static hipError_t attr = hipSuccess;
  std::call_once(once, [&]() { attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); });
  UDET_HIP(attr);
  dim3 grid(p.N * d * d * BY * BX * ((p.Cout + G::BN - 1) / G::BN), 1, p.ksplit > 1 ? p.ksplit : 1);
  UDET_LAUNCH(kern, grid, dim3(EIGHT ? 512 : 256), G::LDS_BYTES, stream, p, d, BY, BX);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
static int launch_half(const ConvParams& p, int d, int BY, int BX, hipStream_t stream) {
  typedef WinoHalfGeom G;
  static std::once_flag once;
  static hipError_t attr = hipSuccess;
  std::call_once(once, [&]() { attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_half_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); });
  UDET_HIP(attr);
  dim3 grid(p.N * d * d * BY * BX * ((p.Cout + G::BN - 1) / G::BN), 1, p.ksplit > 1 ? p.ksplit : 1);
  UDET_LAUNCH(conv_wino_half_kernel, grid, dim3(256), G::LDS_BYTES, stream, p, d, BY, BX);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}
int launch_conv_wino(ConvParams& p, int variant, int ks, hipStream_t stream) {
  int d, w[9], BY, BX;
  if (!conv_wino_ok(p) || !conv_wino_geometry(p, &d, w) || !conv_wino_variant_ok(p, variant)) {
    set_error("wino: launch not eligible (variant %d)", variant);
    return UDET_ERR_UNSUPPORTED;
  }
  variant_blocks(p, variant, d, &BY, &BX);
  const int cap = conv_wino_max_ksplit(p, variant);
  p.ksplit = ks > cap ? cap : (ks < 1 ? 1 : ks);
  p.fold = 0; p.tail_full = 0; p.tail_ks = 0; p.tail_prow0 = 0;
  p.ncls = 1;
  if (p.ksplit > 1) p.ldp = (p.Cout + 3) & ~3;
  int rc;
  if (variant == 0) rc = launch_variant<2, 1, 2, false>(p, d, BY, BX, stream);
  else if (variant == 1) rc = launch_variant<2, 2, 1, false>(p, d, BY, BX, stream);
  else if (variant == 2) rc = launch_variant<2, 1, 2, true>(p, d, BY, BX, stream);
  else if (variant == 4) rc = launch_half(p, d, BY, BX, stream);
  else rc = launch_variant<2, 2, 1, true>(p, d, BY, BX, stream);
  if (rc != UDET_OK) return rc;
  if (p.ksplit > 1) return launch_splitk_second_pass(p, stream);
  return UDET_OK;
}

}  // namespace udet
