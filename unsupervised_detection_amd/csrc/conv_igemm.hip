// Implicit-GEMM convolution on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   M = output pixels (n,qy,qx)   N = output channels   K = (tap, input channel) flattened
//
// 256-thread workgroups (4 wave64), BM x BN output tile.  K is ONE flat axis over the launch's tap list
// (k = tap*Kc + c): a BK-wide LDS stage may straddle taps, so 7x7/5x5 convolutions over 4..16 channels and
// the ragged PWC slab windows run the same BK=32 pipeline as the 128-channel 3x3 layers; the tail of the
// last stage is zero-filled.  Double-buffered LDS, register-staged global->LDS copies (one barrier per stage).
// The A tile is stored K-major in LDS ([BK][BM+pad], pad chosen so that the transposing ds_write_b32 are
// conflict-free) which makes the MFMA A fragment (lane l -> row l&31, k = l>>5) a conflict-free ds_read_b32;
// the B tile ([BK][BN]) is the packed-weight layout itself.
// Stride-2 backward-data / conv2d_transpose: the four output-parity classes are ONE launch (each workgroup
// belongs to one class and walks only that class's taps: no multiplications by structural zeros).
//
// Replaces the TF-1.13 Conv2D / Conv2DBackpropInput kernels the reference calls through
// tf.layers.conv2d / tf.nn.conv2d / tf.layers.conv2d_transpose
// (models/utils/convolution_utils.py:46,81; models/PWCNet/model_pwcnet.py:161-165,286,484-504,562-574).
#include <stdlib.h>

#include <functional>
#include <type_traits>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_host.h"
#include <algorithm>

namespace udet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

// libudet_exp.so only (tools/igemm_stamps.py): per-workgroup cycle stamps of the LDS-DMA kernel -- 0 entry, 1 tables done, 2 first stage
// landed, 3 K loop done, 4 tile stored (issued), 5 stores acknowledged, 6 / 7 block decoded / tables written, 8 / 9 inside the tile store
// (its set-up done / first half block issued; IGEMM_STAMP_B: the x-block index is blockIdx.x -- single launches only)
#ifdef UDET_EXPERIMENT
#define IGEMM_TS 12
__device__ long long g_igemm_ts[1024 * IGEMM_TS];
#define IGEMM_STAMP_AT(b, i)                                                                                                            \
  do {                                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (b) < 1024) g_igemm_ts[(b) * IGEMM_TS + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#define IGEMM_STAMP(i) IGEMM_STAMP_AT(bid_x, i)
#define IGEMM_STAMP_B(i) IGEMM_STAMP_AT((int)blockIdx.x, i)
extern "C" int udet_exp_igemm_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_igemm_ts), (size_t)(n < 1024 * IGEMM_TS ? n : 1024 * IGEMM_TS) * sizeof(long long));
}
#else
#define IGEMM_STAMP(i) do {} while (0)
#define IGEMM_STAMP_B(i) do {} while (0)
#endif

// Flat-K cursor.  K runs channel-block-major: for every block of CB = min(Kc,32) input channels all taps of the launch,
// then the next channel block (the last block may be narrower).  A workgroup therefore re-visits its ~3 input rows
// for all taps of one channel block while they are still in L1/L2, instead of streaming the whole channel depth once
// per tap (9x the algorithmic read traffic out of L2 for a 3x3 layer over 568 channels).
struct KCursor {
  int blk, tap, c, w;
};
struct KOrder {
  int Kc, CB, nblk, wl, ntc;
};
__device__ __forceinline__ KOrder korder(int Kc, int ntc) {
  KOrder o;
  o.Kc = Kc; o.ntc = ntc;
  o.CB = Kc < 32 ? Kc : 32;
  o.nblk = Kc < 32 ? 1 : (Kc + 31) >> 5;  // (= ceil(Kc / CB) without a run-time division)
  o.wl = Kc - (o.nblk - 1) * o.CB;
  return o;
}
__device__ __forceinline__ KCursor kc_init(const KOrder& o, int kf) {
  KCursor k;
  const int per = o.ntc * o.CB;
  int blk = per > 0 ? kf / per : o.nblk;
  if (blk >= o.nblk - 1) {
    const int rem = kf - (o.nblk - 1) * per;
    k.blk = o.nblk - 1; k.w = o.wl;
    k.tap = rem / o.wl; k.c = rem - k.tap * o.wl;
    if (k.tap >= o.ntc) { k.blk = o.nblk; k.tap = 0; }
  } else {
    const int rem = kf - blk * per;
    k.blk = blk; k.w = o.CB;
    k.tap = rem / o.CB; k.c = rem - k.tap * o.CB;
  }
  return k;
}
__device__ __forceinline__ void kc_advance(const KOrder& o, KCursor& k, int step) {
  if (k.w == step) {  // common case (32-channel block, 32-wide stage): same channel offset, next tap
    if (++k.tap == o.ntc) {
      k.tap = 0;
      ++k.blk;
      k.w = k.blk == o.nblk - 1 ? o.wl : o.CB;
    }
  } else {
    k.c += step;
  }
  while (k.c >= k.w) {  // (also re-normalises the offset after stepping into the narrower last block)
    k.c -= k.w;
    if (++k.tap == o.ntc) {
      k.tap = 0;
      ++k.blk;
      k.w = k.blk == o.nblk - 1 ? o.wl : o.CB;
    }
  }
}
__device__ __forceinline__ bool kc_valid(const KOrder& o, const KCursor& k) { return k.blk < o.nblk; }
__device__ __forceinline__ int kc_chan(const KOrder& o, const KCursor& k) { return k.blk * o.CB + k.c; }

// per-wave LDS scratch of the transposing store below: 16 rows x UDET_XP floats, carved out of the (now idle) stage buffers
#define UDET_XP 40
template <size_t SA, size_t SB>
__device__ __forceinline__ float* xpose_scratch(float* a, float* b, int wave) {
  constexpr size_t W = 16 * UDET_XP * sizeof(float);
  if constexpr (SA >= 4 * W) return a + wave * 16 * UDET_XP;
  else if constexpr (SB >= 4 * W) return b + wave * 16 * UDET_XP;
  else if constexpr (SA >= 2 * W && SB >= 2 * W) return (wave < 2 ? a : b) + (wave & 1) * 16 * UDET_XP;
  else return nullptr;
}

// The class (or segment, ConvParams::nseg) an x-block belongs to and that block's place in it
struct TileCls {
  int cls, m0, Mtot, OHWq, OWq, ooy, oox, tap0, ntc, prow0;
  FastDiv fd_ohw, fd_ow;
};
template <int BM>
__device__ __forceinline__ TileCls tile_cls(const ConvParams& p, int bid) {
  TileCls t;
  if (p.nseg == 0) {
    t.OHWq = p.OHq * p.OWq; t.OWq = p.OWq;
    t.Mtot = p.N * t.OHWq;
    const int mtiles = (t.Mtot + BM - 1) / BM;
    t.cls = p.ncls > 1 ? bid / mtiles : 0;  // (one class: no run-time division in front of every launch's first instruction of work)
    t.m0 = (bid - t.cls * mtiles) * BM;
    t.tap0 = p.cls_tap[t.cls];
    t.ntc = p.cls_tap[t.cls + 1] - t.tap0;
    t.ooy = p.ncls > 1 ? (t.cls >> 1) : p.ooy; t.oox = p.ncls > 1 ? (t.cls & 1) : p.oox;
    t.prow0 = t.cls * t.Mtot;
    t.fd_ohw = p.fd_ohw; t.fd_ow = p.fd_ow;
    return t;
  }
  int s = 0, b = bid;
  for (; s < p.nseg - 1; ++s) {
    const int mt = (p.N * p.seg[s].h * p.seg[s].w + BM - 1) / BM;
    if (b < mt) break;
    b -= mt;
  }
  const ConvSeg& g = p.seg[s];
  t.cls = s;
  t.OHWq = g.h * g.w; t.OWq = g.w;
  t.Mtot = p.N * t.OHWq;
  t.m0 = b * BM;
  t.tap0 = p.seg_tap[s];
  t.ntc = p.seg_tap[s + 1] - t.tap0;
  t.ooy = g.oy; t.oox = g.ox;
  t.prow0 = g.prow0;
  t.fd_ohw = g.fd_hw; t.fd_ow = g.fd_w;
  return t;
}
__device__ __forceinline__ ConvTap conv_tap(const ConvParams& p, int i) { return p.nseg ? p.tap_tab[i] : p.taps[i]; }
// output pixel offset of row `ma` of the launch's flat row space (split-K second pass)
__device__ __forceinline__ int row_pixel_off(const ConvParams& p, int ma) {
  int m, OHWq, OWq, ooy, oox;
  FastDiv fa, fb;
  if (p.nseg == 0) {
    OHWq = p.OHq * p.OWq; OWq = p.OWq;
    const int Mtot = p.N * OHWq, cls = ma / Mtot;
    m = ma - cls * Mtot;
    ooy = p.ncls > 1 ? (cls >> 1) : p.ooy; oox = p.ncls > 1 ? (cls & 1) : p.oox;
    fa = p.fd_ohw; fb = p.fd_ow;
  } else {
    int s = 0;
    while (s < p.nseg - 1 && ma >= p.seg[s + 1].prow0) ++s;
    const ConvSeg& g = p.seg[s];
    m = ma - g.prow0;
    OHWq = g.h * g.w; OWq = g.w; ooy = g.oy; oox = g.ox;
    fa = g.fd_hw; fb = g.fd_w;
  }
  const int nb = (int)fdiv(m, fa), rem = m - nb * OHWq;
  const int qy = (int)fdiv(rem, fb), qx = rem - qy * OWq;
  return (nb * p.OH + qy * p.osy + ooy) * p.OW + qx * p.osx + oox;
}

// Result of one workgroup: plain launches run the epilogue; split-K launches store the partial tile into slab blockIdx.z
// (row index = parity class * Mtot + pixel).
// The MFMA accumulator holds COLUMN n = lane of 8+8 rows, so a direct store is one dword per lane and (bias, activation, 64-bit
// address, flag tests) once per element -- ~13,000 instructions for a 128x128 tile, more than the instruction cache holds, and
// ~15 % of the run time of a mid-size layer.  With `xp` (16 x UDET_XP floats of LDS per wave) the tile goes through LDS half a
// 32x32 block at a time and leaves as float4 rows: 8 lanes x 16 B per pixel, epilogue arithmetic once per quad, and the
// store loop is not unrolled (4 x 2 copies of its body instead of 256).
// the float4 path of igemm_store for one (activation, operand) variant: a plain function template, NOT a lambda inside igemm_store --
// with a generic lambda instantiated four ways hipcc copied the whole 1752-byte kernel-argument block to scratch in every kernel with a
// tile larger than 64 x 64 (1760 bytes of scratch per lane; round 6)
template <int TM, int TN, int WTM, int WTN, bool ELU, bool PLAIN>
__device__ __forceinline__ void igemm_store_quads(const ConvParams& p, floatx16 (&acc)[TM][TN], const int* rowoff, int wm, int wn, int li, int lh, int n0,
                                                  int prow0, bool slab, long slab_off, float* xp, const float4* bias_pre) {
  const int lane = lh * 32 + li, rr = lane >> 3, c4 = (lane & 7) * 4;
  // the bias quad of a column block does not depend on the row: loaded once per block, not once per quad behind the previous quad's
  // store; the two passes of a half block request their per-pixel operands together; the activation is selected once, by the caller
  // (conv_epilogue.h: epi4_*, EpiAct -- per element it cost five scalar branches).  PLAIN: a launch with neither residual nor accumulate
  // nor dU emission -- or a K slice writing its slab -- has NO global load in its store loop; with one, every wait for it also drains the
  // stores issued before it (loads and stores share vmcnt on gfx950)
  float4 bias[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + wn * WTN + j * 32 + c4;
    if (bias_pre) bias[j] = bias_pre[j];  // (requested in front of the K loop: igemm_bias_prefetch)
    else bias[j] = (!slab && nb < p.Cout) ? epi4_bias(p, nb) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const EpiAct ea = epi_act(p);
  IGEMM_STAMP_B(8);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + wn * WTN + j * 32 + c4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // accumulator registers 8h .. 8h+7 are rows 16h .. 16h+15 of the block
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) xp[((r & 3) + 8 * (r >> 2) + 4 * lh) * UDET_XP + li] = acc[i][j][8 * h + r];
        __builtin_amdgcn_wave_barrier();  // same wave: LDS serves its instructions in order, only the compiler must not reorder
        int off[2];
        float4 v[2];
        Epi4Req rq[2];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = wm * WTM + i * 32 + h * 16 + pass * 8 + rr;
          off[pass] = rowoff[row];
          v[pass] = *reinterpret_cast<const float4*>(&xp[(pass * 8 + rr) * UDET_XP + c4]);
          if (!PLAIN && !slab && off[pass] >= 0 && nb < p.Cout) epi4_request(p, off[pass], nb, rq[pass]);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = wm * WTM + i * 32 + h * 16 + pass * 8 + rr;
          if (off[pass] < 0) continue;
          if (slab) {
            if (nb < p.ldp) *reinterpret_cast<float4*>(p.partial + (slab_off + (long)(prow0 + row) * p.ldp + nb)) = v[pass];
          } else if (nb < p.Cout) {
            if (PLAIN) epi4_finish_plain<ELU>(p, off[pass], nb, v[pass], bias[j], ea.slope);
            else epi4_finish<ELU>(p, off[pass], nb, v[pass], bias[j], rq[pass], ea);
          }
        }
        if (i == 0 && j == 0 && h == 0) IGEMM_STAMP_B(9);
      }
    }
  }
}
// the bias quads of a wave's column blocks, requested in front of the K loop (the quad of the float4 store path: lane & 7): at the head of
// the tile store the same load is a cold miss of ~1 000-2 000 cycles with nothing to hide behind.  Zero for K slices (the second pass adds
// the bias) and for columns beyond the layer.
template <int TN, int WTN>
__device__ __forceinline__ void igemm_bias_prefetch(const ConvParams& p, int n0, int wn, int lane, bool slab, float4 (&bias)[TN]) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + wn * WTN + j * 32 + (lane & 7) * 4;
    bias[j] = (!slab && p.bias && nb + 3 < ((p.Cout + 3) & ~3) && (p.Cout & 3) == 0 && nb < p.Cout) ? epi4_bias(p, nb) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void igemm_store(const ConvParams& p, floatx16 (&acc)[TM][TN], const int* rowoff, int wm, int wn, int li,
                                            int lh, int n0, int prow0, int Mtot, bool slab, long slab_off, float* xp = nullptr,
                                            const float4* bias_pre = nullptr) {
  // slab: this workgroup holds a K slice; its partial tile goes to p.partial + slab_off + (class row) * ldp
  if (xp != nullptr && !(slab && p.fold) && (slab ? (reinterpret_cast<uintptr_t>(p.partial) & 15) == 0 : epilogue4_out_ok(p))) {
    const bool plain = slab || epi4_plain(p);
    if (!slab && p.act == ACT_ELU) {
      if (plain) igemm_store_quads<TM, TN, WTM, WTN, true, true>(p, acc, rowoff, wm, wn, li, lh, n0, prow0, slab, slab_off, xp, bias_pre);
      else igemm_store_quads<TM, TN, WTM, WTN, true, false>(p, acc, rowoff, wm, wn, li, lh, n0, prow0, slab, slab_off, xp, bias_pre);
    } else {
      if (plain) igemm_store_quads<TM, TN, WTM, WTN, false, true>(p, acc, rowoff, wm, wn, li, lh, n0, prow0, slab, slab_off, xp, bias_pre);
      else igemm_store_quads<TM, TN, WTM, WTN, false, false>(p, acc, rowoff, wm, wn, li, lh, n0, prow0, slab, slab_off, xp, bias_pre);
    }
    return;
  }
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
        const float v = acc[i][j][r];
        if (slab) {
          if (n < p.ldp) {
            float* dst = p.partial + (slab_off + (long)(prow0 + row) * p.ldp + n);
            // folded form: the slab is published write-through (device-scope store, `sc1`): it is in memory when the store is
            // acknowledged, so no L2 write-back fence is needed before the ticket (MI355X_MICROARCH.md "publish-large")
            if (p.fold) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
          }
          continue;
        }
        if (n >= p.Cout) continue;
        conv_epilogue(p, off, n, v);
      }
    }
  }
}

// Split-K without a second launch: every workgroup publishes its partial tile write-through (igemm_store), drains its
// stores, and one lane draws a ticket; the workgroup that draws the last sums the slabs IN SPLIT ORDER (the result does not
// depend on which workgroup arrives last) with device-scope (`sc1`) loads -- they read memory, not a stale line of this XCD's
// L2, which is not coherent with the L2s the other workgroups wrote through -- and runs the epilogue, then resets the ticket
// for the next launch on this stream.  No release / acquire fences: a fence writes back / invalidates the whole L2 and cost
// more than the launch it replaces (r2a: the folded form with __threadfence() lost on every one of 141 split shapes).
// Called by the NT threads [0, NT) of the workgroup that are still alive (the staging waves of the wave-specialised kernels
// have exited: s_barrier counts surviving waves only).
__device__ __forceinline__ float4 load4_device_scope(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                     __uint_as_float((unsigned)(b >> 32)));
}
template <int BM, int BN, int NT>
__device__ __forceinline__ void splitk_fold(const ConvParams& p, const int* rowoff, int* s_last, int t, int n0, int prow0, int Mtot,
                                            int tile_id) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's slab stores are acknowledged (write-through: in memory)
  __syncthreads();
  if (t == 0) *s_last = atomicAdd(p.tickets + tile_id, 1) == p.ksplit - 1;
  __syncthreads();
  if (!*s_last) return;
  constexpr int C4 = BN / 4, ROWS = NT / C4;
  const int c4 = t % C4, n = n0 + c4 * 4;
  const size_t slab = (size_t)p.Mall * p.ldp;
  if (n < p.ldp && t < ROWS * C4) {  // (BN = 96: 240 of the 256 threads tile the [ROWS][C4] grid exactly)
    for (int row = t / C4; row < BM; row += ROWS) {
      const int off = rowoff[row];
      if (off < 0) continue;
      const float* src = p.partial + (size_t)(prow0 + row) * p.ldp + n;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = 0;
      for (; s + 3 < p.ksplit; s += 4) {  // four slabs in flight, added in split order
        const float4 a0 = load4_device_scope(src + (size_t)s * slab);
        const float4 a1 = load4_device_scope(src + (size_t)(s + 1) * slab);
        const float4 a2 = load4_device_scope(src + (size_t)(s + 2) * slab);
        const float4 a3 = load4_device_scope(src + (size_t)(s + 3) * slab);
        v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
        v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w;
        v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w;
        v.x += a3.x; v.y += a3.y; v.z += a3.z; v.w += a3.w;
      }
      for (; s < p.ksplit; ++s) {
        const float4 a0 = load4_device_scope(src + (size_t)s * slab);
        v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
      }
      if (n < p.Cout) conv_epilogue(p, off, n, v.x);
      if (n + 1 < p.Cout) conv_epilogue(p, off, n + 1, v.y);
      if (n + 2 < p.Cout) conv_epilogue(p, off, n + 2, v.z);
      if (n + 3 < p.Cout) conv_epilogue(p, off, n + 3, v.w);
    }
  }
  if (t == 0) p.tickets[tile_id] = 0;
}

// WS (wave specialisation): 512-thread workgroups; waves 0-3 only read fragments from LDS and issue MFMAs, waves
// 4-7 only stage (global -> registers -> LDS) one stage ahead.  The matrix pipe of a SIMD is then fed by waves that
// never wait on HBM/L2 or on address arithmetic; one raw s_barrier per stage hands the buffers over.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool WS>
__global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : 2) void conv_igemm_kernel(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int NT = WS ? 512 : 256;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM * 32 == WTM && TN * 32 == WTN, "wave tile must be a multiple of 32");
  constexpr int KQ = BK / 4;                // float4 per A row per stage
  constexpr int LDA = BM + 32 / BK;         // 4*LDA == 32/KQ (mod 32): conflict-free transposing stores
  constexpr int A_ROWS = 256 / KQ;          // A rows staged per pass
  constexpr int A_LD = BM / A_ROWS;
  static_assert(A_LD * A_ROWS == BM, "BM must be a multiple of 256/(BK/4)");
  constexpr int B_F4_ROW = BN / 4;
  constexpr int B_LD = BK * B_F4_ROW / 256;
  static_assert(B_LD * 256 == BK * B_F4_ROW, "BK*BN/4 must be a multiple of 256");

  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
  __shared__ int rowoff[BM];
  __shared__ int2 tap_yx[UDET_MAX_TAPS];
  __shared__ int tap_w[UDET_MAX_TAPS];
  __shared__ int s_last;

  const int tid = threadIdx.x;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 8);  // WS: 0 = MFMA waves, 1 = staging waves
  const int t = tid & 255;                                    // index inside the role's 256 threads
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: consecutive M tiles (which share input halos) stay on one XCD's L2.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const TileCls tc = tile_cls<BM>(p, bid);
  const int OHWq = tc.OHWq, Mtot = tc.Mtot /* of this class / segment */, m0 = tc.m0, tap0 = tc.tap0, ntc = tc.ntc, ooy = tc.ooy, oox = tc.oox;
  const int n0 = blockIdx.y * BN;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;

  // ---- per-block tables -----------------------------------------------------
  for (int i = tid; i < ntc; i += NT) {
    const ConvTap tp = conv_tap(p, tap0 + i);
    tap_yx[i] = make_int2(tp.dy, tp.dx);
    tap_w[i] = tp.widx;
  }
  for (int r = tid; r < BM; r += NT) {
    const int m = m0 + r;
    int off = -1;
    if (m < Mtot) {
      const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
      const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
      off = (n * p.OH + qy * p.osy + ooy) * p.OW + qx * p.osx + oox;
    }
    rowoff[r] = off;
  }
  const int a_kq = t % KQ;
  int a_base[A_LD], a_iy0[A_LD], a_ix0[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int r = t / KQ + j * A_ROWS;
    const int m = m0 + r;
    if (m < Mtot) {
      const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
      const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
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

  const int Kc = p.Kc;
  const int nchunks = (ntc * Kc + BK - 1) / BK;
  int c_begin = 0, c_end = nchunks;
  if (p.ksplit > 1) {
    c_begin = (int)((unsigned)(nchunks * blockIdx.z) / (unsigned)p.ksplit);  // (32-bit: nchunks * ksplit < 2^31)
    c_end = (int)((unsigned)(nchunks * (blockIdx.z + 1)) / (unsigned)p.ksplit);
  }
  // flat-K cursors of this thread's A float4 and of its B rows; advanced by BK per stage
  const KOrder ko = korder(Kc, ntc);
  KCursor ka = kc_init(ko, c_begin * BK + a_kq * 4), kb[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) kb[j] = kc_init(ko, c_begin * BK + (t + j * 256) / B_F4_ROW);
  __syncthreads();  // tap tables visible

  float4 ra[A_LD], rb[B_LD];
  auto load_chunk = [&]() {
    int dy = 0, dx = 0;
    const bool a_ok = kc_valid(ko, ka);
    const int a_c = kc_chan(ko, ka);
    if (a_ok) {
      const int2 yx = tap_yx[ka.tap];
      dy = yx.x;
      dx = yx.y;
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = a_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        iy >>= p.up_shift;
        ix >>= p.up_shift;
        const size_t off = (size_t)(a_base[j] + iy * Ws + ix) * p.ldx + p.x_coff + a_c;
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
      const int c4 = (t + j * 256) % B_F4_ROW;
      const int n = n0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc_valid(ko, kb[j]) && n < p.ldw)
        v = *reinterpret_cast<const float4*>(p.wp + ((size_t)tap_w[kb[j].tap] * Kc + kc_chan(ko, kb[j])) * p.ldw + n);
      rb[j] = v;
    }
    // advance the cursors to the next stage
    kc_advance(ko, ka, BK);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) kc_advance(ko, kb[j], BK);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int r = t / KQ + j * A_ROWS;
      As[buf][a_kq * 4 + 0][r] = ra[j].x;
      As[buf][a_kq * 4 + 1][r] = ra[j].y;
      As[buf][a_kq * 4 + 2][r] = ra[j].z;
      As[buf][a_kq * 4 + 3][r] = ra[j].w;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256;
      const int krow = idx / B_F4_ROW, c4 = idx - krow * B_F4_ROW;
      *reinterpret_cast<float4*>(&Bs[buf][krow][c4 * 4]) = rb[j];
    }
  };

  // MFMA stage: fragments are double-buffered in registers (reads for k-pair kk+1 are in flight while the matrix
  // pipe works on kk), so a lone wave keeps the pipe fed without waiting out the LDS latency every 4 MFMAs.
  auto compute_chunk = [&](int buf) {
    float a[2][TM], b[2][TN];
    auto frag = [&](int s, int kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[s][i] = As[buf][kk * 2 + lh][wm * WTM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[s][j] = Bs[buf][kk * 2 + lh][wn * WTN + j * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) frag((kk + 1) & 1, kk + 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
      // pin the order: next k-pair's LDS reads are issued BEFORE this k-pair's MFMAs (hipcc otherwise sinks them)
      if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
  };

  if constexpr (WS) {
    // raw barriers: only LDS traffic is drained (lgkmcnt), global loads stay in flight across the hand-over
    auto handover = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    if (role == 1) {
      __builtin_amdgcn_s_setprio(3);  // staging waves first (see conv_igemm_dma_kernel)
      if (c_begin < c_end) {
        load_chunk();
        store_chunk(0);
        if (c_begin + 1 < c_end) load_chunk();
      }
      handover();
      int buf = 0;
      for (int c = c_begin; c < c_end; ++c) {
        if (c + 1 < c_end) {
          store_chunk(buf ^ 1);                  // stage c+1 (loaded during the previous iteration)
          if (c + 2 < c_end) load_chunk();       // stage c+2 stays in flight over the barrier
        }
        handover();
        buf ^= 1;
      }
      return;
    }
    handover();
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
      compute_chunk(buf);
      handover();
      buf ^= 1;
    }
  } else {
    if (c_begin < c_end) {
      load_chunk();
      store_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
      const bool more = c + 1 < c_end;
      if (more) load_chunk();
      compute_chunk(buf);
      if (more) store_chunk(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue -------------------------------------------------------------
  igemm_store<TM, TN, WTM, WTN>(p, acc, rowoff, wm, wn, li, lh, n0, tc.prow0 + m0, Mtot, p.ksplit > 1,
                                (long)blockIdx.z * p.Mall * p.ldp, xpose_scratch<sizeof(As), sizeof(Bs)>(&As[0][0][0], &Bs[0][0][0], wave));
  if (p.ksplit > 1 && p.fold) splitk_fold<BM, BN, 256>(p, rowoff, &s_last, t, n0, tc.prow0 + m0, Mtot, blockIdx.y * gridDim.x + bid);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (forward convolutions and backward-data of linear layers: no act' on the A operand).
// The staging waves issue `global_load_lds_dwordx4` (16 B per lane straight into LDS, no VGPR round trip, no
// ds_write pass); halo / K-tail lanes read a 16-byte zero block instead of branching.  DMA writes LDS lane-linearly,
// so the A stage is row-major [BM][32] (one 128-byte line per pixel) with the 16-byte slot index XOR-swizzled by
// (row>>1)&7 on the SOURCE side; the MFMA waves read their fragment as ONE ds_read_b128 per 32 rows per 4 k-pairs
// (conflict-free under the swizzle) and walk K in the permuted order {4g+e : g = 2*kk+half}, which the B fragment
// reads ([k][n] rows, ds_read_b32) follow.  Same flat-K / parity-class / split-K semantics as conv_igemm_kernel.
// ---------------------------------------------------------------------------------------------------------------
// (the body takes the workgroup's x index and the x extent of ITS problem as arguments: a pair launch -- conv_igemm_dma_pair_kernel below --
// runs two problems of the same tile configuration in one grid, each workgroup seeing only its own problem's parameter block)
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, bool F16>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvParams& p, const int bid_x, const int grid_x) {
  static_assert(NS >= 2 && NS <= 4, "stages");
  static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves");
  constexpr int BK = 32;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM * 32 == WTM && TN * 32 == WTN && BM % 32 == 0, "tile");
  constexpr int A_LD = BM / 32;               // 256 staging threads cover 32 rows x 8 slots per pass
  constexpr int B_F4_ROW = BN / 4;
  constexpr int B_LD = BK * B_F4_ROW / 256;
  static_assert(B_LD * 256 == BK * B_F4_ROW, "BK*BN/4 must be a multiple of 256");
  typedef __attribute__((address_space(3))) void* lds_ptr;

  __shared__ __attribute__((aligned(16))) float As[NS][BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs[NS][BK][BN];
  __shared__ int rowoff[BM];
  __shared__ int2 tap_yx[UDET_MAX_TAPS];
  __shared__ int tap_w[UDET_MAX_TAPS];
  __shared__ int s_last;

  IGEMM_STAMP(0);
  const int tid = threadIdx.x;
  // Speculative tap fetch (round 6): the tap table of an unsegmented single-class launch starts at taps[0], whatever the block decodes
  // to -- its (vector) load from the kernel-argument block goes out HERE, beside the scalar loads of the fields the decode waits for,
  // instead of behind them (two back-to-back cold misses, ~1 us each, in front of every launch's first DMA)
  int spec_dy = 0, spec_dx = 0, spec_widx = 0;  // (three scalars, not a ConvTap copy: hipcc keeps the 12-byte struct in scratch memory)
  if (tid < UDET_MAX_TAPS) {
    spec_dy = p.taps[tid].dy;
    spec_dx = p.taps[tid].dx;
    spec_widx = p.taps[tid].widx;
  }
  const int role = __builtin_amdgcn_readfirstlane(tid >> 8);  // 0 = MFMA waves, 1 = staging waves
  const int t = tid & 255;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  int bid = bid_x;
  int kz = blockIdx.z, knz = p.ksplit;  // K slice of this workgroup / slices of its tile
  {
    int nwg = grid_x;
    if (p.tail_ks > 1) {  // tail split: the x-blocks past tail_full are cut into tail_ks slices, the others run whole
      nwg = p.tail_full;
      knz = 1;
      if (bid >= p.tail_full) {
        const int r = bid - p.tail_full;
        kz = r % p.tail_ks;
        bid = p.tail_full + r / p.tail_ks;
        knz = p.tail_ks;
      }
    }
    if (bid < nwg) {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
  }
  const TileCls tc = tile_cls<BM>(p, bid);
  const int OHWq = tc.OHWq, Mtot = tc.Mtot /* of this class / segment */, m0 = tc.m0, tap0 = tc.tap0, ntc = tc.ntc, ooy = tc.ooy, oox = tc.oox;
  const int n0 = blockIdx.y * BN;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;
  IGEMM_STAMP(6);

  if (p.nseg == 0 && tap0 == 0) {  // (uniform) the speculative fetch is this block's table
    if (tid < ntc) {
      tap_yx[tid] = make_int2(spec_dy, spec_dx);
      tap_w[tid] = spec_widx;
    }
  } else {
    for (int i = tid; i < ntc; i += 512) {
      const ConvTap tp = conv_tap(p, tap0 + i);
      tap_yx[i] = make_int2(tp.dy, tp.dx);
      tap_w[i] = tp.widx;
    }
  }
  // (the output row offsets are read by the tile store only: the MFMA waves fill them while they wait for the first stage -- below --
  // instead of in front of the barrier every wave's first DMA waits behind: round 6, tools/igemm_stamps.py)
  const int Kc = p.Kc;
  // kfast (Kc >= 32, no up-sampled read): a stage is ONE (channel block, tap) pair -- the last, narrower block is padded with zero
  // lanes instead of straddling into the next tap -- so the K cursor is wave-uniform (see the staging waves)
  // kfast bit 2 (Kc in {4, 8, 16}): a stage is 32 / Kc WHOLE taps; the tap of a lane follows from its channel slot (a per-lane constant)
  const int tsh = Kc == 4 ? 3 : (Kc == 8 ? 2 : 1);  // log2(taps per stage) of the packed form (Kc = 4, 8, 16): shifts, not run-time divisions
  const int tps = (p.kfast & 4) ? 1 << tsh : 1;     // taps per stage
  const int nchunks = (p.kfast & 1) ? ntc * ((Kc + 31) >> 5) : ((p.kfast & 4) ? (ntc + tps - 1) >> tsh : (ntc * Kc + BK - 1) / BK);
  int c_begin = 0, c_end = nchunks;
  if (knz > 1) {
    c_begin = (int)((unsigned)(nchunks * kz) / (unsigned)knz);  // (nchunks * knz < 2^31: 32-bit divisions, a third of the 64-bit ones' instructions)
    c_end = (int)((unsigned)(nchunks * (kz + 1)) / (unsigned)knz);
  }
  // slab of this slice: regular split-K keeps whole-output slabs, the tail split only the rows from tail_prow0 on
  const long slab_off = p.tail_ks > 1 ? ((long)kz * (p.Mall - p.tail_prow0) - p.tail_prow0) * p.ldp : (long)kz * p.Mall * p.ldp;
  IGEMM_STAMP(7);
  // (the barrier that publishes the tap tables sits inside the two role paths: the staging waves reach it only after their per-row
  // address arithmetic, which needs no table -- that work runs beside the kernel-argument / tap-table latency instead of behind it)

  if (role == 1) {
    // ------------------------------------------------ staging waves ------------------------------------------------
    // issue priority over the MFMA waves of the same SIMD (this and the co-resident workgroups'): a stage's DMA goes out as soon as
    // its buffer is free instead of waiting for gaps between MFMAs (128x64 tile on the 568-channel layer: 920 -> 750 us)
    __builtin_amdgcn_s_setprio(3);
    const int kq = lane & 7;                                  // LDS slot written by this lane (lane-linear)
    const int kqs = kq ^ ((wave * 4 + (lane >> 4)) & 7);      // channel group it holds: slot ^ ((row>>1)&7)
    int a_base[A_LD], a_iy0[A_LD], a_ix0[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int m = m0 + j * 32 + wave * 8 + (lane >> 3);
      if (m < Mtot) {
        const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
        const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
        a_base[j] = n * Hs * Ws;
        a_iy0[j] = qy * p.isy;
        a_ix0[j] = qx * p.isx;
      } else {
        a_base[j] = 0;
        a_iy0[j] = -(1 << 28);
        a_ix0[j] = 0;
      }
    }
    const float* zero = p.zero16;
    // Generic K cursor (stages may straddle taps: Kc < 32 or an up-sampled read): per-lane (block, tap, channel) cursors advanced
    // with data-dependent control flow -- ~1500 instructions per stage for a 128x128 tile, more than the 4096 MFMA cycles of the
    // stage leave room for on a SIMD that also hosts an MFMA wave.  The uniform cursor below needs ~100.
    // (the generic cursor's set-up is a dozen integer divisions by run-time values, ~40 instructions each: only launches that use it pay
    // for it -- round 6: the staging waves' set-up was ~3 700 cycles in front of EVERY launch's first DMA, tools/igemm_stamps.py)
    const KOrder ko = korder(Kc, ntc);
    KCursor ka, kb[B_LD];
    ka.blk = ka.tap = ka.c = ka.w = 0;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) kb[j] = ka;
    if (!(p.kfast & 5)) {
      ka = kc_init(ko, c_begin * BK + kqs * 4);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) kb[j] = kc_init(ko, c_begin * BK + (t + j * 256) / B_F4_ROW);
    }
    auto issue_generic = [&](int buf) {
      int dy = 0, dx = 0;
      const bool a_ok = kc_valid(ko, ka);
      const int a_c = kc_chan(ko, ka);
      if (a_ok) {
        const int2 yx = tap_yx[ka.tap];
        dy = yx.x;
        dx = yx.y;
      }
#pragma unroll
      for (int j = 0; j < A_LD; ++j) {
        int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
        const bool ok = a_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        iy >>= p.up_shift;
        ix >>= p.up_shift;
        const float* src = ok ? p.x + ((size_t)(a_base[j] + iy * Ws + ix) * p.ldx + p.x_coff + a_c) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)&As[buf][j * 32 + wave * 8][0], 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int c4 = (t + j * 256) % B_F4_ROW;
        const int n = n0 + c4 * 4;
        const bool ok = kc_valid(ko, kb[j]) && n < p.ldw;
        const int wi = ok ? tap_w[kb[j].tap] : 0;
        const float* src = ok ? p.wp + (((size_t)wi * Kc + kc_chan(ko, kb[j])) * p.ldw + n) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
      kc_advance(ko, ka, BK);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) kc_advance(ko, kb[j], BK);
    };
    // Uniform K cursor: stage s = (block s / ntc, tap s % ntc), kept in scalars.  Per lane and row only constants remain: the
    // element offset of the row's pixel at tap (0,0) and channel slot kqs, the weight row / column of each B quad.
    int a_off[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      a_off[j] = a_iy0[j] < -(1 << 27) ? 0 : (a_base[j] + a_iy0[j] * Ws + a_ix0[j]) * p.ldx + p.x_coff + kqs * 4;  // (rows past the grid: never read)
    int b_off[B_LD], b_row[B_LD];
    bool b_col[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int idx = t + j * 256, row = idx / B_F4_ROW, n = n0 + (idx - row * B_F4_ROW) * 4;
      b_row[j] = row;
      b_off[j] = row * p.ldw + n;
      b_col[j] = n < p.ldw;
    }
    int s_blk = c_begin == 0 ? 0 : __builtin_amdgcn_readfirstlane(c_begin / (ntc > 0 ? ntc : 1));  // (unsplit launches: no division)
    int s_tap = __builtin_amdgcn_readfirstlane(c_begin - s_blk * ntc);
    auto issue_fast = [&](int buf) {
      const int2 yx = tap_yx[s_tap];
      const int dy = yx.x, dx = yx.y, c0 = s_blk << 5;
      const int tap_off = (dy * Ws + dx) * p.ldx + c0;
      const bool ch_ok = c0 + kqs * 4 < Kc;
#pragma unroll
      for (int j = 0; j < A_LD; ++j) {
        const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
        const bool ok = ch_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = ok ? p.x + (a_off[j] + tap_off) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)&As[buf][j * 32 + wave * 8][0], 16, 0, 0);
      }
      const float* wrow = p.wp + ((size_t)tap_w[s_tap] * Kc + c0) * p.ldw;
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const bool ok = b_col[j] && c0 + b_row[j] < Kc;
        const float* src = ok ? wrow + b_off[j] : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
      if (++s_tap == ntc) { s_tap = 0; ++s_blk; }
    };
    // Packed taps (Kc < 32): stage s holds taps s * tps .. s * tps + tps - 1; K index k of the stage = (tap k / Kc, channel k % Kc).
    const int ksh = Kc == 4 ? 2 : (Kc == 8 ? 3 : 4);                      // (packed taps exist for Kc = 4, 8, 16 only: shifts, not divisions)
    const int a_sub = (kqs * 4) >> ksh, a_ch = (kqs * 4) - (a_sub << ksh);  // this lane's A slot
    int b_sub[B_LD], b_poff[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      b_sub[j] = b_row[j] >> ksh;
      b_poff[j] = (b_row[j] - (b_sub[j] << ksh)) * p.ldw + (b_off[j] - b_row[j] * p.ldw);  // (channel row, column) inside the tap's weight block
    }
    // (the stage index is the caller's counter, not a captured variable of its own: two captured counters incremented in sibling
    // branches end as a pointer phi that keeps both in scratch memory -- 12 bytes of private segment on every launch of this kernel)
    auto issue_pack = [&](int buf, int s_stage) {
      const int ta = s_stage * tps + a_sub;
      const bool ta_ok = ta < ntc;
      const int2 yx = tap_yx[ta_ok ? ta : 0];
      const int dy = yx.x, dx = yx.y;
      const int tap_off = (dy * Ws + dx) * p.ldx + a_ch - kqs * 4;  // (a_off carries + kqs * 4)
#pragma unroll
      for (int j = 0; j < A_LD; ++j) {
        const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
        const bool ok = ta_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = ok ? p.x + (a_off[j] + tap_off) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)&As[buf][j * 32 + wave * 8][0], 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int tb = s_stage * tps + b_sub[j];
        const bool ok = b_col[j] && tb < ntc;
        const float* src = ok ? p.wp + ((size_t)tap_w[ok ? tb : 0] * Kc * p.ldw + b_poff[j]) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
    };
    auto issue = [&](int buf, int stage) {
      if (p.kfast & 1) issue_fast(buf);
      else if (p.kfast & 4) issue_pack(buf, stage);
      else issue_generic(buf);
    };
    // NS-deep ring: NS - 1 stages are in flight while the MFMA waves work on one, so a stage has (NS - 1) chunk times to land
    // (one 128x128 chunk is 1.7 us of MFMA work, about one loaded-memory latency: with a single stage in flight a workgroup
    // alone on its CU waits at every barrier).  Loads retire in order: waiting for vmcnt <= (stages issued later) * L is
    // waiting for the stage the MFMA waves need next.
    constexpr int L = A_LD + B_LD;  // DMA instructions per lane and stage
    auto landed = [&](int newer) {  // `newer` (uniform): stages issued after the one that has to be in LDS now
      if (NS > 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
      else if (NS > 2 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    __syncthreads();  // tap tables visible (the MFMA waves' counterpart: in front of their accumulator set-up)
    int issued = c_begin, ibuf = 0;
    for (int s = 0; s < NS - 1 && issued < c_end; ++s) {
      issue(ibuf, issued);
      ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
      ++issued;
    }
    landed(issued - c_begin - 1);
    for (int c = c_begin; c < c_end; ++c) {
      if (issued < c_end) {  // its buffer held stage c - 1, which the MFMA waves left at the previous barrier
        issue(ibuf, issued);
        ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
        ++issued;
      }
      landed(issued - c - 2);  // stage c + 1 in LDS (nothing left to wait for after the last one: vmcnt(0) is free)
    }
    return;
  }

  // -------------------------------------------------- MFMA waves --------------------------------------------------
  __syncthreads();  // (pairs with the staging waves' barrier above)
  IGEMM_STAMP(1);
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int swz = (li >> 1) & 7;  // (row>>1)&7 of every row this lane reads (wave / sub-tile offsets are multiples of 16)
  const float xscale = F16 ? p.f16_xscale : 1.f;
  auto handover = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto compute_chunk = [&](int buf) {
    float4 a[2][TM];
    float b[2][4][TN];
    auto frag = [&](int s, int kk) {
      const int g = 2 * kk + lh;  // channel group of this lane half
#pragma unroll
      for (int i = 0; i < TM; ++i) a[s][i] = *reinterpret_cast<const float4*>(&As[buf][wm * WTM + i * 32 + li][(g ^ swz) * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s][e][j] = Bs[buf][g * 4 + e][wn * WTN + j * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) frag((kk + 1) & 1, kk + 1);
      if constexpr (F16) {  // the lane half's four consecutive K values of a fragment are one fp16 operand of the K = 8 MFMA
        halfx4 ah[TM], bh[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          ah[i] = halfx4{(_Float16)(a[kk & 1][i].x * xscale), (_Float16)(a[kk & 1][i].y * xscale), (_Float16)(a[kk & 1][i].z * xscale),
                         (_Float16)(a[kk & 1][i].w * xscale)};
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bh[j] = halfx4{(_Float16)b[kk & 1][0][j], (_Float16)b[kk & 1][1][j], (_Float16)b[kk & 1][2][j], (_Float16)b[kk & 1][3][j]};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = e == 0 ? a[kk & 1][i].x : (e == 1 ? a[kk & 1][i].y : (e == 2 ? a[kk & 1][i].z : a[kk & 1][i].w));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kk & 1][e][j], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (kk + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, TM + 4 * TN, 0);
      if constexpr (!F16) __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
    }
  };
  float4 bias_pre[TN];  // (only the float4 store path reads it: Cout a multiple of 4 -- igemm_bias_prefetch yields zeros otherwise, unused)
  igemm_bias_prefetch<TN, WTN>(p, n0, wn, lane, knz > 1, bias_pre);
  for (int r = t; r < BM; r += 256) {  // rows of the tile -> output pixel offsets (visible to every MFMA wave behind the hand-over barriers)
    const int m = m0 + r;
    int off = -1;
    if (m < Mtot) {
      const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
      const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
      off = (n * p.OH + qy * p.osy + ooy) * p.OW + qx * p.osx + oox;
    }
    rowoff[r] = off;
  }
  handover();
  IGEMM_STAMP(2);
  {
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
      compute_chunk(buf);
      handover();
      buf = buf + 1 == NS ? 0 : buf + 1;
    }
  }
  IGEMM_STAMP(3);
  if (F16 && xscale != 1.f) {
    const float inv = 1.f / xscale;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  igemm_store<TM, TN, WTM, WTN>(p, acc, rowoff, wm, wn, li, lh, n0, tc.prow0 + m0, Mtot, knz > 1, slab_off,
                                xpose_scratch<sizeof(As), sizeof(Bs)>(&As[0][0][0], &Bs[0][0][0], wave), bias_pre);
  if (p.ksplit > 1 && p.fold) splitk_fold<BM, BN, 256>(p, rowoff, &s_last, t, n0, tc.prow0 + m0, Mtot, blockIdx.y * grid_x + bid);
#ifdef UDET_EXPERIMENT
  IGEMM_STAMP(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IGEMM_STAMP(5);
#endif
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, bool F16 = false>
__global__ __launch_bounds__(512, NS == 2 ? 4 : 2) void conv_igemm_dma_kernel(const ConvParams p) {
  conv_igemm_dma_body<BM, BN, WAVES_M, WAVES_N, NS, F16>(p, blockIdx.x, gridDim.x);
}
// Two problems in ONE launch ("pair launch", round 6): same tile configuration, same N blocks and K slices, independent operands --
// x-blocks [0, xa) belong to problem 0, the rest to problem 1.  The recover net's two encoders (nets.py:57-75: aconv_k / bconv_k, same
// geometry per level, separate weights, different batch) and their backward-data passes run this way: each of those launches fills a
// fraction of the chip and costs a launch boundary, two of them side by side cost hardly more than the larger one.  The parameter
// blocks stay in the kernel-argument segment (2 x 1752 bytes of the 4 KB): the workgroup picks its block with one scalar select.
struct ConvPair {
  ConvParams p[2];
  int xa;
};
static_assert(sizeof(ConvPair) <= 4000, "kernel-argument segment");
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS>
__global__ __launch_bounds__(512, NS == 2 ? 4 : 2) void conv_igemm_dma_pair_kernel(const ConvPair pp) {
  const int second = __builtin_amdgcn_readfirstlane((int)blockIdx.x >= pp.xa ? 1 : 0);
  conv_igemm_dma_body<BM, BN, WAVES_M, WAVES_N, NS, false>(pp.p[second], (int)blockIdx.x - (second ? pp.xa : 0),
                                                             second ? (int)gridDim.x - pp.xa : pp.xa);
}

// ---------------------------------------------------------------------------------------------------------------
// Self-staging LDS-DMA variant: 256 threads = 4 MFMA waves that also issue the DMA of the next stage themselves (the
// address arithmetic runs in the shadow of the previous MFMAs), 16-wide K stages.  A 128x128 tile then needs 32 KB of
// LDS and one wave per SIMD, so three to four workgroups share a CU -- the MFMA pipe of a SIMD is fed by waves of
// DIFFERENT workgroups that are at different points of their stage (one waits at its barrier or for its fragments while
// another multiplies).  conv_bench on the 128-channel 3x3 layer: one wave-specialised 128x128 workgroup alone on a CU
// keeps the pipe 44 % busy, two co-resident ones 57 %.
// A stage: row-major [BM][16] (64-byte rows, 4 slots of 16 B), slot XOR-swizzled by (row>>1)&3 on the source side; B: [16][BN].
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool F16 = false>
__global__ __launch_bounds__(256, 3) void conv_igemm_dma4_kernel(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int BK = 16;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM * 32 == WTM && TN * 32 == WTN && BM % 64 == 0 && BN % 32 == 0, "tile");
  constexpr int A_LD = BM / 64;               // 256 threads cover 64 rows x 4 slots per pass
  constexpr int B_F4_ROW = BN / 4;
  constexpr int B_F4 = BK * B_F4_ROW;         // float4 of one weight stage (128 for BN = 32: half of the threads load)
  constexpr int B_LD = (B_F4 + 255) / 256;
  static_assert(B_F4 % 64 == 0, "whole waves issue the weight DMA");
  typedef __attribute__((address_space(3))) void* lds_ptr;

  __shared__ __attribute__((aligned(16))) float As[2][BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
  __shared__ int rowoff[BM];
  __shared__ int2 tap_yx[UDET_MAX_TAPS];
  __shared__ int tap_w[UDET_MAX_TAPS];
  __shared__ int s_last;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const TileCls tc = tile_cls<BM>(p, bid);
  const int OHWq = tc.OHWq, Mtot = tc.Mtot /* of this class / segment */, m0 = tc.m0, tap0 = tc.tap0, ntc = tc.ntc, ooy = tc.ooy, oox = tc.oox;
  const int n0 = blockIdx.y * BN;
  const int Hs = p.H >> p.up_shift, Ws = p.W >> p.up_shift;

  for (int i = t; i < ntc; i += 256) {
    const ConvTap tp = conv_tap(p, tap0 + i);
    tap_yx[i] = make_int2(tp.dy, tp.dx);
    tap_w[i] = tp.widx;
  }
  for (int r = t; r < BM; r += 256) {
    const int m = m0 + r;
    int off = -1;
    if (m < Mtot) {
      const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
      const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
      off = (n * p.OH + qy * p.osy + ooy) * p.OW + qx * p.osx + oox;
    }
    rowoff[r] = off;
  }
  const int Kc = p.Kc;
  // kfast (bit 1: Kc >= 16, no up-sampled read): a stage is ONE (16-channel block, tap) pair -- uniform K cursor, see conv_igemm_dma_kernel
  const bool kfast = (p.kfast & 2) != 0;
  const int nchunks = kfast ? ntc * ((Kc + 15) >> 4) : (ntc * Kc + BK - 1) / BK;
  int c_begin = 0, c_end = nchunks;
  if (p.ksplit > 1) {
    c_begin = (int)((unsigned)(nchunks * blockIdx.z) / (unsigned)p.ksplit);  // (32-bit: nchunks * ksplit < 2^31)
    c_end = (int)((unsigned)(nchunks * (blockIdx.z + 1)) / (unsigned)p.ksplit);
  }
  __syncthreads();

  // ---- staging state of this thread: A_LD rows x one 16-byte slot, B_LD float4 of the weight stage ---------------
  const int kq = lane & 3;                                   // LDS slot written by this lane (lane-linear)
  const int kqs = kq ^ ((wave * 8 + (lane >> 3)) & 3);       // channel group it holds: slot ^ ((row>>1)&3), row = wave*16 + lane>>2
  int a_base[A_LD], a_iy0[A_LD], a_ix0[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int m = m0 + j * 64 + wave * 16 + (lane >> 2);
    if (m < Mtot) {
      const int n = (int)fdiv(m, tc.fd_ohw), rem = m - n * OHWq;
      const int qy = (int)fdiv(rem, tc.fd_ow), qx = rem - qy * tc.OWq;
      a_base[j] = n * Hs * Ws;
      a_iy0[j] = qy * p.isy;
      a_ix0[j] = qx * p.isx;
    } else {
      a_base[j] = 0;
      a_iy0[j] = -(1 << 28);
      a_ix0[j] = 0;
    }
  }
  const float* zero = p.zero16;
  const KOrder ko = korder(Kc, ntc);
  KCursor ka = kc_init(ko, kfast ? 0 : c_begin * BK + kqs * 4), kb[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) kb[j] = kc_init(ko, kfast ? 0 : c_begin * BK + (t + j * 256) / B_F4_ROW);
  auto issue_generic = [&](int buf) {
    int dy = 0, dx = 0;
    const bool a_ok = kc_valid(ko, ka);
    const int a_c = kc_chan(ko, ka);
    if (a_ok) {
      const int2 yx = tap_yx[ka.tap];
      dy = yx.x;
      dx = yx.y;
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = a_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      iy >>= p.up_shift;
      ix >>= p.up_shift;
      const float* src = ok ? p.x + ((size_t)(a_base[j] + iy * Ws + ix) * p.ldx + p.x_coff + a_c) : zero;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr)&As[buf][j * 64 + wave * 16][0], 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      if (j * 256 + wave * 64 < B_F4) {  // wave-uniform
        const int c4 = (t + j * 256) % B_F4_ROW;
        const int n = n0 + c4 * 4;
        const bool ok = kc_valid(ko, kb[j]) && n < p.ldw;
        const int wi = ok ? tap_w[kb[j].tap] : 0;
        const float* src = ok ? p.wp + (((size_t)wi * Kc + kc_chan(ko, kb[j])) * p.ldw + n) : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
    }
    kc_advance(ko, ka, BK);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) kc_advance(ko, kb[j], BK);
  };
  // uniform cursor: stage s = (16-channel block s / ntc, tap s % ntc), kept in scalars
  int a_off[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j)
    a_off[j] = a_iy0[j] < -(1 << 27) ? 0 : (a_base[j] + a_iy0[j] * Ws + a_ix0[j]) * p.ldx + p.x_coff + kqs * 4;
  int b_off[B_LD], b_row[B_LD];
  bool b_col[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int idx = t + j * 256, row = idx / B_F4_ROW, n = n0 + (idx - row * B_F4_ROW) * 4;
    b_row[j] = row;
    b_off[j] = row * p.ldw + n;
    b_col[j] = n < p.ldw;
  }
  const int ntc_ = ntc > 0 ? ntc : 1;
  int s_blk = __builtin_amdgcn_readfirstlane(c_begin / ntc_);
  int s_tap = __builtin_amdgcn_readfirstlane(c_begin - s_blk * ntc_);
  auto issue_fast = [&](int buf) {
    const int2 yx = tap_yx[s_tap];
    const int dy = yx.x, dx = yx.y, c0 = s_blk << 4;
    const int tap_off = (dy * Ws + dx) * p.ldx + c0;
    const bool ch_ok = c0 + kqs * 4 < Kc;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = ch_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float* src = ok ? p.x + (a_off[j] + tap_off) : zero;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr)&As[buf][j * 64 + wave * 16][0], 16, 0, 0);
    }
    const float* wrow = p.wp + ((size_t)tap_w[s_tap] * Kc + c0) * p.ldw;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      if (j * 256 + wave * 64 < B_F4) {  // wave-uniform
        const bool ok = b_col[j] && c0 + b_row[j] < Kc;
        const float* src = ok ? wrow + b_off[j] : zero;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr)(&Bs[buf][0][0] + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
    }
    if (++s_tap == ntc) { s_tap = 0; ++s_blk; }
  };
  auto issue = [&](int buf) {
    if (kfast) issue_fast(buf);
    else issue_generic(buf);
  };
  auto meet = [&]() {  // this wave's DMA has landed and its fragment reads are done, then meet the other waves
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int swz = (li >> 1) & 3;  // (row>>1)&3 of every row this lane reads (wave / sub-tile offsets are multiples of 32)
  const float xscale = F16 ? p.f16_xscale : 1.f;
  auto compute_chunk = [&](int buf) {
    float4 a[2][TM];
    float b[2][4][TN];
    auto frag = [&](int s_, int kk) {
      const int g = 2 * kk + lh;  // channel group of this lane half
#pragma unroll
      for (int i = 0; i < TM; ++i) a[s_][i] = *reinterpret_cast<const float4*>(&As[buf][wm * WTM + i * 32 + li][(g ^ swz) * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s_][e][j] = Bs[buf][g * 4 + e][wn * WTN + j * 32 + li];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk + 1 < 2) frag((kk + 1) & 1, kk + 1);
      if constexpr (F16) {  // the lane half's four consecutive K values of a fragment are one fp16 operand of the K = 8 MFMA
        halfx4 ah[TM], bh[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          ah[i] = halfx4{(_Float16)(a[kk & 1][i].x * xscale), (_Float16)(a[kk & 1][i].y * xscale), (_Float16)(a[kk & 1][i].z * xscale),
                         (_Float16)(a[kk & 1][i].w * xscale)};
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bh[j] = halfx4{(_Float16)b[kk & 1][0][j], (_Float16)b[kk & 1][1][j], (_Float16)b[kk & 1][2][j], (_Float16)b[kk & 1][3][j]};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = e == 0 ? a[kk & 1][i].x : (e == 1 ? a[kk & 1][i].y : (e == 2 ? a[kk & 1][i].z : a[kk & 1][i].w));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kk & 1][e][j], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (kk + 1 < 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + 4 * TN, 0);
      if constexpr (!F16) __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
    }
  };

  if (c_begin < c_end) issue(0);
  meet();
  {
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
      if (c + 1 < c_end) issue(buf ^ 1);  // lands while this stage is multiplied
      compute_chunk(buf);
      meet();
      buf ^= 1;
    }
  }
  if (F16 && xscale != 1.f) {
    const float inv = 1.f / xscale;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  igemm_store<TM, TN, WTM, WTN>(p, acc, rowoff, wm, wn, li, lh, n0, tc.prow0 + m0, Mtot, p.ksplit > 1,
                                (long)blockIdx.z * p.Mall * p.ldp, xpose_scratch<sizeof(As), sizeof(Bs)>(&As[0][0][0], &Bs[0][0][0], wave));
  if (p.ksplit > 1 && p.fold) splitk_fold<BM, BN, 256>(p, rowoff, &s_last, t, n0, tc.prow0 + m0, Mtot, blockIdx.y * gridDim.x + bid);
}

// second pass of a split-K launch: sum the partial slabs and run the epilogue.  SL lanes share one output element
// (each sums every SL-th slab, then a fixed-order shuffle tree): small outputs with many splits stay parallel.
template <int SL>
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvParams p) {
  const int Mall = p.Mall;
  const long total = (long)Mall * p.Cout;
  const int sl = threadIdx.x % SL;
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) / SL; e < total; e += (long)gridDim.x * (256 / SL)) {
    const int ma = (int)(e / p.Cout), n = (int)(e - (long)ma * p.Cout);
    // four slabs in flight per trip (the loads are independent; a plain loop waits for each before the next add);
    // the additions keep the slab order, so the sum is the same number as before
    float v = 0.f;
    const float* src = p.partial + (size_t)ma * p.ldp + n;
    const size_t slab = (size_t)Mall * p.ldp;
    int s = sl;
    for (; s + 3 * SL < p.ksplit; s += 4 * SL) {
      const float a0 = src[(size_t)s * slab], a1 = src[(size_t)(s + SL) * slab];
      const float a2 = src[(size_t)(s + 2 * SL) * slab], a3 = src[(size_t)(s + 3 * SL) * slab];
      v += a0;
      v += a1;
      v += a2;
      v += a3;
    }
    for (; s < p.ksplit; s += SL) v += src[(size_t)s * slab];
#pragma unroll
    for (int d = SL / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, SL);
    if (sl != 0) continue;
    const int off = row_pixel_off(p, ma);
    conv_epilogue(p, off, n, v);
  }
}

// the one-lane-per-element form with four consecutive channels per thread: 16-byte slab loads (a quarter of the load instructions
// and address arithmetic per byte), the same per-element summation order as conv_splitk_epilogue_kernel<1>
__global__ __launch_bounds__(256) void conv_splitk_epilogue4_kernel(const ConvParams p) {
  const int Mall = p.Mall;
  const int nq = p.ldp >> 2;  // quads per partial row (ldp = Cout rounded up to 4)
  const int row0 = p.tail_ks > 1 ? p.tail_prow0 : 0, ksplit = p.tail_ks > 1 ? p.tail_ks : p.ksplit;  // tail split: rows >= tail_prow0 only
  const long total = (long)(Mall - row0) * nq;
  const size_t slab = (size_t)(Mall - row0) * p.ldp;
  const bool vec = epilogue4_out_ok(p);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int mr = (int)(e / nq), n = (int)(e - (long)mr * nq) * 4, ma = row0 + mr;
    const float* src = p.partial + (size_t)mr * p.ldp + n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 3 < ksplit; s += 4) {
      const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
      const float4 a1 = *reinterpret_cast<const float4*>(src + (size_t)(s + 1) * slab);
      const float4 a2 = *reinterpret_cast<const float4*>(src + (size_t)(s + 2) * slab);
      const float4 a3 = *reinterpret_cast<const float4*>(src + (size_t)(s + 3) * slab);
      v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
      v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w;
      v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w;
      v.x += a3.x; v.y += a3.y; v.z += a3.z; v.w += a3.w;
    }
    for (; s < ksplit; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
      v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
    }
    const int off = row_pixel_off(p, ma);
    if (vec) {  // (Cout a multiple of 4: whole quads)
      if (n < p.Cout) conv_epilogue4(p, off, n, v);
      continue;
    }
    if (n < p.Cout) conv_epilogue(p, off, n, v.x);
    if (n + 1 < p.Cout) conv_epilogue(p, off, n + 1, v.y);
    if (n + 2 < p.Cout) conv_epilogue(p, off, n + 2, v.z);
    if (n + 3 < p.Cout) conv_epilogue(p, off, n + 3, v.w);
  }
}

// second pass of a pair launch: blocks [0, nba) reduce problem 0's slabs, the rest problem 1's (each as conv_splitk_epilogue4_kernel)
__device__ __forceinline__ void splitk_epilogue4_body(const ConvParams& p, const int bid_x, const int grid_x) {
  const int Mall = p.Mall;
  const int nq = p.ldp >> 2;
  const long total = (long)Mall * nq;
  const size_t slab = (size_t)Mall * p.ldp;
  const bool vec = epilogue4_out_ok(p);
  for (long e = (long)bid_x * 256 + threadIdx.x; e < total; e += (long)grid_x * 256) {
    const int ma = (int)(e / nq), n = (int)(e - (long)ma * nq) * 4;
    const float* src = p.partial + (size_t)ma * p.ldp + n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 3 < p.ksplit; s += 4) {  // (the summation order of conv_splitk_epilogue4_kernel: slab by slab)
      const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
      const float4 a1 = *reinterpret_cast<const float4*>(src + (size_t)(s + 1) * slab);
      const float4 a2 = *reinterpret_cast<const float4*>(src + (size_t)(s + 2) * slab);
      const float4 a3 = *reinterpret_cast<const float4*>(src + (size_t)(s + 3) * slab);
      v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
      v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w;
      v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w;
      v.x += a3.x; v.y += a3.y; v.z += a3.z; v.w += a3.w;
    }
    for (; s < p.ksplit; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
      v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
    }
    const int off = row_pixel_off(p, ma);
    if (vec) {
      if (n < p.Cout) conv_epilogue4(p, off, n, v);
      continue;
    }
    if (n < p.Cout) conv_epilogue(p, off, n, v.x);
    if (n + 1 < p.Cout) conv_epilogue(p, off, n + 1, v.y);
    if (n + 2 < p.Cout) conv_epilogue(p, off, n + 2, v.z);
    if (n + 3 < p.Cout) conv_epilogue(p, off, n + 3, v.w);
  }
}
__global__ __launch_bounds__(256) void conv_splitk_epilogue4_pair_kernel(const ConvPair pp) {
  const int second = __builtin_amdgcn_readfirstlane((int)blockIdx.x >= pp.xa ? 1 : 0);
  splitk_epilogue4_body(pp.p[second], (int)blockIdx.x - (second ? pp.xa : 0), second ? (int)gridDim.x - pp.xa : pp.xa);
}

// x-blocks of a launch: M tiles of every class / segment
static int conv_xblocks(const ConvParams& p, int bm) {
  if (p.nseg == 0) return p.ncls * ((p.N * p.OHq * p.OWq + bm - 1) / bm);
  int x = 0;
  for (int s = 0; s < p.nseg; ++s) x += (p.N * p.seg[s].h * p.seg[s].w + bm - 1) / bm;
  return x;
}

// second pass of a split-K launch (no tail split, not folded): sums the ksplit slabs of p.partial and runs the epilogue
int launch_splitk_second_pass(const ConvParams& p, hipStream_t stream) {
  const long total = (long)p.Mall * p.Cout;
  // lanes per element: keep >= ~64k threads busy while the split count allows it
  const int sl = (p.ksplit >= 16 && total * 16 <= 262144) ? 16 : ((p.ksplit >= 4 && total * 4 <= 262144) ? 4 : 1);
  long nbl = (total * sl + 255) / 256;
  const int nb = (int)(nbl > 4096 ? 4096 : nbl);
  if (sl == 16) UDET_LAUNCH(conv_splitk_epilogue_kernel<16>, dim3(nb), dim3(256), 0, stream, p);
  else if (sl == 4) UDET_LAUNCH(conv_splitk_epilogue_kernel<4>, dim3(nb), dim3(256), 0, stream, p);
  else if (p.ldp % 4 == 0 && !(reinterpret_cast<uintptr_t>(p.partial) & 15)) {
    const long nb4l = ((long)p.Mall * (p.ldp >> 2) + 255) / 256;
    UDET_LAUNCH(conv_splitk_epilogue4_kernel, dim3((int)(nb4l > 4096 ? 4096 : nb4l)), dim3(256), 0, stream, p);
  } else UDET_LAUNCH(conv_splitk_epilogue_kernel<1>, dim3(nb), dim3(256), 0, stream, p);
  UDET_HIP(hipGetLastError());
  return UDET_OK;
}

static int g_force_bm = 0, g_force_bn = 0, g_force_ks = -1, g_force_ws = -1, g_force_fold = -1, g_force_tail = 0;
static int g_last_cfg = 0;  // kernel family / tile / split count of the most recent launch_conv (debug query)
int conv_last_config() { return g_last_cfg; }
void conv_force_config(int bm, int bn, int ks) {
  g_force_bm = bm & 0xffff; g_force_bn = bn;
  g_force_ks = ks < 0 ? ks : (ks & 0xff);
  g_force_tail = ks < 0 ? 0 : ((ks >> 8) & 0xff);  // ks + 256 r: tail split for r workgroup slots per CU (ks & 255 slices; 0: as many as fill a round)
  // bit 16: non-specialised, 17: LDS-DMA (wave-specialised), 18: tile kernel, 19: self-staging LDS-DMA (4 waves, BK 16);
  // bit 20: split-K through the second launch, bit 21: split-K folded into the last-arriving workgroup;
  // bit 22 / 23: LDS-DMA with a 3 / 4 stage ring; bit 24: the direct 2-channel-head kernels (conv_thin.hip) where a launch is eligible
  g_force_fold = (bm >> 20) & 1 ? 0 : ((bm >> 21) & 1 ? 1 : -1);
  // bit 25: the Winograd F(2x2,3x3) family (conv_wino.hip) where a launch is eligible; bm & 0xffff = variant (bit 0: 64 tiles x 64 channels / 128 x 32, bit 1: four / eight waves)
  g_force_ws = (bm >> 16) & 1 ? 0 : ((bm >> 17) & 1 ? 2 : ((bm >> 18) & 1 ? 3 : ((bm >> 19) & 1 ? 6 : ((bm >> 22) & 1 ? 4 : ((bm >> 23) & 1 ? 5 : ((bm >> 24) & 1 ? 7 : ((bm >> 25) & 1 ? 9 : -1)))))));
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_cfg(ConvParams& p, int ws, hipStream_t stream) {
  const int Mtot = p.N * p.OHq * p.OWq;  // (tail split: unsegmented launches only, run_cfg)
  dim3 grid(conv_xblocks(p, BM), (p.Cout + BN - 1) / BN, p.ksplit > 1 ? p.ksplit : 1);
  if (p.tail_ks > 1) {  // tail split (run_cfg checked the kernel family, the slab capacity and the alignment)
    const int mtiles = (Mtot + BM - 1) / BM;
    p.tail_prow0 = (p.tail_full / mtiles) * Mtot + (p.tail_full % mtiles) * BM;
    grid.x = p.tail_full + (grid.x - p.tail_full) * p.tail_ks;
    grid.z = 1;
  }
  if (ws == 6) {
    if constexpr (BM % 64 == 0 && BN % 64 == 0 && BM <= 128) {
      if (p.f16) UDET_LAUNCH((conv_igemm_dma4_kernel<BM, BN, 2, 2, true>), grid, dim3(256), 0, stream, p);
      else UDET_LAUNCH((conv_igemm_dma4_kernel<BM, BN, 2, 2>), grid, dim3(256), 0, stream, p);
    } else if constexpr (BN == 32 && BM % 128 == 0) {
      if (p.f16) UDET_LAUNCH((conv_igemm_dma4_kernel<BM, BN, 4, 1, true>), grid, dim3(256), 0, stream, p);
      else UDET_LAUNCH((conv_igemm_dma4_kernel<BM, BN, 4, 1>), grid, dim3(256), 0, stream, p);
    } else {
      set_error("conv: no self-staging kernel for tile %dx%d", BM, BN);
      return UDET_ERR_UNSUPPORTED;
    }
  }
  else if (ws == 2 && p.f16) UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 2, true>), grid, dim3(512), 0, stream, p);
  else if ((ws == 4 || ws == 5) && p.f16) UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 3, true>), grid, dim3(512), 0, stream, p);
  else if (ws == 2) UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 2>), grid, dim3(512), 0, stream, p);
  else if (ws == 4) UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 3>), grid, dim3(512), 0, stream, p);
  else if (ws == 5) {
    if constexpr (BM <= 128) UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 4>), grid, dim3(512), 0, stream, p);
    else UDET_LAUNCH((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N, 3>), grid, dim3(512), 0, stream, p);
  }
  else if (ws) UDET_LAUNCH((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, true>), grid, dim3(512), 0, stream, p);
  else UDET_LAUNCH((conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false>), grid, dim3(256), 0, stream, p);
  UDET_HIP(hipGetLastError());
  if (p.tail_ks > 1) {
    const long nb4l = ((long)(p.Mall - p.tail_prow0) * (p.ldp >> 2) + 255) / 256;
    UDET_LAUNCH(conv_splitk_epilogue4_kernel, dim3((int)(nb4l > 4096 ? 4096 : nb4l)), dim3(256), 0, stream, p);
    UDET_HIP(hipGetLastError());
  } else if (p.ksplit > 1 && !p.fold) {
    UDET_TRY(launch_splitk_second_pass(p, stream));
  }
  return UDET_OK;
}

// ---- tile / split-K selection ---------------------------------------------------------------
struct ConvCfg { int bm, bn, ks, ws, fold, tail; };  // fold: split-K summed by the last-arriving workgroup (no second launch);
                                                    // tail > 0: x-blocks [0, tail) unsplit, the rest cut into ks slices (ConvParams::tail_full)
static long cfg_tiles(const ConvParams& p, int bm, int bn) {
  return (long)conv_xblocks(p, bm) * ((p.Cout + bn - 1) / bn);
}
static int max_class_taps(const ConvParams& p) {
  int mx = 0;
  for (int c = 0; c < p.nseg; ++c) mx = p.seg_tap[c + 1] - p.seg_tap[c] > mx ? p.seg_tap[c + 1] - p.seg_tap[c] : mx;
  if (p.nseg) return mx;
  for (int c = 0; c < p.ncls; ++c) mx = p.cls_tap[c + 1] - p.cls_tap[c] > mx ? p.cls_tap[c + 1] - p.cls_tap[c] : mx;
  return mx;
}
static int max_ksplit(const ConvParams& p) {  // capacity / minimum-work bound on the split count
  if (!p.partial) return 1;
  const int nchunks = (max_class_taps(p) * p.Kc + 31) / 32;
  int ks = nchunks / 2 > 64 ? 64 : nchunks / 2;
  const size_t per_split = (size_t)p.Mall * ((p.Cout + 3) & ~3);
  while (ks > 1 && per_split * ks > p.partial_cap) --ks;
  return ks < 1 ? 1 : ks;
}
struct TileGeoms;
size_t conv_tile_lds_bytes(const ConvParams& p, int th, int cbmax, TileGeoms* gout, bool* big);
int launch_conv_tile(const ConvParams& p, int th, int cbmax, hipStream_t stream);
static bool tile_ok(const ConvParams& p, int th, int cb = 32) {
  if (p.nseg) return false;  // segmented launches: implicit-GEMM families only
  if (cb == 16 && p.Kc <= 16) return false;  // (the same launch as cb = 32)
  const size_t b = conv_tile_lds_bytes(p, th, cb, nullptr, nullptr);
  // (<= 256 columns: the kernel walks 32-column blocks as blockIdx.y and re-reads the halo per block -- thin inputs with wide outputs,
  // the backward-data view of the recover decoder: 32 -> 194 channels 163 -> 150 us; beyond that the implicit GEMM always won)
  return b > 0 && b <= 96 * 1024 && p.Kc <= 256 && p.Cout <= 256;
}
static bool self_staging_tile(int bm, int bn) {  // tiles conv_igemm_dma4_kernel is instantiated for
  return ((bm == 128 || bm == 64) && (bn == 64 || bn == 128)) || (bn == 32 && (bm == 128 || bm == 256));
}
static bool dma_ok(const ConvParams& p) { return p.xa == nullptr && p.zero16 != nullptr && !(reinterpret_cast<uintptr_t>(p.zero16) & 15); }
// tail split for workgroups filling r slots per CU: x-blocks of the whole rounds stay unsplit (*full_x of them), the rest is cut
// into *ks slices so that it fills one more round.  false: the tile count is a whole number of rounds, or less than one.
static bool tail_for_rounds(const ConvParams& p, int bm, int bn, int r, int kcap, int* full_x, int* ks) {
  if (p.nseg) return false;
  const int Mtot = p.N * p.OHq * p.OWq, X = p.ncls * ((Mtot + bm - 1) / bm), Y = (p.Cout + bn - 1) / bn;
  const long S = 256L * r, T = (long)X * Y, fullT = T / S * S;
  if (fullT == 0 || fullT == T) return false;
  const int fx = (int)(fullT / Y), rem_x = X - fx;
  if (fx <= 0 || rem_x <= 0) return false;
  long k = S / ((long)rem_x * Y);
  if (k > kcap) k = kcap;
  if (k < 2) return false;
  *full_x = fx;
  *ks = (int)k;
  return true;
}
static ConvCfg heuristic_cfg(const ConvParams& p) {
  // N tile from the channel count; M tile shrunk while the launch would leave CUs without a workgroup
  ConvCfg c;
  c.ws = 1; c.fold = 0; c.tail = 0;
  if (p.Cout <= 32) { c.bn = 32; c.bm = 256; if (cfg_tiles(p, 256, 32) < 384) c.bm = 128; }
  else if (p.Cout <= 64) { c.bn = 64; c.bm = 128; if (cfg_tiles(p, 128, 64) < 384) c.bm = 64; }
  else if (p.Cout <= 96) { c.bn = 96; c.bm = 128; }
  else { c.bn = 128; c.bm = 128; if (cfg_tiles(p, 128, 128) < 320) { c.bn = 64; if (cfg_tiles(p, 128, 64) < 384) c.bm = 64; } }
  const long tiles = cfg_tiles(p, c.bm, c.bn);
  c.ks = 1;
  if (tiles < 256) {
    int ks = (int)((512 + tiles - 1) / tiles);
    const int cap = max_ksplit(p), half = cap / 2 > 0 ? cap / 2 : 1;  // keep >= 4 stages per split
    c.ks = ks > half ? half : ks;
  }
  c.fold = 0;  // measured (r2a): the release / acquire of the folded form costs more than the second launch on almost every shape;
               // the tuner still tries it for its winner
  if (p.f16 && dma_ok(p)) c.ws = 2;  // fp16 multiplication exists in the LDS-DMA families only
  return c;
}
static int run_cfg(ConvParams& p, const ConvCfg& c, hipStream_t stream) {
  if (c.ws == 7 || c.ws == 8) {  // direct kernels for the 2-channel heads (conv_thin.hip): 7 two input channels, 8 two output channels
    p.ksplit = 1; p.fold = 0; p.tail_full = 0; p.tail_ks = 0;
    return c.ws == 7 ? launch_conv_thin_k(p, stream) : launch_conv_thin_n(p, stream);
  }
  if (c.ws == 9) return launch_conv_wino(p, c.bm, c.ks, stream);  // Winograd F(2x2,3x3) (conv_wino.hip); bm carries the variant
  if (c.ws == 3) {  // tile-resident direct convolution (conv_tile.hip); bm carries the tile height
    p.ksplit = 1;
    p.fold = 0;
    return launch_conv_tile(p, c.bm, c.bn == 16 ? 16 : 32, stream);  // bn carries the channels per pass
  }
  p.ksplit = c.ks > 1 ? c.ks : 1;
  p.fold = 0;
  p.tail_full = 0; p.tail_ks = 0; p.tail_prow0 = 0;
  if (p.ksplit > 1) {
    p.ldp = (p.Cout + 3) & ~3;
    p.fold = c.fold && p.tickets && cfg_tiles(p, c.bm, c.bn) <= UDET_MAX_TICKETS;
    const int Mtot = p.N * p.OHq * p.OWq, mtiles = (Mtot + c.bm - 1) / c.bm, xb = p.ncls * mtiles;
    if (c.tail > 0 && c.tail < xb && !p.nseg && (c.ws == 2 || c.ws == 4 || c.ws == 5) && !(reinterpret_cast<uintptr_t>(p.partial) & 15)) {
      const long prow0 = (long)(c.tail / mtiles) * Mtot + (long)(c.tail % mtiles) * c.bm;
      if ((size_t)((long)p.ncls * Mtot - prow0) * p.ldp * p.ksplit <= p.partial_cap) {
        p.tail_full = c.tail; p.tail_ks = p.ksplit; p.ksplit = 1; p.fold = 0;
      }
    }
  }
  if (c.bm == 256 && c.bn == 32) return launch_cfg<256, 32, 32, 4, 1>(p, c.ws, stream);
  if (c.bm == 128 && c.bn == 32) return launch_cfg<128, 32, 32, 4, 1>(p, c.ws, stream);
  if (c.bm == 128 && c.bn == 64) return launch_cfg<128, 64, 32, 2, 2>(p, c.ws, stream);
  if (c.bm == 64 && c.bn == 64) return launch_cfg<64, 64, 32, 2, 2>(p, c.ws, stream);
  if (c.bm == 128 && c.bn == 96) return launch_cfg<128, 96, 32, 4, 1>(p, c.ws, stream);
  if (c.bm == 128 && c.bn == 128) return launch_cfg<128, 128, 32, 2, 2>(p, c.ws, stream);
  set_error("conv: no kernel for tile %dx%d", c.bm, c.bn);
  return UDET_ERR_UNSUPPORTED;
}

// ---- autotuner: while tuning is on, the first launch of every distinct problem shape times its candidate
// (tile, split-K, wave-specialisation) configurations on the caller's stream and caches the fastest -------------
static std::unordered_map<uint64_t, ConvCfg> g_cache;
static std::mutex g_cache_mu;
static int g_tuning = 0;
static void tune_scratch_free();
void conv_set_tuning(int on) { g_tuning = on; if (!on) tune_scratch_free(); }
int conv_tuned_shapes() { std::lock_guard<std::mutex> l(g_cache_mu); return (int)g_cache.size(); }
void conv_clear_tuning() { std::lock_guard<std::mutex> l(g_cache_mu); g_cache.clear(); }
// text form of the cache ("c <problem key> bm bn ks ws fold tail" per line): lets a second process (a rocprofv3 trace of timed
// steps only) run exactly the configurations a tuning run picked
void conv_tune_dump(FILE* f) {
  std::lock_guard<std::mutex> l(g_cache_mu);
  for (auto& kv : g_cache)
    fprintf(f, "c %llu %d %d %d %d %d %d\n", (unsigned long long)kv.first, kv.second.bm, kv.second.bn, kv.second.ks, kv.second.ws, kv.second.fold,
            kv.second.tail);
}
void conv_tune_put(unsigned long long key, int bm, int bn, int ks, int ws, int fold, int tail) {
  std::lock_guard<std::mutex> l(g_cache_mu);
  g_cache[(uint64_t)key] = ConvCfg{bm, bn, ks, ws, fold, tail};
}

// ---- candidate verification -------------------------------------------------------------------------------------------
// The tuner selects on time; a configuration that is fast because it computes something else must never be cached.  Before
// a winner is stored its output on the tuning data is compared (max-abs, relative to the largest reference element) with the
// output of the reference configuration (built-in heuristic, register-staged wave-specialised kernel).  The two result
// buffers are temporary device allocations that live only while tuning is on (the one place where the library allocates).
static float* g_vbuf[3] = {nullptr, nullptr, nullptr};  // two result buffers + {max|a-b|, max|a|}
static size_t g_vcap = 0;
static int g_rejected = 0;
int conv_tune_rejected() { return g_rejected; }
void conv_tune_note_reject() { ++g_rejected; }
float* tune_scratch(size_t floats, int which) {
  if (floats > g_vcap) {
    for (int i = 0; i < 2; ++i) { if (g_vbuf[i]) (void)hipFree(g_vbuf[i]); g_vbuf[i] = nullptr; }
    g_vcap = floats + floats / 4;
    for (int i = 0; i < 2; ++i)
      if (hipMalloc(reinterpret_cast<void**>(&g_vbuf[i]), g_vcap * sizeof(float)) != hipSuccess) { g_vbuf[i] = nullptr; g_vcap = 0; return nullptr; }
  }
  if (!g_vbuf[2] && hipMalloc(reinterpret_cast<void**>(&g_vbuf[2]), 2 * sizeof(float)) != hipSuccess) return nullptr;
  return g_vbuf[which];
}
static void tune_scratch_free() {
  for (int i = 0; i < 3; ++i) { if (g_vbuf[i]) (void)hipFree(g_vbuf[i]); g_vbuf[i] = nullptr; }
  g_vcap = 0;
}
__global__ __launch_bounds__(256) void tune_maxdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
  float d = 0.f, m = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float x = a[e], y = b[e];
    float df = fabsf(x - y);
    if (!(df == df)) df = 3.0e38f;  // NaN in either result
    d = fmaxf(d, df);
    m = fmaxf(m, fabsf(x));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { d = fmaxf(d, __shfl_xor(d, o)); m = fmaxf(m, __shfl_xor(m, o)); }
  if ((threadIdx.x & 63) == 0) {  // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(d));
    atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(m));
  }
}
// max|a-b| <= 2e-4 * max|a| + 1e-6 ?  (a = reference; split-K orders differ by ~1e-6 relative)
bool tune_compare(const float* a, const float* b, size_t n, hipStream_t stream, float* diff_out, float* scale_out) {
  float* res = g_vbuf[2];
  if (!res) return false;
  if (hipMemsetAsync(res, 0, 2 * sizeof(float), stream) != hipSuccess) return false;
  long nbl = ((long)n + 255) / 256;
  hipLaunchKernelGGL(tune_maxdiff_kernel, dim3((int)(nbl > 2048 ? 2048 : nbl)), dim3(256), 0, stream, a, b, (long)n, res);
  float h[2] = {3.0e38f, 0.f};
  if (hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return false;
  if (diff_out) *diff_out = h[0];
  if (scale_out) *scale_out = h[1];
  return h[0] <= 2e-4f * h[1] + 1e-6f;
}

static uint64_t conv_key(const ConvParams& p) {
  const int f[] = {p.N, p.H, p.W, p.up_shift, p.Kc, p.Cout, p.ntaps, p.ncls, p.OHq, p.OWq, p.isy, p.osy, p.xa ? 1 : 0,
                   p.ldx, p.ldy, p.accumulate, p.res ? 1 : 0, p.y2 ? 1 : 0, p.partial ? 1 : 0, p.cls_tap[1], p.uo ? 1 : 0, p.f16 ? 1 : 0, p.kreal,
                   p.wino_u ? p.wino_np : 0, p.ntaps > 0 ? p.taps[0].dy : 0, p.ntaps > 0 ? p.taps[0].dx : 0};  // (first tap: the dilation -- it
                                                                                         // decides what the Winograd sub-lattices look like)
  uint64_t h = 1469598103934665603ull;
  for (int v : f) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
  for (int s = 0; s < p.nseg; ++s)  // segmented launches: the segment grids and their tap counts (ncls / OHq / OWq / taps[] are zero)
    for (int v : {p.seg[s].oy, p.seg[s].ox, p.seg[s].h, p.seg[s].w, p.seg_tap[s + 1]}) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
  return h;
}
static float time_cfg(ConvParams& p, const ConvCfg& c, int reps, hipStream_t stream) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
  if (run_cfg(p, c, stream) != UDET_OK) return 1e30f;  // warm-up
  (void)hipEventRecord(e0, stream);
  for (int r = 0; r < reps; ++r) run_cfg(p, c, stream);
  (void)hipEventRecord(e1, stream);
  if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
static ConvCfg tune_cfg_impl(ConvParams& p, hipStream_t stream);
static ConvCfg tune_cfg(ConvParams& p, hipStream_t stream) {
  // candidates are launched hundreds of times: an accumulating launch would grow its output with every repetition and the layers
  // behind it would be tuned (and verified) on ever larger data -- beyond the fp16 range in fp16 mode.  Timed as plain stores (one
  // read of the output less per element).
  const int acc = p.accumulate;
  p.accumulate = 0;
  const ConvCfg c = tune_cfg_impl(p, stream);
  p.accumulate = acc;
  return c;
}
static ConvCfg tune_cfg_impl(ConvParams& p, hipStream_t stream) {
  const ConvCfg h = heuristic_cfg(p);
  std::vector<ConvCfg> cand;
  static const int TILES[6][2] = {{256, 32}, {128, 32}, {128, 64}, {64, 64}, {128, 96}, {128, 128}};
  const int kcap = max_ksplit(p);
  for (auto& t : TILES) {
    const int bm = t[0], bn = t[1];
    // N tiles wider than needed waste MFMA columns; much narrower ones re-read the A operand
    if (p.Cout <= 32 && bn != 32) continue;
    if (p.Cout > 32 && p.Cout <= 64 && bn > 64) continue;
    if (p.Cout > 64 && p.Cout <= 96 && bn != 96 && bn != 32) continue;
    if (p.Cout > 96 && bn < 64) continue;
    if (p.Cout > 96 && bn == 96 && p.Cout % 96 != 0 && p.Cout <= 128) continue;
    const long tiles = cfg_tiles(p, bm, bn);
    std::vector<int> kss;
    for (int ks = 1; ks <= kcap; ks *= 2) kss.push_back(ks);
    // split counts that fill whole rounds of the 256 CUs (tiles*ks just below a multiple of 256): a 144-tile layer runs
    // at 144/256 of the chip unsplit and at 1008/1024 with 7 splits
    if (tiles < 512)
      for (int k = 1; k <= 6; ++k) {
        const int ks = (int)(256L * k / tiles);
        if (ks >= 3 && ks <= kcap && (ks & (ks - 1)) != 0 && std::find(kss.begin(), kss.end(), ks) == kss.end()) kss.push_back(ks);
      }
    for (int ks : kss) {
      if (ks > 1 && (tiles >= 512 || tiles * ks > 4096)) continue;
      if (tiles * ks < 96 && ks * 2 <= kcap) continue;  // hopelessly under-filled
      cand.push_back({bm, bn, ks, 1, 0});
    }
  }
  cand.push_back(h);
  ConvCfg best = h;
  float best_ms = 1e30f;
  for (auto& c : cand) {
    float ms = time_cfg(p, c, 3, stream);
    if (ms < best_ms * 1.15f && ms < 0.25f) ms = 0.5f * (ms + time_cfg(p, c, 6, stream));  // short launches: re-time the contenders
    if (ms < best_ms) { best_ms = ms; best = c; }
  }
  ConvCfg alt = best;
  alt.ws = 0;
  float a = time_cfg(p, best, 5, stream), b = time_cfg(p, alt, 5, stream);
  if (b < a * 0.97f) { best = alt; a = b; }
  const bool f16 = p.f16 && dma_ok(p);  // only LDS-DMA configurations multiply in fp16: every candidate must, or results differ per shape
  if (f16) { best = h; a = b = 1e30f; }
  if (dma_ok(p)) {  // LDS-DMA staging: re-scan the tiles, the balance between staging and MFMA waves differs
    for (auto& c : cand) {
      ConvCfg d = c;
      for (int ws : {2, 4, 5, 6}) {  // wave-specialised / self-staging (4 waves, 16-wide stages, 3-4 workgroups per CU)
        if (ws == 6 && !self_staging_tile(d.bm, d.bn)) continue;
        d.ws = ws;
        const float ms = time_cfg(p, d, 3, stream);
        if (ms < a * 0.98f) {
          const float ms5 = time_cfg(p, d, 5, stream);
          if (ms5 < a * 0.98f) { a = ms5; best = d; }
        }
        // Tail split of an unsplit candidate: a launch whose workgroups fill r slots per CU for k whole rounds and a fraction of
        // another runs that last round on part of the chip (576 tiles of 64x64 on 256 CUs: three on 64 CUs, two on the rest).
        // Cutting only the LAST round's tiles into K slices makes it a full round of short workgroups, at the slab traffic of
        // those tiles alone.
        if (d.ks <= 1 && ws != 6 && p.partial && kcap >= 2 && ms < a * 1.3f) {
          long seen[4] = {0, 0, 0, 0};
          for (int r = 1; r <= 4; ++r) {
            int full_x = 0, ks = 0;
            if (!tail_for_rounds(p, d.bm, d.bn, r, kcap, &full_x, &ks)) continue;
            const long id = (long)full_x * 1024 + ks;
            if (id == seen[0] || id == seen[1] || id == seen[2]) continue;
            seen[r - 1] = id;
            ConvCfg e = d;
            e.ks = ks; e.tail = full_x; e.fold = 0;
            const float mt = time_cfg(p, e, 3, stream);
            if (mt < a * 0.98f) {
              const float mt5 = time_cfg(p, e, 5, stream);
              if (mt5 < a * 0.98f) { a = mt5; best = e; }
            }
          }
        }
      }
    }
    b = a;
  }
  for (int thcb : {8 * 64 + 32, 4 * 64 + 32, 8 * 64 + 16, 4 * 64 + 16}) {  // thin layers: tile-resident direct convolution (multiplies in fp16 too when asked to)
    const int th = thcb >> 6, cb = thcb & 63;  // tile height x channels resident per pass
    if (!tile_ok(p, th, cb)) continue;
    const ConvCfg d = {th, cb, 1, 3, 0};
    const float ms = time_cfg(p, d, 3, stream);
    if (ms < a * 0.97f) {
      const float ms5 = time_cfg(p, d, 5, stream);
      if (ms5 < a * 0.97f) { a = b = ms5; best = d; }
    }
  }
  for (int ws : {7, 8}) {  // the direct kernels for 2-channel inputs / outputs
    if (!(ws == 7 ? conv_thin_k_ok(p) : conv_thin_n_ok(p)) || p.f16) continue;  // (fp16 mode: every configuration must multiply alike)
    const ConvCfg d = {0, 0, 1, ws, 0, 0};
    const float ms = time_cfg(p, d, 5, stream);
    if (getenv("UDET_TUNE_LOG") && atoi(getenv("UDET_TUNE_LOG")) > 1)
      fprintf(stderr, "[udet tune]   direct family %d: %.1f us against %.1f (N=%d %dx%d Kc=%d taps=%d Cout=%d)\n", ws, ms * 1e3f, (a < b ? a : b) * 1e3f, p.N, p.OHq, p.OWq, p.Kc, p.ntaps, p.Cout);
    if (ms < (a < b ? a : b) * 0.97f) { a = b = ms; best = d; }
  }
  if (conv_wino_ok(p)) {  // Winograd F(2x2,3x3): 2.25x fewer multiplications; K slices where the tiles do not fill the chip
    for (int v = 0; v < 5; ++v) {  // (bit 0: tile shape, bit 1: four / eight waves; 4: the half-size form, two workgroups per CU)
      if (!conv_wino_variant_ok(p, v)) continue;
      const long wgs = conv_wino_workgroups(p, v);
      const int cap = conv_wino_max_ksplit(p, v);
      std::vector<int> kss = {1};
      if (wgs < 256 && cap >= 2) kss.push_back(2);  // (under one round of workgroups: two K slices even where no whole round results)
      if (wgs < 384)
        for (int r = 1; r <= 3; ++r) {
          const int ks = (int)(256L * r / (wgs > 0 ? wgs : 1));
          if (ks >= 2 && ks <= cap && std::find(kss.begin(), kss.end(), ks) == kss.end()) kss.push_back(ks);
        }
      for (int ks : kss) {
        const ConvCfg d = {v, 0, ks, 9, 0, 0};
        const float ms = time_cfg(p, d, 3, stream);
        if (getenv("UDET_TUNE_LOG") && atoi(getenv("UDET_TUNE_LOG")) > 1)
          fprintf(stderr, "[udet tune]   winograd variant %d ks=%d (%ld workgroups): %.1f us against %.1f (N=%d %dx%d Kc=%d Cout=%d)\n", v, ks, wgs, ms * 1e3f,
                  (a < b ? a : b) * 1e3f, p.N, p.OHq, p.OWq, p.Kc, p.Cout);
        if (ms < (a < b ? a : b) * 0.97f) {
          const float ms5 = time_cfg(p, d, 5, stream);
          if (ms5 < (a < b ? a : b) * 0.97f) { a = b = ms5; best = d; }
        }
      }
    }
  }
  if (best.ks > 1 && best.ws != 3 && best.ws < 7 && best.tail == 0) {  // the other way of summing the slabs: last-arriving workgroup <-> second launch
    const ConvCfg w = best;
    for (int ks : {w.ks, w.ks / 2, w.ks / 4}) {  // the folded form sums its slabs in one workgroup: fewer slabs may suit it better
      if (ks < 2) continue;
      ConvCfg d = w;
      d.fold = !w.fold;
      d.ks = ks;
      const float ms = time_cfg(p, d, 5, stream);
      if (ms < (a < b ? a : b) * 0.98f) { a = b = ms; best = d; }
    }
  }
  // verification against the reference configuration on the tuning data (see above)
  if (best.bm != h.bm || best.bn != h.bn || best.ks != h.ks || best.ws != h.ws || best.fold != h.fold || best.tail != h.tail) {
    const int ld = (p.Cout + 3) & ~3;
    const size_t n = (size_t)p.N * p.OH * p.OW * ld;
    float* r0 = tune_scratch(n, 0);
    float* r1 = tune_scratch(n, 1);
    bool ok = false;
    float diff = 0.f, scale = 0.f;
    if (r0 && r1) {
      ConvParams q = p;
      q.ldy = ld; q.y_coff = 0; q.accumulate = 0; q.y2 = nullptr; q.uo = nullptr;
      (void)hipMemsetAsync(r0, 0, n * sizeof(float), stream);
      (void)hipMemsetAsync(r1, 0, n * sizeof(float), stream);
      q.y = r0;
      int rc = run_cfg(q, h, stream);
      q.y = r1;
      if (rc == UDET_OK) rc = run_cfg(q, best, stream);
      ok = rc == UDET_OK && tune_compare(r0, r1, n, stream, &diff, &scale);
    }
    if (!ok) {
      fprintf(stderr, "[udet tune] REJECTED N=%d %dx%d Kc=%d taps=%d cls=%d Cout=%d: %dx%d ks=%d ws=%d fold=%d tail=%d differs from the reference "
              "configuration (max|diff| %.3e, scale %.3e); keeping the heuristic\n", p.N, p.OHq, p.OWq, p.Kc, p.ntaps, p.ncls, p.Cout,
              best.bm, best.bn, best.ks, best.ws, best.fold, best.tail, diff, scale);
      conv_tune_note_reject();
      best = h;
    }
  }
  if (getenv("UDET_TUNE_LOG"))
    fprintf(stderr, "[udet tune] N=%d %dx%d Kc=%d taps=%d cls=%d Cout=%d -> %dx%d ks=%d ws=%d fold=%d tail=%d  %.1f us (heuristic %dx%d ks=%d)\n", p.N,
            p.OHq, p.OWq, p.Kc, p.ntaps, p.ncls, p.Cout, best.bm, best.bn, best.ks, best.ws, best.fold, best.tail, (a < b ? a : b) * 1e3f, h.bm, h.bn,
            h.ks);
  return best;
}

static int g_debug_f16 = 0;  // test hook: fp16 multiplication for the single-operator entry points too
void conv_debug_f16(int on) { g_debug_f16 = on; }
int conv_debug_f16_on() { return g_debug_f16; }
// argument checks + the derived fields every kernel family reads (Mall, fast divisors, uniform-cursor flags, segment rows)
static int conv_prepare(ConvParams& p) {
  if (g_debug_f16) p.f16 = 1;
  if (p.f16 && !(p.f16_xscale > 0.f)) p.f16_xscale = 1.f;
  // the tuning pass repeats every launch hundreds of times on random data, accumulating launches included: its "gradients" are far
  // larger than real ones and would overflow fp16 under the 4096 scale (every candidate NaN, every shape rejected); the scale does
  // not change a launch's duration
  if (p.f16 && g_tuning) p.f16_xscale = 1.f;
  if (p.Kc % 4 != 0 || p.ldx % 4 != 0 || p.x_coff % 4 != 0 || p.ldw % 4 != 0) {
    set_error("conv: Kc=%d ldx=%d x_coff=%d ldw=%d violate the 4-float alignment contract", p.Kc, p.ldx, p.x_coff, p.ldw);
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
  if (p.nseg) {
    if (p.nseg < 0 || p.nseg > UDET_MAX_SEGS || !p.tap_tab || p.up_shift || p.xa) {
      set_error("conv: malformed segmented launch");
      return UDET_ERR_ARG;
    }
    int prow = 0;
    for (int s = 0; s < p.nseg; ++s) {
      ConvSeg& g = p.seg[s];
      if (g.h < 1 || g.w < 1 || p.seg_tap[s + 1] - p.seg_tap[s] > UDET_MAX_TAPS || p.seg_tap[s + 1] < p.seg_tap[s]) {
        set_error("conv: segment %d is empty or has more than %d taps", s, UDET_MAX_TAPS);
        return UDET_ERR_SHAPE;
      }
      g.prow0 = prow;
      prow += p.N * g.h * g.w;
      g.fd_hw = make_fastdiv((unsigned)(g.h * g.w));
      g.fd_w = make_fastdiv((unsigned)g.w);
    }
    p.Mall = prow;
    p.ncls = 1; p.ntaps = 0; p.cls_tap[0] = p.cls_tap[1] = 0;
    p.OHq = p.seg[0].h; p.OWq = p.seg[0].w;  // (what the untuned heuristics and the log lines look at: the first, largest segment)
  } else {
    if (p.ncls != 4) {
      p.ncls = 1;
      p.cls_tap[0] = 0;
      p.cls_tap[1] = p.ntaps;
    }
    p.Mall = p.ncls * p.N * p.OHq * p.OWq;
  }
  p.fd_ohw = make_fastdiv((unsigned)(p.OHq * p.OWq));
  p.fd_ow = make_fastdiv((unsigned)p.OWq);
  // uniform K cursors (no up-sampled read).  bit 0: Kc >= 32, 32-wide stages (wave-specialised kernel); bit 1: Kc >= 16, 16-wide
  // stages (self-staging kernel); bit 2: Kc in {4, 8, 16}, 32 / Kc whole taps per 32-wide stage (wave-specialised kernel)
  p.kfast = 0;
  if (p.up_shift == 0) p.kfast = p.Kc >= 32 ? 3 : ((p.Kc >= 16 ? 2 : 0) | ((p.Kc == 4 || p.Kc == 8 || p.Kc == 16) ? 4 : 0));
  return UDET_OK;
}
int launch_conv(ConvParams& p, hipStream_t stream) {
  UDET_TRY(conv_prepare(p));
  ConvCfg c;
  bool have = false;
  const uint64_t key = conv_key(p);
  {
    std::lock_guard<std::mutex> l(g_cache_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { c = it->second; have = true; }
  }
  if (have) {
    // a cached entry may come from a file (udet_tune_load): never trust it beyond what the launcher would choose itself --
    // only instantiated tiles / families, the split count inside this launch's capacity, folding only where tickets exist
    static const int TILES[6][2] = {{256, 32}, {128, 32}, {128, 64}, {64, 64}, {128, 96}, {128, 128}};
    bool tile = false;
    for (auto& t : TILES) tile = tile || (c.bm == t[0] && c.bn == t[1]);
    if (c.ws == 3) tile = (c.bm == 4 || c.bm == 8) && (c.bn == 16 || c.bn == 32);
    if (c.ws == 7 || c.ws == 8) tile = true;  // (the direct 2-channel kernels carry no tile; eligibility is re-checked below)
    if (c.ws == 9) tile = c.bm >= 0 && c.bm <= 4;  // (Winograd: bm carries the variant; eligibility is re-checked below)
    if (!tile || c.ws < 0 || c.ws > 9) {
      c = heuristic_cfg(p);
    } else {
      const int cap = c.ws == 9 ? 16 : max_ksplit(p);  // (launch_conv_wino clamps to its own capacity)
      if (c.ks < 1) c.ks = 1;
      if (c.ks > cap) { c.ks = cap; c.tail = 0; }
      c.fold = c.fold ? 1 : 0;
      if (c.tail < 0) c.tail = 0;
    }
  }
  if (!have) {
    if (g_tuning) {
      c = tune_cfg(p, stream);
      std::lock_guard<std::mutex> l(g_cache_mu);
      g_cache[key] = c;
    } else {
      c = heuristic_cfg(p);
      // untuned default for the 2-channel heads: the direct kernels (an order of magnitude fewer padded multiplications)
      // (not while a test pins an implicit-GEMM tile: udet_debug_force_conv)
      if (!g_force_bm && !p.f16 && conv_thin_n_ok(p)) c.ws = 8;
      else if (!g_force_bm && !p.f16 && conv_thin_k_ok(p)) c.ws = 7;
    }
  }
  if (g_force_bm && g_force_ws != 9) { c.bm = g_force_bm; c.bn = g_force_bn; }
  if (g_force_ks >= 0) { c.ks = (g_force_ks > max_ksplit(p) && g_force_ws != 9) ? max_ksplit(p) : g_force_ks; c.tail = 0; }
  if (g_force_ws >= 0) c.ws = g_force_ws;
  if (g_force_fold >= 0) c.fold = g_force_fold;
  if (g_force_tail > 0) {
    int fx = 0, k = 0;
    if (tail_for_rounds(p, c.bm, c.bn, g_force_tail, max_ksplit(p), &fx, &k)) { c.tail = fx; c.ks = c.ks >= 2 ? c.ks : k; c.fold = 0; }
  }
  if ((c.ws == 2 || c.ws == 4 || c.ws == 5 || c.ws == 6) && !dma_ok(p)) c.ws = 1;
  if (c.ws == 6 && !self_staging_tile(c.bm, c.bn)) c.ws = 2;
  if (g_force_ws == 3) { c.ws = 3; c.bm = (g_force_bm == 4) ? 4 : 8; c.bn = g_force_bn == 16 ? 16 : 32; }
  if (g_force_ws == 7) c.ws = conv_thin_k_ok(p) ? 7 : (conv_thin_n_ok(p) ? 8 : heuristic_cfg(p).ws);
  if ((c.ws == 7 && !conv_thin_k_ok(p)) || (c.ws == 8 && !conv_thin_n_ok(p)) || ((c.ws == 7 || c.ws == 8) && p.f16)) c = heuristic_cfg(p);
  if (c.ws == 3 && !tile_ok(p, c.bm, c.bn == 16 ? 16 : 32)) { c = heuristic_cfg(p); }
  if (g_force_ws == 9) {  // test / tool hook: a single-operator launch carries no transformed weights -- build them here, from the packed ones
    int d9, w9[9];
    if (!p.wino_u && !p.f16 && !p.xa && p.Kc % 8 == 0 && p.Kc >= 8 && conv_wino_geometry(p, &d9, w9)) {
      static float* g_wino_scratch = nullptr;
      static size_t g_wino_cap = 0;
      static std::mutex mu;
      std::lock_guard<std::mutex> l(mu);
      const int np9 = conv_wino_np(p.Cout);
      const size_t need = (size_t)(p.Kc / 8) * 16 * 2 * np9 * 4;
      if (need > g_wino_cap) {
        (void)hipStreamSynchronize(stream);
        if (g_wino_scratch) (void)hipFree(g_wino_scratch);
        g_wino_scratch = nullptr; g_wino_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&g_wino_scratch), need * sizeof(float)) == hipSuccess) g_wino_cap = need;
      }
      if (g_wino_scratch && launch_wino_from_packed(p, g_wino_scratch, np9, stream) == UDET_OK) { p.wino_u = g_wino_scratch; p.wino_np = np9; }
    }
    const int v9 = (g_force_bm & 7) == 4 ? 4 : (g_force_bm & 3);
    if (conv_wino_ok(p) && (conv_wino_variant_ok(p, v9) || conv_wino_variant_ok(p, v9 ^ 1))) {
      c.ws = 9; c.bm = conv_wino_variant_ok(p, v9) ? v9 : (v9 ^ 1); c.bn = 0; c.fold = 0; c.tail = 0;
      if (g_force_ks < 0) c.ks = 1;
    } else if (c.ws == 9) c = heuristic_cfg(p);
  }
  if (c.ws == 9 && (!conv_wino_ok(p) || !conv_wino_variant_ok(p, c.bm))) c = heuristic_cfg(p);
  g_last_cfg = (c.ws & 0xff) | ((c.bm & 0xfff) << 8) | ((c.ks & 0xff) << 20) | ((c.ks > 1 && c.fold && c.ws != 3 && p.tickets ? 1 : 0) << 28) |
               ((c.ks > 1 && c.tail > 0 ? 1 : 0) << 29);
  return run_cfg(p, c, stream);
}


// ---- pair launches (two problems, one grid; conv_igemm_dma_pair_kernel) -----------------------------------------------------------------
// Eligible: two unsegmented fp32 problems the LDS-DMA family can take, with the same K depth, output width, tap geometry, class
// structure and strides (batch, operands, epilogue may differ).  The pair has ONE configuration (tile, K slices, stage ring), tuned as
// a unit and cached under the pair's own key; ws < 0 in the cache = "these two are faster apart".
static std::unordered_map<uint64_t, ConvCfg> g_pair_cache;
static int g_force_pair = -1;  // test hook (libudet_debug): 1 pairs whatever the tuner thinks, 0 never pairs
void conv_force_pair(int on) { g_force_pair = on; }
static int g_last_pair = 0;
int conv_last_pair() { return g_last_pair; }
static bool pair_compatible(const ConvParams& a, const ConvParams& b) {
  if (a.nseg || b.nseg || a.f16 || b.f16 || a.up_shift || b.up_shift || !dma_ok(a) || !dma_ok(b)) return false;
  if (a.Kc != b.Kc || a.Cout != b.Cout || a.ldw != b.ldw || a.ntaps != b.ntaps || a.ncls != b.ncls) return false;
  if (a.isy != b.isy || a.isx != b.isx || a.osy != b.osy || a.osx != b.osx || a.kfast != b.kfast) return false;
  if (!a.partial || a.partial != b.partial) return false;  // (one scratch region, carved in two below)
  for (int c = 0; c <= a.ncls; ++c)
    if (a.cls_tap[c] != b.cls_tap[c]) return false;
  for (int t = 0; t < a.ntaps; ++t)
    if (a.taps[t].dy != b.taps[t].dy || a.taps[t].dx != b.taps[t].dx) return false;
  return true;
}
static uint64_t pair_key(const ConvParams& a, const ConvParams& b) {
  uint64_t h = conv_key(a) * 1099511628211ull ^ conv_key(b);
  h ^= 0x9e3779b97f4a7c15ull;
  return h * 1099511628211ull;
}
void conv_pair_tune_dump(FILE* f) {
  std::lock_guard<std::mutex> l(g_cache_mu);
  for (auto& kv : g_pair_cache) fprintf(f, "p %llu %d %d %d %d\n", (unsigned long long)kv.first, kv.second.bm, kv.second.bn, kv.second.ks, kv.second.ws);
}
void conv_pair_tune_put(unsigned long long key, int bm, int bn, int ks, int ws) {
  std::lock_guard<std::mutex> l(g_cache_mu);
  g_pair_cache[(uint64_t)key] = ConvCfg{bm, bn, ks, ws, 0, 0};
}
int conv_pair_tuned_shapes() { std::lock_guard<std::mutex> l(g_cache_mu); return (int)g_pair_cache.size(); }
void conv_pair_clear_tuning() { std::lock_guard<std::mutex> l(g_cache_mu); g_pair_cache.clear(); }

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_pair_cfg(ConvParams& a, ConvParams& b, int ws, hipStream_t stream) {
  ConvPair pp;
  pp.p[0] = a; pp.p[1] = b;
  pp.xa = conv_xblocks(a, BM);
  dim3 grid(pp.xa + conv_xblocks(b, BM), (a.Cout + BN - 1) / BN, a.ksplit > 1 ? a.ksplit : 1);
  if (ws == 4) UDET_LAUNCH((conv_igemm_dma_pair_kernel<BM, BN, WAVES_M, WAVES_N, 3>), grid, dim3(512), 0, stream, pp);
  else UDET_LAUNCH((conv_igemm_dma_pair_kernel<BM, BN, WAVES_M, WAVES_N, 2>), grid, dim3(512), 0, stream, pp);
  UDET_HIP(hipGetLastError());
  if (a.ksplit > 1) {
    const long na = ((long)a.Mall * (a.ldp >> 2) + 255) / 256, nb = ((long)b.Mall * (b.ldp >> 2) + 255) / 256;
    pp.xa = (int)(na > 2048 ? 2048 : na);
    const int xb = (int)(nb > 2048 ? 2048 : nb);
    UDET_LAUNCH(conv_splitk_epilogue4_pair_kernel, dim3(pp.xa + xb), dim3(256), 0, stream, pp);
    UDET_HIP(hipGetLastError());
  }
  return UDET_OK;
}
// slab capacity of a pair: both problems' K slices side by side in the launch lane's scratch
static int pair_max_ksplit(const ConvParams& a, const ConvParams& b) {
  const int nchunks = (max_class_taps(a) * a.Kc + 31) / 32;
  int ks = nchunks / 2 > 64 ? 64 : nchunks / 2;
  const size_t per_split = ((size_t)a.Mall + b.Mall) * ((a.Cout + 3) & ~3) + 64;
  while (ks > 1 && per_split * ks > a.partial_cap) --ks;
  return ks < 1 ? 1 : ks;
}
static int run_pair_cfg(ConvParams& a, ConvParams& b, const ConvCfg& c, hipStream_t stream) {
  const int cap = pair_max_ksplit(a, b);
  const int ks = c.ks > cap ? cap : (c.ks < 1 ? 1 : c.ks);
  for (ConvParams* q : {&a, &b}) {
    q->ksplit = ks; q->fold = 0; q->tail_full = 0; q->tail_ks = 0; q->tail_prow0 = 0;
    q->ldp = (q->Cout + 3) & ~3;
  }
  float* const base = a.partial;
  if (ks > 1) {
    if ((reinterpret_cast<uintptr_t>(base) & 15) != 0) { set_error("conv pair: unaligned split-K scratch"); return UDET_ERR_ALIGN; }
    size_t off = (size_t)ks * a.Mall * a.ldp;
    off = (off + 15) & ~(size_t)15;
    b.partial = base + off;
  }
  int rc;
  if (c.bm == 256 && c.bn == 32) rc = launch_pair_cfg<256, 32, 4, 1>(a, b, c.ws, stream);
  else if (c.bm == 128 && c.bn == 32) rc = launch_pair_cfg<128, 32, 4, 1>(a, b, c.ws, stream);
  else if (c.bm == 128 && c.bn == 64) rc = launch_pair_cfg<128, 64, 2, 2>(a, b, c.ws, stream);
  else if (c.bm == 64 && c.bn == 64) rc = launch_pair_cfg<64, 64, 2, 2>(a, b, c.ws, stream);
  else if (c.bm == 128 && c.bn == 96) rc = launch_pair_cfg<128, 96, 4, 1>(a, b, c.ws, stream);
  else if (c.bm == 128 && c.bn == 128) rc = launch_pair_cfg<128, 128, 2, 2>(a, b, c.ws, stream);
  else { set_error("conv pair: no kernel for tile %dx%d", c.bm, c.bn); rc = UDET_ERR_UNSUPPORTED; }
  b.partial = base;
  return rc;
}
static ConvCfg pair_heuristic(const ConvParams& a, const ConvParams& b) {
  ConvCfg c = heuristic_cfg(b.Mall >= a.Mall ? b : a);
  c.ws = 2; c.fold = 0; c.tail = 0;
  const long tiles = cfg_tiles(a, c.bm, c.bn) + cfg_tiles(b, c.bm, c.bn);
  c.ks = 1;
  if (tiles < 256) {
    const int cap = pair_max_ksplit(a, b), half = cap / 2 > 0 ? cap / 2 : 1;
    const int ks = (int)((512 + tiles - 1) / tiles);
    c.ks = ks > half ? half : ks;
  }
  return c;
}
static float time_calls(const std::function<int()>& fn, int reps, hipStream_t stream) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
  if (fn() != UDET_OK) return 1e30f;
  (void)hipEventRecord(e0, stream);
  for (int r = 0; r < reps; ++r) (void)fn();
  (void)hipEventRecord(e1, stream);
  if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
static ConvCfg tune_pair(ConvParams& a, ConvParams& b, hipStream_t stream) {
  const int acc_a = a.accumulate, acc_b = b.accumulate;
  a.accumulate = b.accumulate = 0;  // (see tune_cfg: repeated accumulation would grow the data)
  // what the two cost apart, each on its own tuned configuration (launch_conv tunes a shape the first time it sees it)
  auto apart = [&]() -> int { ConvParams x = a, y = b; int rc = launch_conv(x, stream); return rc != UDET_OK ? rc : launch_conv(y, stream); };
  (void)apart();
  float t_apart = time_calls(apart, 5, stream);
  t_apart = 0.5f * (t_apart + time_calls(apart, 5, stream));
  static const int TILES[6][2] = {{256, 32}, {128, 32}, {128, 64}, {64, 64}, {128, 96}, {128, 128}};
  const int kcap = pair_max_ksplit(a, b);
  ConvCfg best = {0, 0, 1, -1, 0, 0};
  float best_ms = 1e30f;
  for (auto& t : TILES) {
    const int bm = t[0], bn = t[1];
    if (a.Cout <= 32 && bn != 32) continue;
    if (a.Cout > 32 && a.Cout <= 64 && bn > 64) continue;
    if (a.Cout > 64 && a.Cout <= 96 && bn != 96 && bn != 32) continue;
    if (a.Cout > 96 && bn < 64) continue;
    if (a.Cout > 96 && bn == 96 && a.Cout % 96 != 0 && a.Cout <= 128) continue;
    const long tiles = cfg_tiles(a, bm, bn) + cfg_tiles(b, bm, bn);
    std::vector<int> kss;
    for (int ks = 1; ks <= kcap; ks *= 2) kss.push_back(ks);
    if (tiles < 512)
      for (int k = 1; k <= 4; ++k) {
        const int ks = (int)(256L * k / tiles);
        if (ks >= 3 && ks <= kcap && (ks & (ks - 1)) != 0 && std::find(kss.begin(), kss.end(), ks) == kss.end()) kss.push_back(ks);
      }
    for (int ks : kss) {
      if (ks > 1 && (tiles >= 512 || tiles * ks > 4096)) continue;
      if (tiles * ks < 96 && ks * 2 <= kcap) continue;
      for (int ws : {2, 4}) {
        const ConvCfg c = {bm, bn, ks, ws, 0, 0};
        float ms = time_calls([&]() { return run_pair_cfg(a, b, c, stream); }, 3, stream);
        if (ms < best_ms * 1.1f) ms = 0.5f * (ms + time_calls([&]() { return run_pair_cfg(a, b, c, stream); }, 6, stream));
        if (ms < best_ms) { best_ms = ms; best = c; }
      }
    }
  }
  // verification: each problem's output of the pair launch against its own stand-alone launch on the built-in configuration
  bool ok = best.ws >= 0;
  float diff = 0.f, scale = 0.f;
  for (int which = 0; ok && which < 2; ++which) {
    ConvParams& p = which ? b : a;
    const int ld = (p.Cout + 3) & ~3;
    const size_t n = (size_t)p.N * p.OH * p.OW * ld;
    float* r0 = tune_scratch(n, 0);
    float* r1 = tune_scratch(n, 1);
    if (!r0 || !r1) { ok = false; break; }
    (void)hipMemsetAsync(r0, 0, n * sizeof(float), stream);
    (void)hipMemsetAsync(r1, 0, n * sizeof(float), stream);
    ConvParams q = p;
    q.ldy = ld; q.y_coff = 0; q.accumulate = 0; q.y2 = nullptr; q.uo = nullptr;
    q.y = r0;
    int rc = run_cfg(q, heuristic_cfg(q), stream);
    ConvParams qa = a, qb = b;
    ConvParams& qq = which ? qb : qa;
    qq.ldy = ld; qq.y_coff = 0; qq.accumulate = 0; qq.y2 = nullptr; qq.uo = nullptr; qq.y = r1;
    if (rc == UDET_OK) rc = run_pair_cfg(qa, qb, best, stream);
    ok = rc == UDET_OK && tune_compare(r0, r1, n, stream, &diff, &scale);
  }
  if (best.ws >= 0 && !ok) {
    fprintf(stderr, "[udet tune] REJECTED pair N=%d+%d %dx%d Kc=%d taps=%d cls=%d Cout=%d: %dx%d ks=%d ws=%d differs from the stand-alone "
            "launches (max|diff| %.3e, scale %.3e); launching them apart\n", a.N, b.N, a.OHq, a.OWq, a.Kc, a.ntaps, a.ncls, a.Cout, best.bm, best.bn,
            best.ks, best.ws, diff, scale);
    conv_tune_note_reject();
    best.ws = -1;
  }
  if (getenv("UDET_TUNE_LOG"))
    fprintf(stderr, "[udet tune] pair N=%d+%d %dx%d Kc=%d taps=%d cls=%d Cout=%d -> %dx%d ks=%d ws=%d  %.1f us, apart %.1f us%s\n", a.N, b.N, a.OHq,
            a.OWq, a.Kc, a.ntaps, a.ncls, a.Cout, best.bm, best.bn, best.ks, best.ws, best_ms * 1e3f, t_apart * 1e3f,
            best_ms < t_apart * 0.97f ? "" : " (kept apart)");
  if (!(best_ms < t_apart * 0.97f)) best.ws = -1;
  a.accumulate = acc_a; b.accumulate = acc_b;
  return best;
}
// Both problems in one launch where that is eligible and (tuned) faster; otherwise the two ordinary launches, a first.
int launch_conv_pair(ConvParams& a, ConvParams& b, hipStream_t stream) {
  g_last_pair = 0;
  UDET_TRY(conv_prepare(a));
  UDET_TRY(conv_prepare(b));
  const bool forced_family = g_force_bm || g_force_ws >= 0 || g_force_ks >= 0;  // (a test pins a family for single launches: respect it)
  if (g_force_pair == 0 || (forced_family && g_force_pair != 1) || !pair_compatible(a, b)) {
    UDET_TRY(launch_conv(a, stream));
    return launch_conv(b, stream);
  }
  ConvCfg c;
  bool have = false;
  const uint64_t key = pair_key(a, b);
  {
    std::lock_guard<std::mutex> l(g_cache_mu);
    auto it = g_pair_cache.find(key);
    if (it != g_pair_cache.end()) { c = it->second; have = true; }
  }
  if (have && c.ws >= 0) {  // (an entry from a file: instantiated tiles and ring depths only)
    static const int TILES[6][2] = {{256, 32}, {128, 32}, {128, 64}, {64, 64}, {128, 96}, {128, 128}};
    bool tile = false;
    for (auto& t : TILES) tile = tile || (c.bm == t[0] && c.bn == t[1]);
    if (!tile || (c.ws != 2 && c.ws != 4)) c = pair_heuristic(a, b);
    if (c.ks < 1) c.ks = 1;
  }
  if (!have) {
    if (g_tuning) {
      c = tune_pair(a, b, stream);
      std::lock_guard<std::mutex> l(g_cache_mu);
      g_pair_cache[key] = c;
    } else {
      c = pair_heuristic(a, b);
    }
  }
  if (g_force_pair == 1 && c.ws < 0) c = pair_heuristic(a, b);
  if (c.ws < 0) {
    UDET_TRY(launch_conv(a, stream));
    return launch_conv(b, stream);
  }
  g_last_pair = 1;
  g_last_cfg = (c.ws & 0xff) | ((c.bm & 0xfff) << 8) | ((c.ks & 0xff) << 20) | (1 << 30);
  return run_pair_cfg(a, b, c, stream);
}

}  // namespace udet
