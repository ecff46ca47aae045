// Prototype + micro-benchmark of the fused Winograd F(2x2,3x3) fp32-MFMA convolution.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino_bench tools/wino_bench.hip && /tmp/wino_bench [shape ...]
// One workgroup = 4 waves (one per SIMD, 512-register budget).  A wave owns ALL 16 Winograd positions of a 32-tile x 32-channel
// block: 16 accumulators of v_mfma_f32_32x32x2_f32 = 256 registers, so the output transform A^T M A needs no cross-wave traffic.
// Workgroup tile: WT x WN waves = (32 WT) tiles x (32 WN) output channels.  K runs in stages of 8 input channels:
//   input  stage: the raw (2 TH + 2) x (2 TW + 2) pixel halo of the workgroup's tiles, [quad][row][column parity][column / 2][4 ch]
//                 (16-byte slots, row stride == 4 (mod 8) slots: the ds_read_b128 of the 4x4 patch are conflict-free)
//   weight stage: the pre-transformed U = G g G^T, [position][lane half][n][4 ch] = the global layout, 32 KB for 64 channels
// both land by global_load_lds_dwordx4 issued by the MFMA waves themselves (3-stage ring, one barrier per stage); the input
// transform B^T d B (32 adds per channel) sits on the LDS -> VGPR path; lanes 0-31 hold channels 4q..4q+3 of k-group q,
// lanes 32-63 channels 4q+4..4q+7: MFMA j of a position consumes component j of both halves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct WinoParams {
  const float* x;
  int ldx, x_coff, N, H, W, Kc;  // Kc: input channels (multiple of 4)
  const float* u;                // transformed weights [kg][nb][16][2][32*WN][4]
  int nkg;                       // ceil(Kc / 8)
  const float* bias;
  float* y;
  int ldy, y_coff, Cout;
  int dil;     // dilation: the d*d output sub-lattices are independent d = 1 problems
  int Hs, Ws;  // ceil(H / d), ceil(W / d)
  int BY, BX;  // workgroup blocks per sub-lattice image
  const float* zero16;
  float alpha;  // leaky slope
  long long* ts;  // optional [workgroup][8] time stamps (round 6: where the ~13 us of per-launch fixed cost go; v4 kernel only)
};

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int row_slots(int pw) { return (pw + 3) / 8 * 8 + 4; }  // smallest S >= pw with S % 8 == 4

// tiles of a workgroup: (4 WTY) x (8 WTX) with WT = WTY * WTX waves along the tile axes
template <int WTY, int WTX, int WN, int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_kernel(const WinoParams p) {
  static_assert(WTY * WTX * WN == 4, "4 waves");
  constexpr int TH = 4 * WTY, TW = 8 * WTX;       // tiles
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;  // halo pixels
  constexpr int CS = PW / 2;                       // used slots per column parity
  constexpr int S = row_slots(PW);                 // slots per halo row: >= PW, == 4 (mod 8)
  static_assert(S % 8 == 4 && S >= PW, "row stride");
  constexpr int HP = S / 2;                                  // slot offset of the odd-column half
  constexpr int IN_SLOTS = 2 * PH * S;                       // 16-byte slots of one input stage
  constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;           // DMA instructions per wave
  constexpr int IN_BYTES = IN_INSTR * 256 * 16;
  constexpr int BN = 32 * WN;
  constexpr int W_BYTES = 16 * 2 * BN * 16;
  constexpr int W_INSTR = W_BYTES / 4096;
  constexpr int STAGE = IN_BYTES + W_BYTES;
  constexpr int L = IN_INSTR + W_INSTR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wt = wave / WN, wty = wt / WTX, wtx = wt % WTX;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int d = p.dil;
  const int bx = bid % p.BX;
  int rem = bid / p.BX;
  const int by = rem % p.BY;
  rem /= p.BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;  // this sub-lattice's grid
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;                          // first output pixel (sub-lattice coordinates)
  const int nb = blockIdx.y;

  // ---- per-lane DMA sources (constant over the stages up to the channel offset) ----
  int in_off[IN_INSTR], in_q4[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int Lx = (i * 4 + wave) * 64 + lane;
    const int quad = Lx / (PH * S), r2 = Lx - quad * (PH * S);
    const int row = r2 / S, s = r2 - row * S;
    const int par = s / HP, cs = s - par * HP;
    const int col = 2 * cs + par;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool ok = quad < 2 && cs < CS && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    in_off[i] = ok ? ((n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + quad * 4 : -1;
    in_q4[i] = quad * 4;
  }
  const float* zero = p.zero16;
  const float* ubase = p.u + (size_t)nb * (W_BYTES / 4) + (size_t)lane * 4;
  const size_t ustride = (size_t)gridDim.y * (W_BYTES / 4);  // floats per k-group

  auto issue = [&](int kg, int buf) {
    char* sb = smem + buf * STAGE;
    const int c0 = kg * 8;
#pragma unroll
    for (int i = 0; i < IN_INSTR; ++i) {
      const float* src = (in_off[i] >= 0 && c0 + in_q4[i] < p.Kc) ? p.x + (in_off[i] + c0) : zero;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr)(sb + (i * 4 + wave) * 1024), 16, 0, 0);
    }
    const float* us = ubase + (size_t)kg * ustride;
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i)
      __builtin_amdgcn_global_load_lds(us + (i * 4 + wave) * 256, (lds_ptr)(sb + IN_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
  };

  floatx16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  // byte offset of this lane's patch origin inside an input stage
  const int a_base = ((lh * PH + 2 * (wty * 4 + ty)) * S + (wtx * 8 + tx)) * 16;
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;

  const int nkg = p.nkg;
  constexpr int LW = L;  // DMA instructions per wave and stage
  // prologue: NS - 1 stages in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkg) issue(s, s);
  int buf = 0;
  for (int kg = 0; kg < nkg; ++kg) {
    // stage kg has landed (loads retire in order: only the stages issued after it may still be in flight)
    if (NS > 2 && kg + NS - 2 < nkg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LW) : "memory");
    else if (NS > 3 && kg + NS - 3 < nkg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * LW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {  // every wave has left stage kg - 1: its buffer takes stage kg + NS - 1
      const int nk = kg + NS - 1;
      int nbuf = buf + NS - 1;
      nbuf = nbuf >= NS ? nbuf - NS : nbuf;
      if (nk < nkg) issue(nk, nbuf);
    }
    const char* sb = smem + buf * STAGE;
    float4 raw[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        raw[r][c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
    float4 bf[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bf[q] = *reinterpret_cast<const float4*>(sb + b_base + q * (2 * BN * 16));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float dd[4][4], t[4][4], v[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dd[r][c] = j == 0 ? raw[r][c].x : (j == 1 ? raw[r][c].y : (j == 2 ? raw[r][c].z : raw[r][c].w));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        t[0][c] = dd[0][c] - dd[2][c];
        t[1][c] = dd[1][c] + dd[2][c];
        t[2][c] = dd[2][c] - dd[1][c];
        t[3][c] = dd[1][c] - dd[3][c];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i][0] = t[i][0] - t[i][2];
        v[i][1] = t[i][1] + t[i][2];
        v[i][2] = t[i][2] - t[i][1];
        v[i][3] = t[i][1] - t[i][3];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float bv = j == 0 ? bf[q].x : (j == 1 ? bf[q].y : (j == 2 ? bf[q].z : bf[q].w));
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q >> 2][q & 3], bv, acc[q], 0, 0, 0);
      }
    }
    buf = buf + 1 == NS ? 0 : buf + 1;
  }

  // ---- output transform + epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1] ----
  const int co = nb * BN + wn * 32 + li;
  const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;  // tile of the wave's 4 x 8 block
    const int oy = Y0 + 2 * (wty * 4 + (m >> 3)), ox = X0 + 2 * (wtx * 8 + (m & 7));
    float s[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i][0] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
      s[i][1] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float y0 = s[0][b] + s[1][b] + s[2][b];
      const float y1 = s[1][b] - s[2][b] - s[3][b];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = oy + a, xx = ox + b;
        if (yy < Hs && xx < Ws && co < p.Cout) {
          float v = (a ? y1 : y0) + bias;
          v = v > 0.f ? v : v * p.alpha;
          p.y[((size_t)(n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldy + p.y_coff + co] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// v2: the same tile / stage / LDS layout, hand software-pipelined for ONE wave per SIMD.
// A stage is four phases, phase i = position row i (16 MFMAs on 4 accumulators x 4 channel components): it needs patch rows
// {0,2} {1,2} {1,2} {1,3} and the weight fragments of positions 4i..4i+3.  The LDS reads of phase i + 1 are issued before the
// multiplications of phase i; the wait-and-barrier that hands stage k + 1 over sits between phases 2 and 3 of stage k, so
// phase 3 prefetches phase 0 of the NEXT stage and no phase ever starts by waiting out the LDS latency behind a barrier.  The
// DMA of stage k + 2 (into the buffer of stage k - 1, which every wave has left at that barrier) is issued in two parts, inside
// phase 3 of stage k and phase 0 of stage k + 1, and has until the next barrier to land.
// ABL (ablation bits): 1 no DMA after the prologue, 2 no MFMA, 4 no input transform, 8 no LDS fragment reads in the loop
// ---------------------------------------------------------------------------------------------------------------------------
// SCHED: 0 = the compiler interleaves VALU / MFMA freely inside a phase; 2 = runs of 4 MFMAs (one channel component), the 8 VALU of the
// next component between the runs; 3 = all 32 VALU of the phase first, then 16 MFMAs back to back
template <int WTY, int WTX, int WN, int ABL, int SCHED = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino2_kernel(const WinoParams p) {
  static_assert(WTY * WTX * WN == 4, "4 waves");
  constexpr int NS = 3;
  constexpr int TH = 4 * WTY, TW = 8 * WTX;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int CS = PW / 2;
  constexpr int S = row_slots(PW);
  constexpr int HP = S / 2;
  constexpr int IN_SLOTS = 2 * PH * S;
  constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;
  constexpr int IN_BYTES = IN_INSTR * 256 * 16;
  constexpr int BN = 32 * WN;
  constexpr int W_BYTES = 16 * 2 * BN * 16;
  constexpr int W_INSTR = W_BYTES / 4096;
  constexpr int STAGE = IN_BYTES + W_BYTES;
  constexpr int L = IN_INSTR + W_INSTR;
  constexpr int LA = (L + 1) / 2;  // DMA instructions of part A (input first)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wt = wave / WN, wty = wt / WTX, wtx = wt % WTX;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int d = p.dil;
  const int bx = bid % p.BX;
  int rem = bid / p.BX;
  const int by = rem % p.BY;
  rem /= p.BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;
  const int nb = blockIdx.y;

  int in_off[IN_INSTR], in_q4[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int Lx = (i * 4 + wave) * 64 + lane;
    const int quad = Lx / (PH * S), r2 = Lx - quad * (PH * S);
    const int row = r2 / S, s = r2 - row * S;
    const int par = s / HP, cs = s - par * HP;
    const int col = 2 * cs + par;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool ok = quad < 2 && cs < CS && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    in_off[i] = ok ? ((n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldx + p.x_coff + quad * 4 : -1;
    in_q4[i] = quad * 4;
  }
  const float* zero = p.zero16;
  const float* ubase = p.u + (size_t)nb * (W_BYTES / 4) + (size_t)lane * 4;
  const float* ubase0 = p.u + (size_t)nb * (W_BYTES / 4);
  const size_t ustride = (size_t)gridDim.y * (W_BYTES / 4);

  // DMA instructions [i0, i1) of stage kg into buffer buf (instruction index: input first, then weights).  Inline assembly on
  // purpose: behind the builtin hipcc orders every later LDS read after the DMA with s_waitcnt vmcnt(0) (it models the DMA as a
  // store to LDS), which serialises fill and multiplication in a kernel whose waves do both; the hand-over is the explicit
  // vmcnt + barrier below.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane16 = lane * 16;
  auto issue = [&](int kg, int buf, int i0, int i1) {
    const unsigned sb = lds0 + buf * STAGE;
    const int c0 = kg * 8;
    const float* us = ubase0 + (size_t)kg * ustride;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (i < i0 || i >= i1) continue;
      if (i < IN_INSTR) {
        const float* src = (in_off[i] >= 0 && c0 + in_q4[i] < p.Kc) ? p.x + (in_off[i] + c0) : zero;
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(sb + (i * 4 + wave) * 1024) : "m0");
      } else {
        const int w = i - IN_INSTR;
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(us + (w * 4 + wave) * 256),
                     "s"(sb + IN_BYTES + (w * 4 + wave) * 1024)
                     : "m0");
      }
    }
  };

  floatx16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  const int a_base = ((lh * PH + 2 * (wty * 4 + ty)) * S + (wtx * 8 + tx)) * 16;
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;

  float4 row[4][4];
  float4 bfr[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) row[r][c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) bfr[s][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ld_row = [&](const char* sb, int r) {
    if (ABL & 8) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) row[r][c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf = [&](const char* sb, int i, int s) {
    if (ABL & 8) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) bfr[s][q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };

  const int nkg = p.nkg;
  issue(0, 0, 0, L);
  if (nkg > 1) issue(1, 1, 0, L);
  if (nkg > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ld_row(smem, 0);
  ld_row(smem, 2);
  ld_bf(smem, 0, 0);

  int buf = 0;
  for (int kg = 0; kg < nkg; ++kg) {
    const char* sb = smem + buf * STAGE;
    const int b1 = buf + 1 == NS ? 0 : buf + 1, b2 = b1 + 1 == NS ? 0 : b1 + 1;
    const char* sbn = smem + b1 * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // ---- LDS reads of the next phase, DMA parts ----
      if (i == 0) {
        ld_row(sb, 1);
        ld_bf(sb, 1, 1);
        if (!(ABL & 1) && kg > 0 && kg + 1 < nkg) issue(kg + 1, b1, LA, L);  // part B of the stage whose part A went out in the previous phase 3
      } else if (i == 1) {
        ld_bf(sb, 2, 0);
      } else if (i == 2) {
        ld_row(sb, 3);
        ld_bf(sb, 3, 1);
      } else {
        ld_row(sbn, 0);
        ld_row(sbn, 2);
        ld_bf(sbn, 0, 0);
        if (!(ABL & 1) && kg + 2 < nkg) issue(kg + 2, b2, 0, LA);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase i: position row i ----
      constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};  // t_i = d[RA] (+/-) d[RB]: d0-d2, d1+d2, d2-d1, d1-d3
      auto vcalc = [&](int j, float (&v)[4]) {
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = comp(row[RA[i]][c], j), b = comp(row[RB[i]][c], j);
          t[c] = (ABL & 4) ? a : (i == 1 ? a + b : a - b);
        }
        if (ABL & 4) {
          v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
          v[0] = t[0] - t[2];
          v[1] = t[1] + t[2];
          v[2] = t[2] - t[1];
          v[3] = t[1] - t[3];
        }
      };
      auto mrun = [&](int j, const float (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ABL & 2) acc[4 * i + q][j] += v[q] * comp(bfr[i & 1][q], j);
          else acc[4 * i + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], comp(bfr[i & 1][q], j), acc[4 * i + q], 0, 0, 0);
        }
      };
      if (SCHED == 3) {
        float vv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vcalc(j, vv[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) mrun(j, vv[j]);
      } else if (SCHED == 4) {  // as 3, but the four components of ONE accumulator back to back (dependent chain: result forwarding)
        float vv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vcalc(j, vv[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[4 * i + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j][q], comp(bfr[i & 1][q], j), acc[4 * i + q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
      } else if (SCHED == 2) {
        float vv[4][4];
        vcalc(0, vv[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mrun(j, vv[j]);
          __builtin_amdgcn_sched_barrier(0);
          if (j < 3) vcalc(j + 1, vv[j + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
          vcalc(j, v);
          mrun(j, v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 2) {  // stage kg + 1 is in LDS for everybody; stage kg - 1's buffer is free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    buf = b1;
  }

  const int co = nb * BN + wn * 32 + li;
  const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int oy = Y0 + 2 * (wty * 4 + (m >> 3)), ox = X0 + 2 * (wtx * 8 + (m & 7));
    float s[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i][0] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
      s[i][1] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float y0 = s[0][b] + s[1][b] + s[2][b];
      const float y1 = s[1][b] - s[2][b] - s[3][b];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = oy + a, xx = ox + b;
        if (yy < Hs && xx < Ws && co < p.Cout) {
          float v = (a ? y1 : y0) + bias;
          v = v > 0.f ? v : v * p.alpha;
          p.y[((size_t)(n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldy + p.y_coff + co] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// v3: v2's pipeline with every non-MFMA instruction placed by hand.  One wave per SIMD issues in order: a ds_read_b128 or an
// LDS-DMA instruction holds the issue port for tens of cycles, so a clump of them (v2: 8-12 reads and 5-6 DMAs at the head of a
// phase) drains the matrix pipe.  Here a phase is 16 slots = [<= 1 LDS read of the next phase] [<= 1 DMA] [1-5 VALU of the input
// transform] [1 MFMA], fenced by sched_barrier so hipcc keeps the order.  Input DMAs use the saddr form with a per-lane byte offset
// and an EXEC mask of the lanes inside the image (the halo / pad slots of all three buffers are zeroed once): no VALU per DMA.
// ---------------------------------------------------------------------------------------------------------------------------
template <int WTY, int WTX, int WN, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino3_kernel(const WinoParams p) {
  static_assert(WTY * WTX * WN == 4, "4 waves");
  constexpr int NS = 3;
  constexpr int TH = 4 * WTY, TW = 8 * WTX;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int CS = PW / 2;
  constexpr int S = row_slots(PW);
  constexpr int HP = S / 2;
  constexpr int IN_SLOTS = 2 * PH * S;
  constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;
  constexpr int IN_BYTES = IN_INSTR * 256 * 16;
  constexpr int BN = 32 * WN;
  constexpr int W_BYTES = 16 * 2 * BN * 16;
  constexpr int W_INSTR = W_BYTES / 4096;
  constexpr int STAGE = IN_BYTES + W_BYTES;
  constexpr int L = IN_INSTR + W_INSTR;
  static_assert(L <= 12, "DMA slots: 4 in phase 3, 4 in phase 0, 4 in phase 1");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wt = wave / WN, wty = wt / WTX, wtx = wt % WTX;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int d = p.dil;
  const int bx = bid % p.BX;
  int rem = bid / p.BX;
  const int by = rem % p.BY;
  rem /= p.BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;
  const int nb = blockIdx.y;

  unsigned in_voff[IN_INSTR];            // byte offset of this lane's 16 bytes from p.x + channel offset
  unsigned long long in_mask[IN_INSTR];  // lanes inside the image
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
    if (!ok) {  // never written by the DMA: zero once, in every ring buffer
#pragma unroll
      for (int b = 0; b < NS; ++b) *reinterpret_cast<float4*>(smem + b * STAGE + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const float* ubase0 = p.u + (size_t)nb * (W_BYTES / 4);
  const size_t ustride = (size_t)gridDim.y * (W_BYTES / 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane16 = lane * 16;
  // one DMA instruction `i` of stage kg into buffer buf
  auto dma = [&](int kg, int buf, int i) {
    const unsigned sb = lds0 + buf * STAGE;
    if (i < IN_INSTR) {
      const float* base = p.x + kg * 8;
      asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(base),
                   "s"(sb + (i * 4 + wave) * 1024), "s"(in_mask[i])
                   : "m0");
    } else {
      const int w = i - IN_INSTR;
      const float* us = ubase0 + (size_t)kg * ustride + (w * 4 + wave) * 256;
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(us), "s"(sb + IN_BYTES + (w * 4 + wave) * 1024) : "m0");
    }
  };

  floatx16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int ty = li >> 3, tx = li & 7;
  const int a_base = ((lh * PH + 2 * (wty * 4 + ty)) * S + (wtx * 8 + tx)) * 16;
  const int b_base = IN_BYTES + (lh * BN + wn * 32 + li) * 16;

  float4 row[4][4];
  float4 bfr[2][4];
  auto ld_row1 = [&](const char* sb, int r, int c) {
    if (!(ABL & 8)) row[r][c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf1 = [&](const char* sb, int i, int s, int q) {
    if (!(ABL & 8)) bfr[s][q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) row[r][c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) bfr[s][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };

  const int nkg = p.nkg;
#pragma unroll
  for (int i = 0; i < L; ++i) dma(0, 0, i);
  if (nkg > 1) {
#pragma unroll
    for (int i = 0; i < L; ++i) dma(1, 1, i);
  }
  if (nkg > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int c = 0; c < 4; ++c) { ld_row1(smem, 0, c); ld_row1(smem, 2, c); ld_bf1(smem, 0, 0, c); }

  int buf = 0;
  for (int kg = 0; kg < nkg; ++kg) {
    const char* sb = smem + buf * STAGE;
    const int b1 = buf + 1 == NS ? 0 : buf + 1, b2 = b1 + 1 == NS ? 0 : b1 + 1;
    const char* sbn = smem + b1 * STAGE;
    const bool dma_bc = !(ABL & 1) && kg > 0 && kg + 1 < nkg;  // parts B / C of stage kg + 1 (part A went out in the previous phase 3)
    const bool dma_a = !(ABL & 1) && kg + 2 < nkg;             // part A of stage kg + 2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};
      float t[4];
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) {
        const int j = sl >> 2, q = sl & 3;
        // (1) one LDS read of the next phase
        if (i == 0) {
          if (sl < 4) ld_row1(sb, 1, sl);
          else if (sl < 8) ld_bf1(sb, 1, 1, sl - 4);
        } else if (i == 1) {
          if (sl < 4) ld_bf1(sb, 2, 0, sl);
        } else if (i == 2) {
          if (sl < 4) ld_row1(sb, 3, sl);
          else if (sl < 8) ld_bf1(sb, 3, 1, sl - 4);
        } else {
          if (sl < 4) ld_row1(sbn, 0, sl);
          else if (sl < 8) ld_row1(sbn, 2, sl - 4);
          else if (sl < 12) ld_bf1(sbn, 0, 0, sl - 8);
        }
        // (2) one DMA instruction: part A (0..3) in phase 3 slots 12..15, B (4..7) in phase 0 slots 8..11, C (8..11) in phase 1 slots 4..7
        if (i == 3 && sl >= 12 && sl - 12 < L) { if (dma_a) dma(kg + 2, b2, sl - 12); }
        if (i == 0 && sl >= 8 && sl < 12 && sl - 4 < L) { if (dma_bc) dma(kg + 1, b1, sl - 4); }
        if (i == 1 && sl >= 4 && sl < 8 && sl + 4 < L) { if (dma_bc) dma(kg + 1, b1, sl + 4); }
        // (3) input transform of this slot's operand
        if (q == 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float a = comp(row[RA[i]][c], j), b = comp(row[RB[i]][c], j);
            t[c] = (ABL & 4) ? a : (i == 1 ? a + b : a - b);
          }
        }
        const float v = (ABL & 4) ? t[q] : (q == 0 ? t[0] - t[2] : (q == 1 ? t[1] + t[2] : (q == 2 ? t[2] - t[1] : t[1] - t[3])));
        // (4) the multiplication
        if (ABL & 2) acc[4 * i + q][j] += v * comp(bfr[i & 1][q], j);
        else acc[4 * i + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, comp(bfr[i & 1][q], j), acc[4 * i + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    buf = b1;
  }

  const int co = nb * BN + wn * 32 + li;
  const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int oy = Y0 + 2 * (wty * 4 + (m >> 3)), ox = X0 + 2 * (wtx * 8 + (m & 7));
    float s[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i][0] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
      s[i][1] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float y0 = s[0][b] + s[1][b] + s[2][b];
      const float y1 = s[1][b] - s[2][b] - s[3][b];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = oy + a, xx = ox + b;
        if (yy < Hs && xx < Ws && co < p.Cout) {
          float v = (a ? y1 : y0) + bias;
          v = v > 0.f ? v : v * p.alpha;
          p.y[((size_t)(n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldy + p.y_coff + co] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// v4: TWO waves per SIMD.  v2 / v3 showed what one wave per SIMD pays for: every ds_read_b128 and every DMA instruction the wave
// issues costs the matrix pipe ~60 idle cycles (it cannot be hidden behind the wave's own MFMAs, however it is placed).  Here a
// workgroup is 8 waves: waves 0-3 ("role 0") own position rows 1 and 2 of their 32-tile x 32-channel block, waves 4-7 ("role 1")
// rows 0 and 3 -- 8 accumulators = 128 registers per wave, so two waves share a SIMD and one multiplies while the other reads.
// Row i of B^T d needs patch rows {0,2} {1,2} {2,1} {1,3}: role 0 reads rows 1, 2 only (both of its phases), role 1 all four.
// A stage is two phases of 16 MFMAs per wave; the hand-over barrier sits between them.  The output transform adds the roles'
// partial column sums through LDS once, in the epilogue: Y0 = s0 + (s1 + s2), Y1 = (s1 - s2) - s3.
// ---------------------------------------------------------------------------------------------------------------------------
template <int WTY, int WTX, int WN, int ABL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4_kernel(const WinoParams p) {
  static_assert(WTY * WTX * WN == 4, "4 wave tiles");
  constexpr int NS = 3;
  constexpr int TH = 4 * WTY, TW = 8 * WTX;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int CS = PW / 2;
  constexpr int S = row_slots(PW);
  constexpr int HP = S / 2;
  constexpr int IN_SLOTS = 2 * PH * S;
  constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;  // input DMA instructions per role-0 wave
  constexpr int IN_BYTES = IN_INSTR * 256 * 16;
  constexpr int BN = 32 * WN;
  constexpr int W_BYTES = 16 * 2 * BN * 16;
  constexpr int NW = W_BYTES / 1024;                // weight DMA wave-instructions per stage
  constexpr int PER = (IN_INSTR * 4 + NW + 7) / 8;  // target per wave
  constexpr int W0 = PER > IN_INSTR ? PER - IN_INSTR : 0;  // weight instructions per role-0 wave
  constexpr int W1 = (NW - 4 * W0) / 4;                    // ... per role-1 wave
  static_assert(W1 >= 0 && 4 * W0 + 4 * W1 == NW, "weight split");
  constexpr int STAGE = IN_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* const tsb = p.ts ? p.ts + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  auto stamp = [&](int i) {
    if (tsb && threadIdx.x == 0) { tsb[i] = (i == 0 || i == 7) ? (long long)wall_clock64() : (long long)__builtin_readcyclecounter(); }
  };
  stamp(0);
  stamp(1);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, sub = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int wn = sub % WN, wt = sub / WN, wty = wt / WTX, wtx = wt % WTX;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int d = p.dil;
  const int bx = bid % p.BX;
  int rem = bid / p.BX;
  const int by = rem % p.BY;
  rem /= p.BY;
  const int sx = rem % d;
  rem /= d;
  const int sy = rem % d;
  const int n = rem / d;
  const int Hs = (p.H - sy + d - 1) / d, Ws = (p.W - sx + d - 1) / d;
  const int Y0 = by * 2 * TH, X0 = bx * 2 * TW;
  const int nb = blockIdx.y;

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
  stamp(2);
  const float* ubase0 = p.u + (size_t)nb * (W_BYTES / 4);
  const size_t ustride = (size_t)gridDim.y * (W_BYTES / 4);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane16 = lane * 16;
  auto dma_in = [&](int kg, int buf, int i) {
    const float* base = p.x + kg * 8;
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(base),
                 "s"(lds0 + buf * STAGE + (i * 4 + sub) * 1024), "s"(in_mask[i])
                 : "m0");
  };
  auto dma_w = [&](int kg, int buf, int w) {  // w: wave-instruction index of the weight stage
    const float* us = ubase0 + (size_t)kg * ustride + w * 256;
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(us), "s"(lds0 + buf * STAGE + IN_BYTES + w * 1024) : "m0");
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
    for (int c = 0; c < 4; ++c)
      if (!(ABL & 8)) dst[c] = *reinterpret_cast<const float4*>(sb + a_base + (r * S + (c & 1) * HP + (c >> 1)) * 16);
  };
  auto ld_bf = [&](const char* sb, int i, float4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!(ABL & 8)) dst[q] = *reinterpret_cast<const float4*>(sb + b_base + (4 * i + q) * (2 * BN * 16));
  };
  auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };
  const int nkg = p.nkg;

  auto body = [&](auto ROLE) {
    constexpr int R = decltype(ROLE)::value;
    constexpr int LR = R == 0 ? IN_INSTR + W0 : W1;
    constexpr int LA = (LR + 1) / 2;
    constexpr int I0 = R == 0 ? 1 : 0, I1 = R == 0 ? 2 : 3;  // position rows of phase 0 / phase 1
    float4 ra[4], rb[4];   // role 0: patch rows 1, 2; role 1: rows 0, 2 (phase 0)
    float4 rc[4], rd[4];   // role 1: rows 1, 3 (phase 1)
    float4 bf0[4], bf1[4];
    float v1[4][4];        // role 0: phase 1's operands, formed during phase 0 from the same two rows
    float zc = 0.5f;
    asm volatile("" : "+v"(zc));
#pragma unroll
    for (int c = 0; c < 4; ++c) ra[c] = rb[c] = rc[c] = rd[c] = bf0[c] = bf1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    dma_part(ROLE, 0, 0, 0, LR);
    if (nkg > 1) dma_part(ROLE, 1, 1, 0, LR);
    if (nkg > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(3);
    ld_row(smem, R == 0 ? 1 : 0, ra);
    ld_row(smem, 2, rb);
    ld_bf(smem, I0, bf0);
    int buf = 0;
    for (int kg = 0; kg < nkg; ++kg) {
      const char* sb = smem + buf * STAGE;
      const int b1 = buf + 1 == NS ? 0 : buf + 1, b2 = b1 + 1 == NS ? 0 : b1 + 1;
      const char* sbn = smem + b1 * STAGE;
      // ---------------- phase 0 ----------------
      if (R == 1) { ld_row(sb, 1, rc); ld_row(sb, 3, rd); }
      ld_bf(sb, I1, bf1);
      if (!(ABL & 1) && kg > 0 && kg + 1 < nkg) dma_part(ROLE, kg + 1, b1, LA, LR);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t[4], v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = (ABL & 16) ? zc : comp(ra[c], j), b = (ABL & 16) ? zc : comp(rb[c], j);
          t[c] = R == 0 ? a + b : a - b;  // row 1: d1 + d2; row 0: d0 - d2
        }
        v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        if (R == 0) {  // row 2: d2 - d1, kept for phase 1
          float u[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) u[c] = (ABL & 16) ? zc : comp(rb[c], j) - comp(ra[c], j);
          v1[j][0] = u[0] - u[2]; v1[j][1] = u[1] + u[2]; v1[j][2] = u[2] - u[1]; v1[j][3] = u[1] - u[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ABL & 2) acc[q][j] += v[q] * comp(bf0[q], j);
          else acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], (ABL & 16) ? zc : comp(bf0[q], j), acc[q], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(ra[c].x), "v"(rb[c].x), "v"(bf0[c].x), "v"(ra[c].w), "v"(rb[c].w), "v"(bf0[c].w));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage kg + 1 is in LDS for every wave; the buffer of stage kg - 1 is free
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // ---------------- phase 1 ----------------
      ld_row(sbn, R == 0 ? 1 : 0, ra);
      ld_row(sbn, 2, rb);
      ld_bf(sbn, I0, bf0);
      if (!(ABL & 1) && kg + 2 < nkg) dma_part(ROLE, kg + 2, b2, 0, LA);
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
          for (int c = 0; c < 4; ++c) t[c] = (ABL & 16) ? zc : comp(rc[c], j) - comp(rd[c], j);  // row 3: d1 - d3
          v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ABL & 2) acc[4 + q][j] += v[q] * comp(bf1[q], j);
          else acc[4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q], (ABL & 16) ? zc : comp(bf1[q], j), acc[4 + q], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(rc[c].x), "v"(rd[c].x), "v"(bf1[c].x), "v"(rc[c].w), "v"(rd[c].w), "v"(bf1[c].w));
      }
      buf = b1;
    }
  };
  if (role == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());

  // ---- output transform: column sums of the own rows, the roles' halves meet in LDS ----
  stamp(4);
  __syncthreads();
  float* exch = reinterpret_cast<float*>(smem) + sub * (16 * 4 * 64);  // [r][4][lane]
  float s[2][2][16];  // [own row 0/1][b][r]
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s[h][0][r] = acc[h * 4 + 0][r] + acc[h * 4 + 1][r] + acc[h * 4 + 2][r];
      s[h][1][r] = acc[h * 4 + 1][r] - acc[h * 4 + 2][r] - acc[h * 4 + 3][r];
    }
  if (role == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      exch[(r * 4 + 0) * 64 + lane] = s[0][0][r];  // s0[b]
      exch[(r * 4 + 1) * 64 + lane] = s[0][1][r];
      exch[(r * 4 + 2) * 64 + lane] = s[1][0][r];  // s3[b]
      exch[(r * 4 + 3) * 64 + lane] = s[1][1][r];
    }
  }
  __syncthreads();
  if (role == 1) return;
  stamp(5);
  const int co = nb * BN + wn * 32 + li;
  const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int oy = Y0 + 2 * (wty * 4 + (m >> 3)), ox = X0 + 2 * (wtx * 8 + (m & 7));
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float s0 = exch[(r * 4 + b) * 64 + lane], s3 = exch[(r * 4 + 2 + b) * 64 + lane];
      const float y0 = s0 + (s[0][b][r] + s[1][b][r]);
      const float y1 = (s[0][b][r] - s[1][b][r]) - s3;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = oy + a, xx = ox + b;
        if (yy < Hs && xx < Ws && co < p.Cout) {
          float v = (a ? y1 : y0) + bias;
          v = v > 0.f ? v : v * p.alpha;
          p.y[((size_t)(n * p.H + sy + d * yy) * p.W + sx + d * xx) * p.ldy + p.y_coff + co] = v;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(6);
  stamp(7);
}

// U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1];  layout [kg][nb][pos][half][BN][4]
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int nkg, int nnb, int BN) {
  const long total = (long)nkg * nnb * 16 * 2 * BN * 4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long r = e;
    const int j = (int)(r & 3); r >>= 2;
    const int nn = (int)(r % BN); r /= BN;
    const int kh = (int)(r & 1); r >>= 1;
    const int pos = (int)(r & 15); r >>= 4;
    const int nb = (int)(r % nnb);
    const int kg = (int)(r / nnb);
    const int c = kg * 8 + kh * 4 + j, co = nb * BN + nn;
    float val = 0.f;
    if (c < Cin && co < Cout) {
      const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
      const int pi = pos >> 2, pj = pos & 3;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) val += G[pi][a] * G[pj][b] * w[((size_t)(a * 3 + b) * Cin + c) * Cout + co];
    }
    u[e] = val;
  }
}

// plain direct convolution (the check): 3x3, stride 1, dilation d, SAME padding, bias, leaky
__global__ void ref_conv_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int d, float alpha) {
  const long total = (long)N * H * W * Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int co = (int)(e % Cout);
    long pix = e / Cout;
    const int ox = (int)(pix % W);
    pix /= W;
    const int oy = (int)(pix % H), n = (int)(pix / H);
    double acc = bias[co];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        const int iy = oy + (a - 1) * d, ix = ox + (b - 1) * d;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* xp = x + ((size_t)(n * H + iy) * W + ix) * ldx;
        const float* wp = w + (size_t)(a * 3 + b) * Cin * Cout + co;
        for (int c = 0; c < Cin; ++c) acc += (double)xp[c] * wp[(size_t)c * Cout];
      }
    float v = (float)acc;
    y[e] = v > 0.f ? v : v * alpha;
  }
}

struct Shape { const char* name; int N, H, W, Cin, Cout, d; };

template <int WTY, int WTX, int WN, int NS, int V2 = 0, int ABL = 0, int SCHED = 0>
static float run(const Shape& s, const float* x, int ldx, const float* w, const float* bias, float* y, const float* zero, int reps, float* u_buf) {
  constexpr int TH = 4 * WTY, TW = 8 * WTX, BN = 32 * WN;
  WinoParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.ldx = ldx; p.x_coff = 0; p.N = s.N; p.H = s.H; p.W = s.W; p.Kc = (s.Cin + 3) & ~3;
  p.nkg = (p.Kc + 7) / 8;
  const int nnb = (s.Cout + BN - 1) / BN;
  hipLaunchKernelGGL(wino_weights_kernel, dim3(1024), dim3(256), 0, 0, w, u_buf, s.Cin, s.Cout, p.nkg, nnb, BN);
  p.u = u_buf; p.bias = bias; p.y = y; p.ldy = s.Cout; p.y_coff = 0; p.Cout = s.Cout; p.dil = s.d;
  p.Hs = (s.H + s.d - 1) / s.d; p.Ws = (s.W + s.d - 1) / s.d;
  p.BY = (p.Hs + 2 * TH - 1) / (2 * TH); p.BX = (p.Ws + 2 * TW - 1) / (2 * TW);
  p.zero16 = zero; p.alpha = 0.1f;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int S = row_slots(PW);
  constexpr int IN_BYTES = ((2 * PH * S + 255) / 256) * 256 * 16;
  constexpr int STAGE = IN_BYTES + 16 * 2 * BN * 16;
  const int shmem = NS * STAGE;
  auto kern = V2 == 4 ? wino4_kernel<WTY, WTX, WN, ABL> : V2 == 3 ? wino3_kernel<WTY, WTX, WN, ABL> : (V2 ? wino2_kernel<WTY, WTX, WN, ABL, SCHED> : wino_kernel<WTY, WTX, WN, NS>);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
  dim3 grid(s.N * s.d * s.d * p.BY * p.BX, nnb);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 block(V2 == 4 ? 512 : 256);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, p);
  CHECK(hipGetLastError());
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, p);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (V2 == 4 && getenv("WINO_TS")) {
    const size_t nblk = (size_t)grid.x * grid.y;
    long long* ts;
    CHECK(hipMalloc(&ts, nblk * 8 * sizeof(long long)));
    CHECK(hipMemset(ts, 0, nblk * 8 * sizeof(long long)));
    WinoParams q = p;
    q.ts = ts;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, q);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float kms = 0.f;
    CHECK(hipEventElapsedTime(&kms, a, b));
    std::vector<long long> h(nblk * 8);
    CHECK(hipMemcpy(h.data(), ts, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    // stamps: 0 wall clock (100 MHz) at entry, 1 cycle counter at entry, 2 after the halo zero-fill + tables, 3 first stage landed,
    // 4 K loop done, 5 roles exchanged, 6 stores acknowledged, 7 wall clock at exit
    long long w0 = h[0], w1 = h[7];
    for (size_t i = 0; i < nblk; ++i) { w0 = std::min(w0, h[i * 8]); w1 = std::max(w1, h[i * 8 + 7]); }
    double seg[5] = {0, 0, 0, 0, 0}, segmax[5] = {0, 0, 0, 0, 0}, first = 0, firstmax = 0, dur = 0;
    for (size_t i = 0; i < nblk; ++i) {
      for (int k = 0; k < 5; ++k) { const double v = (double)(h[i * 8 + k + 2] - h[i * 8 + k + 1]); seg[k] += v; segmax[k] = std::max(segmax[k], v); }
      const double st = (double)(h[i * 8] - w0) * 10.0;  // ns
      first += st; firstmax = std::max(firstmax, st);
      dur += (double)(h[i * 8 + 7] - h[i * 8]) * 10.0;
    }
    printf("\n      [ts] event-bracket %.1f us; first entry -> last exit %.1f us; workgroup entry after the first: mean %.1f us max %.1f us; workgroup lifetime mean %.1f us\n",
           kms * 1e3, (double)(w1 - w0) * 0.01, first / nblk * 1e-3, firstmax * 1e-3, dur / nblk * 1e-3);
    const char* nm[5] = {"zero-fill+tables", "first stage lands", "K loop", "role exchange", "transform+stores"};
    for (int k = 0; k < 5; ++k) printf("      [ts] %-18s mean %8.0f cycles  max %8.0f\n", nm[k], seg[k] / nblk, segmax[k]);
    CHECK(hipFree(ts));
  }
  printf("    %s<%d,%d,%d,%d> sched %d abl %d grid %dx%d lds %d KB S=%d: ", V2 == 4 ? "v4" : V2 == 3 ? "v3" : (V2 ? "v2" : "v1"), WTY, WTX, WN, NS, SCHED, ABL, grid.x, grid.y, shmem / 1024, S);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const Shape shapes[] = {
      {"pwc.dc_conv21", 4, 96, 160, 565, 128, 1}, {"pwc.conv2_1", 4, 96, 160, 243, 128, 1}, {"pwc.conv2_3", 4, 96, 160, 467, 64, 1},
      {"pwc.conv2_4", 4, 96, 160, 531, 32, 1},    {"pwc.dc_conv22", 4, 96, 160, 128, 128, 2}, {"pwc.dc_conv31", 4, 48, 80, 597, 128, 1},
      {"gen.conv5", 4, 48, 96, 128, 128, 1},      {"gen.conv3", 4, 96, 192, 64, 64, 1},     {"odd", 2, 37, 53, 20, 40, 1},
      {"odd.d3", 1, 41, 50, 36, 70, 3},
      {"ovh.1stage", 4, 48, 96, 8, 128, 1}, {"ovh.2stage", 4, 48, 96, 16, 128, 1}, {"ovh.4stage", 4, 48, 96, 32, 128, 1}};
  const int reps = 20;
  for (const Shape& s : shapes) {
    bool sel = argc <= 1;
    for (int i = 1; i < argc; ++i) sel = sel || strstr(s.name, argv[i]);
    if (!sel) continue;
    const int ldx = (s.Cin + 7) & ~7;
    const size_t nx = (size_t)s.N * s.H * s.W * ldx, nw = (size_t)9 * s.Cin * s.Cout, ny = (size_t)s.N * s.H * s.W * s.Cout;
    std::vector<float> hx(nx), hw(nw), hb(s.Cout);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (size_t i = 0; i < nx; ++i) hx[i] = (int)(i % ldx) < s.Cin ? rnd() : 0.f;
    const float ws = sqrtf(2.f / (9.f * s.Cin)) * 2.f;
    for (auto& v : hw) v = rnd() * ws;
    for (auto& v : hb) v = rnd() * 0.1f;
    float *x, *w, *b, *y, *yr, *zero, *u;
    CHECK(hipMalloc(&x, nx * 4)); CHECK(hipMalloc(&w, nw * 4)); CHECK(hipMalloc(&b, s.Cout * 4));
    CHECK(hipMalloc(&y, ny * 4)); CHECK(hipMalloc(&yr, ny * 4)); CHECK(hipMalloc(&zero, 256));
    const size_t nu = (size_t)((ldx + 7) / 8) * ((s.Cout + 127) / 128 * 128 + 128) * 16 * 8;
    CHECK(hipMalloc(&u, nu * 4));
    CHECK(hipMemset(zero, 0, 256));
    CHECK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), s.Cout * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_conv_kernel, dim3(4096), dim3(256), 0, 0, x, ldx, w, b, yr, s.N, s.H, s.W, s.Cin, s.Cout, s.d, 0.1f);
    CHECK(hipDeviceSynchronize());
    std::vector<float> hy(ny), hr(ny);
    CHECK(hipMemcpy(hr.data(), yr, ny * 4, hipMemcpyDeviceToHost));
    const double gflop = 2.0 * s.N * s.H * s.W * (double)s.Cout * s.Cin * 9 * 1e-9;
    printf("%s  N=%d %dx%d %d->%d d=%d  %.2f GFLOP\n", s.name, s.N, s.H, s.W, s.Cin, s.Cout, s.d, gflop);
    auto report = [&](float us) {
      CHECK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
      double md = 0, mr = 0;
      for (size_t i = 0; i < ny; ++i) { md = fmax(md, fabs((double)hy[i] - hr[i])); mr = fmax(mr, fabs((double)hr[i])); }
      printf("%8.1f us  %6.1f TFLOP/s (direct-equivalent)  max|diff| %.2e / scale %.2e\n", us, gflop / us * 1e3, md, mr);
      CHECK(hipMemset(y, 0, ny * 4));
    };
    report(run<2, 1, 2, 3, 1, 0, 3>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 2, 1, 3, 4>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4, 1>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4, 8>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4, 9>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4, 16>(s, x, ldx, w, b, y, zero, reps, u));
    report(run<2, 1, 2, 3, 4, 17>(s, x, ldx, w, b, y, zero, reps, u));
    if (getenv("WINO_ABL")) {
      report(run<2, 1, 2, 3, 1, 1>(s, x, ldx, w, b, y, zero, reps, u));
      report(run<2, 1, 2, 3, 1, 2>(s, x, ldx, w, b, y, zero, reps, u));
      report(run<2, 1, 2, 3, 1, 4>(s, x, ldx, w, b, y, zero, reps, u));
      report(run<2, 1, 2, 3, 1, 8>(s, x, ldx, w, b, y, zero, reps, u));
      report(run<2, 1, 2, 3, 1, 9>(s, x, ldx, w, b, y, zero, reps, u));
      report(run<2, 1, 2, 3, 1, 13>(s, x, ldx, w, b, y, zero, reps, u));
    }
    hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(yr); hipFree(zero); hipFree(u);
  }
  return 0;
}
