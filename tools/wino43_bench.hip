// Prototype + micro-benchmark of a fused Winograd F(4x4,3x3) fp32-MFMA convolution:
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/wino43_bench tools/wino43_bench.hip && /tmp/wino43_bench [shape ...]
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      36 multiplications per 4x4 output tile and channel pair: 2.25 per output
//                                                        (F(2x2,3x3): 4, direct: 9)
//
// One workgroup = 4 waves = ONE wave per SIMD (512-register budget): 32 tiles (4 x 8 tiles of 4x4 pixels = 16 x 32 output pixels) x 64
// output channels.  Wave (rh, nh) owns position rows 3 rh .. 3 rh + 2 (18 of the 36 positions) of a 32-tile x 32-channel block:
// 18 accumulators of v_mfma_f32_32x32x2_f32 = 288 registers.  K runs in stages of 8 input channels; a stage is THREE PHASES, phase p =
// position row 3 rh + p of each wave (6 positions x 4 MFMAs = 24 MFMAs).
//   input stage : the raw 18 x 34 pixel halo, [channel quad][row][column mod 4][column / 4] in 16-byte slots, row stride 38 slots
//                 (conflict-free ds_read_b128 of a lane's 6 x 6 patch: a 16-lane group reads slots == tx + 8 ty (mod 16)); two buffers
//   weight phase: the pre-transformed U = G g G^T of the two position rows of the phase, [rh][column j][lane half][64 channels][4 ch] =
//                 24 KB (a whole stage would be 72 KB); a ring of four phase buffers
// both land by global_load_lds_dwordx4 issued by the MFMA waves themselves.  The input transform B^T d B sits on the LDS -> VGPR path:
// per phase a wave forms t = (B^T d)[row] from the 3-4 patch rows its B^T row touches (18 or 24 reads) and v = t B (6 quads).
// Output transform: each wave forms its rows' share of A^T M A, the two row halves exchange half a tile each through LDS and each
// stores two of the four output rows.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct W43Params {
  const float* x;
  int ldx, x_coff, N, H, W, Kc;  // Kc: input channels, multiple of 8
  const float* u;                // [Kc / 8][3 phases][2 rh][6 j][2 lane halves][np][4]
  int np;                        // padded output channels (multiple of 64)
  const float* bias;
  float* y;
  int ldy, y_coff, Cout;
  int BY, BX;
  float alpha;
  long long* ts;
};

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

namespace g43 {
constexpr int TH = 4, TW = 8;                      // tiles of a workgroup
constexpr int PH = 4 * TH + 2, PW = 4 * TW + 2;    // 18 x 34 halo pixels
constexpr int PLW = 9;                             // slots per column-phase plane (columns c, c + 4, ...)
constexpr int S = 38;                              // slots per halo row: >= 4 * PLW, 4 S == 8 (mod 16)
constexpr int QS = PH * S;                         // slots per channel quad (684)
constexpr int IN_SLOTS = 2 * QS;                   // 1368
constexpr int IN_INSTR = (IN_SLOTS + 255) / 256;   // 6 DMA instructions per wave and stage
constexpr int IN_BYTES = IN_INSTR * 256 * 16;      // 24 KB
constexpr int BN = 64;
constexpr int UPH_BYTES = 2 * 6 * 2 * BN * 16;     // one weight phase: 24 KB
constexpr int UPH_INSTR = UPH_BYTES / 4096;        // 6 per wave
constexpr int NUB = 4;                             // weight phase buffers
constexpr int LDS_BYTES = 2 * IN_BYTES + NUB * UPH_BYTES;  // 144 KB
static_assert((4 * S) % 16 == 8 && S >= 4 * PLW, "row stride");
static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= 4 * 32 * 1024, "LDS");
}  // namespace g43

// B^T of F(4x4,3x3) (points 0, +-1, +-2, inf)
__device__ __host__ constexpr float BT43(int i, int a) {
  constexpr float M[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  return M[i][a];
}

__device__ __forceinline__ float4 f4_mul(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 f4_fma(float s, const float4& a, const float4& c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 f4_add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino43_kernel(const W43Params p) {
  using namespace g43;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const in_buf = smem;                    // [2][IN_BYTES]
  char* const u_buf = smem + 2 * IN_BYTES;      // [NUB][UPH_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rh = wave >> 1, nh = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int ty = li >> 3, tx = li & 7;

  long long* const tsb = p.ts ? p.ts + (size_t)blockIdx.x * 8 : nullptr;
  auto stamp = [&](int i) {
    if (tsb && threadIdx.x == 0) tsb[i] = (long long)__builtin_readcyclecounter();
  };
  stamp(0);

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int NB = (p.Cout + BN - 1) / BN;
  const int nb = bid % NB;
  bid /= NB;
  const int bx = bid % p.BX;
  int rem = bid / p.BX;
  const int by = rem % p.BY;
  const int n = rem / p.BY;
  const int Y0 = by * 4 * TH, X0 = bx * 4 * TW;  // first output pixel of the workgroup
  const int nkg = p.Kc >> 3;

  // ---- input DMA: per-lane byte offsets + EXEC masks; halo / pad slots are zeroed once in both buffers ----
  unsigned in_voff[IN_INSTR];
  unsigned long long in_mask[IN_INSTR];
#pragma unroll
  for (int i = 0; i < IN_INSTR; ++i) {
    const int Lx = (i * 4 + wave) * 64 + lane;
    const int quad = Lx / QS, r2 = Lx - quad * QS;
    const int row = r2 / S, s = r2 - row * S;
    const int plane = s / PLW, idx = s - plane * PLW;
    const int col = 4 * idx + plane;
    const int yy = Y0 - 1 + row, xx = X0 - 1 + col;
    const bool ok = quad < 2 && s < 4 * PLW && col < PW && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    in_voff[i] = ok ? (unsigned)(((n * p.H + yy) * p.W + xx) * p.ldx + p.x_coff + quad * 4) * 4u : 0u;
    in_mask[i] = __ballot(ok);
    if (!ok) {
      *reinterpret_cast<float4*>(in_buf + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(in_buf + IN_BYTES + Lx * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  stamp(1);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane16 = (unsigned)lane * 16;
  // input instructions [i0, i1) of stage kg into input buffer kg & 1
  auto dma_in = [&](int kg, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < IN_INSTR; ++i) {
      if (i < i0 || i >= i1) continue;
      if (ABL & 1) continue;
      asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(in_voff[i]), "s"(p.x + kg * 8),
                   "s"(lds0 + (kg & 1) * IN_BYTES + (i * 4 + wave) * 1024), "s"(in_mask[i])
                   : "m0");
    }
  };
  // weight phase q = 3 kg + ph into ring buffer q & 3: 24 wave-instructions of 1 KB ((rh, j, lane half) rows of 64 channels)
  const float* const ubase = p.u + (size_t)nb * BN * 4;
  auto dma_u = [&](int q) {
#pragma unroll
    for (int i = 0; i < UPH_INSTR; ++i) {
      if (ABL & 1) continue;
      const int w = i * 4 + wave;
      const float* src = ubase + ((size_t)q * 24 + w) * p.np * 4;
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(lds0 + 2 * IN_BYTES + (q & 3) * UPH_BYTES + w * 1024) : "m0");
    }
  };

  floatx16 acc[18];
#pragma unroll
  for (int q = 0; q < 18; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // byte offset of this lane's patch origin (row 4 ty, column 4 tx of the halo) inside an input stage, of its weight quad inside a phase
  const int a_base = (lh * QS + 4 * ty * S + tx) * 16;
  const int b_base = ((rh * 6) * 2 + lh) * (BN * 16) + (nh * 32 + li) * 16;
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };

  const int Q = 3 * nkg;
  // prologue: input stage 0, weight phases 0 .. 2
  dma_in(0, 0, IN_INSTR);
  dma_u(0);
  if (Q > 1) dma_u(1);
  if (Q > 2) dma_u(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stamp(2);

  // one phase: position row I = 3 RH + PHS of this wave
  auto phase = [&](auto RH_, auto PHS_, int kg) {
    constexpr int RH = decltype(RH_)::value, PHS = decltype(PHS_)::value, I = 3 * RH + PHS;
    const int q = 3 * kg + PHS;
    // ---- DMA of what this phase may refill: weight phase q + 3 (its buffer held phase q - 1), next stage's input in phases 0 and 1 ----
    if (PHS == 0 && kg + 1 < nkg) dma_in(kg + 1, 0, 3);
    if (PHS == 1 && kg + 1 < nkg) dma_in(kg + 1, 3, 6);
    if (q + 3 < Q) dma_u(q + 3);
    const char* sb = in_buf + (kg & 1) * IN_BYTES + a_base;
    const char* ub = u_buf + (q & 3) * UPH_BYTES + b_base;
    // ---- t[b] = sum_a BT[I][a] d[a][b]: only the patch rows row I of B^T touches ----
    float4 t[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      bool first = true;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const float cf = BT43(I, a);
        if (cf == 0.f) continue;
        float4 dv = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
        if (!(ABL & 8)) dv = *reinterpret_cast<const float4*>(sb + (a * S + (b & 3) * PLW + (b >> 2)) * 16);
        if (first) { t[b] = f4_mul(cf, dv); first = false; }
        else t[b] = f4_fma(cf, dv, t[b]);
      }
    }
    // ---- v[j] = sum_b BT[j][b] t[b] ----
    float4 v[6];
    v[0] = f4_fma(4.f, t[0], f4_fma(-5.f, t[2], t[4]));
    {
      const float4 s12 = f4_add(t[1], t[2]), d34 = f4_add(t[3], t[4]);   // v1 = -4 (t1 + t2) + (t3 + t4)
      v[1] = f4_fma(-4.f, s12, d34);
      const float4 d12 = f4_sub(t[1], t[2]), d43 = f4_sub(t[4], t[3]);   // v2 = 4 (t1 - t2) + (t4 - t3)
      v[2] = f4_fma(4.f, d12, d43);
      const float4 e = f4_sub(t[4], t[2]), f = f4_sub(t[3], t[1]);       // v3 = (t4 - t2) + 2 (t3 - t1); v4 = (t4 - t2) - 2 (t3 - t1)
      v[3] = f4_fma(2.f, f, e);
      v[4] = f4_fma(-2.f, f, e);
    }
    v[5] = f4_fma(4.f, t[1], f4_fma(-5.f, t[3], t[5]));
    // ---- 24 MFMAs ----
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float4 bq = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
      if (!(ABL & 8)) bq = *reinterpret_cast<const float4*>(ub + j * (2 * BN * 16));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (ABL & 2) acc[PHS * 6 + j][c] += comp(v[j], c) * comp(bq, c);
        else acc[PHS * 6 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(v[j], c), comp(bq, c), acc[PHS * 6 + j], 0, 0, 0);
      }
    }
    // ---- hand-over: weight phase q + 1 (issued two phases ago) and, behind phase 2, the next stage's input must have landed ----
    if (q + 3 < Q) {  // steady state: every batch went out (9, 9, 6 instructions in phases 0, 1, 2)
      if (PHS == 0) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else if (PHS == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto body = [&](auto RH_) {
    for (int kg = 0; kg < nkg; ++kg) {
      phase(RH_, std::integral_constant<int, 0>(), kg);
      phase(RH_, std::integral_constant<int, 1>(), kg);
      phase(RH_, std::integral_constant<int, 2>(), kg);
    }
  };
  if (rh == 0) body(std::integral_constant<int, 0>());
  else body(std::integral_constant<int, 1>());
  stamp(3);

  // ---- output transform.  M[i][j] = acc[(i - 3 rh) * 6 + j]; w[i][v] = sum_j M[i][j] AT[v][j]; z[u][v] = sum_{own i} AT[u][i] w[i][v] ----
  // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  rh = 0 keeps output rows 0, 1 and sends its share of rows 2, 3;
  // rh = 1 keeps rows 2, 3 and sends its share of rows 0, 1: xch[wave][r][8][lane]
  float* const xch = reinterpret_cast<float*>(smem) + wave * (16 * 8 * 64);
  const float* const xin = reinterpret_cast<const float*>(smem) + (wave ^ 2) * (16 * 8 * 64);
  float keep[16][8];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float w[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float m0 = acc[i * 6 + 0][r], m1 = acc[i * 6 + 1][r], m2 = acc[i * 6 + 2][r], m3 = acc[i * 6 + 3][r], m4 = acc[i * 6 + 4][r],
                  m5 = acc[i * 6 + 5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      w[i][0] = m0 + s12 + s34;
      w[i][1] = d12 + 2.f * d34;
      w[i][2] = s12 + 4.f * s34;
      w[i][3] = d12 + 8.f * d34 + m5;
    }
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) {
      float z0, z1, z2, z3;
      if (rh == 0) {  // rows 0, 1, 2 of M
        z0 = w[0][vv] + w[1][vv] + w[2][vv];
        z1 = w[1][vv] - w[2][vv];
        z2 = w[1][vv] + w[2][vv];
        z3 = w[1][vv] - w[2][vv];
      } else {        // rows 3, 4, 5
        z0 = w[0][vv] + w[1][vv];
        z1 = 2.f * (w[0][vv] - w[1][vv]);
        z2 = 4.f * (w[0][vv] + w[1][vv]);
        z3 = 8.f * (w[0][vv] - w[1][vv]) + w[2][vv];
      }
      if (rh == 0) {
        keep[r][vv] = z0; keep[r][4 + vv] = z1;
        xch[(r * 8 + vv) * 64 + lane] = z2; xch[(r * 8 + 4 + vv) * 64 + lane] = z3;
      } else {
        keep[r][vv] = z2; keep[r][4 + vv] = z3;
        xch[(r * 8 + vv) * 64 + lane] = z0; xch[(r * 8 + 4 + vv) * 64 + lane] = z1;
      }
    }
  }
  __syncthreads();
  stamp(4);
  const int co = nb * BN + nh * 32 + li;
  const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;  // tile of the wave's 4 x 8 block held by accumulator register r
    const int oy0 = Y0 + 4 * (m >> 3) + 2 * rh, ox0 = X0 + 4 * (m & 7);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int yy = oy0 + (e >> 2), xx = ox0 + (e & 3);
      float val = keep[r][e] + xin[(r * 8 + e) * 64 + lane] + bias;
      val = val > 0.f ? val : val * p.alpha;
      if (yy < p.H && xx < p.W && co < p.Cout) p.y[((size_t)(n * p.H + yy) * p.W + xx) * p.ldy + p.y_coff + co] = val;
    }
  }
  stamp(5);
}

// U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1];  layout [kg][phase][rh][j][half][np][4]
__global__ void wino43_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int nkg, int np) {
  const long total = (long)nkg * 3 * 2 * 6 * 2 * np * 4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long r = e;
    const int cj = (int)(r & 3); r >>= 2;
    const int nn = (int)(r % np); r /= np;
    const int kh = (int)(r & 1); r >>= 1;
    const int j = (int)(r % 6); r /= 6;
    const int rh = (int)(r & 1); r >>= 1;
    const int ph = (int)(r % 3);
    const int kg = (int)(r / 3);
    const int i = 3 * rh + ph;
    const int c = kg * 8 + kh * 4 + cj, co = nn;
    float val = 0.f;
    if (c < Cin && co < Cout) {
      const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
      double s = 0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) s += G[i][a] * G[j][b] * (double)w[((size_t)(a * 3 + b) * Cin + c) * Cout + co];
      val = (float)s;
    }
    u[e] = val;
  }
}

// plain direct convolution in double (the check): 3x3, stride 1, SAME padding, bias, leaky
__global__ void ref_conv_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int N,
                                int H, int W, int Cin, int Cout, float alpha) {
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
        const int iy = oy + a - 1, ix = ox + b - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* xp = x + ((size_t)(n * H + iy) * W + ix) * ldx;
        const float* wp = w + (size_t)(a * 3 + b) * Cin * Cout + co;
        for (int c = 0; c < Cin; ++c) acc += (double)xp[c] * wp[(size_t)c * Cout];
      }
    const float v = (float)acc;
    y[e] = v > 0.f ? v : v * alpha;
  }
}

struct Shape { const char* name; int N, H, W, Cin, Cout; };

template <int ABL>
static float run(const Shape& s, const float* x, int ldx, const float* w, const float* bias, float* y, int reps, float* u_buf) {
  using namespace g43;
  W43Params p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.ldx = ldx; p.N = s.N; p.H = s.H; p.W = s.W; p.Kc = ldx;
  const int nkg = p.Kc / 8;
  p.np = (s.Cout + 63) / 64 * 64;
  hipLaunchKernelGGL(wino43_weights_kernel, dim3(2048), dim3(256), 0, 0, w, u_buf, s.Cin, s.Cout, nkg, p.np);
  p.u = u_buf; p.bias = bias; p.y = y; p.ldy = s.Cout; p.Cout = s.Cout; p.alpha = 0.1f;
  p.BY = (s.H + 4 * TH - 1) / (4 * TH); p.BX = (s.W + 4 * TW - 1) / (4 * TW);
  auto kern = wino43_kernel<ABL>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  dim3 grid(s.N * p.BY * p.BX * (p.np / 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, 0, p);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  if (getenv("W43_TS")) {
    long long* ts;
    CHECK(hipMalloc(&ts, (size_t)grid.x * 8 * sizeof(long long)));
    CHECK(hipMemset(ts, 0, (size_t)grid.x * 8 * sizeof(long long)));
    W43Params q = p;
    q.ts = ts;
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, 0, q);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h((size_t)grid.x * 8);
    CHECK(hipMemcpy(h.data(), ts, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double seg[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < grid.x; ++i)
      for (int k = 0; k < 5; ++k) seg[k] += (double)(h[i * 8 + k + 1] - h[i * 8 + k]) / grid.x;
    printf("      [ts] zero-fill %.0f; first stage %.0f; K loop %.0f (%.0f per stage); transform + exchange %.0f; stores %.0f  (cycles, mean)\n", seg[0], seg[1], seg[2],
           seg[2] / nkg, seg[3], seg[4]);
    CHECK(hipFree(ts));
  }
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, 0, p);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("    F(4x4,3x3) abl %d grid %d lds %d KB: ", ABL, grid.x, LDS_BYTES / 1024);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const Shape shapes[] = {{"pwc.dc_conv21", 4, 96, 160, 565, 128}, {"pwc.conv2_1", 4, 96, 160, 243, 128}, {"pwc.conv2_3", 4, 96, 160, 467, 64},
                          {"pwc.dc_conv31", 4, 48, 80, 597, 128},  {"gen.conv5", 4, 48, 96, 128, 128},    {"odd", 2, 37, 53, 20, 40},
                          {"one", 1, 16, 32, 8, 64}};
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
    float *x, *w, *b, *y, *yr, *u;
    CHECK(hipMalloc(&x, nx * 4)); CHECK(hipMalloc(&w, nw * 4)); CHECK(hipMalloc(&b, s.Cout * 4));
    CHECK(hipMalloc(&y, ny * 4)); CHECK(hipMalloc(&yr, ny * 4));
    const int np = (s.Cout + 63) / 64 * 64;
    const size_t nu = (size_t)(ldx / 8) * 3 * 2 * 6 * 2 * np * 4 + 65536;
    CHECK(hipMalloc(&u, nu * 4));
    CHECK(hipMemset(u, 0, nu * 4));
    CHECK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), s.Cout * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_conv_kernel, dim3(4096), dim3(256), 0, 0, x, ldx, w, b, yr, s.N, s.H, s.W, s.Cin, s.Cout, 0.1f);
    CHECK(hipDeviceSynchronize());
    std::vector<float> hy(ny), hr(ny);
    CHECK(hipMemcpy(hr.data(), yr, ny * 4, hipMemcpyDeviceToHost));
    const double gflop = 2.0 * s.N * s.H * s.W * (double)s.Cout * s.Cin * 9 * 1e-9;
    printf("%s  N=%d %dx%d %d->%d  %.2f GFLOP\n", s.name, s.N, s.H, s.W, s.Cin, s.Cout, gflop);
    auto report = [&](float us, bool check) {
      CHECK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
      double md = 0, mr = 0;
      for (size_t i = 0; i < ny; ++i) { md = fmax(md, fabs((double)hy[i] - hr[i])); mr = fmax(mr, fabs((double)hr[i])); }
      printf("%8.1f us  %6.1f TFLOP/s (direct-equivalent)", us, gflop / us * 1e3);
      if (check) printf("  max|diff| %.2e / scale %.2e = %.1e", md, mr, md / mr);
      printf("\n");
      CHECK(hipMemset(y, 0, ny * 4));
    };
    report(run<0>(s, x, ldx, w, b, y, reps, u), true);
    if (getenv("W43_ABL")) {
      report(run<1>(s, x, ldx, w, b, y, reps, u), false);   // no DMA
      report(run<8>(s, x, ldx, w, b, y, reps, u), false);   // no LDS reads
      report(run<9>(s, x, ldx, w, b, y, reps, u), false);   // neither
      report(run<2>(s, x, ldx, w, b, y, reps, u), false);   // no MFMA (4 VALU FMAs on one register instead)
    }
    hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(yr); hipFree(u);
  }
  return 0;
}
