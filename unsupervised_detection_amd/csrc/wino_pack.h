// Winograd weight transform U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], as one work item of the re-layout jobs
// (pack modes 7 / 8: conv_host.hip pack_jobs_kernel, conv_wino.hip wino_pack_kernel).
#pragma once
#include "common.h"

namespace udet {

// Work item `it` of a job = one (8-channel group kg, lane half kh, column n): the four channels' 3x3 filters are read ONCE (36 loads,
// coalesced along n) and all 16 positions' float4 are written (16 stores of 16 bytes, contiguous along n).  dst
// [Kc / 8][16][2][np = j.ldw][4] from the HWIO source [9][R = Cin][C = Cout]; mode 7: K = input channels (gap map of the slab padding),
// N = output channels; mode 8 (backward-data): K = output channels, N = input channels, taps mirrored.  j.total = (Kc / 8) * 2 * np items.
// (The first form computed one float per thread: 9 strided loads and the whole G g G^T per element -- the trainable networks' per-step
// re-layout launch went from 40 to 52 us per network with the Winograd operands in it.)
__device__ __forceinline__ void wino_pack_item(const PackJob& j, const float* __restrict__ src, const float* __restrict__ gamma, float bn_c,
                                               float* __restrict__ dst, long it) {
  const int np = j.ldw;
  const int n = (int)(it % np);
  const long r = it / np;
  const int kh = (int)(r & 1), kg = (int)(r >> 1);
  float g[4][9];  // [channel of the quad][tap a * 3 + b]
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int k = kg * 8 + kh * 4 + jj;
    int ks = k;
    if (k >= j.k_split) ks = (k < j.k_split + j.k_gap) ? -1 : k - j.k_gap;
    const int ci = j.mode == 7 ? ks : n, co = j.mode == 7 ? n : ks;
    const bool ok = ci >= 0 && ci < j.R && co >= 0 && co < j.C;
    const float sc = (ok && gamma) ? gamma[co] * bn_c : 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ky = j.mode == 7 ? a : 2 - a, kx = j.mode == 7 ? b : 2 - b;
        g[jj][a * 3 + b] = ok ? src[((long)(ky * 3 + kx) * j.R + ci) * j.C + co] * sc : 0.f;
      }
  }
  float4* out = reinterpret_cast<float4*>(dst) + ((long)kg * 32 + kh) * np + n;  // position p: + p * 2 * np float4
  float u[4][16];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    // rows: t[i][b] = (G g)[i][b], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float g0 = g[jj][b], g1 = g[jj][3 + b], g2 = g[jj][6 + b];
      t[0][b] = g0;
      t[1][b] = 0.5f * (g0 + g1 + g2);
      t[2][b] = 0.5f * (g0 - g1 + g2);
      t[3][b] = g2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[jj][i * 4 + 0] = t[i][0];
      u[jj][i * 4 + 1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[jj][i * 4 + 2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[jj][i * 4 + 3] = t[i][2];
    }
  }
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) out[(long)pos * 2 * np] = make_float4(u[0][pos], u[1][pos], u[2][pos], u[3][pos]);
}

}  // namespace udet
