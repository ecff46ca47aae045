// Winograd weight transform U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], as one element of the re-layout jobs
// (pack modes 7 / 8: conv_host.hip pack_jobs_kernel, conv_wino.hip wino_pack_kernel).
#pragma once
#include "common.h"

namespace udet {

// element e of dst [Kc / 8][16][2][np = j.ldw][4] from the HWIO source [9][R = Cin][C = Cout]; mode 7: K = input channels (gap map of
// the slab padding), N = output channels; mode 8 (backward-data): K = output channels, N = input channels, taps mirrored
__device__ __forceinline__ float wino_pack_elem(const PackJob& j, const float* __restrict__ src, const float* __restrict__ gamma, float bn_c, long e) {
  const int jj = (int)(e & 3);
  const long r0 = e >> 2;
  const int n = (int)(r0 % j.ldw);
  const long r1 = r0 / j.ldw;
  const int kh = (int)(r1 & 1), pos = (int)((r1 >> 1) & 15), kg = (int)(r1 >> 5);
  const int k = kg * 8 + kh * 4 + jj;
  int ks = k;
  if (k >= j.k_split) ks = (k < j.k_split + j.k_gap) ? -1 : k - j.k_gap;
  const int ci = j.mode == 7 ? ks : n, co = j.mode == 7 ? n : ks;
  if (ci < 0 || ci >= j.R || co < 0 || co >= j.C) return 0.f;
  const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int pi = pos >> 2, pj = pos & 3;
  float val = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int ky = j.mode == 7 ? a : 2 - a, kx = j.mode == 7 ? b : 2 - b;
      val += Gm[pi][a] * Gm[pj][b] * src[((long)(ky * 3 + kx) * j.R + ci) * j.C + co];
    }
  if (gamma) val *= gamma[co] * bn_c;
  return val;
}

}  // namespace udet
