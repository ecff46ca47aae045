// Ceiling probe for the fp32 MFMA pipe of one MI355X (what the implicit-GEMM main loops can reach at best):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak      (profiles/rNN_mfma_peak.txt)
// Every variant issues the same number of v_mfma_f32_32x32x2_f32 per wave on four independent accumulators (the
// 64x64 wave tile of the 128x128 kernels) and differs in what surrounds them:
//   mode 0  MFMAs only
//   mode 1  + the LDS fragment reads of a 128x128x32 stage (2 ds_read_b128 + 8 ds_read_b32 per 16 MFMAs), prefetched one
//             group ahead
//   mode 2  + one s_barrier per 64 MFMAs (the stage handover) among the workgroup's waves
//   mode 3  mode 2 with 4 extra idle waves per workgroup that only take part in the barrier (the staging waves' slot)
//   mode 4  mode 3, and those waves fetch the next stage (NJ x 4 KB) with global_load_lds_dwordx4 -- one stage in flight and vmcnt(0)
//           before the barrier (the protocol of conv_igemm_dma_kernel), or two in flight with a 3-buffer ring: every 8 lanes read one
//           128-byte line, lines `stride` bytes apart, each workgroup cycling through its own `region` bytes (a power of two of
//           lines) of a global buffer.  The address arithmetic is a handful of VALU operations on purpose: with a 64-bit modulo per
//           load the same probe loses a third of its MFMA rate (VALU work of a staging wave is paid by the MFMA wave of its SIMD).
// Reported: TFLOP/s over the whole chip at `wgs_per_cu` resident workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int NJ = 8, int NS = 2>
__global__ __launch_bounds__(512) void probe(float* out, int stages, const float* src = nullptr, long region = 0, int stride = 128) {
  __shared__ __attribute__((aligned(16))) float As[NS][128][32];
  __shared__ __attribute__((aligned(16))) float Bs[NS][32][128];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * 128 * 32; i += blockDim.x) {
    (&As[0][0][0])[i] = 1.0f / (1 + (i & 7));
    (&Bs[0][0][0])[i] = 0.5f;
  }
  __syncthreads();
  if (tid >= 256) {  // mode 3: barrier-only waves; mode 4: staging waves
    if (MODE == 4) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      const int t = tid - 256, w = t >> 6, l = t & 63;
      const char* base = reinterpret_cast<const char*>(src) + (long)blockIdx.x * region;
      unsigned cur = 0;  // line index inside the region
      const unsigned mask = (unsigned)(region / stride) - 1;  // (power of two) lines of the region
      const unsigned lane_off = (l & 7) * 16, lane_line = (unsigned)(l >> 3);
      int buf = 1;
      auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {  // NJ instructions x 4 waves x 1 KB per stage (8: 32 KB = 256 lines)
          const unsigned line = (cur + (j * 4 + w) * 8 + lane_line) & mask;  // (address arithmetic kept to a handful of VALU operations)
          const char* a = base + (line * (unsigned)stride + lane_off);
          float* dst = j < 4 ? &As[buf][0][0] + (j * 4 + w) * 256 : &Bs[buf][0][0] + ((j - 4) * 4 + w) * 256;
          __builtin_amdgcn_global_load_lds(a, (lds_ptr)dst, 16, 0, 0);
        }
        cur += 256;
        buf = buf + 1 == NS ? 0 : buf + 1;
      };
      if (NS > 2) issue();  // a second stage in flight
      for (int s = 0; s < stages; ++s) {
        issue();
        if (NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
    for (int s = 0; s < stages; ++s) __builtin_amdgcn_s_barrier();
    return;
  }
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int swz = (li >> 1) & 7;
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 a[2][2];
  float b[2][4][2];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 2; ++i) {
      a[s][i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
      for (int e = 0; e < 4; ++e) b[s][e][i] = 0.5f;
    }
  int buf = 0;
  for (int s = 0; s < stages; ++s) {
    auto frag = [&](int q, int kk) {
      const int g = 2 * kk + lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[q][i] = *reinterpret_cast<const float4*>(&As[buf][wm * 64 + i * 32 + li][(g ^ swz) * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[q][e][j] = Bs[buf][g * 4 + e][wn * 64 + j * 32 + li];
    };
    if (MODE >= 1) frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1 && kk + 1 < 4) frag((kk + 1) & 1, kk + 1);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float av = e == 0 ? a[kk & 1][i].x : (e == 1 ? a[kk & 1][i].y : (e == 2 ? a[kk & 1][i].z : a[kk & 1][i].w));
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kk & 1][e][j], acc[i][j], 0, 0, 0);
        }
      if (MODE >= 1 && kk + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
    if (MODE >= 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    buf = buf + 1 == NS ? 0 : buf + 1;
  }
  float v = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) v += acc[i][j][r];
  if (v == 12345.678f) out[tid] = v;
}

template <int MODE, int NJ = 8, int NS = 2>
static void run(int threads, int wgs_per_cu, int stages, float* out, const float* src = nullptr, long region = 0, int stride = 128) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  probe<MODE, NJ, NS><<<grid, threads>>>(out, stages, src, region, stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) probe<MODE, NJ, NS><<<grid, threads>>>(out, stages, src, region, stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = (double)grid * 4 * stages * 64 * (2.0 * 32 * 32 * 2);
  printf("mode %d  threads %3d  wg/cu %d  stages %d: %8.1f us  %6.1f TFLOP/s  (%.2f us per stage)", MODE, threads, wgs_per_cu, stages, ms * 1e3,
         flops / ms * 1e-9, ms * 1e3 / stages / wgs_per_cu);
  if (MODE == 4) printf("  %d KB/stage, %d in flight, region %ld KB stride %d: %.1f GB/s per CU, %.2f TB/s chip", NJ * 4, NS - 1, region >> 10, stride,
                        NJ * 4096.0 * stages * wgs_per_cu / ms * 1e-6, NJ * 4096.0 * stages * grid / ms * 1e-9);
  printf("\n");
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int stages = 200;
  float* src;
  const long cap = 1L << 30;
  hipMalloc(&src, cap);
  hipMemset(src, 0, cap);
  for (int w = 1; w <= 2; ++w) {
    run<0>(256, w, stages, out);
    run<1>(256, w, stages, out);
    run<2>(256, w, stages, out);
    run<3>(512, w, stages, out);
    const long regions[] = {256L << 10, 2048L << 10};
    for (long region : regions) {  // L2-resident ... MALL / HBM
      if (region * 256 * w > cap) continue;
      run<4, 8, 2>(512, w, stages, out, src, region, 512);
      run<4, 4, 2>(512, w, stages, out, src, region, 512);
      run<4, 2, 2>(512, w, stages, out, src, region, 512);
      if (w == 1) {
        run<4, 8, 3>(512, w, stages, out, src, region, 512);
        run<4, 4, 3>(512, w, stages, out, src, region, 512);
      }
    }
  }
  return 0;
}
