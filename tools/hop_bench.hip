// Dependent-launch latency on this part: a chain of 600 short kernels on ONE stream vs the same chain hopping between N streams
// through events (hipEventRecord + hipStreamWaitEvent), the host far ahead of the GPU.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hop tools/hop_bench.hip && for n in 1 2 4 7; do /tmp/hop $n 1000; done
//   GPU_MAX_HW_QUEUES=8 /tmp/hop 7 1000      (more than four hardware queues in use: see DESIGN.md 6.3)
// profiles/r03_hop_bench.txt: 3.8 us per same-stream dependent launch, 12 us per cross-stream hop, 79 us once seven streams sit
// on seven hardware queues -- why the plan keeps ROCm's default of four hardware queues for its six lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void spin(long cycles, int* sink) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1024) *sink = 1;
}
int main(int argc, char** argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 2, hops = 600;
  const long cyc = argc > 2 ? atol(argv[2]) : 1000;  // wall_clock64: 100 MHz -> 1000 = 10 us
  std::vector<hipStream_t> ss(ns);
  for (auto& s : ss) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(hops);
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    // a long head kernel lets the host run ahead of the GPU
    spin<<<1, 64, 0, ss[0]>>>(2000000, nullptr);  // 20 ms
    (void)hipEventRecord(a, ss[0]);
    for (int h = 0; h < hops; ++h) {
      hipStream_t s = ss[h % ns];
      if (h && ns > 1) (void)hipStreamWaitEvent(s, ev[h - 1], 0);
      spin<<<1, 64, 0, s>>>(cyc, nullptr);
      (void)hipEventRecord(ev[h], s);
    }
    hipStream_t last = ss[(hops - 1) % ns];
    (void)hipEventRecord(b, last);
    (void)hipStreamSynchronize(last);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep == 2) printf("streams %d  kernel %.1f us  per dependent launch %.2f us  (overhead %.2f us)\n", ns, cyc / 100.0, ms * 1e3 / hops, ms * 1e3 / hops - cyc / 100.0);
  }
  return 0;
}
