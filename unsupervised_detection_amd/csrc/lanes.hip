// Which of a process's streams can actually run side by side?  ROCm maps every stream onto one of GPU_MAX_HW_QUEUES (default 4)
// hardware queues when the stream is created (least-referenced queue first); two streams on one queue execute in submission order.
// The probe: a ~200 us spin kernel on `a`, then an empty kernel on `b`; `b`'s kernel finishing before `a`'s spin means different queues.
#include "lanes.h"

#include "common.h"

namespace udet {

__global__ void lane_spin_kernel(long ticks) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
__global__ void lane_noop_kernel() {}

// one probe: the (100 MHz) tick count of the spin is ~200 us -- a host that needs 100 us between the two launches still sees `b`
// finish first when the streams do overlap
static int probe_once(hipStream_t a, hipStream_t b, bool* concurrent) {
  hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
  struct Guard {  // (the error paths below return early)
    hipEvent_t *a, *b, *c;
    ~Guard() { for (hipEvent_t* e : {a, b, c}) if (*e) (void)hipEventDestroy(*e); }
  } guard{&e0, &ea, &eb};
  UDET_HIP(hipEventCreate(&e0));
  UDET_HIP(hipEventCreate(&ea));
  UDET_HIP(hipEventCreate(&eb));
  // the first launch of a kernel on a device resolves its code object: never inside the timed window (issued every time -- two
  // empty launches cost less than keeping a per-device, thread-safe "warm" flag)
  lane_spin_kernel<<<1, 64, 0, a>>>(1);
  lane_noop_kernel<<<1, 64, 0, b>>>();
  UDET_HIP(hipStreamSynchronize(a));
  UDET_HIP(hipStreamSynchronize(b));
  UDET_HIP(hipEventRecord(e0, a));
  lane_spin_kernel<<<1, 64, 0, a>>>(20000);
  UDET_HIP(hipEventRecord(ea, a));
  lane_noop_kernel<<<1, 64, 0, b>>>();
  UDET_HIP(hipEventRecord(eb, b));
  UDET_HIP(hipEventSynchronize(ea));
  UDET_HIP(hipEventSynchronize(eb));
  float ta = 0.f, tb = 0.f;
  UDET_HIP(hipEventElapsedTime(&ta, e0, ea));
  UDET_HIP(hipEventElapsedTime(&tb, e0, eb));
  *concurrent = tb < 0.6f * ta;
  return UDET_OK;
}

// A positive answer is proof (the empty kernel cannot overtake a spin on its own queue); a negative one may be a slow host or a GPU
// shared with another process: repeated, any positive result counts.
int streams_concurrent(hipStream_t a, hipStream_t b, bool* concurrent) {
  *concurrent = false;
  if (a == b) return UDET_OK;
  for (int attempt = 0; attempt < 3 && !*concurrent; ++attempt) UDET_TRY(probe_once(a, b, concurrent));
  return UDET_OK;
}

}  // namespace udet
