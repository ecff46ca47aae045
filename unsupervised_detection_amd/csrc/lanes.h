#pragma once
#include <hip/hip_runtime.h>

namespace udet {
// *concurrent = work submitted to `a` and `b` can overlap (the streams sit on different hardware queues); synchronises both streams
int streams_concurrent(hipStream_t a, hipStream_t b, bool* concurrent);
}  // namespace udet
