// Optional HIP-event timing of the dominant kernels on their launch stream (bench.py's roofline
// leg).  Off by default; never active under graph capture.  Channel 0 = k_query_attend
// (aggregator), channel 1 = k_conv (embedder).
#pragma once
#include <hip/hip_runtime.h>

namespace dsmil_prof {
constexpr int CH_ATTEND = 0, CH_CONV = 1, NCH = 2;
// records a start event; returns a slot (>= 0) to pass to end(), or -1 when profiling is off
int begin(int channel, hipStream_t st);
void end(int channel, int slot, hipStream_t st);
}  // namespace dsmil_prof
