// In-library kernel timing with HIP events on the launch stream (bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>

namespace odpl {
void prof_begin(int kernel_id, hipStream_t s);   // no-op unless enabled for kernel_id
void prof_end(int kernel_id, hipStream_t s);
}  // namespace odpl
