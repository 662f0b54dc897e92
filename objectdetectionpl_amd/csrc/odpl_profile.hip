// Event-pair pool per profiled kernel.  Disabled (one predictable branch per launch) by default.
#include <vector>
#include "odpl_profile.hpp"
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace odpl {
namespace {
constexpr int kMaxPairs = 4096;
struct Slot {
    bool on = false;
    int used = 0;
    bool open = false;
    std::vector<hipEvent_t> pre, start, stop;
};
Slot g_slots[ODPL_PROF_NUM_KERNELS];
}  // namespace

void prof_begin(int id, hipStream_t s) {
    Slot& sl = g_slots[id];
    if (!sl.on || sl.used >= kMaxPairs) return;
    if ((int)sl.start.size() <= sl.used) {
        hipEvent_t p, a, b;
        if (hipEventCreate(&p) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        sl.pre.push_back(p); sl.start.push_back(a); sl.stop.push_back(b);
    }
    // pre -> start is an empty interval: it measures what an event pair costs on this stream, so that
    // collect() can report (stop - start) - (start - pre), the kernel alone
    sl.open = hipEventRecord(sl.pre[sl.used], s) == hipSuccess && hipEventRecord(sl.start[sl.used], s) == hipSuccess;
}

void prof_end(int id, hipStream_t s) {
    Slot& sl = g_slots[id];
    if (!sl.on || !sl.open) return;
    if (hipEventRecord(sl.stop[sl.used], s) == hipSuccess) sl.used++;
    sl.open = false;
}
}  // namespace odpl

extern "C" int odpl_profile_enable(int32_t kernel_id, int32_t on) {
    if (kernel_id < 0 || kernel_id >= ODPL_PROF_NUM_KERNELS) return ODPL_EINVAL;
    odpl::g_slots[kernel_id].on = on != 0;
    return 0;
}

extern "C" int odpl_profile_collect(int32_t kernel_id, float* host_ms, float* raw_ms, int32_t max_n) {
    if (kernel_id < 0 || kernel_id >= ODPL_PROF_NUM_KERNELS || (max_n > 0 && !host_ms)) return ODPL_EINVAL;
    odpl::Slot& sl = odpl::g_slots[kernel_id];
    int n = 0;
    for (int i = 0; i < sl.used; ++i) {
        if (hipEventSynchronize(sl.stop[i]) != hipSuccess) continue;
        float ms = 0.f, null_ms = 0.f;
        if (hipEventElapsedTime(&ms, sl.start[i], sl.stop[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&null_ms, sl.pre[i], sl.start[i]) != hipSuccess) continue;
        if (raw_ms && n < max_n) raw_ms[n] = ms;
        ms -= null_ms;
        if (n < max_n) host_ms[n] = ms > 0.f ? ms : 0.f;
        n++;
    }
    sl.used = 0;
    return n;
}
