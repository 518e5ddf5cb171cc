// nd_amd/csrc/capi.hip -- version, error reporting and event timing of the C ABI
// declared in include/nd_amd.h.
#include <mutex>
#include <vector>

#include "common.hpp"

namespace nd_amd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-kernel timing ring -------------------------------------------------
struct TimingState {
    std::mutex mu;
    bool enabled = false;
    int capacity = 0;
    int used = 0;
    int dropped = 0;
    uint64_t select = 0;          // 0 = every kernel id
    std::vector<hipEvent_t> start, stop;
    std::vector<int32_t> ids;
};
static TimingState g_timing;

KernelTimer::KernelTimer(int kernel_id, hipStream_t s) : slot(-1), stream(s)
{
    TimingState &t = g_timing;
    if (!t.enabled) return;
    std::lock_guard<std::mutex> lk(t.mu);
    if (!t.enabled) return;
    if (t.select != 0 && (kernel_id < 0 || kernel_id > 63 || !((t.select >> kernel_id) & 1ull))) return;
    if (t.used >= t.capacity) {
        t.dropped++;
        return;
    }
    slot = t.used++;
    t.ids[slot] = kernel_id;
    (void)hipEventRecord(t.start[slot], stream);
}

KernelTimer::~KernelTimer()
{
    if (slot < 0) return;
    (void)hipEventRecord(g_timing.stop[slot], stream);
}

}  // namespace nd_amd

using namespace nd_amd;

extern "C" int nd_amd_abi_version(void) { return ND_AMD_ABI_VERSION; }

extern "C" const char *nd_amd_last_error(void) { return g_err; }

extern "C" int nd_amd_timing_enable(int capacity)
{
    TimingState &t = g_timing;
    std::lock_guard<std::mutex> lk(t.mu);
    for (size_t i = 0; i < t.start.size(); i++) {
        (void)hipEventDestroy(t.start[i]);
        (void)hipEventDestroy(t.stop[i]);
    }
    t.start.clear();
    t.stop.clear();
    t.ids.clear();
    t.used = 0;
    t.dropped = 0;
    t.capacity = 0;
    t.enabled = false;
    t.select = 0;
    if (capacity <= 0) return ND_AMD_OK;
    t.start.resize(capacity);
    t.stop.resize(capacity);
    t.ids.assign(capacity, 0);
    for (int i = 0; i < capacity; i++) {
        ND_HIP_CHECK(hipEventCreate(&t.start[i]));
        ND_HIP_CHECK(hipEventCreate(&t.stop[i]));
    }
    t.capacity = capacity;
    t.enabled = true;
    return ND_AMD_OK;
}

extern "C" int nd_amd_timing_collect(int32_t *kernel_ids, float *ms, int max_n, int *n_out)
{
    TimingState &t = g_timing;
    std::lock_guard<std::mutex> lk(t.mu);
    int n = t.used < max_n ? t.used : max_n;
    for (int i = 0; i < n; i++) {
        ND_HIP_CHECK(hipEventSynchronize(t.stop[i]));
        float v = 0.f;
        ND_HIP_CHECK(hipEventElapsedTime(&v, t.start[i], t.stop[i]));
        kernel_ids[i] = t.ids[i];
        ms[i] = v;
    }
    if (n_out) *n_out = n;
    t.used = 0;
    return ND_AMD_OK;
}

extern "C" int nd_amd_timing_select(uint64_t id_mask)
{
    TimingState &t = g_timing;
    std::lock_guard<std::mutex> lk(t.mu);
    t.select = id_mask;
    return ND_AMD_OK;
}

extern "C" int nd_amd_timing_dropped(void)
{
    TimingState &t = g_timing;
    std::lock_guard<std::mutex> lk(t.mu);
    const int d = t.dropped;
    t.dropped = 0;
    return d;
}
