// ABI version, error text and the optional per-kernel hipEvent profiler.
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace ebfi {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EBFI_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return EBFI_OK;
}

int ensure_dynamic_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> granted;   // (kernel, device) -> bytes already set
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(EBFI_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    int &have = granted[{kernel, dev}];
    if (have >= bytes) return EBFI_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess)
        return fail(EBFI_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev,
                    hipGetErrorString(e));
    have = bytes;
    return EBFI_OK;
}

namespace {
struct Pending {
    int kernel;
    hipEvent_t start, stop;
    double flops, bytes;
};
struct KernelStat {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0.0;
    double flops = 0.0, bytes = 0.0;   // algorithmic work of the timed launches
};
struct Profiler {
    std::mutex mu;
    bool enabled = false;
    std::vector<KernelStat> stats;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;  // recycled events
    int dropped = 0;
    int capacity = EBFI_PROF_MAX_PENDING;   // pending pairs between two collects (ebfi_prof_set_capacity)

    int kernel_id(const char *name) {
        for (size_t i = 0; i < stats.size(); ++i)
            if (stats[i].name == name) return (int)i;
        stats.push_back(KernelStat{name, 0, 0.0, 0.0, 0.0});
        return (int)stats.size() - 1;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
Profiler &prof() {
    static Profiler p;
    return p;
}
}  // namespace

ProfScope::ProfScope(const char *kernel_name, hipStream_t stream, double flops, double bytes)
    : slot_(-1), stream_(stream) {
    Profiler &p = prof();
    if (!p.enabled) return;  // racy read is fine: enable/disable happens between timed regions
    std::lock_guard<std::mutex> lock(p.mu);
    if ((int)p.pending.size() >= p.capacity) {
        ++p.dropped;
        return;
    }
    Pending pe{p.kernel_id(kernel_name), p.get_event(), p.get_event(), flops, bytes};
    if (!pe.start || !pe.stop) return;
    (void)hipEventRecord(pe.start, stream_);
    p.pending.push_back(pe);
    slot_ = (int)p.pending.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot_ < 0) return;
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    if (slot_ < (int)p.pending.size()) (void)hipEventRecord(p.pending[slot_].stop, stream_);
}

}  // namespace ebfi

namespace ebfi {
const char *dev_getenv(const char *name) {
    static const bool dev = [] { const char *e = getenv("EBFI_DEV"); return e != nullptr && e[0] == '1'; }();
    return dev ? getenv(name) : nullptr;
}
}  // namespace ebfi

using namespace ebfi;

extern "C" {

int ebfi_abi_version(void) { return EBFI_ABI_VERSION; }
const char *ebfi_last_error(void) { return g_err; }

void ebfi_prof_enable(int on) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    p.enabled = on != 0;
}

int ebfi_prof_set_capacity(int max_pending) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    if (max_pending < 1) return fail(EBFI_ERR_ARG, "prof capacity %d", max_pending);
    p.capacity = max_pending;
    return EBFI_OK;
}

void ebfi_prof_reset(void) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    for (auto &pe : p.pending) {
        p.pool.push_back(pe.start);
        p.pool.push_back(pe.stop);
    }
    p.pending.clear();
    p.stats.clear();
    p.dropped = 0;
}

int ebfi_prof_collect(int *dropped) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    int n = 0;
    for (auto &pe : p.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.stop) == hipSuccess &&
            hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
            p.stats[pe.kernel].launches += 1;
            p.stats[pe.kernel].total_ms += ms;
            p.stats[pe.kernel].flops += pe.flops;
            p.stats[pe.kernel].bytes += pe.bytes;
            ++n;
        }
        p.pool.push_back(pe.start);
        p.pool.push_back(pe.stop);
    }
    p.pending.clear();
    if (dropped) *dropped = p.dropped;
    return n;
}

int ebfi_prof_num_kernels(void) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    return (int)p.stats.size();
}

int ebfi_prof_get(int index, const char **name, int64_t *launches, double *total_ms) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    if (index < 0 || index >= (int)p.stats.size()) return fail(EBFI_ERR_ARG, "prof index %d out of range", index);
    if (name) *name = p.stats[index].name.c_str();
    if (launches) *launches = p.stats[index].launches;
    if (total_ms) *total_ms = p.stats[index].total_ms;
    return EBFI_OK;
}

int ebfi_prof_get_work(int index, double *flops, double *bytes) {
    Profiler &p = prof();
    std::lock_guard<std::mutex> lock(p.mu);
    if (index < 0 || index >= (int)p.stats.size()) return fail(EBFI_ERR_ARG, "prof index %d out of range", index);
    if (flops) *flops = p.stats[index].flops;
    if (bytes) *bytes = p.stats[index].bytes;
    return EBFI_OK;
}

}  // extern "C"
