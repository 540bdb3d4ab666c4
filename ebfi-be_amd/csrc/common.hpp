// Shared host-side helpers of libebfi_hip.so (error reporting, launch checks, event profiler).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ebfi_hip.h"
#include "ablate_guard.hpp"

namespace ebfi {

// thread-local last-error text; returns `code` so call sites can `return fail(...)`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// hipGetLastError() after a launch -> EBFI_OK / EBFI_ERR_LAUNCH (with message)
int check_launch(const char *what);

// Development switches (EBFI_CONV_*, EBFI_WGRAD_*: kernel selection for A/B runs and tests) are honoured ONLY when the process
// was started with EBFI_DEV=1 (looked up once): a production process never changes kernels, split counts or reduction order
// because of a stray environment variable, and pays no getenv per launch (round-2 advisory).  The switch itself is read on
// every call, so a test can flip it between launches.
const char *dev_getenv(const char *name);

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) to at least `bytes` on the CURRENT
// device, once per (kernel, device), thread-safe; EBFI_OK or EBFI_ERR_LAUNCH with the HIP error text.
int ensure_dynamic_lds(const void *kernel, int bytes);

// RAII hipEvent bracket around one kernel launch (no-op unless ebfi_prof_enable(1)).
class ProfScope {
  public:
    // flops / bytes: ALGORITHMIC work of this launch (0 = not tracked); summed per kernel name
    ProfScope(const char *kernel_name, hipStream_t stream, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;

  private:
    int slot_;
    hipStream_t stream_;
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Dims4 {
    int64_t v[4];
};
static inline Dims4 dims4(const int64_t *p) { return Dims4{{p[0], p[1], p[2], p[3]}}; }

}  // namespace ebfi
