// device_alloc.hpp -- hipMalloc with an optional poison fill (host code only).
// With SEPFWI_POISON=1 in the environment every fresh device allocation of the library is filled with 0xFF bytes (a NaN in every
// float, -1 in every int) before it is handed out: a kernel that reads memory nothing has written yet then poisons its outputs
// instead of silently seeing whatever the allocator left there (zeros in a fresh process, stale data after a free).  GPU
// AddressSanitizer is not available on the target pool; this is the uninitialised-read check that is (scripts/gpu_poison.sh).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

namespace sepfwi {

inline bool poison_allocations() {
    static const bool on = [] {
        const char *e = std::getenv("SEPFWI_POISON");
        return e && std::strcmp(e, "0") != 0 && e[0] != '\0';
    }();
    return on;
}

inline hipError_t dev_malloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison_allocations()) {
        e = hipMemset(*p, 0xFF, bytes);
        // hipMemset on device memory does not block the host, and the session's streams are non-blocking ones that do not wait for
        // the null stream: without this the fill could land AFTER the first kernels that write the buffer
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
}

}  // namespace sepfwi
