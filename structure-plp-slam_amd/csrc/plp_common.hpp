// Error plumbing and small RAII helpers shared by the host drivers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/plp_front.h"

namespace plp {

plp_status set_error(plp_status s, const char* msg);
plp_status set_hip_error(hipError_t e, const char* expr, const char* file, int line);

#define PLP_HIP(expr)                                                                 \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return ::plp::set_hip_error(_e, #expr, __FILE__, __LINE__); \
    } while (0)
#define PLP_TRY(expr)                        \
    do {                                     \
        plp_status _s = (expr);              \
        if (_s != PLP_OK) return _s;         \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    hipError_t upload(const void* src, size_t n, hipStream_t st) {
        hipError_t e = reserve(n);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, st);
    }
};

}  // namespace plp
