// Error plumbing and small RAII helpers shared by the host drivers.
#pragma once
#include <cstring>
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
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }   // (hipFree waits for the device's outstanding work)
    hipError_t upload(const void* src, size_t n, hipStream_t st) {
        hipError_t e = reserve(n);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, st);
    }
};

// Page-locked bounce buffer of a context for the host-pointer entry points: the caller's image is copied into it by the CPU and
// goes to the device from there, so that no DMA / blit ever addresses the caller's pageable memory (whose pages the caller may
// unmap right after the call; a randomised sweep that allocated a fresh image per call hit rare GPU page faults otherwise).
struct HostPinned {
    void* p = nullptr;
    size_t bytes = 0;
    HostPinned() = default;
    HostPinned(const HostPinned&) = delete;
    HostPinned& operator=(const HostPinned&) = delete;
    ~HostPinned() { if (p) (void)hipHostFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        if (n == 0) n = 16;
        // one page of slack behind the payload: a copy engine / blit kernel that fetches its source in 16-byte (or wider) pieces may
        // touch a few bytes past the last requested one, and the page after a host allocation need not be mapped
#ifdef PLP_NO_STAGING_SLACK      // diagnostic build only (tools/build_variant.sh): the exactly-sized staging buffer of early round 2
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
#else
        hipError_t e = hipHostMalloc(&p, n + 4096, hipHostMallocDefault);
#endif
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    // rows x cols bytes from a pitched host image into the buffer at byte offset off, densely packed (pitch = cols)
    void pack(size_t off, const uint8_t* src, size_t step, int rows, int cols) {
        uint8_t* d = static_cast<uint8_t*>(p) + off;
        if (step == (size_t)cols) memcpy(d, src, (size_t)rows * cols);
        else for (int y = 0; y < rows; ++y) memcpy(d + (size_t)y * cols, src + (size_t)y * step, (size_t)cols);
    }
};

}  // namespace plp
