// Diagnostic (not part of the library): a kernel that only OCCUPIES resources -- `lds_bytes` of LDS per one-wave workgroup (and, with
// regs != 0, ~160 VGPRs) -- and sleeps for `cycles`, to measure what the kernels of a step lose to co-resident region-growing waves
// through each resource alone (tools/experiments/corun_probe.py).   hipcc -O3 --offload-arch=gfx950 -shared -fPIC lds_hog.hip -o liblds_hog.so
#include <hip/hip_runtime.h>

template <bool REGS>
__global__ __launch_bounds__(64) void k_hog(long long cycles, int* sink) {
    extern __shared__ int s[];
    if (threadIdx.x == 0) s[0] = 1;
    float acc[REGS ? 150 : 1];
    if (REGS) {
#pragma unroll
        for (int i = 0; i < 150; ++i) acc[i] = (float)(threadIdx.x + i);
    }
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {
        __builtin_amdgcn_s_sleep(32);
        if (REGS) {
#pragma unroll
            for (int i = 0; i < 150; ++i) acc[i] = acc[i] * 1.0001f + 1.f;
        }
    }
    float r = 0;
    if (REGS) {
#pragma unroll
        for (int i = 0; i < 150; ++i) r += acc[i];
    }
    if (r == 12345.678f || s[0] == 77) *sink = 1;
}

extern "C" int hog_launch(void* stream, int blocks, int lds_bytes, long long cycles, int regs, int* sink) {
    if (regs) hipLaunchKernelGGL(k_hog<true>, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, cycles, sink);
    else hipLaunchKernelGGL(k_hog<false>, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, cycles, sink);
    return (int)hipGetLastError();
}
