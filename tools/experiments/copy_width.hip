// What a copy achieves on this chip as a function of the bytes per lane and access (1, 4, 16) and of how the work is cut into workgroups:
// the tile kernels of the library stage one dword per lane and write one dword per lane and all sit near 1.7-1.9 TB/s of algorithmic bytes.
//   hipcc --offload-arch=gfx950 -O3 -o build_exp/copy_width tools/experiments/copy_width.hip && build_exp/copy_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <class T> __global__ __launch_bounds__(256) void k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// one 128 x 32-byte tile per workgroup of 256 threads, a dword per lane and row, as k_blur7 reads and writes (no halo, no compute); pitch 640, 480 rows
__global__ __launch_bounds__(256) void k_tile_dword(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, size_t frame_stride) {
    const int tiles_x = w / 128, t = blockIdx.x, f = blockIdx.y;
    const int tx0 = (t % tiles_x) * 128, ty0 = (t / tiles_x) * 32, c = (threadIdx.x & 31) * 4, r0 = (threadIdx.x >> 5) * 4;
    const uint8_t* s = src + (size_t)f * frame_stride; uint8_t* d = dst + (size_t)f * frame_stride;
    uint32_t v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const uint32_t*>(s + (size_t)(ty0 + r0 + r) * w + tx0 + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<uint32_t*>(d + (size_t)(ty0 + r0 + r) * w + tx0 + c) = v[r];
}
// the same tile with 16 bytes per lane: 8 lanes per row, 32 rows per 256 threads
__global__ __launch_bounds__(256) void k_tile_x4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, size_t frame_stride) {
    const int tiles_x = w / 128, t = blockIdx.x, f = blockIdx.y;
    const int tx0 = (t % tiles_x) * 128, ty0 = (t / tiles_x) * 32, c = (threadIdx.x & 7) * 16, r = threadIdx.x >> 3;
    const uint8_t* s = src + (size_t)f * frame_stride; uint8_t* d = dst + (size_t)f * frame_stride;
    const uint4 v = *reinterpret_cast<const uint4*>(s + (size_t)(ty0 + r) * w + tx0 + c);
    *reinterpret_cast<uint4*>(d + (size_t)(ty0 + r) * w + tx0 + c) = v;
}
template <class F> float time_ms(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int W = 640, H = 480, B = 2048;
    const size_t n = (size_t)W * H * B;
    uint8_t *src, *dst; hipMalloc(&src, n); hipMalloc(&dst, n); hipMemset(src, 1, n); hipMemset(dst, 0, n);
    auto report = [&](const char* what, float ms) { printf("%-58s %7.3f ms  %6.2f TB/s (read + write)\n", what, ms, 2.0 * n / ms / 1e9); };
    for (int wg_per_cu : {8, 16, 32}) {
        const int g = 256 * wg_per_cu; char buf[96];
        snprintf(buf, 96, "grid-stride copy, 1 byte per lane, %d workgroups/CU", wg_per_cu); report(buf, time_ms([&] { hipLaunchKernelGGL(k_copy<uint8_t>, dim3(g), dim3(256), 0, 0, src, dst, n); }));
        snprintf(buf, 96, "grid-stride copy, 4 bytes per lane, %d workgroups/CU", wg_per_cu); report(buf, time_ms([&] { hipLaunchKernelGGL(k_copy<uint32_t>, dim3(g), dim3(256), 0, 0, (const uint32_t*)src, (uint32_t*)dst, n / 4); }));
        snprintf(buf, 96, "grid-stride copy, 16 bytes per lane, %d workgroups/CU", wg_per_cu); report(buf, time_ms([&] { hipLaunchKernelGGL(k_copy<uint4>, dim3(g), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, n / 16); }));
    }
    report("128 x 32 tile per workgroup, dword per lane and row (blur7 shape)", time_ms([&] { hipLaunchKernelGGL(k_tile_dword, dim3((W / 128) * (H / 32), B), dim3(256), 0, 0, src, dst, W, H, (size_t)W * H); }));
    report("128 x 32 tile per workgroup, 16 bytes per lane", time_ms([&] { hipLaunchKernelGGL(k_tile_x4, dim3((W / 128) * (H / 32), B), dim3(256), 0, 0, src, dst, W, H, (size_t)W * H); }));
    return 0;
}
