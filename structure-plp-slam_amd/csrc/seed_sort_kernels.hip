// LSD seed order exactly as a reference built with libstdc++ produces it (closes definition D1; seed_sort_model.hpp has the
// derivation and the host model).  lsd.cpp's ll_angle sorts ALL (sw - 1)(sh - 1) pixels by gradient bin with std::sort; the kernel
// replays std::__introsort_loop on that array as rank-paired partitions, the stable counting sort of k_lsd_order_entries is the final
// insertion sort.
//
// One workgroup of 16 waves per frame, the array in HBM / L2 (`ent`, 4 bytes per pixel: pixel | defined << 19 | bin << 20):
//   G  segments longer than the LDS window (kSsT entries): the whole workgroup partitions one segment in global memory
//      (wg_partition<true>: chunk masks -> prefix sums -> partner positions by rank -> swaps; four barriers)
//   W  a segment that fits is loaded into LDS and finished there:
//      - above kSsTask entries the whole workgroup partitions it (wg_partition<false>, the same code on LDS)
//      - 65 .. kSsTask entries: ONE WAVE per segment, chunk masks in registers (lane c = chunk c), a task queue in LDS, no barriers
//      - 17 .. 64 entries: one LANE per segment runs the library's sequential loop
//      - at most 16 entries: left to the final insertion sort (= the counting sort)
//   The recursion budget travels with every segment; where it runs out one lane runs the library's heap sort (never seen on an
//   image; forced in the tests).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "line_device.hpp"
#include "plp_barrier.hpp"
#include "seed_sort_model.hpp"

namespace plp {

// Two configurations of one body (seed_sort_impl.inc).  Batches: 4 waves per workgroup and a 4096-entry window (35 KB of LDS) -- alone that is the
// slower kernel (6.0 against 5.0 ms per 2048 frames) but four such workgroups share a CU with each other and with the other streams' kernels, and
// the STEP is what counts: 27.6 ms against 32.7 ms with 16 waves and 144 KB (profiles/r04_seed_sort.md).  Small batches (the single-frame call of
// data/frame.cc:1146-1163): 16 waves and a 24576-entry window, the shortest time per frame.
#define SS_NS ss_thr
#ifndef PLP_SS_THR_WAVES
#define PLP_SS_THR_WAVES 4
#endif
#ifndef PLP_SS_THR_T
#define PLP_SS_THR_T 4096
#endif
#define SS_WAVES PLP_SS_THR_WAVES
#define SS_T PLP_SS_THR_T
#ifndef PLP_SS_THR_MINW
#define PLP_SS_THR_MINW 4   /* at most 128 VGPRs: four workgroups of four waves per CU (5.06 against 6.06 ms alone, 27.1 against 27.7 ms in the step) */
#endif
#define SS_MINW PLP_SS_THR_MINW
#ifndef PLP_SS_THR_TASK
#define PLP_SS_THR_TASK 4096
#endif
#define SS_TASK PLP_SS_THR_TASK
#include "seed_sort_impl.inc"
#undef SS_NS
#undef SS_WAVES
#undef SS_T
#undef SS_MINW
#undef SS_TASK
#define SS_NS ss_lat
#define SS_WAVES 16
#define SS_T 24576
#define SS_MINW 1
#define SS_TASK 4096
#include "seed_sort_impl.inc"
#undef SS_NS
#undef SS_WAVES
#undef SS_T
#undef SS_MINW
#undef SS_TASK

constexpr int kSsLatMaxFrames = 256;   // batches up to this many frames take the 16-wave configuration (one workgroup per CU)

size_t seed_sort_ws_entries(size_t nv) { return 2 + 2 * ((nv / 2 + 64 + 1) & ~(size_t)1) + 2 * 2 * ((nv + 63) / 64 + 1); }   // u32 units per frame: n_live, two rank-indexed entry lists, chunk masks

// The final insertion sort = a stable counting sort by bin of the DEFINED entries, in array order (k_lsd_order's steps 2-4 on the
// permuted array instead of the row-major pixel sequence).
__global__ __launch_bounds__(256) void k_lsd_order_entries(LinePlanes P, const uint32_t* __restrict__ ent_all, const uint32_t* __restrict__ ws_all, size_t ws_stride) {
    __shared__ uint32_t cnt[4][1024];
    __shared__ uint32_t s_wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int nv_all = (P.sw - 1) * (P.sh - 1);
    // the sort's live length: behind it lie right parts of partitions whose pivot bin was below the bin of the smallest defined magnitude -- undefined pixels
    // and stale copies of entries that moved to the left (seed_sort_impl.inc wg_partition): not part of the array any more
    const int nv = min(nv_all, (int)__builtin_amdgcn_readfirstlane((int)ws_all[(size_t)b * ws_stride]));
    for (int i = tid; i < 4096; i += 256) (&cnt[0][0])[i] = 0;
    wg_barrier();
    const uint32_t* ent = ent_all + (size_t)b * nv_all;
    uint32_t* order = P.order + (size_t)b * nv_all;
    const int ngroups = (nv + 63) / 64, gper = (ngroups + 3) / 4, g0 = q * gper, g1 = min(ngroups, g0 + gper);
    uint32_t* comp = P.reg + (size_t)b * P.reg_frame_stride + (size_t)g0 * 64;
    int ncomp = 0;
    for (int gb = g0; gb < g1; gb += 4) {
        uint32_t v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (gb + u) * 64 + lane;
            v4[u] = (gb + u < g1 && i < nv) ? ent[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t v = v4[u];
            const unsigned long long defm = __ballot((v & ss_thr::kSsDefBit) != 0);
            if (!defm) continue;
            if (v & ss_thr::kSsDefBit) {
                atomicAdd(&cnt[q][v >> ss_thr::kSsShift], 1u);
                comp[ncomp + __popcll(defm & ((1ull << lane) - 1ull))] = v & ~ss_thr::kSsDefBit;
            }
            ncomp += __popcll(defm);
        }
    }
    wg_barrier();
    {   // thread t owns bins 1023-4t .. 1020-4t (descending)
        const int v0 = 1023 - 4 * tid;
        uint32_t c[4][4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[j][k] = cnt[k][v0 - j]; tot += c[j][k]; }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_wsum[q] = inc;
        wg_barrier();
        uint32_t run = inc - tot;
        for (int k = 0; k < q; ++k) run += s_wsum[k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { cnt[k][v0 - j] = run; run += c[j][k]; }
        if (tid == 255) P.n_order[b] = (int32_t)run;
    }
    wg_barrier();
    for (int ib = 0; ib < ncomp; ib += 256) {
        uint32_t e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e4[u] = ib + 64 * u + lane < ncomp ? comp[ib + 64 * u + lane] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i0 = ib + 64 * u;
            if (i0 >= ncomp) break;
            const bool valid = i0 + lane < ncomp;
            const uint32_t e = e4[u];
            const unsigned v = e >> ss_thr::kSsShift;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 10; ++bit) {
                const unsigned long long m = __ballot((v >> bit) & 1u);
                peers &= ((v >> bit) & 1u) ? m : ~m;
            }
            if (valid) {
                const int rank = __popcll(peers & ((1ull << lane) - 1ull));
                order[cnt[q][v] + rank] = e & kLsdSeedPixMask;
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0) cnt[q][v] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

void launch_seed_order_exact(hipStream_t st, const LinePlanes& P, const LsdParams& lp, int B, uint32_t* ent, uint32_t* ws, size_t ws_stride) {
    const int n = P.sw * P.sh;
    const size_t nv = (size_t)(P.sw - 1) * (P.sh - 1);
    if (B <= kSsLatMaxFrames) hipLaunchKernelGGL(ss_lat::k_lsd_seed_sort, dim3(B), dim3(ss_lat::kSsThreads), ss_lat::lds_bytes(nv), st, P, lp, (n + 255) / 256, ent, ws, ws_stride);
    else hipLaunchKernelGGL(ss_thr::k_lsd_seed_sort, dim3(B), dim3(ss_thr::kSsThreads), ss_thr::lds_bytes(nv), st, P, lp, (n + 255) / 256, ent, ws, ws_stride);
    hipLaunchKernelGGL(k_lsd_order_entries, dim3(B), dim3(256), 0, st, P, (const uint32_t*)ent, (const uint32_t*)ws, ws_stride);
}
hipError_t seed_sort_configure() {
    const void* fns[4] = {reinterpret_cast<const void*>(ss_thr::k_lsd_seed_sort), reinterpret_cast<const void*>(ss_thr::k_seed_sort_debug),
                          reinterpret_cast<const void*>(ss_lat::k_lsd_seed_sort), reinterpret_cast<const void*>(ss_lat::k_seed_sort_debug)};
    const size_t bytes[4] = {ss_thr::lds_bytes(kLsdMaxScaledPixels), ss_thr::lds_bytes(kLsdMaxScaledPixels), ss_lat::lds_bytes(kLsdMaxScaledPixels), ss_lat::lds_bytes(kLsdMaxScaledPixels)};
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes[i]);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// variant 0: the configuration of large batches, 1: of small ones
void launch_seed_sort_debug(hipStream_t st, uint32_t* ent, int n, int depth, uint32_t skip_key, uint32_t* ws, int32_t* status, int* dbg, int variant, int copies) {
    const size_t ws_stride = seed_sort_ws_entries((size_t)n);
    if (variant) hipLaunchKernelGGL(ss_lat::k_seed_sort_debug, dim3(copies), dim3(ss_lat::kSsThreads), ss_lat::lds_bytes((size_t)n), st, ent, n, depth, skip_key, ws, ws_stride, status, dbg);
    else hipLaunchKernelGGL(ss_thr::k_seed_sort_debug, dim3(copies), dim3(ss_thr::kSsThreads), ss_thr::lds_bytes((size_t)n), st, ent, n, depth, skip_key, ws, ws_stride, status, dbg);
}

}  // namespace plp
