// Workgroup barrier whose wait for the wave's own LDS writes cannot be optimised away.
//
// `__syncthreads()` is a workgroup-scope release fence, `s_barrier`, and an acquire fence.  On gfx950 the release fence becomes `s_waitcnt lgkmcnt(0)` -- as a SOFT wait,
// which the compiler's wait-count insertion pass is free to delete wherever its scoreboard shows no LDS operation pending.  ROCm 7.2's pass deletes it at a barrier that
// HEADS A LOOP when nothing is pending on the path from the function's entry, although the loop's back edge arrives with LDS writes in flight (the instruction is gone by
// the time the back edge's state reaches the header).  Round 6 found this in diagnostic builds of the seed sort (profiles/r06_seed_sort.md section 4: the barrier at the top
// of the loop over a frame's global partitions; wave 0's pushes to the segment stack were still in the LDS queue when the other waves read the stack), and in the debug
// entry's kernel of every build: the "failure that needs a second dispatch on the chip" of rounds 4 - 6 -- a busy LDS pipeline stretches the window from never to always.
// Whether the production kernel kept its wait depended on unrelated code before the loop.
//
// wg_barrier() therefore issues the wait itself, as inline assembly the compiler cannot drop (where it would have waited anyway the second wait costs one issue slot).
// tools/isa_barrier_check.py / tests/test_kernel_resources.py prove on the ISA the build keeps (csrc/build/*.s) that EVERY s_barrier of EVERY kernel of the library is
// reached with the wave's LDS writes drained on every path -- kernels that still use __syncthreads() included.
#pragma once
#include <hip/hip_runtime.h>

namespace plp {
#ifdef PLP_SOFT_BARRIERS   // DIAGNOSTIC build only: the barriers as they were until round 6 (with -DPLP_SS_VADDR_GLOBAL this is the reproducer of the failure)
__device__ __forceinline__ void wg_barrier() { __syncthreads(); }
__device__ __forceinline__ void wg_barrier_after_global_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#else
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
// ... and the wave's own global loads / stores too: for data that goes from wave to wave of a workgroup through HBM / L2 (the waves share a CU and its vector L1, which is
// why the memory model lets a workgroup-scope release skip vmcnt; the storing wave still has to have SENT its stores)
__device__ __forceinline__ void wg_barrier_after_global_stores() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); }
#endif
}  // namespace plp
