// Separable fixed-point Gaussian blur of one 128 x 32 output tile (cv::GaussianBlur on CV_8U: 8.8 taps that sum
// to 256, out = (sum_j k[j] * (sum_i k[i] * src) + 32768) >> 16, BORDER_REFLECT_101), shared by the ORB 7-tap
// blur (all pyramid levels) and the 11-/5-tap blurs of the line front-end.  Integer arithmetic throughout, no
// intermediate rounding, so only the sums matter, not how they are grouped.
//
// 256 threads, three phases:
//   stage       input rows with an 8-byte left pad as aligned dwords; rows are reflected by index, the few dwords of a row
//               that straddle the left / right image border are patched byte-wise by a second small pass (border tiles only)
//   horizontal  a work item makes 4 adjacent sums for TWO rows: v_dot4_u32_u8 of aligned LDS dwords against taps that are
//               pre-shifted to the window's byte offset (wave-uniform constants: no v_alignbyte), stored as (row 2p, row 2p+1)
//               u16 pairs -- the layout the vertical pass multiplies directly
//   vertical    a thread owns 4 columns x 4 rows: v_dot2_u32_u16 of the row pairs against tap pairs ((k0,k1),(k2,k3).. for
//               an even output row, (0,k0),(k1,k2).. for an odd one): K/2+1 instructions per output instead of K multiply-adds
//               on unpacked values; the four output bytes are cut out with v_perm_b32 (the sums cannot exceed 255 << 16)
#pragma once
#include <hip/hip_runtime.h>
#include "plp_barrier.hpp"
#include <stdint.h>

namespace plp {

constexpr int kBlurTW = 128, kBlurTH = 32, kBlurPad = 8;
constexpr int kBlurRS = kBlurTH / 8;   // output rows per thread in the vertical pass (8 strips of a 256-thread workgroup)

__device__ __forceinline__ int blur_reflect101(int p, int len) {
    if (len == 1) return 0;
    // one reflection at either border covers every index a tile of an image of at least 48 rows / columns asks for: straight-line code (the general
    // loop below had made every staging load a loop of its own and kept the compiler from issuing a tile's loads together)
    p = p < 0 ? -p : p;
    p = p >= len ? 2 * (len - 1) - p : p;
    if (__builtin_expect(p < 0 || p >= len, 0))
        while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

template <int R>
struct BlurTileLds {
    static constexpr int IW = kBlurTW + 2 * kBlurPad;          // 144 bytes per input row
    static constexpr int IH = kBlurTH + 2 * R;                 // even
    static constexpr int NPR = IH / 2;                         // row pairs
    uint8_t in[IH * IW];
    uint32_t hs2[NPR * kBlurTW];                               // horizontal sums, (row 2p | row 2p+1 << 16) per column
};

typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t blur_dot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, a), __builtin_bit_cast(blur_us2, b), c, false);
}

// horizontal pass, barrier, vertical pass on a staged tile (the caller has synchronised after staging); emit(r0, c4, rows) gets every thread's 4 x 4 bytes
template <int R, class Emit>
__device__ __forceinline__ void blur_tile_compute(BlurTileLds<R>& S, const int* __restrict__ taps, Emit emit) {
    constexpr int K = 2 * R + 1, IW = BlurTileLds<R>::IW, NPR = BlurTileLds<R>::NPR;
    const int tid = threadIdx.x;
    // ---- horizontal
    constexpr int RU = (R + 3) / 4 * 4;           // window start rounded down to a dword: RU - R bytes of slack
    constexpr int OFF = RU - R;                   // byte offset of tap 0 of output 0 inside the first dword
    constexpr int ND = (OFF + 3 + K + 3) / 4;     // dwords covering the 4 windows
    // taps shifted by e = 0..3 bytes, four to a dword (wave-uniform)
    uint32_t tp[4][(K + 3 + 3) / 4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < (K + 3 + 3) / 4; ++m) {
            tp[e][m] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * m + k - e;
                if (t >= 0 && t < K) tp[e][m] |= (uint32_t)taps[t] << (8 * k);
            }
        }
    for (int i = tid; i < NPR * (kBlurTW / 4); i += 256) {
        const int p = i >> 5, c4 = (i & 31) * 4;
        uint32_t a[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(&S.in[(2 * p + rr) * IW + kBlurPad - RU + c4]);
            uint32_t d[ND];
#pragma unroll
            for (int k = 0; k < ND; ++k) d[k] = q[k];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                constexpr int dummy = 0; (void)dummy;
                const int s = OFF + o, q0 = s >> 2, e = s & 3;
                uint32_t acc = 0;
#pragma unroll
                for (int m = 0; m < (e + K + 3) / 4; ++m) acc = __builtin_amdgcn_udot4(d[q0 + m], tp[e][m], acc, false);
                a[rr][o] = acc;
            }
        }
        uint4 v;
        v.x = a[0][0] | (a[1][0] << 16); v.y = a[0][1] | (a[1][1] << 16); v.z = a[0][2] | (a[1][2] << 16); v.w = a[0][3] | (a[1][3] << 16);
        *reinterpret_cast<uint4*>(&S.hs2[p * kBlurTW + c4]) = v;
    }
    wg_barrier();
    // ---- vertical: 4 columns x 4 rows per thread
    const int cg = tid & 31, strip = tid >> 5;
    const int c4 = cg * 4, r0 = strip * kBlurRS;
    constexpr int NP = K / 2 + 1;                 // tap pairs per output row
    uint32_t te[NP], to[NP];                      // even rows: (k0,k1),(k2,k3),..,(k[K-1],0); odd rows: (0,k0),(k1,k2),..
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        te[j] = (uint32_t)taps[2 * j] | (2 * j + 1 < K ? (uint32_t)taps[2 * j + 1] << 16 : 0u);
        to[j] = (j > 0 ? (uint32_t)taps[2 * j - 1] : 0u) | ((uint32_t)taps[2 * j] << 16);
    }
    uint4 wp[NP + 1];
#pragma unroll
    for (int j = 0; j <= NP; ++j) wp[j] = *reinterpret_cast<const uint4*>(&S.hs2[(r0 / 2 + j) * kBlurTW + c4]);
    uint32_t rows[kBlurRS];
#pragma unroll
    for (int rr = 0; rr < kBlurRS; ++rr) {
        uint32_t a0 = 32768u, a1 = 32768u, a2 = 32768u, a3 = 32768u;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t t = (rr & 1) ? to[j] : te[j];
            const uint4 v = wp[rr / 2 + j];
            a0 = blur_dot2(v.x, t, a0); a1 = blur_dot2(v.y, t, a1); a2 = blur_dot2(v.z, t, a2); a3 = blur_dot2(v.w, t, a3);
        }
        // byte 2 of each sum (sum <= 255 * 65536 + 32768): v_perm_b32 selects bytes of {src0, src1}, src1 = bytes 0-3
        rows[rr] = __builtin_amdgcn_perm(a1, a0, 0x0c0c0602u) | __builtin_amdgcn_perm(a3, a2, 0x06020c0cu);
    }
    emit(r0, c4, rows);
}

// ---- the tile in three pieces: issue every load of the tile into registers (blur_prefetch), store them to LDS (blur_stage_prefetched), compute
// (blur_tile_compute).  Written for persistent workgroups that prefetch tile n + 1 during tile n (measured slower, the kernel is
// in profiles/r05_removed_experiment_knobs.patch); what it showed and what stays is that ALL of a tile's loads must be in flight before the first wait.
struct BlurJob { const uint8_t* src; uint8_t* dst; int src_pitch, dst_pitch, w, h, tx0, ty0; };   // wave-uniform

template <int R> struct BlurPrefetch {
    static constexpr int IW = BlurTileLds<R>::IW, IH = BlurTileLds<R>::IH, DW = IW / 4;
    static constexpr int NPRE = (IH * DW + 255) / 256, NPATCH = (IH * 8 + 255) / 256;
    uint32_t pre[NPRE], patch[NPATCH];
};
// (row, dword) of a thread's q-th staging slot: slot i = tid + 256 q of the IH x DW dwords, by additions from the thread's first (256 = QR x DW + QD) instead of a
// division per slot
template <int R> struct BlurSlot {
    static constexpr int DW = BlurTileLds<R>::IW / 4, QR = 256 / DW, QD = 256 % DW;
    static_assert((BlurPrefetch<R>::NPRE - 1) * QD + DW - 1 < 2 * DW, "a thread's dword index wraps at most once per step");
    int r0, d0;
    __device__ __forceinline__ BlurSlot() { const int tid = threadIdx.x; r0 = tid / DW; d0 = tid - r0 * DW; }
    __device__ __forceinline__ void at(int q, int& r, int& d) const { d = d0 + QD * q; r = r0 + QR * q; if (d >= DW) { d -= DW; ++r; } }
};
// issue the loads of a tile (whole dwords inside the image; the dwords that straddle the image border assembled byte-wise, border tiles only).  Rows are
// reflected only where the tile touches the top or bottom border (wave-uniform); addresses are 32-bit offsets from the plane's base.
template <int R> __device__ __forceinline__ void blur_prefetch(BlurPrefetch<R>& F, const BlurJob& J) {
    constexpr int IW = BlurTileLds<R>::IW, IH = BlurTileLds<R>::IH, DW = IW / 4;
    const int tid = threadIdx.x;
    const BlurSlot<R> slot;
    const bool rows_inside = J.ty0 >= R && J.ty0 + kBlurTH + R <= J.h;
#pragma unroll
    for (int q = 0; q < BlurPrefetch<R>::NPRE; ++q) {
        int r, d;
        slot.at(q, r, d);
        const int x = J.tx0 - kBlurPad + 4 * d;
        int y = J.ty0 + r - R;
        if (!rows_inside) y = blur_reflect101(y, J.h);
        F.pre[q] = 0u;
        if (tid + 256 * q < IH * DW && x >= 0 && x + 3 < J.w) F.pre[q] = *reinterpret_cast<const uint32_t*>(J.src + (uint32_t)(__mul24(y, J.src_pitch) + x));
    }
    const int dr0 = (J.w - 3 - (J.tx0 - kBlurPad) + 3) >> 2;
    if (J.tx0 <= 0 || dr0 < DW) {
        uint8_t b[BlurPrefetch<R>::NPATCH][4];                   // all byte loads first, the dwords assembled afterwards: one memory round trip
#pragma unroll
        for (int q = 0; q < BlurPrefetch<R>::NPATCH; ++q) {
            const int i = tid + 256 * q, r = i >> 3, slot = i & 7;
            const int d = slot < 4 ? slot : dr0 + slot - 4, x = J.tx0 - kBlurPad + 4 * d;
            // (unconditional: the reflected indices are valid for every slot, and straight-line loads are what lets all eight be in flight together;
            //  blur_stage_prefetched only uses the slots that are due)
            const uint8_t* row = J.src + (size_t)blur_reflect101(J.ty0 + r - R, J.h) * J.src_pitch;
#pragma unroll
            for (int k = 0; k < 4; ++k) b[q][k] = row[blur_reflect101(x + k, J.w)];
        }
#pragma unroll
        for (int q = 0; q < BlurPrefetch<R>::NPATCH; ++q)
            F.patch[q] = (uint32_t)b[q][0] | ((uint32_t)b[q][1] << 8) | ((uint32_t)b[q][2] << 16) | ((uint32_t)b[q][3] << 24);
    }
}
// the prefetched dwords go to the LDS tile (same slots as blur_tile_core's staging)
template <int R> __device__ __forceinline__ void blur_stage_prefetched(BlurTileLds<R>& S, const BlurPrefetch<R>& F, const BlurJob& J) {
    constexpr int IW = BlurTileLds<R>::IW, IH = BlurTileLds<R>::IH, DW = IW / 4;
    const int tid = threadIdx.x;
    const BlurSlot<R> slot;
#pragma unroll
    for (int q = 0; q < BlurPrefetch<R>::NPRE; ++q) {
        int r, d;
        slot.at(q, r, d);
        const int x = J.tx0 - kBlurPad + 4 * d;
        if (tid + 256 * q < IH * DW && x >= 0 && x + 3 < J.w) *reinterpret_cast<uint32_t*>(&S.in[__mul24(r, IW) + 4 * d]) = F.pre[q];
    }
    const int dr0 = (J.w - 3 - (J.tx0 - kBlurPad) + 3) >> 2;
    if (J.tx0 <= 0 || dr0 < DW) {
#pragma unroll
        for (int q = 0; q < BlurPrefetch<R>::NPATCH; ++q) {
            const int i = tid + 256 * q, r = i >> 3, slot = i & 7;
            const int d = slot < 4 ? slot : dr0 + slot - 4, x = J.tx0 - kBlurPad + 4 * d;
            const bool todo = i < IH * 8 && (slot < 4 ? x < 0 : (d >= 0 && d < DW && x <= J.w + R + 3));
            if (todo) *reinterpret_cast<uint32_t*>(&S.in[r * IW + 4 * d]) = F.patch[q];
        }
    }
}


// Staging of a tile whose 144-byte rows lie inside the image in x (tx0 >= 8 and tx0 + 136 <= w: three tile columns in five of a 640-wide level): every dword
// is a whole aligned dword of the source, so there are no range checks and no byte-wise patches, the (row, dword) of a thread's q-th load follows from its first by
// additions (BlurSlot) instead of a division per load, rows are reflected only where the tile touches the top or bottom border (wave-uniform), and the
// address is a 32-bit offset from the plane's base.  Half of the blur kernels' vector instructions were this bookkeeping (k_blur7: ~230 of 460 per wave); the
// bytes that reach LDS are the same.
template <int R>
__device__ __forceinline__ void blur_stage_inside_x(BlurTileLds<R>& S, const uint8_t* __restrict__ src, int src_pitch, int h, int tx0, int ty0) {
    constexpr int IW = BlurTileLds<R>::IW, IH = BlurTileLds<R>::IH, DW = IW / 4, N = IH * DW, NPRE = BlurPrefetch<R>::NPRE;
    const int tid = threadIdx.x;
    const BlurSlot<R> slot;
    const bool rows_inside = ty0 >= R && ty0 + kBlurTH + R <= h;
    const int xb = tx0 - kBlurPad;
    uint32_t v[NPRE]; int lo[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        int r, d;
        slot.at(q, r, d);
        int y = ty0 + r - R;
        if (!rows_inside) y = blur_reflect101(y, h);
        lo[q] = __mul24(r, IW) + 4 * d;
        const bool live = (q + 1) * 256 <= N || tid + 256 * q < N;
        v[q] = live ? *reinterpret_cast<const uint32_t*>(src + (uint32_t)(__mul24(y, src_pitch) + xb + 4 * d)) : 0u;
    }
#pragma unroll
    for (int q = 0; q < NPRE; ++q)
        if ((q + 1) * 256 <= N || tid + 256 * q < N) *reinterpret_cast<uint32_t*>(&S.in[lo[q]]) = v[q];
}

// Core: stage, horizontal pass, vertical pass; every thread ends with the blurred bytes of its 4 columns x 4 rows (one dword per
// row, tile-local rows r0..r0+3, columns c4..c4+3) and hands them to emit(r0, c4, rows).  The tile origin may lie outside the image
// (tx0 a multiple of 4, possibly negative): the blur is then evaluated on the REFLECT_101 extension of the source, which for a
// symmetric kernel equals the REFLECT_101 extension of the blurred image -- what a following 3x3 filter wants at the border.
// src: plane base (4-byte aligned, pitch % 4 == 0)
template <int R, class Emit>
__device__ __forceinline__ void blur_tile_core(BlurTileLds<R>& S, const uint8_t* __restrict__ src, int src_pitch, int w, int h, int tx0, int ty0,
                                               const int* __restrict__ taps, Emit emit) {
    static_assert(BlurTileLds<R>::IH % 2 == 0 && kBlurRS == 4, "row pairs");
    // ---- stage input rows [ty0-R, ty0+TH+R) x columns [tx0-PAD, tx0+TW+PAD): every thread ISSUES all its loads (6 whole dwords inside the image, and
    // on border tiles 2 dwords assembled byte-wise around x = 0 / x = w), then stores them to LDS.  Until round 4 this was a loop of load -> wait ->
    // LDS store: six serialized memory round trips per tile, which is what every kernel built on this tile was waiting for (profiles/r04_tile_pipelining.md).
    if (tx0 >= kBlurPad && tx0 + kBlurTW + kBlurPad <= w) blur_stage_inside_x<R>(S, src, src_pitch, h, tx0, ty0);   // (wave-uniform)
    else {
        const BlurJob J{src, nullptr, src_pitch, 0, w, h, tx0, ty0};
        BlurPrefetch<R> F;
        blur_prefetch<R>(F, J);
        blur_stage_prefetched<R>(S, F, J);
    }
    wg_barrier();
    blur_tile_compute<R>(S, taps, emit);
}

// The plain blur: the tile's bytes go to the output plane (pitch % 4 == 0, rows padded to a multiple of 4).
template <int R>
__device__ __forceinline__ void blur_tile(BlurTileLds<R>& S, const uint8_t* __restrict__ src, int src_pitch, uint8_t* __restrict__ dst,
                                          int dst_pitch, int w, int h, int tx0, int ty0, const int* __restrict__ taps) {
    blur_tile_core<R>(S, src, src_pitch, w, h, tx0, ty0, taps, [&](int r0, int c4, const uint32_t (&rows)[kBlurRS]) {
        const int x = tx0 + c4;
        if (x >= w) return;
#pragma unroll
        for (int rr = 0; rr < kBlurRS; ++rr) {
            const int y = ty0 + r0 + rr;
            if (y < h) *reinterpret_cast<uint32_t*>(dst + (size_t)y * dst_pitch + x) = rows[rr];
        }
    });
}


}  // namespace plp
