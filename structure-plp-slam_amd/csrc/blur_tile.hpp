// Separable fixed-point Gaussian blur of one 128 x 64 output tile (cv::GaussianBlur on CV_8U: 8.8 taps that sum
// to 256, out = (sum_j k[j] * (sum_i k[i] * src) + 32768) >> 16, BORDER_REFLECT_101), shared by the ORB 7-tap
// blur (all pyramid levels) and the 11-/5-tap blurs of the line front-end.
//
// 256 threads.  LDS: input rows with an 8-byte aligned left pad (so the global reads are aligned dwords),
// horizontal sums as u16.  Horizontal pass: a thread makes 4 adjacent sums from 4+2R byte reads.  Vertical
// pass: a thread owns 4 adjacent columns x 8 rows and slides a (2R+1)-row register window down the strip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plp {

constexpr int kBlurTW = 128, kBlurTH = 64, kBlurPad = 8;

__device__ __forceinline__ int blur_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

template <int R>
struct BlurTileLds {
    static constexpr int IW = kBlurTW + 2 * kBlurPad;          // 144 bytes per input row
    static constexpr int IH = kBlurTH + 2 * R;
    uint8_t in[IH * IW];
    uint16_t hs[IH * kBlurTW];
};

// src: plane base (4-byte aligned, pitch % 4 == 0), dst: output plane (pitch % 4 == 0, rows padded to a multiple of 4)
template <int R>
__device__ __forceinline__ void blur_tile(BlurTileLds<R>& S, const uint8_t* __restrict__ src, int src_pitch, uint8_t* __restrict__ dst,
                                          int dst_pitch, int w, int h, int tx0, int ty0, const int* __restrict__ taps) {
    constexpr int K = 2 * R + 1, IW = BlurTileLds<R>::IW, IH = BlurTileLds<R>::IH, DW = IW / 4;
    const int tid = threadIdx.x;
    // ---- stage input rows [ty0-R, ty0+TH+R) x columns [tx0-PAD, tx0+TW+PAD) as dwords
    for (int i = tid; i < IH * DW; i += 256) {
        const int r = i / DW, d = i - r * DW;
        const int y = blur_reflect101(ty0 + r - R, h);
        const int x = tx0 - kBlurPad + 4 * d;
        const uint8_t* row = src + (size_t)y * src_pitch;
        uint32_t v;
        if (x >= 0 && x + 3 < w) v = *reinterpret_cast<const uint32_t*>(row + x);
        else if (x + 3 < -R || x > w - 1 + R) v = 0;                  // never read by a valid output
        else
            v = (uint32_t)row[blur_reflect101(x, w)] | ((uint32_t)row[blur_reflect101(x + 1, w)] << 8) |
                ((uint32_t)row[blur_reflect101(x + 2, w)] << 16) | ((uint32_t)row[blur_reflect101(x + 3, w)] << 24);
        *reinterpret_cast<uint32_t*>(&S.in[r * IW + 4 * d]) = v;
    }
    __syncthreads();
    // ---- horizontal: 4 adjacent sums per work item
    for (int i = tid; i < IH * (kBlurTW / 4); i += 256) {
        const int r = i / (kBlurTW / 4), c4 = (i - r * (kBlurTW / 4)) * 4;
        const uint8_t* p = &S.in[r * IW + kBlurPad - R + c4];
        uint32_t px[K + 3];
#pragma unroll
        for (int k = 0; k < K + 3; ++k) px[k] = p[k];
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t t = (uint32_t)taps[k];
            a0 += t * px[k]; a1 += t * px[k + 1]; a2 += t * px[k + 2]; a3 += t * px[k + 3];
        }
        uint32_t* o = reinterpret_cast<uint32_t*>(&S.hs[r * kBlurTW + c4]);
        o[0] = a0 | (a1 << 16);
        o[1] = a2 | (a3 << 16);
    }
    __syncthreads();
    // ---- vertical: 4 columns x 8 rows per thread, sliding window of K rows
    const int cg = tid & 31, strip = tid >> 5;
    const int c4 = cg * 4, r0 = strip * 8;
    const int x = tx0 + c4;
    if (x >= w) return;
    uint32_t win[K][2];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(&S.hs[(r0 + k) * kBlurTW + c4]);
        win[k][0] = q[0]; win[k][1] = q[1];
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(&S.hs[(r0 + rr + K - 1) * kBlurTW + c4]);
        win[K - 1][0] = q[0]; win[K - 1][1] = q[1];
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t t = (uint32_t)taps[k];
            a0 += t * (win[k][0] & 0xffffu); a1 += t * (win[k][0] >> 16);
            a2 += t * (win[k][1] & 0xffffu); a3 += t * (win[k][1] >> 16);
        }
        const int y = ty0 + r0 + rr;
        if (y < h) {
            const uint32_t packed = min((a0 + 32768u) >> 16, 255u) | (min((a1 + 32768u) >> 16, 255u) << 8) |
                                    (min((a2 + 32768u) >> 16, 255u) << 16) | (min((a3 + 32768u) >> 16, 255u) << 24);
            *reinterpret_cast<uint32_t*>(dst + (size_t)y * dst_pitch + x) = packed;
        }
#pragma unroll
        for (int k = 0; k < K - 1; ++k) { win[k][0] = win[k + 1][0]; win[k][1] = win[k + 1][1]; }
    }
}

}  // namespace plp
