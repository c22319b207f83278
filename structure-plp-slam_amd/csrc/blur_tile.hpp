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

constexpr int kBlurTW = 128, kBlurTH = 32, kBlurPad = 8;
constexpr int kBlurRS = kBlurTH / 8;   // output rows per thread in the vertical pass (8 strips of a 256-thread workgroup)

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
    // ---- horizontal: 4 adjacent sums per work item.  The taps are packed four to a dword and applied with
    // v_dot4_u32_u8 on byte windows cut from aligned LDS dwords with v_alignbyte (2-3 dot products per output
    // instead of 2R+1 multiply-adds on single bytes).
    constexpr int RU = (R + 3) / 4 * 4;           // window start rounded down to a dword: RU - R bytes of slack
    constexpr int OFF = RU - R;                   // byte offset of tap 0 of output 0 inside the first dword
    constexpr int ND = (OFF + K + 3 + 3) / 4;     // dwords covering the 4 windows
    constexpr int NT = (K + 3) / 4;               // packed tap dwords
    uint32_t tp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        tp[j] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * j + k < K) tp[j] |= (uint32_t)taps[4 * j + k] << (8 * k);
    }
    for (int i = tid; i < IH * (kBlurTW / 4); i += 256) {
        const int r = i / (kBlurTW / 4), c4 = (i - r * (kBlurTW / 4)) * 4;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&S.in[r * IW + kBlurPad - RU + c4]);
        uint32_t d[ND + 1];
#pragma unroll
        for (int k = 0; k < ND; ++k) d[k] = p[k];
        d[ND] = 0;
        uint32_t a[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int sh = OFF + o;               // first byte of this output's window
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int q = (sh + 4 * j) >> 2, e = (sh + 4 * j) & 3;
                const uint32_t wnd = e == 0 ? d[q] : __builtin_amdgcn_alignbyte(d[q + 1], d[q], (uint32_t)e);
                acc = __builtin_amdgcn_udot4(wnd, tp[j], acc, false);
            }
            a[o] = acc;
        }
        uint32_t* o2 = reinterpret_cast<uint32_t*>(&S.hs[r * kBlurTW + c4]);
        o2[0] = a[0] | (a[1] << 16);
        o2[1] = a[2] | (a[3] << 16);
    }
    __syncthreads();
    // ---- vertical: 4 columns x 8 rows per thread; the (2R+1)-row window lives unpacked in registers (the loop is
    // fully unrolled, so sliding it is register renaming)
    const int cg = tid & 31, strip = tid >> 5;
    const int c4 = cg * 4, r0 = strip * kBlurRS;
    const int x = tx0 + c4;
    if (x >= w) return;
    uint32_t win[K + kBlurRS - 1][4];
#pragma unroll
    for (int k = 0; k < K + kBlurRS - 1; ++k) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(&S.hs[(r0 + k) * kBlurTW + c4]);
        const uint32_t q0 = q[0], q1 = q[1];
        win[k][0] = q0 & 0xffffu; win[k][1] = q0 >> 16; win[k][2] = q1 & 0xffffu; win[k][3] = q1 >> 16;
    }
#pragma unroll
    for (int rr = 0; rr < kBlurRS; ++rr) {
        uint32_t a0 = 32768u, a1 = 32768u, a2 = 32768u, a3 = 32768u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t t = (uint32_t)taps[k];
            a0 += t * win[rr + k][0]; a1 += t * win[rr + k][1]; a2 += t * win[rr + k][2]; a3 += t * win[rr + k][3];
        }
        const int y = ty0 + r0 + rr;
        if (y < h) {
            const uint32_t packed = min(a0 >> 16, 255u) | (min(a1 >> 16, 255u) << 8) | (min(a2 >> 16, 255u) << 16) | (min(a3 >> 16, 255u) << 24);
            *reinterpret_cast<uint32_t*>(dst + (size_t)y * dst_pitch + x) = packed;
        }
    }
}

}  // namespace plp
