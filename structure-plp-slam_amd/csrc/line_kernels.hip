// Hand-written HIP kernels (gfx950, wave64) of the batched line front-end: LSD detection + LBD description.
// Restates LineFeatureTracker::extract_LSD_LBD (reference src/PLPSLAM/feature/line_extractor.cc:88-160),
// the LSDDetectorC wrapper (line_descriptor/LSDDetector_custom.cpp:216-320), OpenCV's LineSegmentDetector
// (lsd.cpp, LSD_REFINE_STD; third-party) and BinaryDescriptor::computeLBD
// (line_descriptor/binary_descriptor_custom.cpp:1018-1364).
//
//   k_blur_plane<R>    cv::GaussianBlur u8 fixed point, (2R+1) taps (R=5: LSD sigma 1.2, R=2: LBD sigma 1)
//   k_resize_exact     cv::resize(x0.5, INTER_LINEAR_EXACT)
//   k_lsd_gradient     ll_angle: 2x2 gradient, magnitude (f64), level-line angle, max magnitude
//   k_lsd_order        counting sort of the seeds: bin descending, row-major inside a bin
//   k_lsd_grow         region_grow / region2rect / refine, ONE wave per frame (the algorithm is a
//                      sequential scan over seeds; frames run in parallel)
//   k_keylines         KeyLine assembly + min_length filter
//   k_blur_sobel       cv::GaussianBlur(5x5, sigma 1) + cv::Sobel 3x3 -> s16 (dx, dy), the blurred image stays in LDS
//   k_blur_half        11-tap blur + x0.5 resize in one pass (even frame sizes)
//   k_lbd              63-row band descriptor, one wave per line (lane = row; each row is a strictly
//                      sequential f32 sum, as in the reference), normalisation, 32-byte binarisation
//   k_line_finalize    lineLength >= 60 filter, 2-D line function (f64), ordered compaction
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "blur_tile.hpp"
#include "plp_barrier.hpp"
#include "line_device.hpp"
#include "plp_common.hpp"
#include "sincos_ziv.hpp"
#include "xcd_map.hpp"

#ifndef PLP_CHAIN_BATCH         // addends per LDS round trip in the rectangle fit's sequential sums (rect_from_ring); 0 / 4 / 8 / 16 measured: profiles/r03_lsd_grow.md
#define PLP_CHAIN_BATCH 8
#endif

namespace plp {

// the two divisions of a tile kernel's workgroup index by launch constants, as multipliers (xcd_map.hpp plp_div_magic)
struct TileDiv { uint32_t gx_magic, tiles_x_magic; };
static inline TileDiv tile_div(int tiles, int tiles_x, int B) { return TileDiv{plp_div_magic((uint32_t)tiles, (uint64_t)tiles * B), plp_div_magic((uint32_t)tiles_x, (uint64_t)tiles)}; }

__device__ __forceinline__ int reflect101_l(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ float fast_atan2_deg_l(float y, float x) {   // cv::fastAtan2, f32, no FMA
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        const float c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        const float c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        const float c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// ------------------------------------------------------------------------------------------ blur
// (2R+1)-tap fixed-point Gaussian of one plane per frame (blur_tile.hpp); 128 x 64 output tile per workgroup.
template <int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur_plane(const uint8_t* __restrict__ src, size_t src_fs, int src_pitch,
                                                    uint8_t* __restrict__ dst, size_t dst_fs, int dst_pitch, int w, int h, BlurTapsN taps, TileDiv td) {
    __shared__ BlurTileLds<R> S;
    const int tiles_x = (w + kBlurTW - 1) / kBlurTW;
    unsigned t, f;
    xcd_frame_major(t, f, td.gx_magic);
    const int trow = (int)plp_div(t, (unsigned)tiles_x, td.tiles_x_magic);
    blur_tile<R>(S, src + (size_t)f * src_fs, src_pitch, dst + (size_t)f * dst_fs, dst_pitch, w, h,
                 ((int)t - trow * tiles_x) * kBlurTW, trow * kBlurTH, taps.k);
}

// ------------------------------------------------------------------------------------------ x0.5 INTER_LINEAR_EXACT
__global__ __launch_bounds__(256) void k_resize_exact(const uint8_t* __restrict__ src, size_t src_fs, int src_pitch,
                                                      uint8_t* __restrict__ dst, size_t dst_fs, int dst_pitch, int dw, int dh,
                                                      ResizeExactTab t) {
    const int dy = blockIdx.y * 4 + threadIdx.y, dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= dh || dx0 >= dw) return;
    const uint8_t* s = src + (size_t)blockIdx.z * src_fs;
    const int yo = t.yo[dy], yc = t.yc[dy];
    const uint8_t* S0 = s + (size_t)yo * src_pitch;
    const uint8_t* S1 = yc < 0 ? S0 : S0 + src_pitch;
    const uint32_t b1 = yc < 0 ? 0u : (uint32_t)yc, b0 = 256u - b1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = min(dx0 + i, dw - 1);
        const int xo = t.xo[dx], xc = t.xc[dx];
        uint32_t h0, h1;
        if (xc < 0) { h0 = (uint32_t)S0[xo] * 256u; h1 = (uint32_t)S1[xo] * 256u; }
        else {
            h0 = (uint32_t)S0[xo] * (uint32_t)(256 - xc) + (uint32_t)S0[xo + 1] * (uint32_t)xc;
            h1 = (uint32_t)S1[xo] * (uint32_t)(256 - xc) + (uint32_t)S1[xo + 1] * (uint32_t)xc;
        }
        const uint32_t v = (b0 * h0 + b1 * h1 + 32768u) >> 16;
        packed |= min(v, 255u) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)blockIdx.z * dst_fs + (size_t)dy * dst_pitch + dx0) = packed;
}

// ------------------------------------------------------------------------------------------ blur11 + x0.5 in one pass
// The plain case of the LSD front: even frame size, so every entry of the INTER_LINEAR_EXACT tables is (2d, weight 128) and a scaled
// pixel is (a + b + c + d + 2) >> 2 of a 2x2 block of the blurred image -- exactly what k_resize_exact computes from those tables
// ((128 * (128 a + 128 b) + 128 * (128 c + 128 d) + 32768) >> 16).  A thread of the blur's vertical pass owns 4 columns x 4 rows:
// its own 2x2 blocks, so the blurred plane never goes to HBM.  The host checks the tables (line_context.hip) and falls back to
// k_blur_plane<5> + k_resize_exact otherwise.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur_half(const uint8_t* __restrict__ src, size_t src_fs, int src_pitch,
                                                                                           uint8_t* __restrict__ dst, size_t dst_fs, int dst_pitch, int w, int h, BlurTapsN taps, TileDiv td) {
    __shared__ BlurTileLds<5> S;
    const int tiles_x = (w + kBlurTW - 1) / kBlurTW;
    unsigned t, f;
    xcd_frame_major(t, f, td.gx_magic);
    const int trow = (int)plp_div(t, (unsigned)tiles_x, td.tiles_x_magic);
    const int tx0 = ((int)t - trow * tiles_x) * kBlurTW, ty0 = trow * kBlurTH;
    uint8_t* d = dst + (size_t)f * dst_fs;
    blur_tile_core<5>(S, src + (size_t)f * src_fs, src_pitch, w, h, tx0, ty0, taps.k, [&](int r0, int c4, const uint32_t (&rows)[kBlurRS]) {
        const int x = tx0 + c4;
        if (x >= w) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = ty0 + r0 + 2 * j;
            if (y >= h) continue;
            const uint32_t lo = __builtin_amdgcn_udot4(rows[2 * j], 0x00000101u, __builtin_amdgcn_udot4(rows[2 * j + 1], 0x00000101u, 2u, false), false) >> 2;
            const uint32_t hi = __builtin_amdgcn_udot4(rows[2 * j], 0x01010000u, __builtin_amdgcn_udot4(rows[2 * j + 1], 0x01010000u, 2u, false), false) >> 2;
            *reinterpret_cast<uint16_t*>(d + (size_t)(y >> 1) * dst_pitch + (x >> 1)) = (uint16_t)(lo | (hi << 8));
        }
    });
}

// ------------------------------------------------------------------------------------------ ll_angle
// One thread per scaled pixel.  "Defined" (magnitude > rho, lsd.cpp ll_angle) is an integer test: the magnitude
// sqrt(g2 / 4.0) is monotone in the integer g2 = gx^2 + gy^2, so the host finds the smallest defined g2 once
// (LsdParams::g2_def_min, same f64 operations).  Only defined pixels get a record (nothing ever reads the others);
// the g2 plane holds g2 for defined pixels and 0 otherwise -- the seed sort works from it.
__global__ __launch_bounds__(256) void k_lsd_gradient(LinePlanes P, LsdParams lp) {
    // About a third of a frame's pixels are "defined" and need the angle, its cosine and sine (~100 instructions); one pixel per lane, every
    // wave paid them for its few defined lanes.  The workgroup's defined pixels are compacted through LDS first (position in the block + the two
    // 10-bit gradient components: one dword), then dense lanes do the arithmetic: about a third of the heavy instructions.
    __shared__ uint32_t s_list[256];
    __shared__ uint32_t s_max[4], s_cnt[4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int idx = blockIdx.x * 256 + tid;
    const int n = P.sw * P.sh;
    uint32_t g2_def = 0, g2_any = 0, packed = 0;
    if (idx < n) {
        const int y = P.sw_magic ? (int)__umulhi((uint32_t)idx, P.sw_magic) : idx / P.sw, x = idx - y * P.sw;   // idx / sw (line_context.hip: when the product form is exact)
        if (x < P.sw - 1 && y < P.sh - 1) {
            const uint8_t* s = P.scaled + ((size_t)b * P.sh + y) * P.spitch + x;
            const int DA = (int)s[P.spitch + 1] - (int)s[0];
            const int BC = (int)s[1] - (int)s[P.spitch];
            const int gx = DA + BC, gy = DA - BC;                  // each in [-510, 510]
            const uint32_t g2 = (uint32_t)(gx * gx + gy * gy);
            g2_any = g2;
            if (g2 >= lp.g2_def_min) {
                g2_def = g2;
                packed = (uint32_t)tid | ((uint32_t)(gx + 512) << 8) | ((uint32_t)(gy + 512) << 18);
            }
        }
        P.g2[(size_t)b * n + idx] = lp.seed_exact ? g2_any : g2_def;   // the exact seed order sorts the undefined pixels too (they have bins)
    }
    // one 64-bit word per wave: pixels that can never seed or join a region (angle NOTDEF)
    const unsigned long long undef = __ballot(g2_def == 0);
    if (lane == 0 && blockIdx.x * 256 + tid < ((n + 63) / 64) * 64) P.undef[(size_t)b * ((n + 63) / 64) + (blockIdx.x * 256 + tid) / 64] = undef;
    const unsigned long long defm = ~undef;
    const int rank = __popcll(defm & ((1ull << lane) - 1ull));
    // max magnitude over the defined pixels = max of g2 (the magnitude is monotone in it).  One plain store per workgroup; k_lsd_order
    // reduces the per-workgroup values (2.4 M same-line atomics per launch had made this kernel wait 86 % of its time).
    uint32_t mx = g2_def;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 0) { s_max[wv] = mx; s_cnt[wv] = (uint32_t)__popcll(defm); }
    wg_barrier();
    const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    const int base = (wv > 0 ? (int)c0 : 0) + (wv > 1 ? (int)c1 : 0) + (wv > 2 ? (int)c2 : 0);
    if (g2_def) s_list[base + rank] = packed;
    if (tid == 0) P.blockmax[(size_t)b * gridDim.x + blockIdx.x] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    wg_barrier();
    const int total = (int)(c0 + c1 + c2 + c3);
    if (tid < total) {
        const uint32_t e = s_list[tid];
        const int gx = (int)((e >> 8) & 1023u) - 512, gy = (int)(e >> 18) - 512;
        LsdPix px;
        px.deg = fast_atan2_deg_l((float)gx, (float)-gy);
        px.g2 = (uint32_t)(gx * gx + gy * gy);
        const float fa = (float)((double)px.deg * (3.14159265358979323846 / 180));
        if (!sincos_ziv(fa, &px.cs.x, &px.cs.y)) px.cs = make_float2((float)cos((double)fa), (float)sin((double)fa));   // ~1 pixel in a million
        P.pix[(size_t)b * n + blockIdx.x * 256 + (int)(e & 255u)] = px;
    }
}

// ------------------------------------------------------------------------------------------ seed ordering
// The reference's pseudo-ordering: all pixels by gradient bin, descending (definition D1: row-major inside a bin).  Region
// growing starts from defined pixels only, so only those are ordered -- about a third of a frame; n_order[b] of them.
// One workgroup per frame; wave q owns the q-th quarter of the row-major pixel sequence.
//   1  frame maximum from the gradient kernel's per-workgroup values -> bin_coef
//   2  per 64 pixels: bin = (int)(magnitude * bin_coef) for the defined ones, per-wave histogram (LDS atomics), the defined
//      pixels compacted as (pixel | bin << kLsdSeedPixBits) into the frame's region-list scratch (free until region growing)
//   3  exclusive scan over (bin descending, wave ascending): 4 bins per thread, shuffles, one LDS hop across waves
//   4  per 64 compacted entries: rank among equal bins by 10 ballots (stable), scatter
__global__ __launch_bounds__(256) void k_lsd_order(LinePlanes P, LsdParams lp, int n_grad_blocks) {
    __shared__ uint32_t cnt[4][1024];
    __shared__ uint32_t s_red[4], s_wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int n = P.sw * P.sh, nv = (P.sw - 1) * (P.sh - 1);
    uint32_t mx = 0;
    for (int i = tid; i < n_grad_blocks; i += 256) mx = max(mx, P.blockmax[(size_t)b * n_grad_blocks + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 0) s_red[q] = mx;
    for (int i = tid; i < 4096; i += 256) (&cnt[0][0])[i] = 0;
    wg_barrier();
    mx = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    const double max_grad = sqrt((double)mx / 4.0);
    const double bin_coef = (max_grad > 0) ? (double)(lp.n_bins - 1) / max_grad : 0;
    const uint32_t* g2 = P.g2 + (size_t)b * n;
    uint32_t* order = P.order + (size_t)b * nv;
    const int ngroups = (n + 63) / 64, gper = (ngroups + 3) / 4, g0 = q * gper, g1 = min(ngroups, g0 + gper);
    uint32_t* comp = P.reg + (size_t)b * P.reg_frame_stride + (size_t)g0 * 64;
    int ncomp = 0;
    for (int gb = g0; gb < g1; gb += 4) {   // four groups per trip: their loads are in flight together (a wave walks ~300 groups, one frame per workgroup)
        uint32_t v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pix = (gb + u) * 64 + lane;
            v4[u] = (gb + u < g1 && pix < n) ? g2[pix] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t v = v4[u];
            const unsigned long long defm = __ballot(v != 0);
            if (!defm) continue;
            if (v) {
                const uint32_t bin = (uint32_t)(int)(sqrt((double)v / 4.0) * bin_coef);
                atomicAdd(&cnt[q][bin], 1u);
                comp[ncomp + __popcll(defm & ((1ull << lane) - 1ull))] = (uint32_t)((gb + u) * 64 + lane) | (bin << kLsdSeedPixBits);
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
    for (int ib = 0; ib < ncomp; ib += 256) {   // four groups of entries loaded together, ranked one after the other
    uint32_t e4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e4[u] = ib + 64 * u + lane < ncomp ? comp[ib + 64 * u + lane] : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i0 = ib + 64 * u;
        if (i0 >= ncomp) break;
        const bool valid = i0 + lane < ncomp;
        const uint32_t e = e4[u];
        const unsigned v = e >> kLsdSeedPixBits;
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
        if (valid && (peers & ((1ull << lane) - 1ull)) == 0) cnt[q][v] += (uint32_t)__popcll(peers);   // one leader per bin
        __builtin_amdgcn_wave_barrier();
    }
    }
}

// ------------------------------------------------------------------------------------------ region growing
__device__ __forceinline__ double angle_diff_signed(double a, double b) {
    const double PI = 3.14159265358979323846, TWO_PI = 2 * 3.14159265358979323846;
    double diff = a - b;
    while (diff <= -PI) diff += TWO_PI;
    while (diff > PI) diff -= TWO_PI;
    return diff;
}
__device__ __forceinline__ bool aligned_to(double a, double theta, double prec) {
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > (3 * 3.14159265358979323846) / 2) {
        n_theta -= 2 * 3.14159265358979323846;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}
__device__ __forceinline__ double shfl_d(double v, int src) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = __shfl((unsigned)bits, src), hi = __shfl((unsigned)((unsigned long long)bits >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// broadcasts from a wave-uniform lane index: v_readlane (no LDS round trip, unlike ds_bpermute-based __shfl)
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(bcast_i(__float_as_int(v), src)); }
__device__ __forceinline__ double bcast_d(double v, int src) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)bcast_i((int)(unsigned)bits, src), hi = (unsigned)bcast_i((int)(unsigned)((unsigned long long)bits >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

struct GrowCtx {
    const LsdPix* pix;
    uint32_t* reg; uint32_t* used;   // used: LDS bitmap (one wave per frame: the USED map; several waves per frame: this wave's OWN marks)
    uint32_t* ring;                  // LDS: the last kRing region points (the breadth-first frontier lives here)
    int sw, sh, lane, ring_mask;
    // several waves per frame (k_lsd_grow_mw) only:
    const uint32_t* comm;            // LDS: the COMMITTED USED map -- read here, written by the main wave when a region is final
    uint32_t* tent;                  // LDS: 4 bits per pixel, who CLAIMS the pixel (advisory: a claim can be overwritten; what a wave itself holds is in
                                     // its private map `used`): 0 nobody; h = 1..7 the region helper h is growing; 7 + h a finished region of helper
                                     // h that waits for its turn; kMwMainId the region the main wave is growing
    const int* tent_pos;             // LDS: per helper, the seed position (rank in the seed order) of its latest attempt, then [kMwMaxWaves ..) the lowest
                                     // and [2 kMwMaxWaves ..) the highest seed position among its finished regions that still wait for their turn
    int policy;                      // 0: a finished region's claim is judged like a growing one's; 1: by the position range of that helper's finished regions
    int tent_id, my_pos;             // this wave's id and the seed position of the region it is growing
    int reg_cap;                     // entries the list at `reg` can take
    uint32_t* assumed; int assumed_cap;   // LDS: pixels this attempt treated as USED because an earlier seed's unfinished region holds them
};
constexpr int kMwMainId = 15, kMwPending = 7;   // owner ids: helpers 1..7, their finished-but-uncommitted regions 8..14, main 15
struct Rect { double x1, y1, x2, y2, width; };

__device__ __forceinline__ bool is_used(const GrowCtx& g, int p) { return (g.used[p >> 5] >> (p & 31)) & 1u; }
__device__ __forceinline__ void set_used(const GrowCtx& g, int p) { atomicOr(&g.used[p >> 5], 1u << (p & 31)); }   // fire-and-forget ds_or
__device__ __forceinline__ int tent_owner(const GrowCtx& g, int p) { return (int)((g.tent[p >> 3] >> ((unsigned)(p & 7) * 4u)) & 15u); }
// The three updates below change a nibble with a check or an atomicAnd followed by a separate atomicOr: two waves claiming one pixel at the same moment can
// leave the OR of their ids (a phantom owner: another helper, a finished region, main) or wipe the other's claim.  That is tolerated, not overlooked: a claim
// is ADVISORY (policy) -- the main wave validates every speculative region against the committed map C, and every index derived from an
// owner id stays in range for all 16 values -- so a wrong nibble can only cause a spurious give-up or a wasted attempt.  tests/test_spec_grow_model.py
// (test_protocol_is_exact_whatever_the_claim_nibbles_say) scrambles the model's claims at random and requires the sequential result all the same.
__device__ __forceinline__ void tent_release(const GrowCtx& g, int p, int id) {   // give the pixel back if it still carries `id`
    const unsigned sh = (unsigned)(p & 7) * 4u;
    if (((g.tent[p >> 3] >> sh) & 15u) == (unsigned)id) atomicAnd(&g.tent[p >> 3], ~(15u << sh));
}
__device__ __forceinline__ void tent_retag(const GrowCtx& g, int p, int from, int to) {
    const unsigned sh = (unsigned)(p & 7) * 4u;
    if (((g.tent[p >> 3] >> sh) & 15u) == (unsigned)from) { atomicAnd(&g.tent[p >> 3], ~(15u << sh)); atomicOr(&g.tent[p >> 3], (unsigned)to << sh); }
}
template <bool MW> __device__ __forceinline__ void set_used_t(const GrowCtx& g, int p) {
    set_used(g, p);
    if (MW) {   // and the claim nibble becomes mine (two atomics: the transient 0 only hides a claim for a moment, which costs nothing but a wasted attempt)
        const unsigned sh = (unsigned)(p & 7) * 4u;
        atomicAnd(&g.tent[p >> 3], ~(15u << sh));
        atomicOr(&g.tent[p >> 3], (unsigned)g.tent_id << sh);
    }
}
// un-mark (refinement, radius reduction): out of this wave's own map, and the claim is given back if it is still this wave's
__device__ __forceinline__ void mw_unmark(const GrowCtx& g, int p) {
    atomicAnd(&g.used[p >> 5], ~(1u << (p & 31)));
    tent_release(g, p, g.tent_id);
}

// region_grow (lsd.cpp).  The region list is processed breadth-first, SEVEN region points at a time: lanes
// 9c..9c+8 hold the 3x3 neighbourhood of the c-th point of the batch (63 lanes), so one round of gathers serves
// seven points and its HBM latency is paid once.  Acceptances are applied strictly in the reference's order
// (point by point, row by row inside a neighbourhood = ascending lane): the lowest passing lane is accepted,
// reg_angle is updated, and only HIGHER lanes are re-tested with the new angle; a pixel that sits in two
// neighbourhoods of the batch is invalidated in the later one once accepted.  Undefined pixels are pre-marked
// USED, so no NOTDEF test; a pixel that is USED when the batch is fetched needs no data at all (USED bits are
// only ever set while a region grows), so a region interior costs almost no HBM sectors.
// The seed's own record arrives with the call (fetched for 64 seeds at once by the caller).  Neighbour prefetching
// and a look-ahead over the next batch were tried and removed again: the gathers mostly hit in L2 (~800 cycles per
// round) while the extra broadcasts cost 16 % more instructions in a kernel that is issue-bound once the other
// streams fill the SIMDs (profiles/r01g_sq_counters.md).
// The centroid sums of region2rect are not accumulated here any more: most regions are smaller than min_reg_size
// and are dropped, the others get them from centroid_sums() (same order of additions).
// The angle test `|a - theta| <= prec` with theta = fastAtan2(sum) is decided WITHOUT computing theta whenever the
// pixel is not within 0.02 deg of the tolerance: cos of the true angle between the pixel's unit vector and the sum
// vector is compared with cos(prec -+ band); the band covers the polynomial error of cv::fastAtan2 (<= 0.0096 deg) and
// all f32 roundings with a factor two to spare, so outside it both tests agree by construction.  theta itself (one
// f32 division + polynomial + two f64 operations on the critical path of EVERY accepted pixel before) is evaluated
// only for a pixel inside the band that could be the next one accepted, and once when the region is complete.
// seed_cs: (float)cos / (float)sin of the seed's f64 angle when the caller has them (it evaluates them for 64 seeds in
// one go: a sincos here runs with 64 lanes computing the same value), else NULL.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "region_grow's hand-scheduled acceptance block (wait states, wave64, v_readlane hazards) is verified for gfx950 only: port it before building for another target"
#endif
// MW (several waves per frame, k_lsd_grow_mw): a pixel is used when it is COMMITTED (g.comm, written by the main wave only) or part of
// the region this wave is growing (its private map g.used).  A pixel CLAIMED by somebody else (the claim nibbles, g.tent):
//   by a FINISHED region that waits for its turn (this helper's own earlier ones included): its pixels are known and will most likely be
//     committed by the time this seed's turn comes -- treated as used and written to the `assumed` list, which the main wave checks at
//     this seed's turn (every assumed pixel must be committed by then);
//   by a region still GROWING from an earlier seed (the main wave's always is): the sequential scan gives that region precedence and it
//     will most likely take this seed's pixels too -- when such a pixel is about to be accepted the wave GIVES UP, cheaply;
//   by a region growing from a LATER seed: ignored, and overwritten on acceptance (that speculation will be found invalid at its turn).
// The main wave ignores every claim: it IS the sequential scan.  Giving up (also when a list outgrows its space) returns
// -1 - (entries written and marked so far).  Model of the protocol: tests/test_spec_grow_model.py.  Measured alternatives
// (profiles/r03_lsd_grow_mw.md): yielding to every earlier claim, and assuming every earlier claim used.
template <bool MW>
__device__ int region_grow(const GrowCtx& g, int seed, bool have_deg, float seed_deg, const float2* seed_cs, double prec, float c_pass,
                           float c_fail, double& reg_angle, int* n_exact_tests = nullptr, long long* racc = nullptr) {
    const int lane = g.lane;
#ifdef PLP_GROW_PROF_ROUND   // diagnostic build: cycles of a round by phase (frame 0 only; every stamp drains the wave's memory counters first and costs ~540 cycles itself)
    long long rp_t = racc ? clock64() : 0;
#define PLP_RSTAMP(k) do { if (racc) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t_ = clock64(); racc[k] += t_ - rp_t; rp_t = t_; } } while (0)
#else
#define PLP_RSTAMP(k)
#endif
    int nreg = 1;
    const int sx = seed % g.sw, sy = seed / g.sw;
    reg_angle = (double)(have_deg ? seed_deg : g.pix[seed].deg) * (3.14159265358979323846 / 180);
    float sumdx, sumdy;
    if (seed_cs) { sumdx = seed_cs->x; sumdy = seed_cs->y; }
    else {
        double s_sin, s_cos;
        sincos(reg_angle, &s_sin, &s_cos);
        sumdx = (float)s_cos; sumdy = (float)s_sin;
    }
    bool theta_valid = true;   // reg_angle is the seed's own angle until the first acceptance
    if (lane == 0) {
        const uint32_t c = (uint32_t)sx | ((uint32_t)sy << 16);
        g.reg[0] = c; g.ring[0] = c;
        set_used_t<MW>(g, seed);
    }
    __builtin_amdgcn_wave_barrier();
    const int slot = lane / 9, k9 = lane - slot * 9;   // slot 0..6 (lane 63: slot 7, idle)
    // the band thresholds as scalars for the hand-scheduled block; a tolerance without a band (c_pass = 2) never enters it
    const bool banded = c_pass <= 1.f;
    const int s_cpass = __builtin_amdgcn_readfirstlane(__float_as_int(c_pass)), s_cfail = __builtin_amdgcn_readfirstlane(__float_as_int(c_fail));
    const int ddx = k9 % 3 - 1, ddy = k9 / 3 - 1;
    const int npix_m1 = g.sw * g.sh - 1;
    bool gave_up = false;
    const int list_cap = MW ? __builtin_amdgcn_readfirstlane(g.reg_cap) : 0;   // wave-uniform, and said so (the block below keeps nreg in a scalar register)
    for (int i = 0; i < nreg && !(MW && gave_up);) {
        const int nb = min(7, nreg - i);
        bool cand = slot < nb;
        int nx = 0, ny = 0, np = 0;
        float deg = 0.f;
        float2 ncs = make_float2(0.f, 0.f);
        {
            uint32_t c = 0;
            if (nreg > i + g.ring_mask + 1) {   // frontier outgrew the LDS ring: read the HBM copy (uniform branch)
                if (cand) c = __hip_atomic_load(&g.reg[i + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else c = g.ring[(i + min(slot, 6)) & g.ring_mask];
            nx = (int)(c & 0xffff) + ddx; ny = (int)(c >> 16) + ddy;
        }
        cand = cand && nx >= 0 && ny >= 0 && nx < g.sw && ny < g.sh;
        np = ny * g.sw + nx;
        PLP_RSTAMP(0);   // the batch's points are here (ring read)
        bool foreign = false;   // claimed by the growing region of an earlier seed (or of the main wave)
        if (MW) {
            bool assume = false;
            if (cand) {
                bool used = ((g.comm[np >> 5] | g.used[np >> 5]) >> (np & 31)) & 1u;
                const int owner = tent_owner(g, np);
                if (!used && owner != 0 && owner != g.tent_id && g.tent_id != kMwMainId) {   // somebody else's claim
                    if (owner > kMwPending && owner != kMwMainId) {   // a FINISHED region that waits for its turn: its pixels are known
                        const int x = owner - kMwPending - 1;
                        if (g.policy == 0) foreign = g.tent_pos[x] < g.my_pos;
                        else if (g.tent_pos[2 * kMwMaxWaves + x] < g.my_pos) { assume = true; used = true; }   // all of that helper's finished regions come earlier: used by my turn
                        else foreign = !(g.tent_pos[kMwMaxWaves + x] > g.my_pos);                              // all later: ignore the claim; mixed: yield
                    } else foreign = owner == kMwMainId || g.tent_pos[owner - 1] < g.my_pos;   // a region still growing from an earlier seed; a later seed's claim is ignored
                }
                cand = !used;
            }
            const unsigned long long am = __builtin_amdgcn_ballot_w64(assume);
            if (am) {   // rare: regions of different waves touch
                // the fill count lives in LDS behind the list (g.assumed[g.assumed_cap]): read back as a wave-uniform value, so that nothing
                // per-lane flows into the give-up decision (the hand-scheduled block wants its loop state in scalar registers)
                const int na = __builtin_amdgcn_readfirstlane((int)g.assumed[g.assumed_cap]);
                if (na + __popcll(am) > g.assumed_cap) gave_up = true;
                else {
                    if (assume) g.assumed[na + __popcll(am & ((1ull << lane) - 1ull))] = (uint32_t)nx | ((uint32_t)ny << 16);
                    if (lane == 0) g.assumed[g.assumed_cap] = (uint32_t)(na + __popcll(am));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        else {   // the USED word of every lane's pixel, read unconditionally from a clamped index: no predicated block (three instructions and a branch) around one LDS read
            const int pc = min(max(np, 0), npix_m1);
            uint32_t uw = g.used[pc >> 5];
            asm volatile("" : "+v"(uw));   // (keeps the read where it is: the optimiser otherwise sinks it into a block predicated on `cand`)
            cand = cand & !((uw >> (pc & 31)) & 1u);
        }
        PLP_RSTAMP(1);   // USED test
        if (cand) {   // 16 bytes per live neighbour; a 32-bit byte offset from the frame's (scalar) base: one shift instead of three 64-bit additions per round
            const LsdPix px = *reinterpret_cast<const LsdPix*>(reinterpret_cast<const char*>(g.pix) + ((uint32_t)np << 4));
            deg = px.deg; ncs = px.cs;
        }
        PLP_RSTAMP(2);   // record gather
        // ---- acceptances in order.  Accepted lanes are strictly increasing, so the set of accepted lanes (a bit mask)
        // already is the order: the list append and the USED bits are written by the accepted lanes themselves after
        // the loop, in parallel (the loop used to hand every acceptance to lane 0: two more broadcasts and a
        // predicated block per accepted pixel of a kernel that is instruction-bound).
        const int n_before = nreg;
        unsigned long long acc = 0;
        // candidate / "comes after the last accepted lane" sets live in scalar registers: per iteration the vector
        // unit only evaluates the two compares of the guard-banded test
        unsigned long long candmask = __builtin_amdgcn_ballot_w64(cand), gt = ~0ull;
        while (true) {
            // ---- the certain decisions, hand-scheduled: 26 instructions per acceptance where the compiler's rendering of the
            // same loop took 40 (a "continue" flag kept as a lane mask, wait states for packed f32 results, the test for "band
            // pixel before the first certain one").  The block decides lanes only while NO eligible lane is inside the guard
            // band; otherwise (`careful`) one decision is taken by the C++ below and the block is entered again.  Its test uses
            // fused multiply-adds: the band covers every rounding of either form, so a lane that is certain here is certain.
            // The sum itself is accumulated with plain f32 additions in acceptance order (bit-exact with the reference).
            // Hazards honoured by hand (the compiler does not look into the block): two VALU between v_rsq and its consumer;
            // lane selects of v_readlane come from the scalar unit (never from a VALU-written SGPR); an SGPR written by
            // v_readlane is read by the VALU three or more instructions later; trailing s_nop before the compiler's code
            // resumes.  EXEC is all ones here: every branch around this point is wave-uniform.
            int careful = 1;
            if (banded) {
                const unsigned long long acc0 = acc;
                float ta, tb;
                unsigned long long t_elig, t_p, t_f;
                int t_k, t_ap, t_rx, t_ry;
                asm volatile(
                    "Lgrow_top_%=:\n\t"
                    "s_and_b64 %[elig], %[cand], %[gt]\n\t"
                    "s_cbranch_scc0 Lgrow_done_%=\n\t"
                    "v_mul_f32 %[a], %[sx], %[sx]\n\t"
                    "v_fmac_f32 %[a], %[sy], %[sy]\n\t"
                    "v_rsq_f32 %[a], %[a]\n\t"
                    "v_mul_f32 %[b], %[cx], %[sx]\n\t"
                    "v_fmac_f32 %[b], %[cy], %[sy]\n\t"
                    "v_mul_f32 %[a], %[b], %[a]\n\t"
                    "v_cmp_le_f32 vcc, %[cp], %[a]\n\t"
                    "v_cmp_gt_f32 %[f], %[cf], %[a]\n\t"
                    "s_or_b64 %[f], %[f], vcc\n\t"
                    "s_andn2_b64 %[f], %[elig], %[f]\n\t"
                    "s_cbranch_scc1 Lgrow_band_%=\n\t"
                    "s_and_b64 %[p], vcc, %[elig]\n\t"
                    "s_cbranch_scc0 Lgrow_done_%=\n\t"
                    "s_ff1_i32_b64 %[k], %[p]\n\t"
                    "v_readlane_b32 %[ap], %[np], %[k]\n\t"
                    "v_readlane_b32 %[rx], %[cx], %[k]\n\t"
                    "v_readlane_b32 %[ry], %[cy], %[k]\n\t"
                    "s_bitset1_b64 %[acc], %[k]\n\t"
                    "v_cmp_eq_u32 vcc, %[ap], %[np]\n\t"
                    "s_lshl_b64 %[gt], -2, %[k]\n\t"
                    "v_add_f32 %[sx], %[rx], %[sx]\n\t"
                    "v_add_f32 %[sy], %[ry], %[sy]\n\t"
                    "s_andn2_b64 %[cand], %[cand], vcc\n\t"
                    "s_branch Lgrow_top_%=\n"
                    "Lgrow_band_%=:\n\t"
                    "s_mov_b32 %[careful], 1\n\t"
                    "s_branch Lgrow_end_%=\n"
                    "Lgrow_done_%=:\n\t"
                    "s_mov_b32 %[careful], 0\n"
                    "Lgrow_end_%=:\n\t"
                    "s_nop 4"
                    : [sx] "+v"(sumdx), [sy] "+v"(sumdy), [cand] "+s"(candmask), [gt] "+s"(gt), [acc] "+s"(acc),
                      [careful] "=&s"(careful), [a] "=&v"(ta), [b] "=&v"(tb), [elig] "=&s"(t_elig), [p] "=&s"(t_p), [f] "=&s"(t_f),
                      [k] "=&s"(t_k), [ap] "=&s"(t_ap), [rx] "=&s"(t_rx), [ry] "=&s"(t_ry)
                    : [cx] "v"(ncs.x), [cy] "v"(ncs.y), [np] "v"(np), [cp] "s"(s_cpass), [cf] "s"(s_cfail)
                    : "vcc", "scc");
                if (acc != acc0) theta_valid = false;
            }
            if (!careful) break;
            // ---- one decision the careful way (an eligible pixel is inside the band, or the tolerance has no band)
            const unsigned long long elig = candmask & gt;
            if (!elig) break;
            const float inv = __builtin_amdgcn_rsqf(sumdx * sumdx + sumdy * sumdy);   // |sum| >= 0.9: no denormal care needed
            const float cosang = (ncs.x * sumdx + ncs.y * sumdy) * inv;
            const unsigned long long P = __builtin_amdgcn_ballot_w64(cosang >= c_pass) & elig;
            const unsigned long long U = elig & ~P & ~__builtin_amdgcn_ballot_w64(cosang < c_fail);
            // a pixel inside the band matters only if it comes before the first certain acceptance
            const unsigned long long before = P ? ((P & (0ull - P)) - 1ull) : ~0ull;
            unsigned long long bal = P;
            if (U & before) {
                if (n_exact_tests) ++*n_exact_tests;
                if (!theta_valid) { reg_angle = (double)fast_atan2_deg_l(sumdy, sumdx) * (3.14159265358979323846 / 180); theta_valid = true; }
                float deg_here = deg;   // (opaque to the optimiser: the f64 conversion of every lane's angle is work for the dozen band cases of a frame, not for every round)
                asm volatile("" : "+v"(deg_here));
                bal = __builtin_amdgcn_ballot_w64(aligned_to((double)deg_here * (3.14159265358979323846 / 180), reg_angle, prec)) & elig;
            }
            if (!bal) break;
            const int k = __builtin_ctzll(bal);
            const float ccos = bcast_f(ncs.x, k), csin = bcast_f(ncs.y, k);
            const int ap = bcast_i(np, k);
            acc |= 1ull << k;
            sumdx = __fadd_rn(sumdx, ccos);
            sumdy = __fadd_rn(sumdy, csin);
            theta_valid = false;
            gt = ~((2ull << k) - 1ull);                                       // lanes above k
            candmask &= ~__builtin_amdgcn_ballot_w64(np == ap);              // the same pixel seen from a later point of the batch
        }
        nreg = n_before + __popcll(acc);   // (the block keeps the accepted SET; the count follows from it: one scalar instruction per acceptance less)
        if (MW) {   // give up before anything of this round is written or marked
            if (gave_up || nreg > list_cap || (acc & __builtin_amdgcn_ballot_w64(foreign))) { gave_up = true; nreg = n_before; acc = 0; }
        }
        PLP_RSTAMP(3);   // acceptance loop
        if ((acc >> lane) & 1ull) {
            const int pos = n_before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));   // accepted lanes below this one
            const uint32_t c = (uint32_t)nx | ((uint32_t)ny << 16);
            g.ring[pos & g.ring_mask] = c;
            set_used_t<MW>(g, np);
            // HBM copy of the list (a 32-bit byte offset from the list's base, which is scalar in k_lsd_grow)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(g.reg) + ((uint32_t)pos << 2)), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_wave_barrier();   // LDS operations of one wave complete in order: the next round's reads see these writes
        PLP_RSTAMP(4);   // appends, marks
        if (racc) { racc[5] += 1; racc[6] += __popcll(acc); }
        i += nb;
    }
    if (MW && gave_up) return -1 - nreg;
    if (!theta_valid) reg_angle = (double)fast_atan2_deg_l(sumdy, sumdx) * (3.14159265358979323846 / 180);
    return nreg;
}
// the region list in HBM is read by all lanes after growing (only regions that are kept get that far)
__device__ __forceinline__ void region_list_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// region2rect (lsd.cpp): centroid given; inertia + extents.  Sequential f64 sums in region order, fed
// 64 region points at a time (coalesced gathers, then a shuffle-driven serial chain).
__device__ void region2rect(const GrowCtx& g, int nreg, double reg_angle, double prec, const double cen[3], Rect& rec) {
    const int lane = g.lane;
    const double x = cen[0] / cen[2], y = cen[1] / cen[2];
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int base = 0; base < nreg; base += 64) {
        const int j = base + lane;
        double txx = 0, tyy = 0, txy = 0;   // this lane's addends (same roundings as the reference's per-point products)
        if (j < nreg) {
            const uint32_t c = g.reg[j];
            const double w = pix_mod(g.pix[(int)(c >> 16) * g.sw + (int)(c & 0xffff)]);
            const double dx = (double)(int)(c & 0xffff) - x, dy = (double)(int)(c >> 16) - y;
            txx = dy * dy * w; tyy = dx * dx * w; txy = dx * dy * w;
        }
        const int cnt = min(64, nreg - base);
        for (int t = 0; t < cnt; ++t) {   // strictly sequential adds, region order
            Ixx += bcast_d(txx, t);
            Iyy += bcast_d(tyy, t);
            Ixy -= bcast_d(txy, t);
        }
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg_l((float)(lambda - Ixx), (float)Ixy)
                                           : (double)fast_atan2_deg_l((float)Ixy, (float)(lambda - Iyy));
    theta *= (3.14159265358979323846 / 180);
    if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += 3.14159265358979323846;
    double dx, dy;
    sincos(theta, &dy, &dx);
    // extents: min / max are order independent -> lane-parallel
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int j = lane; j < nreg; j += 64) {
        const uint32_t c = g.reg[j];
        const double rdx = (double)(int)(c & 0xffff) - x, rdy = (double)(int)(c >> 16) - y;
        const double l = rdx * dx + rdy * dy, w = -rdx * dy + rdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l);
        w_max = fmax(w_max, w); w_min = fmin(w_min, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l_max = fmax(l_max, shfl_d(l_max, lane ^ o)); l_min = fmin(l_min, shfl_d(l_min, lane ^ o));
        w_max = fmax(w_max, shfl_d(w_max, lane ^ o)); w_min = fmin(w_min, shfl_d(w_min, lane ^ o));
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    if (rec.width < 1.0) rec.width = 1.0;
}

// region2rect for a region that still sits completely in the LDS frontier ring (nreg <= ring size, at most 256 points):
// coordinates come from LDS, every lane keeps its (up to four) points and their weights in registers, so the centroid
// sums, the inertia sums and the extents need ONE round of gathers instead of three passes over the HBM copy of the
// list.  Same additions in the same order as centroid_sums() + region2rect().
// The ~55 rectangle fits per frame that follow a shrink step of reduce_region_radius also take the fit that keeps the points in registers, reading
// the coordinates from the reordered list (FROM_LIST): the general path (centroid_sums + region2rect) costs nine broadcasts and adds per point.  Same
// additions in the same order by construction (rect_from_ring == centroid_sums + region2rect is what the first fits already rely on).  Round 5:
// k_lsd_grow 12.05 -> 11.75 ms per 2048 frames for four registers more, line tests + fuzz bit-exact.
constexpr int kRingPts = 4;   // points per lane the fit from the ring keeps in registers: regions up to 64 * kRingPts points take it
// FROM_LIST: the coordinates come from the HBM copy of the list instead of the ring (after reduce_region_radius has reordered the list; the ring still
// serves as scratch).
template <bool FROM_LIST = false> __device__ void rect_from_ring(const GrowCtx& g, int nreg, double reg_angle, double prec, Rect& rec) {
    const int lane = g.lane;
    // per lane up to four points, kept small across the two sequential passes (this function is the kernel's register peak, and what
    // two of these waves leave of a SIMD's registers decides how many waves of the neighbouring kernels fit): coordinates stay packed as
    // in the list
    uint32_t pc[kRingPts];
    double w[kRingPts];
#pragma unroll
    for (int u = 0; u < kRingPts; ++u) {
        const int j = lane + 64 * u;
        pc[u] = 0; w[u] = 0;
        if (64 * u >= nreg) continue;   // uniform: most fitted regions have fewer than 64 points
        if (j < nreg) {
            pc[u] = FROM_LIST ? __hip_atomic_load(&g.reg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : g.ring[j & g.ring_mask];
            w[u] = pix_mod(g.pix[(int)(pc[u] >> 16) * g.sw + (int)(pc[u] & 0xffff)]);
        }
    }
    auto PX = [&](int u) -> int { return (int)(pc[u] & 0xffff); };
    auto PY = [&](int u) -> int { return (int)(pc[u] >> 16); };
    auto WT = [&](int u) -> double { return w[u]; };
    // The three running sums of a pass are three lanes of one loop: the addends go through LDS (3 doubles per point,
    // 32 points at a time), lane k adds stream k in region order -- 2 instructions per point instead of 9 broadcasts and
    // adds.  The scratch is the frontier ring itself: its coordinates are in registers by now and the next region_grow
    // starts it afresh (256 words = 32 x 3 doubles + slack).
    double* sc = reinterpret_cast<double*>(g.ring);
    __builtin_amdgcn_wave_barrier();
    auto chain3 = [&](auto addend) -> double {   // addend(u, k): stream k of this lane's point u
        double acc = 0;
        const int nchunk = (nreg + 31) >> 5;
        for (int c = 0; c < nchunk; ++c) {
            const int u = c >> 1;
            if ((lane >> 5) == (c & 1) && lane + 64 * u < nreg) {
                double* o = sc + 3 * (lane & 31);
#pragma unroll
                for (int uu = 0; uu < kRingPts; ++uu)
                    if (uu == u) { o[0] = addend(uu, 0); o[1] = addend(uu, 1); o[2] = addend(uu, 2); }
            }
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(32, nreg - 32 * c);
            if (lane < 3) {   // PLP_CHAIN_BATCH addends are fetched together, then added one after the other: the chain is the additions, not one LDS
                int t = 0;        // round trip per point (the loop used to be ds_read -> wait -> add per point: ~130 cycles each, a sixth of the kernel)
#if PLP_CHAIN_BATCH >= 16
                for (; t + 16 <= cnt; t += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = sc[3 * (t + u) + lane];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += v[u];
                }
#endif
#if PLP_CHAIN_BATCH >= 8
                for (; t + 8 <= cnt; t += 8) {
                    const double v0 = sc[3 * t + lane], v1 = sc[3 * t + 3 + lane], v2 = sc[3 * t + 6 + lane], v3 = sc[3 * t + 9 + lane];
                    const double v4 = sc[3 * t + 12 + lane], v5 = sc[3 * t + 15 + lane], v6 = sc[3 * t + 18 + lane], v7 = sc[3 * t + 21 + lane];
                    acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
                }
#elif PLP_CHAIN_BATCH >= 4
                for (; t + 4 <= cnt; t += 4) {
                    const double v0 = sc[3 * t + lane], v1 = sc[3 * t + 3 + lane], v2 = sc[3 * t + 6 + lane], v3 = sc[3 * t + 9 + lane];
                    acc += v0; acc += v1; acc += v2; acc += v3;
                }
#endif
                for (; t < cnt; ++t) acc += sc[3 * t + lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
        return acc;
    };
    double acc = chain3([&](int u, int k) -> double { const double w = WT(u); return k == 0 ? (double)PX(u) * w : k == 1 ? (double)PY(u) * w : w; });
    const double sx = bcast_d(acc, 0), sy = bcast_d(acc, 1), sw = bcast_d(acc, 2);
    const double x = sx / sw, y = sy / sw;
    acc = chain3([&](int u, int k) -> double {
        const double dx = (double)PX(u) - x, dy = (double)PY(u) - y, w = WT(u);
        return k == 0 ? dy * dy * w : k == 1 ? dx * dx * w : -(dx * dy * w);   // Ixy -= v  ==  Ixy += -v
    });
    const double Ixx = bcast_d(acc, 0), Iyy = bcast_d(acc, 1), Ixy = bcast_d(acc, 2);
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg_l((float)(lambda - Ixx), (float)Ixy)
                                           : (double)fast_atan2_deg_l((float)Ixy, (float)(lambda - Iyy));
    theta *= (3.14159265358979323846 / 180);
    if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += 3.14159265358979323846;
    double dx, dy;
    sincos(theta, &dy, &dx);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
#pragma unroll
    for (int u = 0; u < kRingPts; ++u)
        if (lane + 64 * u < nreg) {
            const double rdx = (double)PX(u) - x, rdy = (double)PY(u) - y;
            const double l = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
            l_max = fmax(l_max, l); l_min = fmin(l_min, l);
            w_max = fmax(w_max, ww); w_min = fmin(w_min, ww);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l_max = fmax(l_max, shfl_d(l_max, lane ^ o)); l_min = fmin(l_min, shfl_d(l_min, lane ^ o));
        w_max = fmax(w_max, shfl_d(w_max, lane ^ o)); w_min = fmin(w_min, shfl_d(w_min, lane ^ o));
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    if (rec.width < 1.0) rec.width = 1.0;
}

__device__ __forceinline__ double rect_density(int nreg, const Rect& r) {
    const double d = sqrt((r.x2 - r.x1) * (r.x2 - r.x1) + (r.y2 - r.y1) * (r.y2 - r.y1));
    return (double)nreg / (d * r.width);
}

// centroid sums in region order (needed after reduce_region_radius reorders the list)
__device__ void centroid_sums(const GrowCtx& g, int nreg, double cen[3]) {
    cen[0] = cen[1] = cen[2] = 0;
    for (int base = 0; base < nreg; base += 64) {
        const int j = base + g.lane;
        double tx = 0, ty = 0, w = 0;
        if (j < nreg) {
            const uint32_t c = g.reg[j];
            w = pix_mod(g.pix[(int)(c >> 16) * g.sw + (int)(c & 0xffff)]);
            tx = (double)(int)(c & 0xffff) * w; ty = (double)(int)(c >> 16) * w;
        }
        const int cnt = min(64, nreg - base);
        for (int t = 0; t < cnt; ++t) { cen[0] += bcast_d(tx, t); cen[1] += bcast_d(ty, t); cen[2] += bcast_d(w, t); }
    }
}

// One pass of reduce_region_radius (lsd.cpp): the points farther than sqrt(radSq) from the seed leave the list by SWAP-WITH-LAST, in list order --
//     for (i = 0; i < m; ++i) if (far(reg[i])) { used = NOTUSED; reg[i] = reg[m - 1]; --m; --i; }
// The order it leaves decides the f64 sums of the next rectangle fit, so it is replayed exactly, but not one point after the other (one lane, a
// dependent global load per point: ~340 cycles each, 5 % of the kernel): the loop is a two-pointer walk in which every far position below the final
// length m_f = #kept receives a kept point from beyond m_f -- the k-th such position from the LEFT the k-th such point from the RIGHT (points pulled in
// that are far themselves are dropped on arrival, which is why only kept ones count).  Ranks by ballots and running counts over 64-point chunks; the
// pairs meet in the frontier ring (idle here): up to ring/2 moves, else the sequential form.  Returns the new length.
// MW (k_lsd_grow_mw): the far points stay in the list BEHIND the live part (its second list must remain a permutation of what was accepted: the claims
// are released from it); their order there is free, so the pairs are swapped.
template <bool MW> __device__ int reduce_radius_pass(const GrowCtx& g, int nreg, double xc, double yc, double radSq) {
    const int lane = g.lane, cap = (g.ring_mask + 1) >> (MW ? 2 : 1);
    auto far_of = [&](uint32_t c) -> bool {
        const double ddx = (double)(int)(c & 0xffff) - xc, ddy = (double)(int)(c >> 16) - yc;
        return ddx * ddx + ddy * ddy > radSq;
    };
    // pass 1: un-mark the far points, count the kept ones
    int m_f = 0;
    for (int base = 0; base < nreg; base += 64) {
        const int j = base + lane;
        bool keep = false;
        if (j < nreg) {
            const uint32_t c = __hip_atomic_load(&g.reg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            keep = !far_of(c);
            if (!keep) {
                const int p = (int)(c >> 16) * g.sw + (int)(c & 0xffff);
                if (MW) mw_unmark(g, p); else atomicAnd(&g.used[p >> 5], ~(1u << (p & 31)));
            }
        }
        m_f += __popcll(__ballot(keep));
    }
    if (m_f == nreg) return nreg;
    // pass 2: far positions below m_f (destinations, by rank from the left) and kept points at or beyond m_f (sources, by rank from the left; the
    // pairing wants them from the right: their number equals the number of destinations, so rank r from the left is rank n_mv - 1 - r from the right)
    uint32_t* dst = g.ring; uint32_t* src = g.ring + cap; uint32_t* dval = g.ring + 2 * cap; uint32_t* spos = g.ring + 3 * cap;   // the last two: MW only
    int n_dst = 0, n_src = 0;
    bool fits = true;
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < nreg; base += 64) {
        const int j = base + lane;
        uint32_t c = 0;
        bool is_dst = false, is_src = false;
        if (j < nreg) {
            c = __hip_atomic_load(&g.reg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const bool far = far_of(c);
            is_dst = far && j < m_f; is_src = !far && j >= m_f;
        }
        const unsigned long long md = __ballot(is_dst), ms = __ballot(is_src);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (n_dst + __popcll(md) > cap || n_src + __popcll(ms) > cap) { fits = false; break; }
        if (is_dst) { const int k = n_dst + __popcll(md & below); dst[k] = (uint32_t)j; if (MW) dval[k] = c; }
        if (is_src) { const int k = n_src + __popcll(ms & below); src[k] = c; if (MW) spos[k] = (uint32_t)j; }
        n_dst += __popcll(md); n_src += __popcll(ms);
    }
    __builtin_amdgcn_wave_barrier();
    if (fits) {   // n_dst == n_src
        for (int k = lane; k < n_dst; k += 64) {
            __hip_atomic_store(&g.reg[dst[k]], src[n_dst - 1 - k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (MW) __hip_atomic_store(&g.reg[spos[n_dst - 1 - k]], dval[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_wave_barrier();
        return m_f;
    }
    int m = nreg;   // more moves than the ring holds: the sequential form (the points are un-marked already)
    if (lane == 0) {
        for (int i = 0; i < m; ++i) {
            const uint32_t c = __hip_atomic_load(&g.reg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (far_of(c)) {
                const uint32_t lastv = __hip_atomic_load(&g.reg[m - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&g.reg[i], lastv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (MW) __hip_atomic_store(&g.reg[m - 1], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                --m; --i;
            }
        }
    }
    return __shfl(m, 0);
}

// grid = (ceil(B / wpb)), block = 64 * wpb: wave w of a block handles frame wpb*blockIdx.x + w (wpb = waves whose
// USED bitmap + frontier ring fit 64 KB of LDS together, at most 4).
__global__ __launch_bounds__(256) void k_lsd_grow(LinePlanes P, LsdParams lp, int B, int wpb, int ring) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    // the wave's index is wave-uniform, and said so: every per-frame pointer below then lives in scalar registers (global addresses become an SGPR base + a
    // 32-bit lane offset instead of 64-bit vector arithmetic)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * wpb + wv;
    if (b >= B) return;
    const int n = P.sw * P.sh, nwords = (n + 31) / 32, nv = (P.sw - 1) * (P.sh - 1);
    GrowCtx g;
    g.comm = nullptr; g.tent = nullptr; g.tent_pos = nullptr; g.tent_id = 0; g.my_pos = 0; g.reg_cap = n; g.assumed = nullptr; g.assumed_cap = 0; g.policy = 0;
    g.pix = P.pix + (size_t)b * n;
    g.reg = P.reg + (size_t)b * P.reg_frame_stride; const int nw_al = (nwords + 1) & ~1;   // the ring doubles as f64 scratch: 8-byte aligned
    g.used = s_bits + (size_t)wv * (nw_al + ring); g.ring = g.used + nw_al; g.ring_mask = ring - 1;
    g.sw = P.sw; g.sh = P.sh; g.lane = lane;
    // USED map starts as the NOTDEF mask: an undefined pixel is never a seed and never aligned, so
    // treating it as used is equivalent and spares a global load per rejected seed
    {
        const uint32_t* u32 = reinterpret_cast<const uint32_t*>(P.undef + (size_t)b * ((n + 63) / 64));
        for (int i = lane; i < nwords; i += 64) g.used[i] = u32[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint32_t* order = P.order + (size_t)b * nv;
    const int n_ord = P.n_order[b];   // defined pixels only (k_lsd_order)
    float4* raw = P.raw + (size_t)b * kLineCap;
    int n_lines = 0;
    long long t_grow = 0, t_rect = 0, t_refine = 0, n_seed = 0, n_pix = 0;
#ifdef PLP_GROW_PROF_REFINE   // diagnostic build: what the refinement is made of (costs 14 registers: not in the shipped kernel)
    long long t_regrow = 0, t_reduce_lane = 0, n_refined = 0, n_reduce_iter = 0, n_reduce_pts = 0, n_fitted = 0;
#define PLP_PR(x) x
#else
#define PLP_PR(x)
#endif
    int n_exact_tests = 0;
#ifdef PLP_GROW_PROF_ROUND   // (tools/build_variant.sh rprof -DPLP_GROW_PROF_ROUND; python tools/grow_profile.py 2048: 'more' = cycles of {ring read, USED test, gather, acceptance loop, appends}, rounds)
    long long racc_v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long* const racc = (P.prof != nullptr && b == 0) ? racc_v : nullptr;
#else
    long long* const racc = nullptr;
#endif
    // phase clocks only for the frame that reports them (every s_memtime is a scalar-memory round trip the wave waits for)
    const bool prof_on = P.prof != nullptr && b == 0;
    auto tick = [&]() -> long long { return prof_on ? clock64() : 0ll; };
    const long long t_begin = tick();
    uint32_t mine_next = lane < n_ord ? order[lane] : 0u;
    for (int base = 0; base < n_ord; base += 64) {
        const bool in_range = base + lane < n_ord;
        const uint32_t mine = mine_next;
        mine_next = base + 64 + lane < n_ord ? order[base + 64 + lane] : 0u;   // the next group's seeds are on their way while this group's regions grow
        // most seeds are already inside an earlier region: test the 64 USED bits in parallel, visit the rest in order
        const bool fresh = in_range && !is_used(g, (int)mine);
        unsigned long long todo = __ballot(fresh);
        // every still-unused seed of this group of 64 fetches its own record now: one round trip for up to 64 regions
        const float s_deg = fresh ? g.pix[mine].deg : 0.f;
        float2 s_cs = make_float2(0.f, 0.f);
        if (fresh) {   // (float)cos / (float)sin of the f64 seed angle, 64 seeds per evaluation
            double sn, cs;
            sincos((double)s_deg * (3.14159265358979323846 / 180), &sn, &cs);
            s_cs = make_float2((float)cs, (float)sn);
        }
        while (todo) {
            const int t = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int seed = bcast_i((int)mine, t);
            if (is_used(g, seed)) continue;   // claimed by a region grown since the ballot
            double reg_angle, cen[3];
            long long t0 = tick();
            const float2 seed_cs = make_float2(bcast_f(s_cs.x, t), bcast_f(s_cs.y, t));
            int nreg = region_grow<false>(g, seed, true, bcast_f(s_deg, t), &seed_cs, lp.prec, lp.c_pass, lp.c_fail, reg_angle, &n_exact_tests, racc);
            t_grow += tick() - t0; ++n_seed; n_pix += nreg;
            if (nreg < lp.min_reg_size) continue;
            Rect rec;
            t0 = tick();
            const int ring_cap = g.ring_mask + 1 >= 256 ? 64 * kRingPts : 0;   // the fit from the ring needs the whole 256-word ring as scratch
            if (nreg <= ring_cap) rect_from_ring(g, nreg, reg_angle, lp.prec, rec);
            else {
                region_list_fence();
                centroid_sums(g, nreg, cen);
                region2rect(g, nreg, reg_angle, lp.prec, cen, rec);
            }
            t_rect += tick() - t0;
            t0 = tick();
            bool keep = true;
            if (lp.refine > 0) {
                double density = rect_density(nreg, rec);
                PLP_PR(++n_fitted;)
                if (density < lp.density_th) {
                    PLP_PR(++n_refined;)
                    // ---- refine: tighter angle tolerance from the points near the seed
                    region_list_fence();
                    const uint32_t c0 = g.reg[0];
                    const double xc = (double)(int)(c0 & 0xffff), yc = (double)(int)(c0 >> 16);
                    const double ang_c = pix_ang(g.pix[(int)(c0 >> 16) * g.sw + (int)(c0 & 0xffff)]);
                    double sum = 0, s_sum = 0;
                    int nn = 0;
                    for (int rb = 0; rb < nreg; rb += 64) {
                        const int j = rb + lane;
                        double a = 0;
                        bool near = false;
                        if (j < nreg) {
                            const uint32_t c = g.reg[j];
                            const int px = (int)(c & 0xffff), py = (int)(c >> 16);
                            atomicAnd(&g.used[(py * g.sw + px) >> 5], ~(1u << ((py * g.sw + px) & 31)));   // *(reg[i].used) = NOTUSED
                            const double ddx = (double)px - xc, ddy = (double)py - yc;
                            near = sqrt(ddx * ddx + ddy * ddy) < rec.width;
                            a = pix_ang(g.pix[py * g.sw + px]);
                        }
                        const double my_d = near ? angle_diff_signed(a, ang_c) : 0.0;
                        const double my_d2 = my_d * my_d;
                        unsigned long long nb = __ballot(near);
                        while (nb) {   // near points in region order
                            const int u = __ffsll((long long)nb) - 1;
                            nb &= nb - 1;
                            sum += bcast_d(my_d, u); s_sum += bcast_d(my_d2, u); ++nn;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const double mean_angle = sum / (double)nn;
                    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)nn + mean_angle * mean_angle);
                    // guard band of the angle test for this tolerance (disabled = every test takes the exact path)
                    float cp = 2.f, cf = -2.f;
                    if (tau >= kLsdBandMinPrec && tau < kLsdBandMaxPrec) { cp = (float)cos(tau - kLsdAngleBand); cf = (float)cos(tau + kLsdAngleBand); }
                    PLP_PR(const long long tr0 = tick();)
                    nreg = region_grow<false>(g, (int)(c0 >> 16) * g.sw + (int)(c0 & 0xffff), false, 0.f, nullptr, tau, cp, cf, reg_angle, nullptr, racc);
                    PLP_PR(t_regrow += tick() - tr0;)
                    region_list_fence();
                    if (nreg < 2) keep = false;
                    else {
                        if (nreg <= ring_cap) rect_from_ring(g, nreg, reg_angle, lp.prec, rec);
                        else {
                            centroid_sums(g, nreg, cen);
                            region2rect(g, nreg, reg_angle, lp.prec, cen, rec);
                        }
                        density = rect_density(nreg, rec);
                        if (density < lp.density_th) {
                            // ---- reduce_region_radius: shrink around the seed until dense enough
                            const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc);
                            const double r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
                            double radSq = r1 > r2 ? r1 : r2;
                            while (density < lp.density_th) {
                                radSq *= 0.75 * 0.75;
                                PLP_PR(++n_reduce_iter; n_reduce_pts += nreg; const long long tl0 = tick();)
                                nreg = reduce_radius_pass<false>(g, nreg, xc, yc, radSq);
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                                __builtin_amdgcn_wave_barrier();
                                PLP_PR(t_reduce_lane += tick() - tl0;)
                                if (nreg < 2) { keep = false; break; }
                                if (nreg <= ring_cap) rect_from_ring<true>(g, nreg, reg_angle, lp.prec, rec);
                                else
                                {
                                    centroid_sums(g, nreg, cen);
                                    region2rect(g, nreg, reg_angle, lp.prec, cen, rec);
                                }
                                density = rect_density(nreg, rec);
                            }
                        }
                    }
                }
            }
            t_refine += tick() - t0;
            if (!keep) continue;
            // +0.5 offset, undo the sub-sampling, cast to f32 (lsd.cpp flsd)
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            if (lp.scale != 1) { rec.x1 /= lp.scale; rec.y1 /= lp.scale; rec.x2 /= lp.scale; rec.y2 /= lp.scale; }
            if (n_lines < kLineCap) { if (lane == 0) raw[n_lines] = make_float4((float)rec.x1, (float)rec.y1, (float)rec.x2, (float)rec.y2); }
            else if (lane == 0) atomicOr(P.status, 4);
            ++n_lines;
        }
    }
    if (lane == 0) P.n_raw[b] = min(n_lines, kLineCap);
    if (lane == 0) { int32_t* gs = P.grow_stats + (size_t)b * 4; gs[0] = (int32_t)n_seed; gs[1] = (int32_t)n_pix; gs[2] = n_exact_tests; gs[3] = 0; }
    if (lane == 0 && prof_on) {   // diagnostics of frame 0: cycles per phase
        P.prof[0] = clock64() - t_begin; P.prof[1] = t_grow; P.prof[2] = t_rect; P.prof[3] = t_refine; P.prof[4] = n_seed; P.prof[5] = n_pix;
#ifdef PLP_GROW_PROF_ROUND
        for (int k = 0; k < 6; ++k) P.prof[6 + k] = racc_v[k];
        P.prof[5] = racc_v[6];   // acceptances (instead of pixels)
#endif
        PLP_PR(P.prof[6] = t_regrow; P.prof[7] = t_reduce_lane; P.prof[8] = n_refined; P.prof[9] = n_reduce_iter; P.prof[10] = n_reduce_pts; P.prof[11] = n_fitted;)
    }
}

// ------------------------------------------------------------------------------------------ region growing, several waves per frame
// The latency path (plp_line_extract brings ONE frame, data/frame.cc:1146-1163): k_lsd_grow's single wave is a sequential scan over the
// seeds; here a workgroup of W waves shares a frame.  Wave 0 (MAIN) is that sequential scan and the only writer of the committed USED map
// C and of the output; waves 1.. (HELPERS) claim groups of 64 seeds ahead of it and grow their regions SPECULATIVELY: they read C, write
// their id into the owner nibble (map T) of every pixel they accept, and leave per region {every pixel they ever accepted, the pixels
// they ASSUMED used because an earlier seed's unfinished region held them, the final list, the rectangle}.  When main reaches such a
// seed it takes the result iff none of the ever-accepted pixels is committed by then and every assumed pixel is -- C only grows, and a
// helper that read C(x) = 0 where the sequential scan would see USED(x) = 1 differs from it only if it ACCEPTED x (a rejected pixel
// leaves no trace), while one that skipped x as used is right iff x is used at its turn; so this is exactly the condition under which the
// sequential scan grows the same region -- and otherwise grows the region itself.  Because main publishes a region's pixels only when the
// region is final (a refinement un-marks and regrows), C is monotone.  Whatever the helpers decide among themselves (whose claim to
// respect) only changes how much speculation is wasted.  Model with random interleavings: tests/test_spec_grow_model.py.  Results equal
// k_lsd_grow's bit for bit (tests/test_gpu_line.py runs both).
// kMwHeap (line_device.hpp): list entries per helper and group buffer (both lists + the assumed pixels of all its regions of one group); kMwBufs buffers per helper
#ifndef PLP_MW_GROUP      // tuning knobs of the several-waves path (tools/build_variant.sh; measured: profiles/r03_lsd_grow.md)
#define PLP_MW_GROUP 64
#endif
#ifndef PLP_MW_BUFS
#define PLP_MW_BUFS 2
#endif
#ifndef PLP_MW_ENTRIES
#define PLP_MW_ENTRIES 16
#endif
constexpr int kMwGroup = PLP_MW_GROUP;   // seeds per ownership unit (a helper claims a group, main walks them in order; its own loads are 64 seeds)
static_assert(PLP_MW_BUFS == kMwHeapBufs && 64 % PLP_MW_GROUP == 0 && PLP_MW_ENTRIES <= 64, "line_device.hpp sizes the helpers' lists by kMwHeapBufs");
constexpr int kMwBufs = PLP_MW_BUFS;         // group buffers per helper: groups it may have finished before main has walked through them
constexpr int kMwEntries = PLP_MW_ENTRIES;      // results per group buffer; beyond them the rest of the group is main's (16 / 32 / 64 seeds x 8 / 4 / 2 buffers x 4 / 8 / 16 entries measured)
constexpr int kMwInline = 8;         // list entries of a small region kept in the LDS entry itself (main then never touches HBM for it)
constexpr int kMwAssumed = 192;      // assumed-used pixels per attempt (a pixel is listed once per time it is looked at: up to 8 times)
struct MwResult { int n1, n2, nfinal, na; bool second, keep; float4 line; };
struct alignas(16) MwEntry { int pos, n1, n2, nfinal; uint32_t flags, off; int na; uint32_t pad1; float4 line; uint32_t inl[kMwInline]; };   // flags: 1 keep, 2 final list = second
static_assert(sizeof(MwEntry) == 80, "MwEntry layout");
struct MwLayout { int waves, ring, nw_al, n_groups_cap, lookahead, policy, prof; };   // LDS (words): C | T (4 bits per pixel) | waves x (O | ring | assumed) | control | owner bytes | entries

// control words are read by all lanes from one address: the value is wave-uniform, and said so (readfirstlane) -- the hand-scheduled
// block of region_grow wants its loop state in scalar registers, which the compiler only grants to values it can prove uniform
__device__ __forceinline__ int lds_ld(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// (the lists in HBM go from one wave of the workgroup to another: WORKGROUP scope -- the waves share the CU's vector cache.  Until round 5 the
// loads, the stores and the release fence of the hand-over had agent scope: cache-bypassing loads and a write-back of the L2 (buffer_wbl2) per
// finished region, for readers that do not exist)
// REQUIREMENT: workgroup scope is enough only while the waves of a workgroup share one CU's vector L1 -- not in threadgroup-split mode.  The build pins
// -mno-tgsplit (csrc/Makefile; checked by tests/test_kernel_resources.py); a tgsplit build must define PLP_MW_AGENT_SCOPE.
#ifdef PLP_MW_AGENT_SCOPE      // diagnostic build only (tools/build_variant.sh): the scope of rounds 3 / 4
#define PLP_MW_SCOPE __HIP_MEMORY_SCOPE_AGENT
#define PLP_MW_SCOPE_NAME "agent"
#else
#define PLP_MW_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#define PLP_MW_SCOPE_NAME "workgroup"
#endif
__device__ __forceinline__ uint32_t heap_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PLP_MW_SCOPE); }
__device__ __forceinline__ int pix_of(uint32_t c, int sw) { return (int)(c >> 16) * sw + (int)(c & 0xffff); }

// give back the claim nibbles of list[0..n) that still carry `id` (another wave may have taken a pixel over)
__device__ __forceinline__ void mw_release(const GrowCtx& g, const uint32_t* list, int n, int id) {
    for (int j = g.lane; j < n; j += 64) tent_release(g, pix_of(heap_ld(list + j), g.sw), id);
}
// the attempt is over: its pixels leave this wave's own map (and, release_id != 0, its claims are given back)
__device__ __forceinline__ void mw_forget(const GrowCtx& g, const uint32_t* list, int n, int release_id) {
    for (int j = g.lane; j < n; j += 64) {
        const int p = pix_of(heap_ld(list + j), g.sw);
        atomicAnd(&g.used[p >> 5], ~(1u << (p & 31)));
        if (release_id) tent_release(g, p, release_id);
    }
}

// One seed through region_grow -> rectangle -> refinement (k_lsd_grow's per-seed body) with TWO lists: the refinement's regrowth is
// written behind the first growth's list instead of over it, and reduce_region_radius swaps a removed point with the last one instead
// of overwriting it, so that list 1 [0, n1) and list 2 [n1, n1 + n2) together hold every pixel this attempt ever accepted.
// Returns false when the speculation gave up (r.n1 / r.n2 = entries marked so far in either list).
__device__ bool mw_process_seed(const GrowCtx& g, const LsdParams& lp, int seed, float seed_deg, float2 seed_cs, MwResult& r) {
    const int lane = g.lane;
    double reg_angle, cen[3];
    r.n2 = 0; r.second = false; r.keep = false;
    if (lane == 0) g.assumed[g.assumed_cap] = 0;   // fill count of the assumed-used list
    __builtin_amdgcn_wave_barrier();
    auto n_assumed = [&]() -> int { return __builtin_amdgcn_readfirstlane((int)g.assumed[g.assumed_cap]); };
    int nreg = region_grow<true>(g, seed, true, seed_deg, &seed_cs, lp.prec, lp.c_pass, lp.c_fail, reg_angle);
    r.na = n_assumed();
    if (nreg < 0) { r.n1 = -1 - nreg; r.nfinal = 0; return false; }
    r.n1 = r.nfinal = nreg;
    if (nreg < lp.min_reg_size) return true;
    Rect rec;
    const int ring_cap = g.ring_mask + 1 >= 256 ? 64 * kRingPts : 0;
    if (nreg <= ring_cap) rect_from_ring(g, nreg, reg_angle, lp.prec, rec);
    else {
        region_list_fence();
        centroid_sums(g, nreg, cen);
        region2rect(g, nreg, reg_angle, lp.prec, cen, rec);
    }
    bool keep = true;
    if (lp.refine > 0) {
        double density = rect_density(nreg, rec);
        if (density < lp.density_th) {
            region_list_fence();
            const uint32_t c0 = g.reg[0];
            const double xc = (double)(int)(c0 & 0xffff), yc = (double)(int)(c0 >> 16);
            const double ang_c = pix_ang(g.pix[pix_of(c0, g.sw)]);
            double sum = 0, s_sum = 0;
            int nn = 0;
            for (int rb = 0; rb < nreg; rb += 64) {
                const int j = rb + lane;
                double a = 0;
                bool near = false;
                if (j < nreg) {
                    const uint32_t c = g.reg[j];
                    const int px = (int)(c & 0xffff), py = (int)(c >> 16);
                    mw_unmark(g, py * g.sw + px);   // *(reg[i].used) = NOTUSED
                    const double ddx = (double)px - xc, ddy = (double)py - yc;
                    near = sqrt(ddx * ddx + ddy * ddy) < rec.width;
                    a = pix_ang(g.pix[py * g.sw + px]);
                }
                const double my_d = near ? angle_diff_signed(a, ang_c) : 0.0;
                const double my_d2 = my_d * my_d;
                unsigned long long nb = __ballot(near);
                while (nb) {
                    const int u = __ffsll((long long)nb) - 1;
                    nb &= nb - 1;
                    sum += bcast_d(my_d, u); s_sum += bcast_d(my_d2, u); ++nn;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const double mean_angle = sum / (double)nn;
            const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)nn + mean_angle * mean_angle);
            float cp = 2.f, cf = -2.f;
            if (tau >= kLsdBandMinPrec && tau < kLsdBandMaxPrec) { cp = (float)cos(tau - kLsdAngleBand); cf = (float)cos(tau + kLsdAngleBand); }
            GrowCtx g2 = g;
            g2.reg = g.reg + r.n1; g2.reg_cap = g.reg_cap - r.n1;
            r.second = true;
            nreg = region_grow<true>(g2, pix_of(c0, g.sw), false, 0.f, nullptr, tau, cp, cf, reg_angle);
            r.na = n_assumed();
            if (nreg < 0) { r.n2 = -1 - nreg; r.nfinal = 0; return false; }
            r.n2 = r.nfinal = nreg;
            region_list_fence();
            if (nreg < 2) keep = false;
            else {
                if (nreg <= ring_cap) rect_from_ring(g2, nreg, reg_angle, lp.prec, rec);
                else {
                    centroid_sums(g2, nreg, cen);
                    region2rect(g2, nreg, reg_angle, lp.prec, cen, rec);
                }
                density = rect_density(nreg, rec);
                if (density < lp.density_th) {
                    const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc);
                    const double r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
                    double radSq = r1 > r2 ? r1 : r2;
                    while (density < lp.density_th) {
                        radSq *= 0.75 * 0.75;
                        nreg = reduce_radius_pass<true>(g2, nreg, xc, yc, radSq);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        if (nreg < 2) { keep = false; break; }
                        centroid_sums(g2, nreg, cen);   // (the fit from the reordered list that k_lsd_grow takes here costs this kernel ten registers: 182)
                        region2rect(g2, nreg, reg_angle, lp.prec, cen, rec);
                        density = rect_density(nreg, rec);
                    }
                    r.nfinal = nreg;
                }
            }
        }
    }
    if (!keep) return true;
    rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
    if (lp.scale != 1) { rec.x1 /= lp.scale; rec.y1 /= lp.scale; rec.x2 /= lp.scale; rec.y2 /= lp.scale; }
    r.keep = true;
    r.line = make_float4((float)rec.x1, (float)rec.y1, (float)rec.x2, (float)rec.y2);
    return true;
}

// grid = B, block = 64 * L.waves, dynamic LDS per MwLayout (host: launch_line_front).
__global__ __launch_bounds__(64 * kMwMaxWaves) void k_lsd_grow_mw(LinePlanes P, LsdParams lp, MwLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mw[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.x;
    const int n = P.sw * P.sh, nwords = (n + 31) / 32, nv = (P.sw - 1) * (P.sh - 1);
    const int W = L.waves;
    uint32_t* C = s_mw;
    uint32_t* T = C + L.nw_al;
    uint32_t* O = T + 4 * L.nw_al + (size_t)wv * (L.nw_al + L.ring + kMwAssumed + 2);   // this wave's own marks (private: a claim nibble can be overwritten)
    uint32_t* my_ring = O + L.nw_al;
    int* ctrl = reinterpret_cast<int*>(T + 4 * L.nw_al + (size_t)W * (L.nw_al + L.ring + kMwAssumed + 2));
    // control words: 0 next_group, 1 main_group, 2 done, 3 abort (watchdog), then per-wave arrays
    int* next_group = ctrl; int* main_group = ctrl + 1; int* done = ctrl + 2; int* wd_abort = ctrl + 3;
    int* hstate = ctrl + 4; int* buf_group = hstate + kMwMaxWaves; int* buf_n = buf_group + kMwBufs * kMwMaxWaves;
    int* hcount = buf_n + kMwBufs * kMwMaxWaves;                                                       // diagnostics: helper attempts, give-ups
    int* cur_pos = hcount + 4;                                                                      // [helpers]: seed position of each helper's latest attempt
    int* pend_lo = cur_pos + kMwMaxWaves; int* pend_hi = pend_lo + kMwMaxWaves;                      // [helpers]: lowest / highest seed position among its finished, waiting regions
    uint8_t* owner = reinterpret_cast<uint8_t*>(pend_hi + kMwMaxWaves);                          // [n_groups_cap]: 0 unpublished, 1 main, 2 + (h * kMwBufs + k)
    // [helpers][kMwBufs][kMwEntries], 16-byte aligned.  The rounding is done on the OFFSET from the (16-byte aligned) LDS base, not on the address as an integer: a pointer
    // made from an integer has no address space the compiler knows, and the hand-over records were reached through FLAT instructions until round 6 -- the one way of
    // reaching LDS that a seed-sort build failed with inside the overlapped step while the same layout through DS instructions never did (profiles/r06_seed_sort.md)
    unsigned char* const mw_base = reinterpret_cast<unsigned char*>(s_mw);
    MwEntry* entries = reinterpret_cast<MwEntry*>(mw_base + (((size_t)((owner + ((L.n_groups_cap + 15) & ~15)) - mw_base) + 15) & ~(size_t)15));
    const int n_ord = P.n_order[b];
    const int n_groups = (n_ord + kMwGroup - 1) / kMwGroup;   // ownership units: kMwGroup consecutive seeds
    {   // C = NOTDEF mask (an undefined pixel is never a seed and never aligned), no pixel held by anybody
        const uint32_t* u32 = reinterpret_cast<const uint32_t*>(P.undef + (size_t)b * ((n + 63) / 64));
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) C[i] = u32[i];
        for (int i = threadIdx.x; i < 4 * L.nw_al; i += blockDim.x) T[i] = 0;
        for (int i = lane; i < L.nw_al; i += 64) O[i] = 0;
        for (int i = threadIdx.x; i < L.n_groups_cap; i += blockDim.x) owner[i] = 0;
        constexpr int kCtrl = 8 + (4 + 2 * kMwBufs) * kMwMaxWaves;
        for (int i = threadIdx.x; i < kCtrl; i += blockDim.x) {   // buf_group = -1 (free); no finished region waits: lowest = "infinity", highest = -1
            const int* pi = ctrl + i;
            ctrl[i] = (pi >= buf_group && pi < buf_n) ? -1 : (pi >= pend_lo && pi < pend_hi) ? 0x7fffffff : (pi >= pend_hi) ? -1 : 0;
        }
    }
    wg_barrier();
    const bool is_main = wv == 0;
    const int h = wv - 1;
    GrowCtx g;
    g.pix = P.pix + (size_t)b * n; g.used = O; g.ring = my_ring; g.ring_mask = L.ring - 1;
    g.sw = P.sw; g.sh = P.sh; g.lane = lane; g.comm = C; g.tent = T;
    g.tent_pos = cur_pos; g.tent_id = is_main ? kMwMainId : wv; g.my_pos = 0; g.policy = L.policy;
    g.assumed = my_ring + L.ring; g.assumed_cap = kMwAssumed;
    uint32_t* const my_heap = is_main ? P.reg + (size_t)b * P.reg_frame_stride : P.mw_heap + (size_t)b * P.mw_heap_frame_stride + (size_t)h * kMwBufs * kMwHeap;
    g.reg = my_heap; g.reg_cap = is_main ? 2 * n : kMwHeap;
    const uint32_t* order = P.order + (size_t)b * nv;
    float4* raw = P.raw + (size_t)b * kLineCap;
    auto committed = [&](int p) -> bool { return (C[p >> 5] >> (p & 31)) & 1u; };
    // a spin that lasts longer than any legitimate wait (tens of milliseconds) is a protocol error: every wave leaves, the batch reports it
    // the main wave's own clocks (plp_line_set_profiling; tools/experiments/latency_profile.py): a clock read is a scalar-memory operation the wave waits
    // for, and five of them around every seed the main wave deals with were taken whether anybody asked or not -- now only when profiling is on
    const bool prof_on = L.prof != 0;
    auto now = [&]() -> long long { return prof_on ? (long long)clock64() : 0ll; };
    long long wd_t0 = 0;
    auto spin = [&]() -> bool {
        __builtin_amdgcn_s_sleep(2);
        if (wd_t0 == 0) wd_t0 = (long long)clock64();
        if ((long long)clock64() - wd_t0 > 400000000ll) { lds_st(wd_abort, 1); if (lane == 0) atomicOr(P.status, 16); }
        return lds_ld(wd_abort) == 0;
    };
    // state of the group this wave works on (main: `mine` holds 64 seeds = four groups, `sub` is the one it is in; helpers: lanes 0..kMwGroup-1)
    constexpr int kSubs = 64 / kMwGroup;
    constexpr unsigned long long kGroupMask = kMwGroup == 64 ? ~0ull : ((1ull << (kMwGroup & 63)) - 1ull);
    int grp = -1, own = 0, kbuf = 0, hoff = 0, nent = 0, n_lines = 0, ld = -1, sub = kSubs - 1;
    int n_self = 0, n_spec_ok = 0, n_spec_bad = 0;           // main: regions it grew itself, results it took / had to reject
    long long c_wait = 0, c_self = 0, c_commit = 0, c_pub = 0, c_grp = 0;   // main: cycles waiting for helpers / growing regions itself / taking results / publishing its own / group set-up
    const long long c_begin = now();
    uint32_t mine = 0, mine_next = 0; float s_deg = 0.f; float2 s_cs = make_float2(0.f, 0.f);
    unsigned long long todo = 0;      // helpers: the group's seeds still to look at; main: those of its 64-seed load
    unsigned long long bits = 0;      // main: those of the current group
    bool have_cs = false;
    // main's view of the helper that owns the current group: progress inside the group, entries announced and their headers (lane i: entry i)
    int prog = 0, ne = 0, pos_vec = -1, hv_n1 = 0, hv_n2 = 0, hv_nf = 0, hv_na = 0, hv_fl = 0, hv_off = 0;
    const MwEntry* eb = entries;
    auto refresh = [&](int hh, int hk) {   // hk = helper * kMwBufs + buffer
        const int st = lds_ld(&hstate[hh]);
        prog = (st >> 8) > grp ? kMwGroup : ((st >> 8) == grp ? (st & 255) : 0);
        ne = lds_ld(&buf_n[hk]);
        eb = entries + (size_t)hk * kMwEntries;
        const MwEntry* e = eb + min(lane, kMwEntries - 1);
        pos_vec = lane < ne ? e->pos : -1;
        hv_n1 = e->n1; hv_n2 = e->n2; hv_nf = e->nfinal; hv_na = e->na; hv_fl = (int)e->flags; hv_off = (int)e->off;   // one round trip for all of them
    };
    if (is_main) mine_next = lane < n_ord ? order[lane] : 0u;
    for (;;) {
        int t = -1;
        wd_t0 = 0;
        if (is_main) {
            for (;;) {   // (the watchdog's flag is looked at where a wave waits: spin())
                if (!bits) {   // on to the next group that still has a seed to look at
                    if (++sub >= kSubs) {
                        if (++ld * 64 >= n_ord) break;
                        mine = mine_next;
                        mine_next = (ld + 1) * 64 + lane < n_ord ? order[(ld + 1) * 64 + lane] : 0u;   // the next load's seeds are on their way while this one is dealt with
                        todo = __ballot(ld * 64 + lane < n_ord && !committed((int)mine));
                        sub = 0;
                    }
                    bits = (todo >> (kMwGroup * sub)) & kGroupMask;
                    if (!bits) continue;
                    const long long cg0 = now();
                    grp = ld * kSubs + sub;
                    lds_st(main_group, grp);
                    // whose group is it?  unclaimed (nobody's fetch-and-add has reached it): take it (and every skipped one before it); else wait
                    // until its helper has said so
                    own = 0;
                    while (!own) {
                        own = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&owner[grp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (own) break;
                        int nu = lds_ld(next_group);
                        if (nu <= grp) {
                            int okc = 0;
                            if (lane == 0) okc = __hip_atomic_compare_exchange_strong(next_group, &nu, grp + 1, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
                            if (__builtin_amdgcn_readfirstlane(okc)) {
                                if (lane == 0) __hip_atomic_store(&owner[grp], (uint8_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                own = 1; break;
                            }
                            continue;
                        }
                        if (!spin()) break;
                    }
                    if (!own) break;
                    wd_t0 = 0;
                    have_cs = false;   // main grows few seeds itself: their angles are fetched when that happens
                    if (own > 1) refresh((own - 2) / kMwBufs, own - 2);
                    c_grp += now() - cg0;
                    continue;
                }
                const int tt = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const int seed = bcast_i((int)mine, kMwGroup * sub + tt);
                if (committed(seed)) continue;
                if (own > 1) {
                    const int hk = own - 2, hh = hk / kMwBufs;
                    if (prog <= tt) {   // until the helper has dealt with position tt
                        bool alive = true;
                        const long long cw0 = now();
                        for (;;) {
                            refresh(hh, hk);
                            if (prog > tt) break;
                            if (!(alive = spin())) break;
                        }
                        if (!alive) break;
                        wd_t0 = 0;
                        c_wait += now() - cw0;
                    }
                    const unsigned long long hit = __ballot(pos_vec == tt);
                    const long long cc0 = now();
                    if (hit) {
                        const int ei = __ffsll((long long)hit) - 1;
                        const MwEntry* e = eb + ei;
                        const int n1 = bcast_i(hv_n1, ei), n2 = bcast_i(hv_n2, ei), nf = bcast_i(hv_nf, ei), na = bcast_i(hv_na, ei), acc_n = n1 + n2, tot = acc_n + na;
                        const uint32_t fl = (uint32_t)bcast_i(hv_fl, ei);
                        const int fb = (fl & 2u) ? n1 : 0;
                        bool bad = false;   // an accepted pixel that is committed by now, or an assumed-used one that is not
                        if (tot <= kMwInline) {   // a small region: its pixels are in the entry, one LDS round trip each for the list and for the committed bits
                            int p = 0;
                            if (lane < tot) { p = pix_of(e->inl[lane], g.sw); bad = committed(p) != (lane >= acc_n); }
                            if (!__ballot(bad)) { if (lane >= fb && lane < fb + nf) atomicOr(&C[p >> 5], 1u << (p & 31)); }
                            else bad = true;
                        } else {
                            const uint32_t* hl = P.mw_heap + (size_t)b * P.mw_heap_frame_stride + (size_t)hk * kMwHeap + bcast_i(hv_off, ei);
                            for (int j0 = 0; j0 < tot; j0 += 64) {
                                const int j = j0 + lane;
                                if (j < tot) bad |= committed(pix_of(heap_ld(hl + j), g.sw)) != (j >= acc_n);
                            }
                            if (!__ballot(bad)) {
                                for (int j0 = 0; j0 < nf; j0 += 64) {
                                    const int j = j0 + lane;
                                    if (j < nf) { const int p = pix_of(heap_ld(hl + fb + j), g.sw); atomicOr(&C[p >> 5], 1u << (p & 31)); }
                                }
                            }
                        }
                        if (!__ballot(bad)) {   // the sequential scan grows exactly this region here: it is committed
                            if (fl & 1u) {
                                if (n_lines < kLineCap) { if (lane == 0) raw[n_lines] = e->line; }
                                else if (lane == 0) atomicOr(P.status, 4);
                                ++n_lines;
                            }
                            __builtin_amdgcn_wave_barrier();
                            ++n_spec_ok;
                            if (nf > 1) {   // the seeds this region covers leave the list together
                                todo &= __ballot(!committed((int)mine));
                                bits &= (todo >> (kMwGroup * sub)) & kGroupMask;
                            }
                            c_commit += now() - cc0;
                            continue;
                        }
                        ++n_spec_bad;
                        c_commit += now() - cc0;
                    }
                }
                t = kMwGroup * sub + tt;
                break;
            }
        } else {
            for (;;) {
                if (lds_ld(done) || lds_ld(wd_abort)) break;
                if (grp >= 0 && todo) {
                    const int tt = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int seed = bcast_i((int)mine, tt);
                    // committed, or held by an earlier seed's unfinished region (mine included): it will most likely be swallowed -- skip;
                    // a LATER seed's claim does not count
                    const int ow = __builtin_amdgcn_readfirstlane(tent_owner(g, seed));
                    const int owh = ow > kMwPending ? ow - kMwPending : ow;   // the helper behind the claim (8 for main: never read)
                    if (committed(seed) || (ow != 0 && (ow == kMwMainId || owh == wv || lds_ld(&cur_pos[owh - 1]) < grp * kMwGroup + tt))) {
                        lds_st(&hstate[h], (grp << 8) | (tt + 1));
                        continue;
                    }
                    if (nent >= kMwEntries || kMwHeap - hoff < 512 + kMwAssumed) { todo = 0; continue; }   // out of room: main does the rest of the group
                    t = tt;
                    break;
                }
                if (grp >= 0) {   // group finished
                    lds_st(&hstate[h], (grp << 8) | kMwGroup);
                    grp = -1;
                }
                // recycle: a buffer whose group main has left gives its claims back
                const int mg = lds_ld(main_group);
                int free_k = -1, lo = 0x7fffffff;
                for (int k = 0; k < kMwBufs; ++k) {
                    const int bg = lds_ld(&buf_group[h * kMwBufs + k]);
                    if (bg >= 0 && bg < mg) {
                        const int nek = lds_ld(&buf_n[h * kMwBufs + k]);
                        const MwEntry* ek = entries + (size_t)(h * kMwBufs + k) * kMwEntries;
                        for (int i = 0; i < nek; ++i) mw_release(g, my_heap + (size_t)k * kMwHeap + ek[i].off, ek[i].n1 + ek[i].n2, wv + kMwPending);
                        __builtin_amdgcn_wave_barrier();
                        lds_st(&buf_n[h * kMwBufs + k], 0);
                        lds_st(&buf_group[h * kMwBufs + k], -1);
                        free_k = k;
                    } else if (bg < 0) free_k = k;
                    else if (lds_ld(&buf_n[h * kMwBufs + k]) > 0) lo = min(lo, bg * kMwGroup + entries[(size_t)(h * kMwBufs + k) * kMwEntries].pos);
                }
                if (lo != lds_ld(&pend_lo[h])) { lds_st(&pend_lo[h], lo); if (lo == 0x7fffffff) lds_st(&pend_hi[h], -1); }   // what still waits for its turn
                const int ng = lds_ld(next_group);
                if (ng >= n_groups) { if (!spin()) break; wd_t0 = 0; continue; }          // nothing left to claim: wait for `done`
                if (free_k < 0 || ng >= mg + L.lookahead) { if (!spin()) break; wd_t0 = 0; continue; }
                int gi = 0;
                if (lane == 0) gi = __hip_atomic_fetch_add(next_group, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                gi = __builtin_amdgcn_readfirstlane(gi);
                if (gi >= n_groups || gi < lds_ld(main_group)) continue;   // past the end, or main has gone past it already (it skips groups without an open seed)
                grp = gi; kbuf = free_k; nent = 0; hoff = 0;
                lds_st(&hstate[h], grp << 8);
                lds_st(&buf_n[h * kMwBufs + kbuf], 0);
                lds_st(&buf_group[h * kMwBufs + kbuf], grp);
                if (lane == 0) __hip_atomic_store(&owner[grp], (uint8_t)(2 + h * kMwBufs + kbuf), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                {   // the group's seeds and the angle records of those not yet committed, all at once
                    const bool in_range = lane < kMwGroup && grp * kMwGroup + lane < n_ord;
                    mine = in_range ? order[grp * kMwGroup + lane] : 0u;
                    const bool fresh = in_range && !committed((int)mine);
                    todo = __ballot(fresh);
                    have_cs = true;
                    s_deg = fresh ? g.pix[mine].deg : 0.f;
                    if (fresh) { double sn, cs; sincos((double)s_deg * (3.14159265358979323846 / 180), &sn, &cs); s_cs = make_float2((float)cs, (float)sn); }
                }
            }
        }
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < 0) break;
        // ---- one region (the only call site of the per-seed code)
        const int seed = bcast_i((int)mine, t);
        if (!is_main) {
            g.reg = my_heap + (size_t)kbuf * kMwHeap + hoff; g.reg_cap = kMwHeap - hoff - kMwAssumed;
            g.my_pos = grp * kMwGroup + t;
            lds_st(&cur_pos[h], g.my_pos);
        }
        MwResult r;
        const long long cs0 = now();
        float seed_deg; float2 seed_cs;
        if (have_cs) { seed_deg = bcast_f(s_deg, t); seed_cs = make_float2(bcast_f(s_cs.x, t), bcast_f(s_cs.y, t)); }
        else {
            seed_deg = g.pix[seed].deg;
            double sn, cs;
            sincos((double)seed_deg * (3.14159265358979323846 / 180), &sn, &cs);
            seed_cs = make_float2((float)cs, (float)sn);
        }
        const bool ok = mw_process_seed(g, lp, seed, seed_deg, seed_cs, r);
        region_list_fence();
        if (is_main) { c_self += now() - cs0; ++n_self; }
        else if (lane == 0) { atomicAdd(&hcount[0], 1); if (!ok) atomicAdd(&hcount[1], 1); }
        const int acc_n = r.n1 + r.n2;
        if (is_main) {
            const long long cp0 = now();
            const int fb = r.second ? r.n1 : 0;
            for (int j = lane; j < r.nfinal; j += 64) { const int p = pix_of(heap_ld(g.reg + fb + j), g.sw); atomicOr(&C[p >> 5], 1u << (p & 31)); }
            mw_forget(g, g.reg, acc_n, kMwMainId);
            todo &= __ballot(!committed((int)mine));
            bits &= (todo >> (kMwGroup * sub)) & kGroupMask;
            c_pub += now() - cp0;
            if (r.keep) {
                if (n_lines < kLineCap) { if (lane == 0) raw[n_lines] = r.line; }
                else if (lane == 0) atomicOr(P.status, 4);
                ++n_lines;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            if (ok) {
                const int tot = acc_n + r.na;
                // the region is finished: its pixels change from "helper h is growing this" to "a finished region of helper h" -- the next
                // attempts of this helper treat them like anybody else's finished region (assumed used, and checked at their turn)
                for (int j = lane; j < r.nfinal; j += 64) tent_retag(g, pix_of(heap_ld(g.reg + (r.second ? r.n1 : 0) + j), g.sw), wv, wv + kMwPending);
                for (int j = lane; j < r.na; j += 64) __hip_atomic_store(g.reg + acc_n + j, g.assumed[j], __ATOMIC_RELAXED, PLP_MW_SCOPE);   // the assumed-used pixels follow the two lists
                MwEntry* e = entries + (size_t)(h * kMwBufs + kbuf) * kMwEntries + nent;
                if (lane == 0) {
                    e->pos = t; e->n1 = r.n1; e->n2 = r.n2; e->nfinal = r.nfinal; e->flags = (r.keep ? 1u : 0u) | (r.second ? 2u : 0u); e->off = (uint32_t)hoff;
                    e->na = r.na; e->line = r.line;
                }
                if (tot <= kMwInline) {
                    if (lane < acc_n) e->inl[lane] = heap_ld(g.reg + lane);
                    else if (lane < tot) e->inl[lane] = g.assumed[lane - acc_n];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, PLP_MW_SCOPE_NAME);   // the lists in HBM are complete before the entry is announced
                __builtin_amdgcn_wave_barrier();
                ++nent; hoff += (tot + 1) & ~1;
                if (lds_ld(&pend_hi[h]) < 0) lds_st(&pend_lo[h], g.my_pos);
                lds_st(&pend_hi[h], g.my_pos);
                lds_st(&buf_n[h * kMwBufs + kbuf], nent);
                mw_forget(g, g.reg, acc_n, 0);
            } else mw_forget(g, g.reg, acc_n, wv);
            __builtin_amdgcn_wave_barrier();
            {   // the seeds of this group that the new region (or anybody's) covers or claims leave the list together (same rule as one by one above)
                const int p = (int)mine;
                const int ow = tent_owner(g, p), owh = ow > kMwPending ? ow - kMwPending : ow;
                const bool skip = committed(p) || (ow != 0 && (ow == kMwMainId || owh == wv || cur_pos[min(owh, kMwMaxWaves) - 1] < grp * kMwGroup + lane));
                todo &= __ballot(!skip);
            }
            lds_st(&hstate[h], (grp << 8) | (t + 1));
        }
    }
    if (is_main) {
        if (lane == 0) {
            P.n_raw[b] = min(n_lines, kLineCap);
            int32_t* gs = P.grow_stats + (size_t)b * 4; gs[0] = n_self; gs[1] = n_spec_ok; gs[2] = n_spec_bad; gs[3] = W;
            if (P.prof && b == 0 && prof_on) {   // diagnostics of frame 0: cycles {total, waiting for helpers, growing itself}, helper attempts | give-ups << 32, regions grown by main, results taken
                P.prof[0] = now() - c_begin; P.prof[1] = c_wait; P.prof[2] = c_self;
                P.prof[3] = (long long)hcount[0] | ((long long)hcount[1] << 32); P.prof[4] = n_self; P.prof[5] = (long long)n_spec_ok | ((long long)n_spec_bad << 32);
                P.prof[6] = c_commit; P.prof[7] = c_pub; P.prof[8] = c_grp; P.prof[9] = P.prof[10] = P.prof[11] = 0;
            }
        }
        lds_st(done, 1);
    }
}

// ------------------------------------------------------------------------------------------ KeyLine assembly
// LSDDetector_custom.cpp:262-303 (octave 0).  One wave per frame; class_id = running index of kept lines.
__global__ __launch_bounds__(64) void k_keylines(LinePlanes P, LsdParams lp) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = P.n_raw[b];
    const float4* raw = P.raw + (size_t)b * kLineCap;
    plp_keyline* out = P.all_kl + (size_t)b * kLineCap;
    int run = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool keep = false;
        plp_keyline kl{};
        if (i < n) {
            float4 e = raw[i];
            const float fw = (float)P.W, fh = (float)P.H;
            if (e.x < 0) e.x = 0;
            if (e.x >= fw) e.x = fw - 1.0f;
            if (e.z < 0) e.z = 0;
            if (e.z >= fw) e.z = fw - 1.0f;
            if (e.y < 0) e.y = 0;
            if (e.y >= fh) e.y = fh - 1.0f;
            if (e.w < 0) e.w = 0;
            if (e.w >= fh) e.w = fh - 1.0f;
            const double ddx = (double)__fsub_rn(e.x, e.z), ddy = (double)__fsub_rn(e.y, e.w);
            const double length = (double)(float)sqrt(ddx * ddx + ddy * ddy);
            keep = length > (double)lp.min_length;
            kl.startPointX = e.x; kl.startPointY = e.y; kl.endPointX = e.z; kl.endPointY = e.w;
            kl.sPointInOctaveX = e.x; kl.sPointInOctaveY = e.y; kl.ePointInOctaveX = e.z; kl.ePointInOctaveY = e.w;
            kl.lineLength = (float)length;
            const int x1 = __float2int_rn(e.x), y1 = __float2int_rn(e.y), x2 = __float2int_rn(e.z), y2 = __float2int_rn(e.w);
            kl.numOfPixels = max(abs(x2 - x1), abs(y2 - y1)) + 1;
            kl.angle = (float)atan2((double)__fsub_rn(e.w, e.y), (double)__fsub_rn(e.z, e.x));
            kl.octave = 0;
            kl.size = __fmul_rn(__fsub_rn(e.z, e.x), __fsub_rn(e.w, e.y));
            kl.response = __fdiv_rn(kl.lineLength, (float)max(P.W, P.H));
            kl.pt_x = __fdiv_rn(__fadd_rn(e.z, e.x), 2.0f);
            kl.pt_y = __fdiv_rn(__fadd_rn(e.w, e.y), 2.0f);
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            kl.class_id = run + __popcll(bal & ((1ull << lane) - 1ull));
            out[kl.class_id] = kl;
            // the band descriptor's direction (binary_descriptor_custom.cpp:1126-1127), here with one line per LANE: in k_lbd, one line per wave,
            // the f64 cos / sin (64 lanes computing one value) was that kernel's register peak
            P.all_kl_dir[(size_t)b * kLineCap + kl.class_id] = make_float2((float)cos((double)kl.angle), (float)sin((double)kl.angle));
        }
        run += __popcll(bal);
    }
    if (lane == 0) P.n_all[b] = run;
}

// ------------------------------------------------------------------------------------------ blur5 + Sobel
// cv::GaussianBlur(5x5, sigma 1) followed by cv::Sobel 3x3 (dx and dy, CV_16S, BORDER_REFLECT_101) in one pass: the blurred image
// is only ever read by the Sobel filter, so it never goes to HBM.  A workgroup blurs a 128 x 32 tile whose origin lies 4 columns
// and 1 row before its 120 x 30 block of outputs (blur_tile_core evaluates the blur on the reflected extension of the source, which
// is the reflected extension of the blurred image: the Sobel border needs nothing else), keeps the bytes in LDS and filters them:
// a work item makes 4 pixels x 2 rows from 4 rows x 3 aligned dwords.  The two derivatives of a pixel are stored side by side (one
// 4-byte gather per LBD sample instead of two 2-byte ones).  grid = (tiles, B) through xcd_frame_major, block = 256.
constexpr int kSobelTW = 120, kSobelTH = 30;
// The blurred tile (4 KB) takes the place of the staged input rows (5 KB): nobody reads those after the horizontal pass, and the barrier between
// the two passes of blur_tile_core lies before the first blurred byte is written.  14.4 instead of 18.5 KB: five workgroups instead of four in the
// 77 KB two region-growing workgroups leave on a CU.
static_assert(sizeof(BlurTileLds<2>::in) >= kBlurTH * (kBlurTW / 4) * 4, "the blurred tile must fit the input rows it replaces");
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur_sobel(const uint8_t* __restrict__ src, size_t src_fs, int src_pitch,
                                                                                            short2* __restrict__ dxy, int w, int h, BlurTapsN taps, TileDiv td) {
    __shared__ __attribute__((aligned(16))) BlurTileLds<2> Sb;
    uint32_t* const bt = reinterpret_cast<uint32_t*>(Sb.in);
    const int tiles_x = (w + kSobelTW - 1) / kSobelTW;
    unsigned t, f;
    xcd_frame_major(t, f, td.gx_magic);
    const int trow = (int)plp_div(t, (unsigned)tiles_x, td.tiles_x_magic);
    const int bx0 = ((int)t - trow * tiles_x) * kSobelTW - 4, by0 = trow * kSobelTH - 1;
    blur_tile_core<2>(Sb, src + (size_t)f * src_fs, src_pitch, w, h, bx0, by0, taps.k, [&](int r0, int c4, const uint32_t (&rows)[kBlurRS]) {
#pragma unroll
        for (int rr = 0; rr < kBlurRS; ++rr) bt[(r0 + rr) * (kBlurTW / 4) + c4 / 4] = rows[rr];
    });
    wg_barrier();
    short2* out_frame = dxy + (size_t)f * dxy_frame_entries(w, h);
    const int tiles8 = (w + 7) / 8;
    for (int i = threadIdx.x; i < (kSobelTW / 4) * (kSobelTH / 2); i += 256) {
        const int cgp = i % (kSobelTW / 4), rp = i / (kSobelTW / 4);
        const int x = bx0 + 4 + 4 * cgp, y = by0 + 1 + 2 * rp;
        if (x >= w || y >= h) continue;
        const int cd = 1 + cgp;
        int d[4][4], sm[4][4];   // per blurred row: right - left and left + 2 mid + right of the 4 pixels
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t* q = &bt[(2 * rp + r) * (kBlurTW / 4) + cd];
            const uint32_t A = q[-1], Bv = q[0], Cv = q[1];
            const int p[6] = {(int)(A >> 24), (int)(Bv & 255u), (int)((Bv >> 8) & 255u), (int)((Bv >> 16) & 255u), (int)(Bv >> 24), (int)(Cv & 255u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[r][k] = p[k + 2] - p[k]; sm[r][k] = p[k] + 2 * p[k + 1] + p[k + 2]; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (y + j >= h) continue;
            uint32_t out[4];   // (dx, dy) as two s16 in one dword
#pragma unroll
            for (int k = 0; k < 4; ++k)
                out[k] = ((uint32_t)(d[j][k] + 2 * d[j + 1][k] + d[j + 2][k]) & 0xffffu) | ((uint32_t)(sm[j + 2][k] - sm[j][k]) << 16);
            // x is a multiple of 4: the four pixels are one aligned 16-byte piece of a tile row (pixels beyond the image edge land in the tile's padding)
            *reinterpret_cast<uint4*>(out_frame + dxy_index(x, y + j, tiles8)) = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------ LBD
__constant__ int c_comb[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                                  {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                                  {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

// grid = (2, B), block = 256: one wave per line, 8 lines of a frame in flight (more waves thrash L1/L2: every wave
// keeps 63 image rows live).
#ifndef PLP_LBD_U
#define PLP_LBD_U 8
#endif
constexpr int kLbdU = PLP_LBD_U;   // band steps whose gathers are in flight together (per lane = band row)
__global__ __launch_bounds__(256) void k_lbd(LinePlanes P, LbdWeightsDev W) {
    __shared__ float s_row[4][63][8];   // per row: pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2 (after the global weight)
    __shared__ float s_des[4][72], s_des2[4][72], s_norm[4][4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), b = blockIdx.y;   // (wave-uniform, and said so: the line's record and addresses are scalar)
    const int n_lines = P.n_all[b];
    for (int li = blockIdx.x * 4 + wv; li < n_lines; li += gridDim.x * 4) {   // a few resident waves walk the frame's lines
    const plp_keyline kl = P.all_kl[(size_t)b * kLineCap + li];
    const short2* dxyImg = P.dxy + (size_t)b * dxy_frame_entries(P.W, P.H);
    const int tiles8 = (P.W + 7) / 8;
    const short imageWidth = (short)(P.W - 1), imageHeight = (short)(P.H - 1);
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2), halfHeight = 31;
    const float midX = (float)(0.5 * (double)__fadd_rn(kl.sPointInOctaveX, kl.ePointInOctaveX));
    const float midY = (float)(0.5 * (double)__fadd_rn(kl.sPointInOctaveY, kl.ePointInOctaveY));
    const float2 dir = P.all_kl_dir[(size_t)b * kLineCap + li];   // ((float)cos((double)kl.angle), (float)sin((double)kl.angle)), from k_keylines
    const float dL0 = dir.x, dL1 = dir.y;
    const float dO0 = -dL1, dO1 = dL0;
    if (lane < 63) {
        // start corner of row hID = lane: the reference steps it row by row (sCorX0 -= dL1; sCorY0 += dL0)
        float sCorX0 = __fadd_rn(__fadd_rn(__fmul_rn(-dL0, (float)halfWidth), __fmul_rn(dL1, (float)halfHeight)), midX);
        float sCorY0 = __fadd_rn(__fsub_rn(__fmul_rn(-dL1, (float)halfWidth), __fmul_rn(dL0, (float)halfHeight)), midY);
        for (int r = 0; r < lane; ++r) { sCorX0 = __fsub_rn(sCorX0, dL1); sCorY0 = __fadd_rn(sCorY0, dL0); }
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdL = 0, ngdL = 0, pgdO = 0, ngdO = 0;
        // the gathers do not depend on the running sums: addresses of 8 steps first, 16 loads in flight, then the
        // strictly ordered accumulation
        for (int w0 = 0; w0 < lengthOfLSP; w0 += kLbdU) {
            int off[kLbdU];
#pragma unroll
            for (int u = 0; u < kLbdU; ++u) {
                short t = (short)roundf(sCorX);
                const short xCor = (t < 0) ? 0 : (t > imageWidth) ? imageWidth : t;
                t = (short)roundf(sCorY);
                const short yCor = (t < 0) ? 0 : (t > imageHeight) ? imageHeight : t;
                off[u] = dxy_index(xCor, yCor, tiles8);
                sCorX = __fadd_rn(sCorX, dL0);
                sCorY = __fadd_rn(sCorY, dL1);
            }
            short2 vxy[kLbdU];
#pragma unroll
            for (int u = 0; u < kLbdU; ++u) vxy[u] = dxyImg[off[u]];
#pragma unroll
            for (int u = 0; u < kLbdU; ++u) {
                if (w0 + u < lengthOfLSP) {
                    const float fx = (float)vxy[u].x, fy = (float)vxy[u].y;
                    const float gDL = __fadd_rn(__fmul_rn(fx, dL0), __fmul_rn(fy, dL1));
                    const float gDO = __fadd_rn(__fmul_rn(fx, dO0), __fmul_rn(fy, dO1));
                    if (gDL > 0) pgdL = __fadd_rn(pgdL, gDL); else ngdL = __fsub_rn(ngdL, gDL);
                    if (gDO > 0) pgdO = __fadd_rn(pgdO, gDO); else ngdO = __fsub_rn(ngdO, gDO);
                }
            }
        }
        const float cg = W.g[lane];
        pgdL = __fmul_rn(cg, pgdL); ngdL = __fmul_rn(cg, ngdL); pgdO = __fmul_rn(cg, pgdO); ngdO = __fmul_rn(cg, ngdO);
        float* r = s_row[wv][lane];
        r[0] = pgdL; r[1] = ngdL; r[2] = __fmul_rn(pgdL, pgdL); r[3] = __fmul_rn(ngdL, ngdL);
        r[4] = pgdO; r[5] = ngdO; r[6] = __fmul_rn(pgdO, pgdO); r[7] = __fmul_rn(ngdO, ngdO);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // band sums: lane = (band, component); contributions arrive in row order hID (neighbour-below band first,
    // own rows, then neighbour-above band), exactly the order of the reference's accumulation
    for (int item = lane; item < 72; item += 64) {
        const int band = item >> 3, comp = item & 7;
        const bool sq = (comp == 2 || comp == 3 || comp == 6 || comp == 7);
        float acc = 0;
        for (int hID = max(0, (band - 1) * 7); hID < min(63, (band + 2) * 7); ++hID) {
            const int rb = hID / 7;
            const float c = rb == band ? W.l[hID % 7 + 7] : (rb == band + 1 ? W.l[hID % 7 + 14] : W.l[hID % 7]);
            const float v = s_row[wv][hID][comp];
            acc = sq ? __fadd_rn(acc, __fmul_rn(__fmul_rn(c, c), v)) : __fadd_rn(acc, __fmul_rn(c, v));
        }
        // mean / std per band (components reordered to the descriptor layout below)
        s_des[wv][item] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // mean / std per band, the two-stage normalisation and the binary string.  Element-wise work is spread over the
    // lanes; the three sums whose order matters (tempM, tempS, temp) are chained by lane 0 over LDS values.
    float* des = s_des2[wv];
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    if (lane < 36) {
        const int band = lane >> 2, k = lane & 3;
        const float invN = (band == 0 || band == 8) ? (float)(1.0 / (7 * 2.0)) : (float)(1.0 / (7 * 3.0));
        const float* sb = &s_des[wv][band * 8];   // pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2
        const int mi = (k < 2) ? k : k + 2, qi = mi + 2;   // mean from 0,1,4,5; second moment from 2,3,6,7
        const float temp = __fmul_rn(sb[mi], invN);
        des[band * 8 + k] = temp;
        des[band * 8 + 4 + k] = sqrtf(__fsub_rn(__fmul_rn(sb[qi], invN), __fmul_rn(temp, temp)));
    }
    wave_sync();
    if (lane == 0) {
        float tempM = 0, tempS = 0;
        for (int band = 0; band < 9; ++band) {
            const float* v = des + 8 * band;
            for (int k = 0; k < 4; ++k) tempM = __fadd_rn(tempM, __fmul_rn(v[k], v[k]));
            for (int k = 4; k < 8; ++k) tempS = __fadd_rn(tempS, __fmul_rn(v[k], v[k]));
        }
        s_norm[wv][0] = __fdiv_rn(1.0f, sqrtf(tempM));
        s_norm[wv][1] = __fdiv_rn(1.0f, sqrtf(tempS));
    }
    wave_sync();
    for (int i = lane; i < 72; i += 64) {
        float v = __fmul_rn(des[i], s_norm[wv][(i >> 2) & 1]);
        if ((double)v > 0.4) v = (float)0.4;
        des[i] = v;
    }
    wave_sync();
    if (lane == 0) {
        float temp = 0;
        for (int i = 0; i < 72; ++i) temp = __fadd_rn(temp, __fmul_rn(des[i], des[i]));
        s_norm[wv][2] = __fdiv_rn(1.0f, sqrtf(temp));
    }
    wave_sync();
    for (int i = lane; i < 72; i += 64) des[i] = __fmul_rn(des[i], s_norm[wv][2]);
    wave_sync();
    if (lane < 32) {
        const float* f1 = des + 8 * c_comb[lane][0];
        const float* f2 = des + 8 * c_comb[lane][1];
        unsigned r = 0;
        for (int i = 0; i < 8; ++i) r += (f1[i] > f2[i]) ? (1u << i) : 0u;
        P.all_lbd[((size_t)b * kLineCap + li) * 32 + lane] = (uint8_t)r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    }   // line loop
}

// ------------------------------------------------------------------------------------------ finalize
// line_extractor.cc:134-159: keep octave 0 && lineLength >= 60, 2-D line function (f64).  One wave per frame.
__global__ __launch_bounds__(64) void k_line_finalize(LinePlanes P, LsdParams lp, plp_keyline* __restrict__ out_kl,
                                                      uint8_t* __restrict__ out_lbd, double* __restrict__ out_fn, int cap,
                                                      int32_t* __restrict__ out_counts) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = P.n_all[b];
    int run = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        plp_keyline kl{};
        bool keep = false;
        if (i < n) { kl = P.all_kl[(size_t)b * kLineCap + i]; keep = kl.octave == 0 && kl.lineLength >= lp.keep_length; }
        const unsigned long long bal = __ballot(keep);
        const int o = run + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep && o < cap) {
            out_kl[(size_t)b * cap + o] = kl;
            const uint4* s = reinterpret_cast<const uint4*>(P.all_lbd + ((size_t)b * kLineCap + i) * 32);
            uint4* d = reinterpret_cast<uint4*>(out_lbd + ((size_t)b * cap + o) * 32);
            d[0] = s[0]; d[1] = s[1];
            const double sx = kl.startPointX, sy = kl.startPointY, ex = kl.endPointX, ey = kl.endPointY;
            const double a = sy * 1.0 - 1.0 * ey, bb = 1.0 * ex - sx * 1.0, c = sx * ey - sy * ex;
            const double nrm = sqrt(a * a + bb * bb);
            double* f = out_fn + ((size_t)b * cap + o) * 3;
            f[0] = a / nrm; f[1] = bb / nrm; f[2] = c / nrm;
        }
        run += __popcll(bal);
    }
    if (lane == 0) {
        out_counts[b] = min(run, cap);
        if (run > cap) atomicOr(P.status, 1);
    }
}

// ------------------------------------------------------------------------------------------ launch sequence
void launch_line_front(hipStream_t st, const LinePlanes& P, const LsdParams& lp, const ResizeExactTab& rt, const BlurTapsN& t11,
                       const BlurTapsN& t5, const LbdWeightsDev& w, plp_keyline* out_kl, uint8_t* out_lbd, double* out_fn, int cap,
                       int32_t* out_counts, int B, hipEvent_t* ev, const LineSideStream* side, int grow_waves, const SeedSortBufs* seed_exact, bool mw_ok, bool grow_on_side) {
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], st); };
    // The LBD image pass (5-tap blur + Sobel) does not depend on LSD.  Unless per-stage timing is requested it is launched FIRST, on the
    // same stream: it then overlaps whatever the other streams of the caller run, and nothing has to join before k_lbd.  (A side
    // stream per context did this until round 3: with two line contexts, the ORB stream and the matcher stream that made six streams
    // on the runtime's four hardware queues, and a context's Sobel pass queued behind 8 ms of matcher kernels of an unrelated stream
    // while its k_lbd waited for it -- profiles/r03c_step_timeline.md.  PLP_LINE_SIDE_STREAM=1 brings the side stream back.)
    static const bool use_side = [] { const char* e = getenv("PLP_LINE_SIDE_STREAM"); return e && e[0] == '1'; }();
    const bool fork = side && !ev && use_side && !grow_on_side;
    const bool sobel_first = !ev && !fork;
    hipStream_t st2 = fork ? side->stream : st;
    const size_t plane_fs = (size_t)P.pitch * P.H, splane_fs = (size_t)P.spitch * P.sh;
    const int tiles = ((P.W + 127) / 128) * ((P.H + kBlurTH - 1) / kBlurTH);
    const int sobel_tiles = ((P.W + kSobelTW - 1) / kSobelTW) * ((P.H + kSobelTH - 1) / kSobelTH);
    const TileDiv blur_td = tile_div(tiles, (P.W + 127) / 128, B), sobel_td = tile_div(sobel_tiles, (P.W + kSobelTW - 1) / kSobelTW, B);
    mark(0);
    if (fork) {
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(st2, side->fork, 0);
        hipLaunchKernelGGL(k_blur_sobel, dim3(sobel_tiles, B), dim3(256), 0, st2, P.img, P.img_frame_stride, P.img_pitch, P.dxy, P.W, P.H, t5, sobel_td);
        (void)hipEventRecord(side->join, st2);
    }
    if (sobel_first) hipLaunchKernelGGL(k_blur_sobel, dim3(sobel_tiles, B), dim3(256), 0, st, P.img, P.img_frame_stride, P.img_pitch, P.dxy, P.W, P.H, t5, sobel_td);
    if (P.half_exact)
        hipLaunchKernelGGL(k_blur_half, dim3(tiles, B), dim3(256), 0, st, P.img, P.img_frame_stride, P.img_pitch, P.scaled, splane_fs, P.spitch, P.W, P.H, t11, blur_td);
    else {
        hipLaunchKernelGGL(k_blur_plane<5>, dim3(tiles, B), dim3(256), 0, st, P.img, P.img_frame_stride, P.img_pitch, P.blur11, plane_fs,
                           P.pitch, P.W, P.H, t11, blur_td);
        hipLaunchKernelGGL(k_resize_exact, dim3((P.sw + 255) / 256, (P.sh + 3) / 4, B), dim3(64, 4), 0, st, P.blur11, plane_fs, P.pitch,
                           P.scaled, splane_fs, P.spitch, P.sw, P.sh, rt);
    }
    mark(1);
    const int n = P.sw * P.sh;
    hipLaunchKernelGGL(k_lsd_gradient, dim3((n + 255) / 256, B), dim3(256), 0, st, P, lp);
    mark(2);
    if (seed_exact) launch_seed_order_exact(st, P, lp, B, seed_exact->ent, seed_exact->ws, seed_exact->ws_stride);
    else hipLaunchKernelGGL(k_lsd_order, dim3(B), dim3(256), 0, st, P, lp, (n + 255) / 256);
    mark(3);
    // per wave: USED bitmap + frontier ring (a power of two; the HBM copy of the region list backs larger frontiers).
    // ~10.6 KB of LDS per wave; at 2048 frames every SIMD carries two of these latency-bound waves.
    static const int ring = [] { const char* e = getenv("PLP_LSD_RING"); int r = e ? atoi(e) : 256; return (r >= 64 && (r & (r - 1)) == 0) ? r : 256; }();
    // four waves per workgroup = one per SIMD of the CU that takes the workgroup: with 147 registers three of these waves fit a SIMD, and 2048
    // one-wave workgroups had spread unevenly (13.2 -> 14.2 ms); profiles/r03_lsd_grow.md section 2
    static const int wpb_env = [] { const char* e = getenv("PLP_LSD_WPB"); return e ? atoi(e) : 4; }();
    const size_t per_wave = (size_t)((((n + 31) / 32 + 1) & ~1) + ring) * 4;
    const int wpb = (int)std::max<size_t>(1, std::min<size_t>(std::min(4, std::max(1, wpb_env)), 65536 / per_wave));   // (one wave of the largest admitted frame: kLsdGrowLdsBytes, above 64 KB)
    // diagnostic only (what the rest of a step costs without region growing): PLP_LSD_SKIP_GROW=k leaves the kernel out after
    // the k-th launch; the later stages then chew on the previous launch's segments
    static const int skip_after = [] { const char* e = getenv("PLP_LSD_SKIP_GROW"); return e ? atoi(e) : -1; }();
    static int n_launch = 0;
    // Few frames (plp_line_extract brings one): a workgroup of several waves per frame (k_lsd_grow_mw: one main wave + helpers that
    // speculate ahead); many frames: one wave per frame, the chip is full of independent scans anyway.
    static const int mw_max_b = [] { const char* e = getenv("PLP_LSD_MW_MAX_B"); return std::min(e ? atoi(e) : 256, kLsdMwMaxFrames); }();   // 256 = one workgroup per CU
    static const int mw_waves = [] { const char* e = getenv("PLP_LSD_MW_WAVES"); int r = e ? atoi(e) : kMwMaxWaves; return std::min(std::max(r, 0), kMwMaxWaves); }();
    static const int mw_policy = [] { const char* e = getenv("PLP_LSD_MW_POLICY"); return e ? atoi(e) : 0; }();   // how helpers treat the claims of finished regions (region_grow)
    MwLayout L{};
    size_t mw_bytes = 0;
    const int want_waves = grow_waves > 0 ? std::min(grow_waves, kMwMaxWaves) : mw_waves;          // plp_line_set_grow_waves overrides the automatic choice
    if (mw_ok && B <= (grow_waves > 1 ? kLsdMwMaxFrames : mw_max_b) && want_waves >= 2 && P.mw_heap && P.reg_frame_stride >= 2 * (size_t)n) {
        const int nw_al = (((n + 31) / 32 + 1) & ~1), groups_cap = ((P.sw - 1) * (P.sh - 1) + kMwGroup - 1) / kMwGroup + 64 / kMwGroup;
        for (int w = want_waves; w >= 2; --w) {
            const size_t bytes = (size_t)5 * nw_al * 4 + (size_t)w * (nw_al + 256 + kMwAssumed + 2) * 4 + (8 + (4 + 2 * kMwBufs) * kMwMaxWaves) * 4 + ((groups_cap + 15) & ~15) + 16 +
                                 (size_t)(w - 1) * kMwBufs * kMwEntries * sizeof(MwEntry);
            if (bytes <= 160 * 1024) { L.waves = w; L.ring = 256; L.nw_al = nw_al; L.n_groups_cap = groups_cap; L.lookahead = kMwBufs * (w - 1); L.policy = mw_policy; L.prof = ev ? 1 : 0; mw_bytes = bytes; break; }
        }
    }
    hipStream_t st_main = st;
    const bool grow_fork = grow_on_side && side && L.waves < 2;
    if (grow_fork) { (void)hipEventRecord(side->fork, st); (void)hipStreamWaitEvent(side->stream, side->fork, 0); st = side->stream; }
    if (skip_after < 0 || n_launch++ < skip_after) {
        if (L.waves >= 2) {
            hipLaunchKernelGGL(k_lsd_grow_mw, dim3(B), dim3(64 * L.waves), mw_bytes, st, P, lp, L);
        } else
            hipLaunchKernelGGL(k_lsd_grow, dim3((B + wpb - 1) / wpb), dim3(64 * wpb), per_wave * wpb, st, P, lp, B, wpb, ring);
    }
    if (grow_fork) { (void)hipEventRecord(side->join, side->stream); st = st_main; (void)hipStreamWaitEvent(st, side->join, 0); }
    mark(4);
    hipLaunchKernelGGL(k_keylines, dim3(B), dim3(64), 0, st, P, lp);
    mark(5);
    if (fork) (void)hipStreamWaitEvent(st, side->join, 0);
    else if (!sobel_first) {
        hipLaunchKernelGGL(k_blur_sobel, dim3(sobel_tiles, B), dim3(256), 0, st, P.img, P.img_frame_stride, P.img_pitch, P.dxy, P.W, P.H, t5, sobel_td);
    }
    mark(6);
    // few resident waves per frame: their 63-row working sets have to stay in L1 / L2.  Workgroups of four waves per frame, kernel alone / step: 1: 1.41 ms /
    // 87.4 k frames/s, 2: 1.37 / 88.0 k, 3: 1.30 / 86.7 k, 4: 1.40 / 87.5 k, 8: 1.64 / 86.8 k, 16: 1.78 / 86.8 k (sessions 35, 36; the step numbers are within
    // their run-to-run spread of each other below 8)
    static const int lbd_blocks_env = [] { const char* e = getenv("PLP_LBD_BLOCKS"); int r = e ? atoi(e) : 0; return r > 0 ? r : 0; }();
    const int lbd_blocks = lbd_blocks_env ? lbd_blocks_env : (B >= 64 ? 2 : 16);   // a single frame (plp_line_extract): the chip is empty, one wave per line
    hipLaunchKernelGGL(k_lbd, dim3(lbd_blocks, B), dim3(256), 0, st, P, w);
    mark(7);
    hipLaunchKernelGGL(k_line_finalize, dim3(B), dim3(64), 0, st, P, lp, out_kl, out_lbd, out_fn, cap, out_counts);
    mark(8);
}

// The several-waves-per-frame kernel takes up to 160 KB of dynamic LDS: the limit belongs to the function ON THE CURRENT DEVICE, so every context
// raises it for its own device when it is created (plp_line_create); a context whose device refuses runs one wave per frame.
hipError_t grow_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_grow), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLsdGrowLdsBytes);
}
hipError_t grow_mw_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_grow_mw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
}

}  // namespace plp
