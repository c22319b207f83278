// Hand-written HIP kernels (gfx950 / CDNA4, wave64) of the batched ORB front-end.
// Every kernel carries a leading frame dimension (blockIdx.y or .z) so one launch covers
// the whole batch; integer stages are bit-exact by construction, the two f32 stages
// (fastAtan2, rBRIEF rotation) use explicit non-contracted f32 ops.
//
//   k_resize_linear   cv::resize(INTER_LINEAR) level l-1 -> l      orb_extractor.cc:315-326
//   k_fast_cells      per-cell cv::FAST(thr 20 -> 7) + NMS + mask   orb_extractor.cc:365-437
//   k_blur7           cv::GaussianBlur(7x7, sigma 2, REFLECT_101)   orb_extractor.cc:148-149
//   k_orient_rbrief   ic_angle + 256-bit rBRIEF + KeyPoint assembly orb_extractor.cc:450-458,708-807
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blur_tile.hpp"
#include "plp_barrier.hpp"
#include "orb_device.hpp"
#include "plp_common.hpp"
#include "xcd_map.hpp"

namespace plp {

// ------------------------------------------------------------------------------------------
// K1  bilinear 11-bit fixed-point down-scale.  Every WAVE makes a 64 x 32 destination strip on its own: the source rectangle it needs
// (<= 44 rows x 96 bytes at scale 1.2) is staged in the wave's quarter of the workgroup's LDS with 16-byte loads, then every lane makes
// 4 pixels x 8 rows with byte reads from LDS (one u32 store per row); lanes 16 k .. 16 k + 15 own rows 8 k .. 8 k + 7 of the strip.
// Horizontal sums of a source row are kept for the next destination row (at scale 1.2 five of six destination rows share a source row with
// their predecessor).  History: the first version gathered single bytes from global memory, 16 per output dword, and was bound by the
// texture addresser (profiles/r01g_sq_counters.md); the second staged a 256 x 32 tile per workgroup, one 256-pixel row of lanes per wave,
// and left 30 % of its lanes without a pixel on the narrow pyramid levels (257 = 256 + 1 columns at level 5: the second tile column ran
// for one pixel per row); strips of 64 columns x 32 rows per wave leave 12 %.
// grid = (ceil(dw / 64), ceil(dh / 128), B), block = 256: the four waves of a workgroup take four vertically adjacent strips.
// ------------------------------------------------------------------------------------------
// table entry i of a wave-uniform int16 table: the byte offset is formed in 32 bits, so the load is "scalar base + lane offset" (an index widened first costs
// a 64-bit shift and a 64-bit add per load: 48 loads per lane in k_resize_linear)
__device__ __forceinline__ int rs_tab(const int16_t* __restrict__ t, int i) { return *reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(t) + ((uint32_t)i << 1)); }
constexpr int kRsRows = 44, kRsPitch = 96;   // staged source rectangle of one strip (scale factors >= 1.1 fit; larger ones fall back to global reads)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_resize_linear(const uint8_t* __restrict__ src_base, size_t src_frame_stride,
                                                       int src_pitch, int sw, uint8_t* __restrict__ dst_base,
                                                       size_t dst_frame_stride, int dst_pitch, int dw, int dh,
                                                       const int16_t* __restrict__ xofs0, const int16_t* __restrict__ xofs1,
                                                       const int16_t* __restrict__ a0, const int16_t* __restrict__ a1,
                                                       const int16_t* __restrict__ yofs0, const int16_t* __restrict__ yofs1,
                                                       const int16_t* __restrict__ b0, const int16_t* __restrict__ b1) {
    constexpr int ROWS = 8;
    __shared__ __attribute__((aligned(16))) uint8_t tiles[4 * kRsRows * kRsPitch];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: strip bounds and row tables in scalar registers
    const int strip_x0 = blockIdx.x * 64, strip_y0 = (blockIdx.y * 4 + wv) * 32;
    if (strip_y0 >= dh) return;   // the whole wave (no workgroup barrier below)
    uint8_t* tile = tiles + wv * (kRsRows * kRsPitch);
    const int strip_x1 = min(strip_x0 + 64, dw) - 1, strip_y1 = min(strip_y0 + 32, dh) - 1;
    const uint8_t* src = src_base + (size_t)blockIdx.z * src_frame_stride;
    uint8_t* dst = dst_base + (size_t)blockIdx.z * dst_frame_stride;
    const int xs = (int)xofs0[strip_x0] & ~15, xe = xofs1[strip_x1];
    const int ys = yofs0[strip_y0], ye = yofs1[strip_y1];
    const int nrows = ye - ys + 1, nchunks = (xe - xs) / 16 + 1;
    const bool staged = nrows <= kRsRows && nchunks * 16 <= kRsPitch;   // wave-uniform
    if (staged) {
        const bool wide = (((uintptr_t)src | (uintptr_t)src_pitch) & 15) == 0;
        const int c = lane & 7, r0 = lane >> 3;      // <= 6 chunks of 16 bytes per row, 8 rows per trip
        if (c < nchunks) {
            const int x = xs + 16 * c;
            if (wide && x + 16 <= src_pitch) {       // all of a lane's (up to six) loads are issued before the first is stored: one memory round trip, not six
                constexpr int NT = (kRsRows + 7) / 8;
                uint4 v[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) v[q] = *reinterpret_cast<const uint4*>(src + (uint32_t)(__mul24(ys + min(r0 + 8 * q, nrows - 1), src_pitch) + x));   // (clamped row: no branch between the loads)
#pragma unroll
                for (int q = 0; q < NT; ++q) { const int r = r0 + 8 * q; if (r < nrows) *reinterpret_cast<uint4*>(tile + r * kRsPitch + 16 * c) = v[q]; }
            } else
                for (int r = r0; r < nrows; r += 8) {
                    const uint8_t* g = src + (size_t)(ys + r) * src_pitch + x;
                    uint8_t* t = tile + r * kRsPitch + 16 * c;
                    for (int k = 0; k < 16; ++k) t[k] = x + k < sw ? g[k] : (uint8_t)0;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();   // the strip belongs to this wave alone: its own LDS writes are all it waits for
    }
    const int dy0 = strip_y0 + (lane >> 4) * ROWS;
    const int dx0 = strip_x0 + (lane & 15) * 4;
    if (dy0 >= dh || dx0 >= dw) return;
    int x0[4], x1[4], wa0[4], wa1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = min(dx0 + i, dw - 1);
        x0[i] = rs_tab(xofs0, dx); x1[i] = rs_tab(xofs1, dx); wa0[i] = rs_tab(a0, dx); wa1[i] = rs_tab(a1, dx);
    }
    if (staged) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] -= xs; x1[i] -= xs; }
        int cached_row = -1, hc[4] = {0, 0, 0, 0};
        auto hrow = [&](int y, int (&h)[4]) {
            const uint8_t* S = tile + (y - ys) * kRsPitch;
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = S[x0[i]] * wa0[i] + S[x1[i]] * wa1[i];
        };
        // the rows' table entries are loaded together, packed two to a register (they were four loads and a wait per row)
        uint32_t tyy[ROWS], tbb[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int dy = min(dy0 + r, dh - 1);
            tyy[r] = (uint32_t)(uint16_t)rs_tab(yofs0, dy) | ((uint32_t)(uint16_t)rs_tab(yofs1, dy) << 16);
            tbb[r] = (uint32_t)(uint16_t)rs_tab(b0, dy) | ((uint32_t)(uint16_t)rs_tab(b1, dy) << 16);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int dy = dy0 + r;
            if (dy >= dh) break;
            const int y0 = (int)(tyy[r] & 0xffffu), y1 = (int)(tyy[r] >> 16);   // one value per 16 lanes
            const int wb0 = (int)(int16_t)(tbb[r] & 0xffffu), wb1 = (int)(int16_t)(tbb[r] >> 16);
            int h0[4], h1[4];
            if (y0 == cached_row) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h0[i] = hc[i];
            } else hrow(y0, h0);
            if (y1 == y0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h1[i] = h0[i];
            } else hrow(y1, h1);
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = (((wb0 * (h0[i] >> 4)) >> 16) + ((wb1 * (h1[i] >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 255) << (8 * i);
                hc[i] = h1[i];
            }
            cached_row = y1;
            // rows are padded to a 64-byte pitch, so the 4-byte store never leaves the row
            *reinterpret_cast<uint32_t*>(dst + (uint32_t)(__mul24(dy, dst_pitch) + dx0)) = packed;
        }
        return;
    }
    for (int r = 0; r < ROWS; ++r) {   // generic scale factors: gather from global memory
        const int dy = dy0 + r;
        if (dy >= dh) break;
        const uint8_t* S0 = src + (size_t)yofs0[dy] * src_pitch;
        const uint8_t* S1 = src + (size_t)yofs1[dy] * src_pitch;
        const int wb0 = b0[dy], wb1 = b1[dy];
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h0 = S0[x0[i]] * wa0[i] + S0[x1[i]] * wa1[i];
            const int h1 = S1[x0[i]] * wa0[i] + S1[x1[i]] * wa1[i];
            const int v = (((wb0 * (h0 >> 4)) >> 16) + ((wb1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(dst + (size_t)dy * dst_pitch + dx0) = packed;
    }
}

// ------------------------------------------------------------------------------------------
// K2+K3  FAST-9/16 score + 3x3 NMS + threshold fallback + mask, one workgroup per cell ROI.
//
// The corner score s(p) = max over the 16 arcs of 9 contiguous circle pixels of
// min(|I - v|, sign-consistent) - 1 is what cv::cornerScore<16> returns and does not depend
// on the threshold; cv::FAST(thr) keeps p iff s(p) >= thr and s(p) is strictly greater than
// the 8 neighbouring scores (non-corners and pixels outside the ROI's tested interior count 0).
// Hence one score map serves both the thr=ini pass and the thr=min fallback of an empty cell.
// grid = (n_cells, B), block = 256.
// ------------------------------------------------------------------------------------------
constexpr int kFastQ1 = 1536, kFastQ2 = 512;
constexpr int kTileW = 76;            // LDS row pitch of the u8 ROI tile (>= 70 + 3 alignment slack)
constexpr int kScoreW = 68;           // 64 tested + 2 zero border, padded

__device__ __forceinline__ int min3(int a, int b, int c) { return min(a, min(b, c)); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast_cells(OrbPlanes pl, const CellDesc* __restrict__ cells,
                                                    const LevelDev* __restrict__ lv, int ini_thr, int min_thr,
                                                    const uint8_t* __restrict__ mask, size_t mask_step,
                                                    size_t mask_frame_stride, uint32_t* __restrict__ cell_cand,
                                                    int32_t* __restrict__ cell_count, int n_cells, uint32_t gx_magic) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[70 * kTileW];
    __shared__ __attribute__((aligned(16))) uint8_t score[66 * kScoreW];   // score map of the tested interior, +1 zero ring
    __shared__ unsigned long long keepbits[64];                            // NMS survivors, one bit per tested position
    // work queues between the passes.  They are capped (a 64 x 64 cell of a textured frame yields ~500 compass survivors
    // and ~100 corners); the position bit masks are always complete, and a pass whose queue overflowed walks all positions
    // and tests the bit instead.  (Uncapped queues cost 16 KB: beside region growing's LDS only two of these workgroups
    // fitted a CU and the kernel ran three times slower there than alone.)
    __shared__ uint16_t queue[kFastQ1], queue2[kFastQ2];
    __shared__ unsigned long long bits1[64], bits2[64];
    __shared__ int q_count, q2_count, n_ini, wave_tot[4], run_base;

    unsigned ucell, uframe;
    xcd_frame_major(ucell, uframe, gx_magic);   // neighbouring cell ROIs overlap by 6 pixels and share cache lines
    const int tid = threadIdx.x, frame = (int)uframe, cell = (int)ucell;
    const CellDesc cd = cells[cell];
    const LevelDev L = lv[cd.level];
    const int out_slot = frame * n_cells + cell;
    const float sf = L.scale;

    const uint8_t* mk = mask ? mask + (size_t)frame * mask_frame_stride : nullptr;
    auto masked = [&](int y, int x) -> bool {   // mask.at<uchar>(y * sf, x * sf) == 0
        return mk[(size_t)(int)((float)y * sf) * mask_step + (int)((float)x * sf)] == 0;
    };
    if (mk) {   // skip the cell when one ROI corner is masked (orb_extractor.cc:395-401)
        const int x0 = cd.min_x, x1 = cd.min_x + cd.w, y0 = cd.min_y, y1 = cd.min_y + cd.h;
        if (masked(y0, x0) || masked(y1, x0) || masked(y0, x1) || masked(y1, x1)) {
            if (tid == 0) cell_count[out_slot] = 0;
            return;
        }
    }

    const uint8_t* img = pl.level_ptr(frame, cd.level, L);
    const int pitch = pl.level_pitch(cd.level, L);
    const int w = cd.w, h = cd.h;
    // stage the ROI through LDS so that ROI column j sits at tile column j + 1: the tested position tx (ROI column
    // tx + 3) is then at tile column tx + 4, dword-aligned for tx % 4 == 0 (pass 1 handles 4 positions per thread from
    // whole dwords).  Global reads are aligned dwords, shifted into place with v_alignbyte.
    {
        const int a0 = cd.min_x - 1;                 // image column of tile column 0
        const int sh = a0 & 3;
        const int ndw = (w + 2 + 3) >> 2;            // tile columns 0 .. w + 1
        const uint8_t* row0 = img + (size_t)cd.min_y * pitch + (a0 - sh);
        // (the loop stays a load -> LDS store per trip: issuing a thread's five dword pairs together was measured and is SLOWER here, 2.31 -> 2.40 ms --
        //  the kernel sits at its register limit and the staging is a small part of it; profiles/r04_tile_pipelining.md)
        // (row, dword) of a thread's next slot by additions: a division by the runtime row length per trip was a fifth of this kernel's vector instructions
        const int qr = 256 / ndw, qd = 256 - qr * ndw;   // wave-uniform
        int r = tid / ndw, c = tid - r * ndw;
        for (int i = tid; i < ndw * h; i += 256, c += qd, r += qr) {
            if (c >= ndw) { c -= ndw; ++r; }
            const uint32_t* g = reinterpret_cast<const uint32_t*>(row0 + (ptrdiff_t)(__mul24(r, pitch) + 4 * c));
            const uint32_t lo = g[0];
            const uint32_t v = sh ? __builtin_amdgcn_alignbyte(g[1], lo, (uint32_t)sh) : lo;
            *reinterpret_cast<uint32_t*>(&tile[r * kTileW + 4 * c]) = v;
        }
    }
    const int tw = w - 6, th = h - 6;   // tested interior (ROI x,y in [3, w-3) x [3, h-3))
    // cv::FAST(ini_thr), and cv::FAST(min_thr) only when that finds nothing in this cell (:404-412).  A corner at
    // threshold t has score >= t, and scores below t never win a 3x3 comparison against one >= t, so each attempt may
    // simply ignore everything below its own threshold.
    int thr = ini_thr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        thr = attempt == 0 ? ini_thr : min_thr;
        for (int i = tid; i < 66 * kScoreW / 4; i += 256) reinterpret_cast<uint32_t*>(score)[i] = 0;
        if (tid < 64) { keepbits[tid] = 0ull; bits1[tid] = 0ull; bits2[tid] = 0ull; }
        if (tid == 0) { q_count = 0; q2_count = 0; n_ini = 0; run_base = 0; }
        wg_barrier();
        // pass 1: every arc of 9 contains two neighbouring compass pixels (0, 4, 8, 12): five dword reads serve four
        // positions and reject most of them
        for (int i = tid; i < 1024; i += 256) {   // 64 rows x 16 groups of 4 positions
            const int ty = i >> 4, tx4 = (i & 15) * 4;
            if (tx4 >= tw || ty >= th) continue;
            const uint32_t* rowc = reinterpret_cast<const uint32_t*>(&tile[(ty + 3) * kTileW]) + (tx4 >> 2);
            const uint32_t C = rowc[1], Lw = rowc[0], Rw = rowc[2];
            const uint32_t Lq = __builtin_amdgcn_alignbyte(C, Lw, 1u);     // columns -3 of the four positions
            const uint32_t Rq = __builtin_amdgcn_alignbyte(Rw, C, 3u);     // columns +3
            const uint32_t U = *(reinterpret_cast<const uint32_t*>(&tile[ty * kTileW]) + (tx4 >> 2) + 1);          // row -3
            const uint32_t D = *(reinterpret_cast<const uint32_t*>(&tile[(ty + 6) * kTileW]) + (tx4 >> 2) + 1);    // row +3
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tx4 + k >= tw) break;
                const int v = (int)((C >> (8 * k)) & 255u);
                const int hi = v + thr, lo = v - thr;
                const int p0 = (int)((D >> (8 * k)) & 255u), p4 = (int)((Rq >> (8 * k)) & 255u), p8 = (int)((U >> (8 * k)) & 255u),
                          p12 = (int)((Lq >> (8 * k)) & 255u);
                const bool b0 = p0 > hi, b4 = p4 > hi, b8 = p8 > hi, b12 = p12 > hi;
                const bool d0 = p0 < lo, d4 = p4 < lo, d8 = p8 < lo, d12 = p12 < lo;
                if (((b0 | b8) & (b4 | b12)) | ((d0 | d8) & (d4 | d12))) {
                    atomicOr(&bits1[ty], 1ull << (tx4 + k));
                    const int slot = atomicAdd(&q_count, 1);
                    if (slot < kFastQ1) queue[slot] = (uint16_t)((ty << 6) | (tx4 + k));
                }
            }
        }
        wg_barrier();
        // pass 2: 16-bit brighter / darker masks of the survivors -> "has an arc of 9" -> second queue
        const int nq1 = q_count;
        const bool dense1 = nq1 <= kFastQ1;
        for (int j = tid; j < (dense1 ? nq1 : 4096); j += 256) {
            const int i = dense1 ? (int)queue[j] : j;
            const int ty = i >> 6, tx = i & 63;
            if (!dense1 && !((bits1[ty] >> tx) & 1ull)) continue;
            const uint8_t* c = &tile[(ty + 3) * kTileW + tx + 4];
            const int v = c[0];
            const int hi = v + thr, lo = v - thr;
            uint32_t B = 0, D = 0;
#define PLP_T(k, dx, dy) { const int p = c[(dy) * kTileW + (dx)]; B |= (uint32_t)(p > hi) << k; D |= (uint32_t)(p < lo) << k; }
            PLP_T(0, 0, 3) PLP_T(1, 1, 3) PLP_T(2, 2, 2) PLP_T(3, 3, 1) PLP_T(4, 3, 0) PLP_T(5, 3, -1) PLP_T(6, 2, -2) PLP_T(7, 1, -3)
            PLP_T(8, 0, -3) PLP_T(9, -1, -3) PLP_T(10, -2, -2) PLP_T(11, -3, -1) PLP_T(12, -3, 0) PLP_T(13, -3, 1) PLP_T(14, -2, 2) PLP_T(15, -1, 3)
#undef PLP_T
            auto arc9 = [](uint32_t m) -> bool {
                m |= m << 16;
                uint32_t x = m & (m >> 1);
                x &= x >> 2;
                x &= x >> 4;
                x &= m >> 8;
                return (x & 0xffffu) != 0;
            };
            if (arc9(B) || arc9(D)) {
                atomicOr(&bits2[ty], 1ull << tx);
                const int slot = atomicAdd(&q2_count, 1);
                if (slot < kFastQ2) queue2[slot] = (uint16_t)i;
            }
        }
        wg_barrier();
        // pass 3: exact score of the corners (dense over the queue: no lane idles on non-corners)
        const int nq = q2_count;
        const bool dense2 = nq <= kFastQ2;
        for (int j = tid; j < (dense2 ? nq : 4096); j += 256) {
            const int i = dense2 ? (int)queue2[j] : j;
            const int ty = i >> 6, tx = i & 63;
            if (!dense2 && !((bits2[ty] >> tx) & 1ull)) continue;
            const uint8_t* c = &tile[(ty + 3) * kTileW + tx + 4];
            const int v = c[0];
            // The 16 ring differences d[k] = p_k - v (|d| <= 255) as eight i16 pairs P[k] = (d[k], d[k + 8]): one packed min / max
            // works on two ring positions, and a pair whose first index runs past 7 is the half-swapped pair of index - 8.  Sliding
            // min / max over windows of 9 by doubling, as before (mn2, mn4, then mn9[k] = min(mn4[k], mn4[k + 4], d[k + 8])), in 40
            // packed registers at most instead of 80 scalar ones: the kernel keeps 8 waves per SIMD without scratch (the scalar
            // version spilled 9 VGPRs, and its scratch stores were this kernel's whole HBM write stream).
            typedef short v2s __attribute__((ext_vector_type(2)));
            auto pair = [&](int o_lo, int o_hi) -> v2s { v2s r; r.x = (short)((int)c[o_lo] - v); r.y = (short)((int)c[o_hi] - v); return r; };
            auto sw = [](v2s x) -> v2s { return __builtin_shufflevector(x, x, 1, 0); };
            v2s P[8];
            P[0] = pair(3 * kTileW, -3 * kTileW);          P[1] = pair(3 * kTileW + 1, -3 * kTileW - 1);
            P[2] = pair(2 * kTileW + 2, -2 * kTileW - 2);  P[3] = pair(kTileW + 3, -kTileW - 3);
            P[4] = pair(3, -3);                            P[5] = pair(-kTileW + 3, kTileW - 3);
            P[6] = pair(-2 * kTileW + 2, 2 * kTileW - 2);  P[7] = pair(-3 * kTileW + 1, 3 * kTileW - 1);
            v2s n2[8], x2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v2s nxt = k < 7 ? P[k + 1] : sw(P[0]);
                n2[k] = __builtin_elementwise_min(P[k], nxt); x2[k] = __builtin_elementwise_max(P[k], nxt);
            }
            v2s n4[8], x4[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                n4[k] = __builtin_elementwise_min(n2[k], k < 6 ? n2[k + 2] : sw(n2[k - 6]));
                x4[k] = __builtin_elementwise_max(x2[k], k < 6 ? x2[k + 2] : sw(x2[k - 6]));
            }
            v2s br, dk;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v2s d8 = sw(P[k]);                                  // (d[k + 8], d[k])
                const v2s mn9 = __builtin_elementwise_min(__builtin_elementwise_min(n4[k], k < 4 ? n4[k + 4] : sw(n4[k - 4])), d8);
                const v2s mx9 = __builtin_elementwise_max(__builtin_elementwise_max(x4[k], k < 4 ? x4[k + 4] : sw(x4[k - 4])), d8);
                br = k ? __builtin_elementwise_max(br, mn9) : mn9;
                dk = k ? __builtin_elementwise_min(dk, mx9) : mx9;
            }
            const int bright = max((int)br.x, (int)br.y), dark = min((int)dk.x, (int)dk.y);
            const int sc = max(bright, -dark) - 1;
            score[(ty + 1) * kScoreW + tx + 1] = (uint8_t)(sc >= thr ? sc : 0);
        }
        wg_barrier();
        // pass 4: 3x3 strict NMS, only around the corners
        int my_ini = 0;
        for (int j = tid; j < (dense2 ? nq : 4096); j += 256) {
            const int i = dense2 ? (int)queue2[j] : j;
            const int ty = i >> 6, tx = i & 63;
            if (!dense2 && !((bits2[ty] >> tx) & 1ull)) continue;
            const uint8_t* sp = &score[(ty + 1) * kScoreW + tx + 1];
            const int v = sp[0];
            const bool ok = v > 0 && v > sp[-1] && v > sp[1] && v > sp[-kScoreW - 1] && v > sp[-kScoreW] && v > sp[-kScoreW + 1] &&
                            v > sp[kScoreW - 1] && v > sp[kScoreW] && v > sp[kScoreW + 1];
            if (ok) { atomicOr(&keepbits[ty], 1ull << tx); ++my_ini; }
        }
        if (my_ini) atomicAdd(&n_ini, my_ini);
        wg_barrier();
        const int found = n_ini;
        wg_barrier();
        if (found > 0 || min_thr >= ini_thr) break;
    }

    // pass 5: ordered (row-major) compaction of the survivors.  The NMS survivors of a position row are one 64-bit mask.
    uint32_t* out = cell_cand + (size_t)out_slot * kCellCap;
    const int lane = tid & 63, wv = tid >> 6;
    if (!mk) {
        // no image mask: one wave does it all.  Lane = row: popcount, exclusive scan over the 64 rows, then every lane
        // writes the few survivors of its own row (a cell keeps ~20 of its 4096 positions).
        if (wv != 0) return;
        unsigned long long m = keepbits[lane];
        const int cnt = __popcll(m);
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        int pos = inc - cnt;
        const uint32_t y = (uint32_t)(lane + 3 + cd.cy * kCellSize);
        while (m) {
            const int tx = __builtin_ctzll(m);
            m &= m - 1;
            // border-relative position: ROI coordinate + cell index * 64 (orb_extractor.cc:426-427)
            const uint32_t v = score[(lane + 1) * kScoreW + tx + 1];
            if (pos < kCellCap) out[pos] = (uint32_t)(tx + 3 + cd.cx * kCellSize) | (y << 12) | (v << 24);
            ++pos;
        }
        if (lane == 63) cell_count[out_slot] = inc;
        return;
    }
    // with an image mask every survivor is tested against it (orb_extractor.cc:429): wave w owns rows 16w .. 16w+15,
    // counts by ballot, one barrier for the wave bases, then the writes
    auto row_mask = [&](int ty) -> unsigned long long {
        const unsigned long long m = keepbits[ty];
        return __ballot(((m >> lane) & 1ull) && !masked(cd.min_y + ty + 3, cd.min_x + lane + 3));
    };
    int cnt = 0;
    for (int ty = 16 * wv; ty < 16 * wv + 16; ++ty) cnt += __popcll(row_mask(ty));
    if (lane == 0) wave_tot[wv] = cnt;
    wg_barrier();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += wave_tot[k];
    for (int ty = 16 * wv; ty < 16 * wv + 16; ++ty) {
        const unsigned long long m = row_mask(ty);
        if ((m >> lane) & 1ull) {
            const uint32_t v = score[(ty + 1) * kScoreW + lane + 1];
            const uint32_t x = (uint32_t)(lane + 3 + cd.cx * kCellSize), y = (uint32_t)(ty + 3 + cd.cy * kCellSize);
            out[off + __popcll(m & ((1ull << lane) - 1ull))] = x | (y << 12) | (v << 24);
        }
        off += __popcll(m);
    }
    if (tid == 0) run_base = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    wg_barrier();
    if (tid == 0) cell_count[out_slot] = run_base;
}

// ------------------------------------------------------------------------------------------
// K6  7x7 sigma-2 Gaussian, 8.8 fixed-point taps (sum 256), exact separable integer passes (blur_tile.hpp).
// One workgroup = 128 x 64 output tile of one level of one frame; all levels in one launch.
// grid = (tiles of all levels, B), block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur7(OrbPlanes pl, uint8_t* __restrict__ blur_base, size_t blur_frame_stride,
                                               const LevelDev* __restrict__ lv, int n_levels, BlurTaps taps, uint32_t gx_magic) {
    __shared__ BlurTileLds<3> S;
    unsigned ut, uf;
    xcd_frame_major(ut, uf, gx_magic);   // all tiles of a frame share halo rows: one L2 per frame
    int t = (int)ut;
    const int frame = (int)uf;
    int level = 0;
    while (level + 1 < n_levels && t >= lv[level].blur_tiles) { t -= lv[level].blur_tiles; ++level; }
    const LevelDev L = lv[level];
    const int tiles_x = (L.w + kBlurTW - 1) / kBlurTW;
    const int trow = (int)plp_div((unsigned)t, (unsigned)tiles_x, L.blur_tiles_x_magic);
    blur_tile<3>(S, pl.level_ptr(frame, level, L), pl.level_pitch(level, L), blur_base + (size_t)frame * blur_frame_stride + L.off, L.pitch,
                 L.w, L.h, (t - trow * tiles_x) * kBlurTW, trow * kBlurTH, taps.k);
}

// ------------------------------------------------------------------------------------------
// K5+K7  orientation + rBRIEF + KeyPoint assembly, 16 lanes per selected key point (4 key points per wave64).
// The first version spent a whole wave on one key point: 4.2 M waves of ~300 instructions each, three dependent memory
// round trips per wave, latency-bound at full occupancy.  Now a round trip serves four key points:
//   ic_angle   lane s owns disc rows s-15 and s+1: eight unaligned dword loads per row, masked moments with
//              v_dot4_u32_u8 against per-row weight tables (sum of (u+15)*I and of I; m10 = first - 15 * second)
//   rBRIEF     lane s owns pairs 16s .. 16s+15 = descriptor bytes 2s, 2s+1
// grid = (ceil(total_sel_cap / 16), B), block = 256.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {   // cv::fastAtan2, no FMA
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

// util::cos / util::sin of the reference (src/PLPSLAM/util/trigonometric.h:43-78)
__device__ __forceinline__ float ref_poly_cos(float v) {
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(0.99940307f, __fmul_rn(v2, __fadd_rn(-0.49558072f, __fmul_rn(0.03679168f, v2))));
}
__device__ __forceinline__ float ref_cos(float v) {
    const float PI = 3.14159265358979f, PI_2 = PI / 2.0f, TWO_PI = 2.0f * PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    const float q = __fmul_rn(v, INV_TWO_PI);
    int fl = (int)q;
    fl -= (fl > q);
    v = __fsub_rn(v, __fmul_rn((float)fl, TWO_PI));
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return ref_poly_cos(v);
    if (v < PI) return -ref_poly_cos(__fsub_rn(PI, v));
    if (v < THREE_PI_2) return -ref_poly_cos(__fsub_rn(v, PI));
    return ref_poly_cos(__fsub_rn(TWO_PI, v));
}
__device__ __forceinline__ float ref_sin(float v) { return ref_cos(__fsub_rn(3.14159265358979f / 2.0f, v)); }

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "rbrief_pattern.inc"
};

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);   // global loads take any byte address on gfx950 (unaligned access mode)
    return v;
}
__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false);
    return v;
}

__global__ __launch_bounds__(256) void k_orient_rbrief(OrbPlanes pl, const uint8_t* __restrict__ blur_base,
                                                       size_t blur_frame_stride, const LevelDev* __restrict__ lv,
                                                       int n_levels, const int32_t* __restrict__ sel,   // [B][total_sel_cap] packed x|y<<12|score<<24
                                                       const int32_t* __restrict__ sel_count,           // [B][kMaxLevels]
                                                       int total_sel_cap, UMax um, plp_keypoint* __restrict__ out_kps,
                                                       uint8_t* __restrict__ out_desc, int cap, int32_t* __restrict__ out_counts,
                                                       int32_t* __restrict__ status, uint32_t gx_magic) {
    __shared__ uint32_t s_w0[16][8], s_w1[16][8];   // per |v|: byte weights 1 / (u + 15) inside the disc, 0 outside
    __shared__ uint32_t s_pat[256];                 // the 256 test pairs (ax, ay, bx, by as int8)
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[16 * 37 * 48];   // per key point: the disc of the level, then the patch of the blurred level
    unsigned ublk, uframe;
    xcd_frame_major(ublk, uframe, gx_magic);   // a frame's patches (two planes, ~2.6 MB) stay in one L2
    const int tid = threadIdx.x, sub = tid & 15, frame = (int)uframe;
    // Key point slots are dealt DENSELY: group d of 16 lanes takes the d-th selected key point of the frame (levels in order), which is also its
    // output row.  The selection array is laid out by level CAPACITY (2032 slots for ~1000 key points at K = 1000): dealt by slot, half of
    // the 16-lane groups -- the tail of every level's range -- loaded the tables, passed the barrier and left.
    const int32_t* cnt = sel_count + frame * kMaxLevels;
    int total = 0;
    for (int l = 0; l < n_levels; ++l) total += cnt[l];
    if (ublk == 0 && tid == 0) {
        out_counts[frame] = min(total, cap);
        if (total > cap) atomicOr(status, 1);
    }
    if ((int)ublk * 16 >= min(total, cap)) return;   // the whole workgroup, before it touches LDS
    if (tid < 128) {
        const int av = tid >> 3, j = tid & 7, half = um.v[av];
        uint32_t w0 = 0, w1 = 0;
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * j + k - 15;
            if (u <= 15 && (u < 0 ? -u : u) <= half) { w0 |= 1u << (8 * k); w1 |= (uint32_t)(u + 15) << (8 * k); }
        }
        s_w0[av][j] = w0; s_w1[av][j] = w1;
    }
    s_pat[tid] = reinterpret_cast<const uint32_t*>(c_pattern)[tid];
    wg_barrier();
    const int out_idx = (int)ublk * 16 + (tid >> 4);
    if (out_idx >= total || out_idx >= cap) return;   // the 16 lanes of a key point leave together
    int level = 0, i = out_idx;
    while (level + 1 < n_levels && i >= cnt[level]) { i -= cnt[level]; ++level; }
    const LevelDev L = lv[level];
    const int g = L.sel_base + i;

    const uint32_t pk = (uint32_t)sel[(size_t)frame * total_sel_cap + g];
    const int cx = (int)(pk & 0xfff) + kOrbBorder, cy = (int)((pk >> 12) & 0xfff) + kOrbBorder;   // level pixel
    const int resp = (int)(pk >> 24);
    if (cx < kOrbBorder || cy < kOrbBorder || cx > L.w - 1 - kOrbBorder || cy > L.h - 1 - kOrbBorder) {   // cannot happen: a selected point is a FAST corner
        if (sub == 0) atomicOr(status, 8);
        return;
    }

    // Both patches of a key point go through LDS (s_patch: 37 rows x 48 bytes per key point): the 16 lanes fetch them with 16-byte
    // loads, neighbouring lanes taking neighbouring pieces of a row, so that one load instruction touches ~20 cache lines instead
    // of 64 -- the kernel used to be bound by L1 tag look-ups (32 single-byte gathers per lane for the test pairs alone).
    uint8_t* patch = s_patch + (tid >> 4) * (37 * 48);
    // intensity centroid over the radius-15 disc (key points keep 19 pixels from the border: every load is inside)
    const uint8_t* img = pl.level_ptr(frame, level, L);
    const int pitch = pl.level_pitch(level, L);
    {   // rows cy-15 .. cy+15, columns cx-15 .. cx+16
        const uint8_t* p0 = img + (size_t)(cy - 15) * pitch + cx - 15;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = sub + 16 * it, row = item >> 1, half = item & 1;
            if (row < 31) *reinterpret_cast<uint4*>(patch + row * 48 + 16 * half) = load_u128_unaligned(p0 + (size_t)row * pitch + 16 * half);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int m10 = 0, m01 = 0;
    {
        const int vA = sub - 15, vB = sub + 1;   // rows -15..0 and 1..15 (lane 15 has no second row)
        const uint4* pA = reinterpret_cast<const uint4*>(patch + (vA + 15) * 48);
        const uint4* pB = reinterpret_cast<const uint4*>(patch + (min(vB, 15) + 15) * 48);
        const uint4 a0 = pA[0], a1 = pA[1], b0 = pB[0], b1 = pB[1];
        const uint32_t dA[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, dB[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t sA = 0, tA = 0, sB = 0, tB = 0;   // s = sum I, t = sum (u + 15) I
        const int aA = -vA, aB = min(vB, 15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sA = __builtin_amdgcn_udot4(dA[j], s_w0[aA][j], sA, false); tA = __builtin_amdgcn_udot4(dA[j], s_w1[aA][j], tA, false);
            sB = __builtin_amdgcn_udot4(dB[j], s_w0[aB][j], sB, false); tB = __builtin_amdgcn_udot4(dB[j], s_w1[aB][j], tB, false);
        }
        if (vB > 15) { sB = 0; tB = 0; }
        m10 = (int)tA - 15 * (int)sA + (int)tB - 15 * (int)sB;
        m01 = vA * (int)sA + vB * (int)sB;
    }
    m10 = row16_sum(m10); m01 = row16_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // rBRIEF on the blurred level: rows cy-18 .. cy+18, columns cx-18 .. cx+29 (the rotated pattern stays within +-18)
    const float arad = (float)((double)angle * 3.14159265358979323846 / 180.0);
    const float ca = ref_cos(arad), sa = ref_sin(arad);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();   // every lane of the key point has read its disc rows
    {
        const uint8_t* b0 = blur_base + (size_t)frame * blur_frame_stride + L.off + (size_t)(cy - 18) * L.pitch + cx - 18;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int item = sub + 16 * it, row = item / 3, seg = item - 3 * row;
            if (row < 37) *reinterpret_cast<uint4*>(patch + row * 48 + 16 * seg) = load_u128_unaligned(b0 + (size_t)row * L.pitch + 16 * seg);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint8_t* bl = patch + 18 * 48 + 18;
    uint32_t bits = 0;
    int ta[16], tb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t pt = s_pat[16 * sub + r];
        const float ax = (float)(int8_t)(pt & 0xff), ay = (float)(int8_t)((pt >> 8) & 0xff), bx = (float)(int8_t)((pt >> 16) & 0xff),
                    by = (float)(int8_t)(pt >> 24);
        const int ar = __float2int_rn(__fadd_rn(__fmul_rn(ax, sa), __fmul_rn(ay, ca)));
        const int ac = __float2int_rn(__fsub_rn(__fmul_rn(ax, ca), __fmul_rn(ay, sa)));
        const int br = __float2int_rn(__fadd_rn(__fmul_rn(bx, sa), __fmul_rn(by, ca)));
        const int bc = __float2int_rn(__fsub_rn(__fmul_rn(bx, ca), __fmul_rn(by, sa)));
        ta[r] = bl[ar * 48 + ac]; tb[r] = bl[br * 48 + bc];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) bits |= (uint32_t)(ta[r] < tb[r]) << r;
    reinterpret_cast<uint16_t*>(out_desc + ((size_t)frame * cap + out_idx) * 32)[sub] = (uint16_t)bits;
    if (sub == 0) {
        plp_keypoint k;
        const float s = L.scale;
        k.x = level ? __fmul_rn((float)cx, s) : (float)cx;
        k.y = level ? __fmul_rn((float)cy, s) : (float)cy;
        k.size = (float)(unsigned)(__fmul_rn((float)kFastPatch, s));
        k.angle = angle;
        k.response = (float)resp;
        k.octave = level;
        k.class_id = -1;
        out_kps[(size_t)frame * cap + out_idx] = k;
    }
}

// ------------------------------------------------------------------------------------------ launch wrappers
void launch_resize(hipStream_t st, const OrbPlanes& pl, const LevelDev* h_lv, int level, int B, const ResizeDev& rs) {
    const LevelDev& S = h_lv[level - 1];
    const LevelDev& D = h_lv[level];
    const uint8_t* src = level - 1 == 0 ? pl.l0 : pl.pyr + S.off;
    const size_t sstride = level - 1 == 0 ? pl.l0_frame_stride : pl.pyr_frame_stride;
    const int spitch = level - 1 == 0 ? pl.l0_pitch : S.pitch;
    dim3 grid((D.w + 63) / 64, (D.h + 127) / 128, B), block(256);
    hipLaunchKernelGGL(k_resize_linear, grid, block, 0, st, src, sstride, spitch, S.w, pl.pyr + D.off, pl.pyr_frame_stride, D.pitch,
                       D.w, D.h, rs.xofs0 + rs.col_base[level], rs.xofs1 + rs.col_base[level], rs.a0 + rs.col_base[level],
                       rs.a1 + rs.col_base[level], rs.yofs0 + rs.row_base[level], rs.yofs1 + rs.row_base[level],
                       rs.b0 + rs.row_base[level], rs.b1 + rs.row_base[level]);
}

void launch_fast(hipStream_t st, const OrbPlanes& pl, const CellDesc* d_cells, int n_cells, const LevelDev* d_lv, int B,
                 int ini_thr, int min_thr, const uint8_t* d_mask, size_t mask_step, size_t mask_frame_stride,
                 uint32_t* cell_cand, int32_t* cell_count) {
    hipLaunchKernelGGL(k_fast_cells, dim3(n_cells, B), dim3(256), 0, st, pl, d_cells, d_lv, ini_thr, min_thr, d_mask, mask_step,
                       mask_frame_stride, cell_cand, cell_count, n_cells, plp_div_magic((uint32_t)n_cells, (uint64_t)n_cells * B));
}

void launch_blur(hipStream_t st, const OrbPlanes& pl, uint8_t* blur, size_t blur_frame_stride, const LevelDev* d_lv,
                 int n_levels, int total_tiles, int B, const BlurTaps& taps, const LevelDev* h_lv) {
    (void)h_lv;
    hipLaunchKernelGGL(k_blur7, dim3(total_tiles, B), dim3(256), 0, st, pl, blur, blur_frame_stride, d_lv, n_levels, taps, plp_div_magic((uint32_t)total_tiles, (uint64_t)total_tiles * B));
}

void launch_orient_rbrief(hipStream_t st, const OrbPlanes& pl, const uint8_t* blur, size_t blur_frame_stride,
                          const LevelDev* d_lv, int n_levels, const int32_t* sel, const int32_t* sel_count,
                          int total_sel_cap, const UMax& um, plp_keypoint* kps, uint8_t* desc, int cap, int32_t* counts,
                          int32_t* status, int B) {
    hipLaunchKernelGGL(k_orient_rbrief, dim3((total_sel_cap + 15) / 16, B), dim3(256), 0, st, pl, blur, blur_frame_stride, d_lv,
                       n_levels, sel, sel_count, total_sel_cap, um, kps, desc, cap, counts, status,
                       plp_div_magic((uint32_t)((total_sel_cap + 15) / 16), (uint64_t)((total_sel_cap + 15) / 16) * B));
}

// ------------------------------------------------------------------------------------------
// match::stereo::compute (reference src/PLPSLAM/match/stereo.cc:45-301), array form.
// k_stereo_match: one wave64 per LEFT key point: best right key point in the row band (Hamming < 75, octave +-1,
// disparity range; first minimum in right-index order), 11x11 L1 patch slide over 11 offsets on the pyramid level
// of the left key point (integer arithmetic: the reference's float patches hold integers), parabola refinement.
// k_stereo_median: per frame, the median of the (int-truncated) correlations and the 2x-median rejection.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_match(OrbPlanes pl_l, OrbPlanes pl_r, const LevelDev* __restrict__ lv, StereoArgs A) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int il = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nl = A.cnt_l ? min(A.cnt_l[b], A.cap) : A.cap, nr = A.cnt_r ? min(A.cnt_r[b], A.cap) : A.cap;
    if (il >= A.cap) return;
    float* xr_out = A.x_right + (size_t)b * A.cap;
    float* dp_out = A.depth + (size_t)b * A.cap;
    int32_t* corr_out = A.corr + (size_t)b * A.cap;
    if (lane == 0) { xr_out[il] = -1.0f; dp_out[il] = -1.0f; corr_out[il] = -1; }
    if (il >= nl) return;
    const plp_keypoint kp = A.kps_l[(size_t)b * A.cap + il];
    const plp_keypoint* kr = A.kps_r + (size_t)b * A.cap;
    const float max_disp = __fdiv_rn(A.fxb, A.tb), min_disp = 0.0f;
    const int row = (int)(unsigned long long)kp.y;              // indices_right_in_row.at(y_left): float -> size_t
    const float min_x = __fsub_rn(kp.x, max_disp), max_x = __fsub_rn(kp.x, min_disp);
    if (max_x < 0) return;
    const uint4* ql = reinterpret_cast<const uint4*>(A.desc_l + ((size_t)b * A.cap + il) * 32);
    const uint4 q0 = ql[0], q1 = ql[1];
    unsigned best = (75u << 16) | 0xffffu;                      // hamm_dist_thr_ = (100 + 50) / 2; strict < keeps the first minimum
    for (int ir = lane; ir < nr; ir += 64) {
        const plp_keypoint k = kr[ir];
        const float r = __fmul_rn(2.0f, lv[k.octave].scale);
        const float lo = __fsub_rn(k.y, r), hi = __fadd_rn(k.y, r);
        int min_r = (int)lo; min_r -= (min_r > lo);
        int max_r = (int)hi; max_r += (max_r < hi);
        if (row < min_r || row > max_r) continue;
        if (k.octave < kp.octave - 1 || k.octave > kp.octave + 1) continue;
        if (k.x < min_x || max_x < k.x) continue;
        const uint4* d = reinterpret_cast<const uint4*>(A.desc_r + ((size_t)b * A.cap + ir) * 32);
        const uint4 d0 = d[0], d1 = d[1];
        const unsigned dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                              __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
        best = min(best, (dist << 16) | (unsigned)ir);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
    if ((best >> 16) >= 75u) return;
    const plp_keypoint kpr = kr[best & 0xffffu];
    // ---- compute_subpixel_disparity (stereo.cc:225-301)
    const LevelDev L = lv[kp.octave];
    const float isf = A.inv_scale[kp.octave];
    const int sxl = __float2int_rn(__fmul_rn(kp.x, isf)), syl = __float2int_rn(__fmul_rn(kp.y, isf)), sxr = __float2int_rn(__fmul_rn(kpr.x, isf));
    constexpr int win = 5, slide = 5;
    if (sxr - slide - win < 0 || L.w <= sxr + slide + win) return;
    const uint8_t* IL = pl_l.level_ptr(b, kp.octave, L);
    const uint8_t* IR = pl_r.level_ptr(b, kp.octave, L);
    const int pitch_l = pl_l.level_pitch(kp.octave, L), pitch_r = pl_r.level_pitch(kp.octave, L);
    const int lc = IL[(size_t)syl * pitch_l + sxl];
    int lv0 = 0, lv1 = 0;                                       // this lane's two pixels of the 11 x 11 left patch, centre removed
    const int p0 = lane, p1 = lane + 64;
    const int dy0 = p0 / 11 - win, dx0 = p0 % 11 - win, dy1 = p1 / 11 - win, dx1 = p1 % 11 - win;
    lv0 = (int)IL[(size_t)(syl + dy0) * pitch_l + sxl + dx0] - lc;
    if (p1 < 121) lv1 = (int)IL[(size_t)(syl + dy1) * pitch_l + sxl + dx1] - lc;
    int corr[2 * slide + 1];
    int best_corr = 0x7fffffff, best_off = 0;
#pragma unroll
    for (int off = -slide; off <= slide; ++off) {
        const int rc = IR[(size_t)syl * pitch_r + sxr + off];
        int s = abs(lv0 - ((int)IR[(size_t)(syl + dy0) * pitch_r + sxr + off + dx0] - rc));
        if (p1 < 121) s += abs(lv1 - ((int)IR[(size_t)(syl + dy1) * pitch_r + sxr + off + dx1] - rc));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        corr[off + slide] = s;
        if (s < best_corr) { best_corr = s; best_off = off; }
    }
    if (best_off == -slide || best_off == slide) return;
    float c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int i = 1; i < 2 * slide; ++i)
        if (i == best_off + slide) { c1 = (float)corr[i - 1]; c2 = (float)corr[i]; c3 = (float)corr[i + 1]; }
    const float x_delta = (float)((double)__fsub_rn(c1, c3) / (2.0 * (double)__fadd_rn(c1, c3) - 4.0 * (double)c2));
    if (x_delta < -1.0 || 1.0 < x_delta) return;
    float best_x_right = __fmul_rn(L.scale, __fadd_rn((float)(sxr + best_off), x_delta));
    float best_disp = __fsub_rn(kp.x, best_x_right);
    if (best_disp < min_disp || max_disp <= best_disp) return;
    if (best_disp <= 0.0f) { best_disp = 0.01f; best_x_right = __fsub_rn(kp.x, best_disp); }
    if (lane == 0) { dp_out[il] = __fdiv_rn(A.fxb, best_disp); xr_out[il] = best_x_right; corr_out[il] = best_corr; }
}

// grid = (B), block = 256: k-th smallest (k = count / 2) of the valid correlations by two 8-bit histogram passes
__global__ __launch_bounds__(256) void k_stereo_median(StereoArgs A) {
    __shared__ int hist[256], s_sel[3];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int nl = A.cnt_l ? min(A.cnt_l[b], A.cap) : A.cap;
    int32_t* corr = A.corr + (size_t)b * A.cap;
    float* xr_out = A.x_right + (size_t)b * A.cap;
    float* dp_out = A.depth + (size_t)b * A.cap;
    hist[tid] = 0;
    wg_barrier();
    for (int i = tid; i < nl; i += 256) { const int c = corr[i]; if (c >= 0) atomicAdd(&hist[min(c >> 8, 255)], 1); }
    wg_barrier();
    if (tid == 0) {
        int total = 0;
        for (int i = 0; i < 256; ++i) total += hist[i];
        int k = total / 2, hb = 0;
        if (total) { while (k >= hist[hb]) { k -= hist[hb]; ++hb; } }
        s_sel[0] = total; s_sel[1] = hb; s_sel[2] = k;
    }
    wg_barrier();
    const int total = s_sel[0], hb = s_sel[1], k = s_sel[2];
    if (total == 0) return;
    wg_barrier();
    hist[tid] = 0;
    wg_barrier();
    for (int i = tid; i < nl; i += 256) { const int c = corr[i]; if (c >= 0 && min(c >> 8, 255) == hb) atomicAdd(&hist[c & 255], 1); }
    wg_barrier();
    if (tid == 0) {
        int kk = k, lb = 0;
        while (kk >= hist[lb]) { kk -= hist[lb]; ++lb; }
        s_sel[1] = (hb << 8) | lb;
    }
    wg_barrier();
    const float median = (float)s_sel[1];
    const float thr = (float)(2.0 * (double)median);
    for (int i = tid; i < nl; i += 256) { const int c = corr[i]; if (c >= 0 && thr < (float)c) { xr_out[i] = -1.0f; dp_out[i] = -1.0f; } }
}

void launch_stereo(hipStream_t st, const OrbPlanes& pl_l, const OrbPlanes& pl_r, const LevelDev* d_lv, const StereoArgs& A, int B) {
    hipLaunchKernelGGL(k_stereo_match, dim3((A.cap + 3) / 4, B), dim3(256), 0, st, pl_l, pl_r, d_lv, A);
    hipLaunchKernelGGL(k_stereo_median, dim3(B), dim3(256), 0, st, A);
}

}  // namespace plp
