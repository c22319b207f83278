// Host driver + C ABI of the ORB extractor (include/plp_front.h).  One plp_orb per
// feature::orb_extractor instance; it owns a HIP stream, the constant tables for the current
// image geometry and the HBM work planes sized for the largest batch seen so far.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "blur_tile.hpp"
#include "orb_device.hpp"
#include "xcd_map.hpp"
#include "plp_common.hpp"
#include "quadtree_model.hpp"

namespace plp {
void launch_quadtree(hipStream_t st, const LevelDev* d_lv, int n_levels, int n_cells_total, const uint32_t* cell_cand,
                     const int32_t* cell_count, int32_t* sel, int32_t* sel_count, int total_sel_cap, uint32_t* qt_scratch,
                     size_t qt_scratch_frame_stride, int32_t* status, int B, int max_quota);
size_t quadtree_scratch_bytes_per_frame(const LevelDev* h_lv, int n_levels);
}

using namespace plp;

struct plp_orb {
    HostPinned pin;            // staging of the host-pointer entry point's image and mask
    plp_orb_params p{};
    std::vector<float> rects;
    int device = 0;
    hipStream_t stream = nullptr;
    OrbScaleTables st;
    OrbGeometry geo;           // for (geo.rows, geo.cols); rows==0 -> not built yet
    std::vector<LevelDev> h_lv;
    int total_blur_tiles = 0;
    // constant tables in HBM
    DevBuf d_lv, d_cells, d_rs;
    ResizeDev rs{};
    // work planes (capB frames)
    int capB = 0;
    DevBuf pyr, blur, l0copy, cell_cand, cell_count, sel, sel_count, status, qt_scratch, d_mask;
    size_t l0copy_frame_stride = 0, qt_frame_stride = 0;
    int max_quota = 0;
    // single-frame host API staging
    DevBuf s_kps, s_desc, s_counts;
    DevBuf stereo_corr, stereo_stage;
    int s_cap = 0;
    // reference state: rectangle mask is created once, at the first frame's size
    bool mask_is_initialized = false;
    std::vector<uint8_t> rect_mask;
    int rect_mask_rows = 0, rect_mask_cols = 0;
    bool rect_mask_uploaded = false;
    DevBuf d_rect_mask;
    // last batch
    int last_B = 0;
    OrbPlanes last_planes{};
    hipStream_t last_stream = nullptr;
    // per-stage HIP-event timing (profiling mode only; makes every batch synchronous)
    bool profiling = false;
    hipEvent_t ev[8] = {};
    double stage_ms[7] = {0, 0, 0, 0, 0, 0, 0};   // l0 copy, pyramid, FAST, blur, quadtree, orient+rBRIEF, total
    long stage_batches = 0;
    std::mutex mu;
};

namespace {

plp_status validate(const plp_orb_params* p) {
    if (!p) return set_error(PLP_ERR_INVALID_ARG, "params is NULL");
    if (p->num_levels == 0 || p->num_levels > (uint32_t)kMaxLevels) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
    if (!(p->scale_factor > 1.0f)) return set_error(PLP_ERR_INVALID_ARG, "scale_factor must be > 1");
    if (p->n_mask_rects < 0 || (p->n_mask_rects > 0 && !p->mask_rects)) return set_error(PLP_ERR_INVALID_ARG, "mask_rects");
    for (int i = 0; i < p->n_mask_rects; ++i) {   // orb_params.cc:40-54
        const float* v = p->mask_rects + 4 * i;
        if (v[0] >= v[1]) return set_error(PLP_ERR_INVALID_ARG, "x_max must be greater than x_min");
        if (v[2] >= v[3]) return set_error(PLP_ERR_INVALID_ARG, "y_max must be greater than x_min");
    }
    return PLP_OK;
}

void reinitialize(plp_orb* c) {   // orb_extractor::initialize()
    c->st = make_scale_tables(c->p.max_num_keypts, c->p.scale_factor, c->p.num_levels);
    c->geo = OrbGeometry{};   // force a geometry rebuild at the next frame
}

plp_status build_geometry_impl(plp_orb* c, int rows, int cols);

// A failed build must not leave a half-initialised geometry behind: the next call at the same size would pass the early-out
// below and launch kernels on tables that were never uploaded.
plp_status build_geometry(plp_orb* c, int rows, int cols) {
    if (c->geo.rows == rows && c->geo.cols == cols && c->geo.n_levels == (int)c->p.num_levels) return PLP_OK;
    const plp_status s = build_geometry_impl(c, rows, cols);
    if (s != PLP_OK) { c->geo = OrbGeometry{}; c->h_lv.clear(); c->capB = 0; }
    return s;
}

plp_status build_geometry_impl(plp_orb* c, int rows, int cols) {
    const int nl = (int)c->p.num_levels;
    // every level must keep a non-empty FAST region; the reference underflows (UB) otherwise
    {
        OrbGeometry g = make_geometry(rows, cols, c->st, nl);
        for (int l = 0; l < nl; ++l)
            if (g.lv[l].w <= 2 * kOrbBorder + 6 || g.lv[l].h <= 2 * kOrbBorder + 6)
                return set_error(PLP_ERR_INVALID_ARG, "image too small for the requested pyramid (level narrower than 45 px)");
        if (cols - 2 * kOrbBorder > 4095 || rows - 2 * kOrbBorder > 4095)
            return set_error(PLP_ERR_UNSUPPORTED, "images larger than 4133 px are not supported (12-bit candidate packing)");
        c->geo = std::move(g);
    }
    c->h_lv.assign(nl, LevelDev{});
    c->total_blur_tiles = 0;
    size_t qt_off = 0;
    int max_cells = 0;
    for (int l = 0; l < nl; ++l) {
        const LevelGeom& G = c->geo.lv[l];
        LevelDev& L = c->h_lv[l];
        L.w = G.w; L.h = G.h; L.pitch = G.pitch; L.off = G.off;
        L.blur_tiles = ((G.w + 127) / 128) * ((G.h + plp::kBlurTH - 1) / plp::kBlurTH);
        L.blur_tiles_x_magic = plp::plp_div_magic((uint32_t)((G.w + 127) / 128), (uint64_t)L.blur_tiles);
        c->total_blur_tiles += L.blur_tiles;
        L.scale = c->st.sf[l];
        L.sel_base = G.sel_base; L.sel_cap = G.sel_cap;
        L.cell_base = G.cell_base; L.n_cells = G.n_cell_rows * G.n_cell_cols;
        L.quota = (int)c->st.quota[l];
        c->max_quota = l == 0 ? L.quota : std::max(c->max_quota, L.quota);
        L.n_init_x = G.n_init_x; L.delta_x = G.delta_x; L.delta_y = G.delta_y;
        // quadtree scratch + the key bits that can differ between two candidates of this level
        L.qt_cap = std::min(std::max(L.n_cells, 1) * kCellCap, 65535);   // FAST's own bound, capped by the u16 node ranges
        L.qt_off = qt_off;
        qt_off += (size_t)L.qt_cap * 20;   // cand, key, two index arrays, radix counters of very large levels
        const int extent = (int)std::ceil(std::max(G.delta_x, G.delta_y)) + 1;
        int t1 = 0;
        while ((1 << t1) < extent) ++t1;
        const int d_eff = std::min(t1 + 2, kQtDepth);            // splits until every rectangle is resolved (+1 slack)
        const int n_init = G.n_init_x * G.n_init_y;
        int nb = 0;
        while ((1 << nb) < n_init) ++nb;
        L.sort_lo = 2 * (kQtDepth - d_eff);
        L.sort_hi = 2 * kQtDepth + nb;
        if (L.n_cells > 2048 || 3 * L.quota + 8 > kQtMaxNodesLdsSmallBlock || n_init > 32)
            return set_error(PLP_ERR_UNSUPPORTED, "frame/keypoint budget exceeds the quadtree kernel limits (quota per level <= 2010, 2048 cells per level, 32 initial nodes)");
        max_cells = std::max(max_cells, L.n_cells);
    }
    if (3 * c->max_quota + 8 > kQtMaxNodesLds && max_cells > 1024)   // a quota above 1960 halves the kernel's cell-prefix block (quadtree_kernel.hip kQtBlkWordsSmall)
        return set_error(PLP_ERR_UNSUPPORTED, "frame/keypoint budget exceeds the quadtree kernel limits (a quota above 1960 per level needs levels of at most 1024 cells)");
    PLP_HIP(c->d_lv.upload(c->h_lv.data(), sizeof(LevelDev) * nl, c->stream));
    PLP_HIP(c->d_cells.upload(c->geo.cells.data(), sizeof(CellDesc) * c->geo.cells.size(), c->stream));
    // resize tables: 8 int16 arrays back to back
    const ResizeTables& R = c->geo.rs;
    const size_t nc = R.xofs0.size(), nr = R.yofs0.size();
    std::vector<int16_t> blob;
    blob.reserve(4 * nc + 4 * nr + 8);
    auto push = [&](const std::vector<int16_t>& v) { size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return o; };
    const size_t o0 = push(R.xofs0), o1 = push(R.xofs1), o2 = push(R.a0), o3 = push(R.a1);
    const size_t o4 = push(R.yofs0), o5 = push(R.yofs1), o6 = push(R.b0), o7 = push(R.b1);
    if (blob.empty()) blob.push_back(0);
    PLP_HIP(c->d_rs.upload(blob.data(), blob.size() * 2, c->stream));
    const int16_t* base = (const int16_t*)c->d_rs.p;
    c->rs.xofs0 = base + o0; c->rs.xofs1 = base + o1; c->rs.a0 = base + o2; c->rs.a1 = base + o3;
    c->rs.yofs0 = base + o4; c->rs.yofs1 = base + o5; c->rs.b0 = base + o6; c->rs.b1 = base + o7;
    for (int l = 0; l < nl; ++l) { c->rs.col_base[l] = R.col_base[l]; c->rs.row_base[l] = R.row_base[l]; }
    c->capB = 0;   // planes depend on the geometry
    PLP_HIP(hipStreamSynchronize(c->stream));
    return PLP_OK;
}

plp_status ensure_capacity(plp_orb* c, int B) {
    if (B <= c->capB) return PLP_OK;
    const OrbGeometry& g = c->geo;
    const size_t n_cells = g.cells.size();
    c->l0copy_frame_stride = ((size_t)g.lv[0].pitch * g.rows + 255) / 256 * 256;
    c->qt_frame_stride = quadtree_scratch_bytes_per_frame(c->h_lv.data(), g.n_levels);
    PLP_HIP(c->pyr.reserve(g.frame_plane_bytes * B));
    PLP_HIP(c->blur.reserve(g.frame_plane_bytes * B));
    PLP_HIP(c->l0copy.reserve(c->l0copy_frame_stride * B));
    PLP_HIP(c->cell_cand.reserve(sizeof(uint32_t) * kCellCap * n_cells * B));
    PLP_HIP(c->cell_count.reserve(sizeof(int32_t) * n_cells * B));
    PLP_HIP(c->sel.reserve(sizeof(int32_t) * g.total_sel_cap * B));
    PLP_HIP(c->sel_count.reserve(sizeof(int32_t) * kMaxLevels * B));
    PLP_HIP(c->status.reserve(sizeof(int32_t) * 4));
    PLP_HIP(c->qt_scratch.reserve(c->qt_frame_stride * B));
    c->capB = B;
    return PLP_OK;
}

// create_rectangle_mask (orb_extractor.cc:297-313): built once, at the first frame's size
void build_rect_mask(plp_orb* c, int rows, int cols) {
    if (c->mask_is_initialized || c->rects.empty()) return;
    if (c->rect_mask.empty()) { c->rect_mask.assign((size_t)rows * cols, 255); c->rect_mask_rows = rows; c->rect_mask_cols = cols; }
    for (size_t i = 0; i + 3 < c->rects.size(); i += 4) {
        const unsigned x0 = (unsigned)std::round((float)cols * c->rects[i]), x1 = (unsigned)std::round((float)cols * c->rects[i + 1]);
        const unsigned y0 = (unsigned)std::round((float)rows * c->rects[i + 2]), y1 = (unsigned)std::round((float)rows * c->rects[i + 3]);
        for (unsigned y = y0; y <= y1 && y < (unsigned)c->rect_mask_rows; ++y)
            for (unsigned x = x0; x <= x1 && x < (unsigned)c->rect_mask_cols; ++x) c->rect_mask[(size_t)y * c->rect_mask_cols + x] = 0;
    }
    c->mask_is_initialized = true;
    c->rect_mask_uploaded = false;
}

plp_status run_batch(plp_orb* c, const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frame_stride,
                     const uint8_t* d_mask, size_t mask_step, size_t mask_frame_stride, plp_keypoint* d_kps, uint8_t* d_desc,
                     int cap, int32_t* d_counts, hipStream_t st) {
    PLP_HIP(hipSetDevice(c->device));
    PLP_TRY(build_geometry(c, rows, cols));
    PLP_TRY(ensure_capacity(c, B));
    const OrbGeometry& g = c->geo;
    const int nl = g.n_levels;

    const bool prof = c->profiling;
    // PLP_TRACE=1 (diagnostic): every stage boundary waits for the stream and is logged, so that a device fault names its stage
    static const bool trace = getenv("PLP_TRACE") != nullptr;
    auto mark = [&](int i) {
        if (prof) (void)hipEventRecord(c->ev[i], st);
        if (trace) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "plp_orb: stage boundary %d reached (%dx%d, B=%d)%s\n", i, cols, rows, B, e == hipSuccess ? "" : " ERROR"); fflush(stderr); }
    };
    mark(0);
    OrbPlanes pl{};
    pl.pyr = (uint8_t*)c->pyr.p; pl.pyr_frame_stride = g.frame_plane_bytes;
    const bool aligned = ((uintptr_t)d_imgs % 4 == 0) && (step % 4 == 0) && (frame_stride % 4 == 0);
    if (aligned) { pl.l0 = d_imgs; pl.l0_frame_stride = frame_stride; pl.l0_pitch = (int)step; }
    else {
        for (int f = 0; f < B; ++f)
            PLP_HIP(hipMemcpy2DAsync((uint8_t*)c->l0copy.p + f * c->l0copy_frame_stride, g.lv[0].pitch, d_imgs + f * frame_stride, step,
                                     cols, rows, hipMemcpyDeviceToDevice, st));
        pl.l0 = (const uint8_t*)c->l0copy.p; pl.l0_frame_stride = c->l0copy_frame_stride; pl.l0_pitch = g.lv[0].pitch;
    }
    // mask selection (orb_extractor.cc:97-115): image mask, else rectangle mask, else none
    build_rect_mask(c, rows, cols);
    if (!d_mask && !c->rect_mask.empty()) {
        if (c->rect_mask_rows != rows || c->rect_mask_cols != cols)
            return set_error(PLP_ERR_INVALID_ARG, "rectangle mask was built for a different frame size");
        if (!c->rect_mask_uploaded) {
            PLP_HIP(c->d_rect_mask.upload(c->rect_mask.data(), c->rect_mask.size(), st));
            c->rect_mask_uploaded = true;
        }
        d_mask = (const uint8_t*)c->d_rect_mask.p; mask_step = cols; mask_frame_stride = 0;
    }

    PLP_HIP(hipMemsetAsync(c->status.p, 0, 16, st));
    mark(1);
    for (int l = 1; l < nl; ++l) launch_resize(st, pl, c->h_lv.data(), l, B, c->rs);
    mark(2);
    launch_fast(st, pl, (const CellDesc*)c->d_cells.p, (int)g.cells.size(), (const LevelDev*)c->d_lv.p, B, (int)c->p.ini_fast_thr,
                (int)c->p.min_fast_thr, d_mask, mask_step, mask_frame_stride, (uint32_t*)c->cell_cand.p, (int32_t*)c->cell_count.p);
    mark(3);
    BlurTaps taps{{18, 34, 48, 56, 48, 34, 18}};   // 7 taps, sigma 2, 8.8 fixed point, sum 256
    launch_blur(st, pl, (uint8_t*)c->blur.p, g.frame_plane_bytes, (const LevelDev*)c->d_lv.p, nl, c->total_blur_tiles, B, taps, c->h_lv.data());
    mark(4);
    launch_quadtree(st, (const LevelDev*)c->d_lv.p, nl, (int)g.cells.size(), (const uint32_t*)c->cell_cand.p,
                    (const int32_t*)c->cell_count.p, (int32_t*)c->sel.p, (int32_t*)c->sel_count.p, g.total_sel_cap,
                    (uint32_t*)c->qt_scratch.p, c->qt_frame_stride, (int32_t*)c->status.p, B, c->max_quota);
    mark(5);
    UMax um;
    for (int v = 0; v <= kHalfPatch; ++v) um.v[v] = c->st.u_max[v];
    launch_orient_rbrief(st, pl, (const uint8_t*)c->blur.p, g.frame_plane_bytes, (const LevelDev*)c->d_lv.p, nl, (const int32_t*)c->sel.p,
                         (const int32_t*)c->sel_count.p, g.total_sel_cap, um, d_kps, d_desc, cap, d_counts, (int32_t*)c->status.p, B);
    PLP_HIP(hipGetLastError());
    if (trace) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "plp_orb: batch done%s\n", e == hipSuccess ? "" : " ERROR"); fflush(stderr); }
    if (prof) {
        mark(6);
        PLP_HIP(hipEventSynchronize(c->ev[6]));
        for (int i = 0; i < 6; ++i) { float ms = 0; PLP_HIP(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1])); c->stage_ms[i] += ms; }
        float tot = 0; PLP_HIP(hipEventElapsedTime(&tot, c->ev[0], c->ev[6])); c->stage_ms[6] += tot;
        ++c->stage_batches;
    }
    c->last_B = B; c->last_planes = pl; c->last_stream = st;
    return PLP_OK;
}

}  // namespace

extern "C" {

void plp_orb_default_params(plp_orb_params* p) {
    if (!p) return;
    p->max_num_keypts = 2000; p->scale_factor = 1.2f; p->num_levels = 8; p->ini_fast_thr = 20; p->min_fast_thr = 7;
    p->mask_rects = nullptr; p->n_mask_rects = 0;
}

plp_status plp_orb_create(const plp_orb_params* params, int device, plp_orb** out) {
    if (!out) return set_error(PLP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    PLP_TRY(validate(params));
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return set_error(PLP_ERR_NO_DEVICE, "no HIP device visible: the ORB path has no CPU fallback");
    if (device < 0 || device >= n) return set_error(PLP_ERR_INVALID_ARG, "device index out of range");
    PLP_HIP(hipSetDevice(device));
    plp_orb* c = new plp_orb();
    c->p = *params;
    c->rects.assign(params->mask_rects, params->mask_rects + 4 * (size_t)params->n_mask_rects);
    c->p.mask_rects = nullptr;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_error(PLP_ERR_HIP, "hipStreamCreate failed"); }
    reinitialize(c);
    *out = c;
    return PLP_OK;
}

void plp_orb_destroy(plp_orb* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

plp_status plp_orb_set_param(plp_orb* c, plp_orb_param_id id, double v) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    switch (id) {
        case PLP_ORB_MAX_NUM_KEYPOINTS: c->p.max_num_keypts = (uint32_t)v; reinitialize(c); break;
        case PLP_ORB_SCALE_FACTOR: if (!(v > 1.0)) return set_error(PLP_ERR_INVALID_ARG, "scale_factor must be > 1"); c->p.scale_factor = (float)v; reinitialize(c); break;
        case PLP_ORB_NUM_SCALE_LEVELS: if (v < 1 || v > kMaxLevels) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]"); c->p.num_levels = (uint32_t)v; reinitialize(c); break;
        case PLP_ORB_INITIAL_FAST_THRESHOLD: c->p.ini_fast_thr = (uint32_t)v; break;   // no initialize(), as the reference
        case PLP_ORB_MINIMUM_FAST_THRESHOLD: c->p.min_fast_thr = (uint32_t)v; break;
        default: return set_error(PLP_ERR_INVALID_ARG, "unknown parameter id");
    }
    return PLP_OK;
}

plp_status plp_orb_get_param(const plp_orb* c, plp_orb_param_id id, double* v) {
    if (!c || !v) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    switch (id) {
        case PLP_ORB_MAX_NUM_KEYPOINTS: *v = c->p.max_num_keypts; break;
        case PLP_ORB_SCALE_FACTOR: *v = c->p.scale_factor; break;
        case PLP_ORB_NUM_SCALE_LEVELS: *v = c->p.num_levels; break;
        case PLP_ORB_INITIAL_FAST_THRESHOLD: *v = c->p.ini_fast_thr; break;
        case PLP_ORB_MINIMUM_FAST_THRESHOLD: *v = c->p.min_fast_thr; break;
        default: return set_error(PLP_ERR_INVALID_ARG, "unknown parameter id");
    }
    return PLP_OK;
}

plp_status plp_orb_get_tables(const plp_orb* c, int32_t* n_levels, float* sf, float* isf, float* s2, float* is2, uint32_t* quota) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    const size_t n = c->p.num_levels;
    if (n_levels) *n_levels = (int32_t)n;
    if (sf) memcpy(sf, c->st.sf.data(), n * 4);
    if (isf) memcpy(isf, c->st.isf.data(), n * 4);
    if (s2) memcpy(s2, c->st.sigma2.data(), n * 4);
    if (is2) memcpy(is2, c->st.isigma2.data(), n * 4);
    if (quota) memcpy(quota, c->st.quota.data(), n * 4);
    return PLP_OK;
}

plp_status plp_orb_extract_batch_device(plp_orb* c, const uint8_t* d_imgs, int32_t B, int32_t rows, int32_t cols, size_t step,
                                        size_t frame_stride, const uint8_t* d_mask, size_t mask_step, size_t mask_frame_stride,
                                        plp_keypoint* d_kps, uint8_t* d_desc, int32_t cap, int32_t* d_counts, void* hip_stream) {
    if (!c || !d_imgs || !d_kps || !d_desc || !d_counts) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0 || rows <= 0 || cols <= 0 || cap <= 0 || step < (size_t)cols) return set_error(PLP_ERR_INVALID_ARG, "bad batch geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    return run_batch(c, d_imgs, B, rows, cols, step, frame_stride, d_mask, mask_step, mask_frame_stride, d_kps, d_desc, cap, d_counts,
                     (hipStream_t)hip_stream);
}

plp_status plp_orb_last_batch_status(plp_orb* c) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B) return PLP_OK;
    PLP_HIP(hipSetDevice(c->device));
    int32_t s[4] = {0, 0, 0, 0};
    PLP_HIP(hipMemcpyAsync(s, c->status.p, 16, hipMemcpyDeviceToHost, c->last_stream));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    if (s[0] & 1) return set_error(PLP_ERR_CAPACITY, "a frame produced more key points than `cap`; output truncated");
    if (s[0] & 8) return set_error(PLP_ERR_HIP, "internal consistency check failed in the ORB kernels (please report the frame)");
    if (s[0] & 2) return set_error(PLP_ERR_OVERFLOW, "per-level candidate scratch overflow");
    return PLP_OK;
}

plp_status plp_orb_extract(plp_orb* c, const uint8_t* img, int32_t rows, int32_t cols, size_t step, const uint8_t* mask,
                           size_t mask_step, plp_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    if (!img || rows <= 0 || cols <= 0) return PLP_OK;   // in_image.empty(): silent return (orb_extractor.cc:76-79)
    if (!kps || !desc || !n_out || cap < 0 || step < (size_t)cols) return set_error(PLP_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    PLP_TRY(build_geometry(c, rows, cols));
    PLP_TRY(ensure_capacity(c, 1));
    hipStream_t st = c->stream;
    const int need = 2 * (int)c->p.max_num_keypts + 4 * kMaxLevels;   // the tree may overshoot each quota by < 2x
    if (c->s_cap < need) {
        PLP_HIP(c->s_kps.reserve(sizeof(plp_keypoint) * need));
        PLP_HIP(c->s_desc.reserve(32 * (size_t)need));
        PLP_HIP(c->s_counts.reserve(16));
        c->s_cap = need;
    }
    // image (and mask) through the page-locked staging buffer into the aligned level-0 plane
    const size_t img_bytes = (size_t)rows * cols;
    PLP_HIP(c->pin.reserve(img_bytes * (mask ? 2 : 1)));
    c->pin.pack(0, img, step, rows, cols);
    PLP_HIP(hipMemcpy2DAsync(c->l0copy.p, c->geo.lv[0].pitch, c->pin.p, cols, cols, rows, hipMemcpyHostToDevice, st));
    const uint8_t* d_mask = nullptr;
    if (mask) {
        PLP_HIP(c->d_mask.reserve(img_bytes));
        c->pin.pack(img_bytes, mask, mask_step, rows, cols);
        PLP_HIP(hipMemcpyAsync(c->d_mask.p, static_cast<const uint8_t*>(c->pin.p) + img_bytes, img_bytes, hipMemcpyHostToDevice, st));
        d_mask = (const uint8_t*)c->d_mask.p;
    }
    PLP_TRY(run_batch(c, (const uint8_t*)c->l0copy.p, 1, rows, cols, c->geo.lv[0].pitch, c->l0copy_frame_stride, d_mask, cols, 0,
                      (plp_keypoint*)c->s_kps.p, (uint8_t*)c->s_desc.p, c->s_cap, (int32_t*)c->s_counts.p, st));
    int32_t n = 0;
    PLP_HIP(hipMemcpyAsync(&n, c->s_counts.p, 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    *n_out = n;
    if (n > cap) return set_error(PLP_ERR_CAPACITY, "caller buffers too small");
    if (n > 0) {
        // results come back through the page-locked buffer as well
        const size_t b_kps = sizeof(plp_keypoint) * (size_t)n, b_desc = 32 * (size_t)n;
        PLP_HIP(c->pin.reserve(b_kps + b_desc));
        uint8_t* hp = static_cast<uint8_t*>(c->pin.p);
        PLP_HIP(hipMemcpyAsync(hp, c->s_kps.p, b_kps, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipMemcpyAsync(hp + b_kps, c->s_desc.p, b_desc, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipStreamSynchronize(st));
        memcpy(kps, hp, b_kps); memcpy(desc, hp + b_kps, b_desc);
    }
    int32_t s[4];
    PLP_HIP(hipMemcpy(s, c->status.p, 16, hipMemcpyDeviceToHost));
    if (s[0] & 8) return set_error(PLP_ERR_HIP, "internal consistency check failed in the ORB kernels (please report the frame)");
    if (s[0] & 2) return set_error(PLP_ERR_OVERFLOW, "per-level candidate scratch overflow");
    return PLP_OK;
}

plp_status plp_orb_set_profiling(plp_orb* c, int32_t enable) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    if (enable && !c->ev[0]) for (auto& e : c->ev) PLP_HIP(hipEventCreate(&e));
    c->profiling = enable != 0;
    for (auto& v : c->stage_ms) v = 0;
    c->stage_batches = 0;
    return PLP_OK;
}

plp_status plp_orb_get_stage_times(plp_orb* c, double* ms7, int64_t* n_batches) {
    if (!c || !ms7) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    for (int i = 0; i < 7; ++i) ms7[i] = c->stage_ms[i];
    if (n_batches) *n_batches = c->stage_batches;
    return PLP_OK;
}

plp_status plp_orb_pyramid_level_size(const plp_orb* c, int32_t level, int32_t* rows, int32_t* cols) {
    if (!c || !rows || !cols) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (c->geo.rows == 0 || level < 0 || level >= c->geo.n_levels) return set_error(PLP_ERR_INVALID_ARG, "no frame processed yet / bad level");
    *rows = c->geo.lv[level].h; *cols = c->geo.lv[level].w;
    return PLP_OK;
}

plp_status plp_orb_pyramid_host(plp_orb* c, int32_t frame, int32_t level, uint8_t* dst, size_t dst_step) {
    if (!c || !dst) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B || frame < 0 || frame >= c->last_B || level < 0 || level >= c->geo.n_levels) return set_error(PLP_ERR_INVALID_ARG, "bad frame/level");
    PLP_HIP(hipSetDevice(c->device));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    const LevelDev& L = c->h_lv[level];
    PLP_HIP(hipMemcpy2D(dst, dst_step, c->last_planes.level_ptr(frame, level, L), c->last_planes.level_pitch(level, L), L.w, L.h,
                        hipMemcpyDeviceToHost));
    return PLP_OK;
}

plp_status plp_orb_debug_read(plp_orb* c, plp_orb_debug_id what, int32_t frame, int32_t level, void* dst, size_t dst_bytes, int64_t* n_out) {
    if (!c || !dst || !n_out) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B || frame < 0 || frame >= c->last_B || level < 0 || level >= c->geo.n_levels) return set_error(PLP_ERR_INVALID_ARG, "bad frame/level");
    PLP_HIP(hipSetDevice(c->device));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    const LevelDev& L = c->h_lv[level];
    const OrbGeometry& g = c->geo;
    if (what == PLP_ORB_DBG_BLURRED) {
        if (dst_bytes < (size_t)L.w * L.h) return set_error(PLP_ERR_CAPACITY, "dst too small");
        PLP_HIP(hipMemcpy2D(dst, L.w, (const uint8_t*)c->blur.p + (size_t)frame * g.frame_plane_bytes + L.off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
        *n_out = (int64_t)L.w * L.h;
        return PLP_OK;
    }
    int32_t* o = (int32_t*)dst;
    const size_t cap = dst_bytes / 12;
    size_t n = 0;
    if (what == PLP_ORB_DBG_CANDIDATES) {
        const size_t n_cells = g.cells.size();
        std::vector<int32_t> cnt(L.n_cells);
        PLP_HIP(hipMemcpy(cnt.data(), (const int32_t*)c->cell_count.p + (size_t)frame * n_cells + L.cell_base, 4 * (size_t)L.n_cells, hipMemcpyDeviceToHost));
        std::vector<uint32_t> buf(kCellCap);
        for (int k = 0; k < L.n_cells; ++k) {
            if (!cnt[k]) continue;
            PLP_HIP(hipMemcpy(buf.data(), (const uint32_t*)c->cell_cand.p + ((size_t)frame * n_cells + L.cell_base + k) * kCellCap, 4 * (size_t)cnt[k], hipMemcpyDeviceToHost));
            for (int i = 0; i < cnt[k]; ++i, ++n) {
                if (n >= cap) return set_error(PLP_ERR_CAPACITY, "dst too small");
                o[3 * n] = buf[i] & 0xfff; o[3 * n + 1] = (buf[i] >> 12) & 0xfff; o[3 * n + 2] = buf[i] >> 24;
            }
        }
    } else if (what == PLP_ORB_DBG_SELECTED) {
        int32_t cnt = 0;
        PLP_HIP(hipMemcpy(&cnt, (const int32_t*)c->sel_count.p + (size_t)frame * kMaxLevels + level, 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> buf(cnt > 0 ? cnt : 1);
        if (cnt) PLP_HIP(hipMemcpy(buf.data(), (const int32_t*)c->sel.p + (size_t)frame * g.total_sel_cap + L.sel_base, 4 * (size_t)cnt, hipMemcpyDeviceToHost));
        for (int i = 0; i < cnt; ++i, ++n) {
            if (n >= cap) return set_error(PLP_ERR_CAPACITY, "dst too small");
            o[3 * n] = buf[i] & 0xfff; o[3 * n + 1] = (buf[i] >> 12) & 0xfff; o[3 * n + 2] = buf[i] >> 24;
        }
    } else return set_error(PLP_ERR_INVALID_ARG, "unknown debug id");
    *n_out = (int64_t)n;
    return PLP_OK;
}

// match::stereo(left pyramid, right pyramid, keypts, descriptors, scale tables, fx*baseline, baseline).compute(...)
// (reference src/PLPSLAM/match/stereo.cc:30-150; constructed in data/frame.cc:277-281).  The pyramids are the ones the
// two extractors built in their LAST extract / extract_batch call (the reference reads orb_extractor::image_pyramid_).
static plp_status stereo_run(plp_orb* l, plp_orb* r, const StereoArgs& A0, int B, hipStream_t st) {
    if (!l->last_B || !r->last_B || B > l->last_B || B > r->last_B) return set_error(PLP_ERR_INVALID_ARG, "both extractors must have processed the frames first");
    if (l->geo.rows != r->geo.rows || l->geo.cols != r->geo.cols || l->geo.n_levels != r->geo.n_levels) return set_error(PLP_ERR_INVALID_ARG, "left/right geometry differs");
    StereoArgs A = A0;
    for (int i = 0; i < kMaxLevels; ++i) A.inv_scale[i] = i < (int)l->st.isf.size() ? l->st.isf[i] : 1.0f;
    PLP_HIP(l->stereo_corr.reserve((size_t)B * A.cap * 4));
    A.corr = (int32_t*)l->stereo_corr.p;
    launch_stereo(st, l->last_planes, r->last_planes, (const LevelDev*)l->d_lv.p, A, B);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_stereo_compute_batch_device(plp_orb* left, plp_orb* right, const plp_keypoint* d_kps_l, const int32_t* d_cnt_l,
                                           const plp_keypoint* d_kps_r, const int32_t* d_cnt_r, const uint8_t* d_desc_l, const uint8_t* d_desc_r,
                                           int32_t cap, int32_t B, float focal_x_baseline, float true_baseline, float* d_x_right, float* d_depths,
                                           void* hip_stream) {
    if (!left || !right || !d_kps_l || !d_kps_r || !d_desc_l || !d_desc_r || !d_x_right || !d_depths) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (cap <= 0 || cap > 65535 || B <= 0) return set_error(PLP_ERR_INVALID_ARG, "bad sizes");
    std::lock_guard<std::mutex> lk(left->mu);
    PLP_HIP(hipSetDevice(left->device));
    StereoArgs A{};
    A.kps_l = d_kps_l; A.kps_r = d_kps_r; A.desc_l = d_desc_l; A.desc_r = d_desc_r; A.cnt_l = d_cnt_l; A.cnt_r = d_cnt_r; A.cap = cap;
    A.fxb = focal_x_baseline; A.tb = true_baseline; A.x_right = d_x_right; A.depth = d_depths;
    return stereo_run(left, right, A, B, (hipStream_t)hip_stream);
}

plp_status plp_stereo_compute(plp_orb* left, plp_orb* right, const plp_keypoint* kps_l, int32_t n_l, const plp_keypoint* kps_r, int32_t n_r,
                              const uint8_t* desc_l, const uint8_t* desc_r, float focal_x_baseline, float true_baseline, float* stereo_x_right,
                              float* depths) {
    if (!left || !right || !stereo_x_right || !depths) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (n_l <= 0) return PLP_OK;
    if (n_r < 0 || (n_r > 0 && (!kps_r || !desc_r)) || !kps_l || !desc_l) return set_error(PLP_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(left->mu);
    PLP_HIP(hipSetDevice(left->device));
    hipStream_t st = left->stream;
    PLP_HIP(hipStreamSynchronize(right->last_stream ? right->last_stream : right->stream));
    const int cap = std::max(n_l, std::max(n_r, 1));
    const size_t bk = ((size_t)cap * sizeof(plp_keypoint) + 255) / 256 * 256, bd = ((size_t)cap * 32 + 255) / 256 * 256, bf = ((size_t)cap * 4 + 255) / 256 * 256;
    PLP_HIP(left->stereo_stage.reserve(2 * bk + 2 * bd + 2 * bf + 512));
    uint8_t* base = (uint8_t*)left->stereo_stage.p;
    int32_t cnts[2] = {n_l, n_r};
    PLP_HIP(hipMemcpyAsync(base, kps_l, (size_t)n_l * sizeof(plp_keypoint), hipMemcpyHostToDevice, st));
    if (n_r) PLP_HIP(hipMemcpyAsync(base + bk, kps_r, (size_t)n_r * sizeof(plp_keypoint), hipMemcpyHostToDevice, st));
    PLP_HIP(hipMemcpyAsync(base + 2 * bk, desc_l, (size_t)n_l * 32, hipMemcpyHostToDevice, st));
    if (n_r) PLP_HIP(hipMemcpyAsync(base + 2 * bk + bd, desc_r, (size_t)n_r * 32, hipMemcpyHostToDevice, st));
    uint8_t* d_cnt = base + 2 * bk + 2 * bd + 2 * bf;
    PLP_HIP(hipMemcpyAsync(d_cnt, cnts, 8, hipMemcpyHostToDevice, st));
    StereoArgs A{};
    A.kps_l = (const plp_keypoint*)base; A.kps_r = (const plp_keypoint*)(base + bk);
    A.desc_l = base + 2 * bk; A.desc_r = base + 2 * bk + bd;
    A.cnt_l = (const int32_t*)d_cnt; A.cnt_r = (const int32_t*)d_cnt + 1; A.cap = cap;
    A.fxb = focal_x_baseline; A.tb = true_baseline;
    A.x_right = (float*)(base + 2 * bk + 2 * bd); A.depth = (float*)(base + 2 * bk + 2 * bd + bf);
    PLP_TRY(stereo_run(left, right, A, 1, st));
    PLP_HIP(hipMemcpyAsync(stereo_x_right, A.x_right, (size_t)n_l * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipMemcpyAsync(depths, A.depth, (size_t)n_l * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    return PLP_OK;
}

// Host model of the quadtree kernel, callable without a GPU (tests/test_quadtree_model.py).
// xys: n int32 triples (x, y, score), border-relative; level_w/level_h: size of the pyramid level.
int32_t plp_model_quadtree_host(const int32_t* xys, int32_t n, int32_t level_w, int32_t level_h, uint32_t quota, int32_t* out_idx) {
    OrbScaleTables st = make_scale_tables(quota, 1.2f, 1);
    OrbGeometry g = make_geometry(level_h, level_w, st, 1);
    std::vector<QtCand> cs(n);
    for (int i = 0; i < n; ++i) cs[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
    std::vector<int> pick = quadtree_select(cs.data(), n, g.lv[0], quota);
    for (size_t i = 0; i < pick.size(); ++i) out_idx[i] = pick[i];
    return (int32_t)pick.size();
}

}  // extern "C"
