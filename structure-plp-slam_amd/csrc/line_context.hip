// Host driver + C ABI of the line front-end (include/plp_front.h): LSD + LBD, batched over frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "line_device.hpp"
#include "seed_sort_model.hpp"
#include "sincos_ziv.hpp"
#include "plp_common.hpp"

using namespace plp;

struct plp_line {
    HostPinned pin;            // staging of the host-pointer entry point's image
    int device = 0;
    hipStream_t stream = nullptr;
    int rows = 0, cols = 0, capB = 0;
    int grow_waves = 0;   // plp_line_set_grow_waves
    int seed_order = PLP_SEED_ORDER_LIBSTDCXX;   // plp_line_set_seed_order.  The reference's order is the default on every device; one that refuses the sort's LDS
                                                 // (seed_sort_ok false) gets an error from extract until the caller selects PLP_SEED_ORDER_STABLE -- never a silent change of results
    bool grow_on_side = false;                 // PLP_GROW_CUS: region growing on the CU-masked side stream
    bool mw_ok = false, seed_sort_ok = false;  // this device accepted the large dynamic-LDS limits of k_lsd_grow_mw / k_lsd_seed_sort
    int mw_capB = 0, seed_capB = 0;            // frames the lazily allocated buffers of those two paths hold
    LinePlanes P{};
    LsdParams lp{};
    ResizeExactTab rt{};
    BlurTapsN t11{}, t5{};
    LbdWeightsDev w{};
    DevBuf tabs, blur11, scaled, pix, g2, maxgrad, undef, order, n_order, reg, mw_heap, seed_ent, seed_ws, raw, n_raw, dx, dy, all_kl, all_kl_dir, all_lbd, n_all, status, prof, grow_stats;
    DevBuf l0copy, s_kl, s_lbd, s_fn, s_cnt;   // host-API staging
    DevBuf aligned;                             // aligned copy of odd-pitch device frames
    int s_cap = 0;
    int last_B = 0;
    bool last_profiled = false;   // the last batch ran with profiling on: only then does `prof` hold that batch's counters
    hipStream_t last_stream = nullptr;
    bool profiling = false;
    bool grow_big_ok = false;
    hipEvent_t ev[9] = {};
    LineSideStream side{};                      // blur5 + Sobel beside the LSD chain
    double stage_ms[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // 8 stages + total
    long stage_batches = 0;
    std::mutex mu;
};

namespace {

// 8.8 fixed-point Gaussian taps that sum to 256 (edge -> centre error diffusion, centre takes the remainder)
void gaussian_taps(int n, double sigma, int* out) {
    std::vector<double> k(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = std::exp(-0.5 / (sigma * sigma) * x * x); sum += k[i]; }
    double err = 0;
    int acc = 0;
    for (int i = 0; i < n / 2; ++i) {
        const double adj = k[i] / sum * 256.0 + err;
        const int v = (int)std::nearbyint(adj);
        err = adj - v;
        out[i] = out[n - 1 - i] = v;
        acc += v;
    }
    out[n / 2] = 256 - 2 * acc;
}

int floor_d(double v) { int i = (int)v; return i - (i > v); }

void exact_coeffs(int ssize, int dsize, std::vector<int16_t>& ofs, std::vector<int16_t>& c1) {
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        const double val = ((double)d + 0.5) * scale - 0.5;
        const int iv = floor_d(val);
        if (iv >= 0 && ssize > 1) {
            if (iv < ssize - 1) { ofs.push_back((int16_t)iv); c1.push_back((int16_t)std::nearbyint((val - (double)iv) * 256.0)); }
            else { ofs.push_back((int16_t)(ssize - 1)); c1.push_back(-1); }
        } else { ofs.push_back(0); c1.push_back(-2); }
    }
}

plp_status build(plp_line* c, int rows, int cols) {
    if (c->rows == rows && c->cols == cols) return PLP_OK;
    if (rows < 16 || cols < 16 || rows > 16000 || cols > 16000) return set_error(PLP_ERR_INVALID_ARG, "unsupported frame size for the line front-end");
    LinePlanes& P = c->P;
    P.W = cols; P.H = rows;
    P.sw = (int)std::nearbyint(cols * 0.5); P.sh = (int)std::nearbyint(rows * 0.5);   // cvRound(ssize * scale)
    // idx / sw as mulhi(idx, ceil(2^32 / sw)): exact while idx * sw < 2^32, i.e. (idx < sw * sh <= kLsdMaxScaledPixels < 2^19) for sw <= 8192; 0 = divide
    P.sw_magic = (P.sw >= 2 && P.sw <= 8192) ? (uint32_t)(((1ull << 32) + (uint64_t)P.sw - 1) / (uint64_t)P.sw) : 0u;
    if (P.sw >= 65536 || P.sh >= 65536 || (size_t)P.sw * P.sh > kLsdMaxScaledPixels || (!c->grow_big_ok && (size_t)P.sw * P.sh > 516065))
        return set_error(PLP_ERR_UNSUPPORTED, "frame too large for the LSD region-growing kernel (the half-resolution image must not exceed 524,257 pixels: a 1920 x 1080 frame is the largest common one)");
    P.pitch = (cols + 63) / 64 * 64; P.spitch = (P.sw + 63) / 64 * 64;
    // LSD constants (line_extractor.cc:113-122, lsd.cpp flsd)
    LsdParams& lp = c->lp;
    const double ang_th = 22.5, quant = 2.0;
    lp.prec = M_PI * ang_th / 180; lp.p = ang_th / 180; lp.rho = quant / std::sin(lp.prec);
    lp.c_pass = (float)std::cos(lp.prec - kLsdAngleBand); lp.c_fail = (float)std::cos(lp.prec + kLsdAngleBand);
    {   // lsd.cpp ll_angle: a pixel is undefined when norm = sqrt((gx^2 + gy^2) / 4.0) <= rho; gx^2 + gy^2 is an integer <= 2 * 510^2
        uint32_t g = 0;
        while (g < 600000u && std::sqrt((double)g / 4.0) <= lp.rho) ++g;
        lp.g2_def_min = g;
    }
    lp.density_th = 0.6; lp.scale = 0.5; lp.n_bins = 1024; lp.refine = 1;
    const double LOG_NT = 5 * (std::log10((double)P.sw) + std::log10((double)P.sh)) / 2 + std::log10(11.0);
    lp.min_reg_size = (int)(size_t)(-LOG_NT / std::log10(lp.p));
    lp.min_length = (float)(0.125 * std::min(cols, rows));
    lp.keep_length = 60.f;
    const double sigma = 0.6 / 0.5;
    const unsigned h = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
    if (h != 5) return set_error(PLP_ERR_UNSUPPORTED, "unexpected LSD kernel size");
    gaussian_taps(11, sigma, c->t11.k);
    for (int i = 0; i < 11; ++i) c->t5.k[i] = 0;
    gaussian_taps(5, 1.0, c->t5.k);
    // LBD weights (binary_descriptor_custom.cpp:217-258; integer divisions as in the reference)
    {
        double u = (7 * 3 - 1) / 2, sg = (7 * 2 + 1) / 2, inv = -1 / (2 * sg * sg);
        for (int i = 0; i < 21; ++i) { const double d = i - u; c->w.l[i] = (float)std::exp(d * d * inv); }
        u = (9 * 7 - 1) / 2; sg = u; inv = -1 / (2 * sg * sg);
        for (int i = 0; i < 63; ++i) { const double d = i - u; c->w.g[i] = (float)std::exp(d * d * inv); }
    }
    std::vector<int16_t> xo, xc, yo, yc;
    exact_coeffs(cols, P.sw, xo, xc);
    exact_coeffs(rows, P.sh, yo, yc);
    {   // the plain x0.5 case: both kernels of the LSD front as one (k_blur_half)
        bool plain = cols == 2 * P.sw && rows == 2 * P.sh;
        for (int i = 0; plain && i < P.sw; ++i) plain = xo[i] == 2 * i && xc[i] == 128;
        for (int i = 0; plain && i < P.sh; ++i) plain = yo[i] == 2 * i && yc[i] == 128;
        P.half_exact = plain ? 1 : 0;
    }
    std::vector<int16_t> blob;
    blob.insert(blob.end(), xo.begin(), xo.end()); blob.insert(blob.end(), xc.begin(), xc.end());
    blob.insert(blob.end(), yo.begin(), yo.end()); blob.insert(blob.end(), yc.begin(), yc.end());
    PLP_HIP(c->tabs.upload(blob.data(), blob.size() * 2, c->stream));
    PLP_HIP(hipStreamSynchronize(c->stream));
    const int16_t* base = (const int16_t*)c->tabs.p;
    c->rt.xo = base; c->rt.xc = base + P.sw; c->rt.yo = base + 2 * P.sw; c->rt.yc = base + 2 * P.sw + P.sh;
    c->rows = rows; c->cols = cols; c->capB = 0; c->mw_capB = 0; c->seed_capB = 0;
    return PLP_OK;
}

plp_status ensure(plp_line* c, int B) {
    LinePlanes& P = c->P;
    const size_t n = (size_t)P.sw * P.sh, nv = (size_t)(P.sw - 1) * (P.sh - 1);
    const int Bmw = std::min(B, kLsdMwMaxFrames);
    // Region lists: one per frame; two where several waves share a frame (k_lsd_grow_mw writes the refinement's regrowth behind the first growth),
    // which only batches of at most kLsdMwMaxFrames frames do.  The buffer holds either layout, the stride is chosen per launch.
    if (B > c->capB) {
        if (!P.half_exact) PLP_HIP(c->blur11.reserve((size_t)P.pitch * P.H * B));   // only the two-kernel fallback of the LSD front writes the blurred plane
        PLP_HIP(c->scaled.reserve((size_t)P.spitch * P.sh * B));
        PLP_HIP(c->pix.reserve(n * sizeof(LsdPix) * B));
        PLP_HIP(c->g2.reserve(n * 4 * B)); PLP_HIP(c->n_order.reserve(4 * (size_t)B)); PLP_HIP(c->maxgrad.reserve(4 * ((n + 255) / 256) * (size_t)B)); PLP_HIP(c->undef.reserve((n + 63) / 64 * 8 * B));
        PLP_HIP(c->order.reserve(nv * 4 * B)); PLP_HIP(c->reg.reserve(std::max(n * (size_t)B, 2 * n * (size_t)Bmw) * 4));
        PLP_HIP(c->raw.reserve(sizeof(float4) * kLineCap * B)); PLP_HIP(c->n_raw.reserve(4 * (size_t)B));
        PLP_HIP(c->dx.reserve(dxy_frame_entries(P.W, P.H) * 4 * B));
        PLP_HIP(c->all_kl.reserve(sizeof(plp_keyline) * kLineCap * B)); PLP_HIP(c->all_kl_dir.reserve(sizeof(float2) * kLineCap * B)); PLP_HIP(c->all_lbd.reserve((size_t)32 * kLineCap * B));
        PLP_HIP(c->n_all.reserve(4 * (size_t)B)); PLP_HIP(c->status.reserve(16)); PLP_HIP(c->prof.reserve(128)); PLP_HIP(c->grow_stats.reserve(16 * (size_t)B));
        P.blur11 = (uint8_t*)c->blur11.p; P.scaled = (uint8_t*)c->scaled.p;
        P.pix = (LsdPix*)c->pix.p; P.g2 = (uint32_t*)c->g2.p; P.n_order = (int32_t*)c->n_order.p;
        P.blockmax = (uint32_t*)c->maxgrad.p; P.undef = (unsigned long long*)c->undef.p;
        P.order = (uint32_t*)c->order.p; P.reg = (uint32_t*)c->reg.p; P.raw = (float4*)c->raw.p; P.n_raw = (int32_t*)c->n_raw.p;
        P.dxy = (short2*)c->dx.p; P.all_kl = (plp_keyline*)c->all_kl.p; P.all_kl_dir = (float2*)c->all_kl_dir.p; P.all_lbd = (uint8_t*)c->all_lbd.p;
        P.n_all = (int32_t*)c->n_all.p; P.status = (int32_t*)c->status.p; P.prof = (long long*)c->prof.p; P.grow_stats = (int32_t*)c->grow_stats.p;
        c->capB = B;
    }
    P.reg_frame_stride = B <= kLsdMwMaxFrames ? 2 * n : n;
    // the helper waves' lists: only a batch that can take the several-waves path needs them (3.7 MB per frame: not for the 1024-frame contexts of a replay)
    P.mw_heap_frame_stride = (size_t)(kMwMaxWaves - 1) * kMwHeapBufs * kMwHeap;
    if (c->mw_ok && B <= kLsdMwMaxFrames && B > c->mw_capB) {
        PLP_HIP(c->mw_heap.reserve(P.mw_heap_frame_stride * 4 * (size_t)B));
        c->mw_capB = B;
    }
    P.mw_heap = (B <= c->mw_capB) ? (uint32_t*)c->mw_heap.p : nullptr;
    if (c->seed_order == PLP_SEED_ORDER_LIBSTDCXX && B > c->seed_capB) {
        PLP_HIP(c->seed_ent.reserve(nv * 4 * (size_t)B)); PLP_HIP(c->seed_ws.reserve(seed_sort_ws_entries(nv) * 4 * (size_t)B));
        c->seed_capB = B;
    }
    return PLP_OK;
}

plp_status run(plp_line* c, const uint8_t* d_imgs, int B, int rows, int cols, size_t step, size_t frame_stride, plp_keyline* d_kl,
               uint8_t* d_lbd, double* d_fn, int cap, int32_t* d_counts, hipStream_t st) {
    PLP_HIP(hipSetDevice(c->device));
    if (c->seed_order == PLP_SEED_ORDER_LIBSTDCXX && !c->seed_sort_ok)   // (the context exists, so that the caller CAN select the other order)
        return set_error(PLP_ERR_UNSUPPORTED, "this device refused the dynamic LDS size of the exact seed sort (144 KB per workgroup): select PLP_SEED_ORDER_STABLE with plp_line_set_seed_order");
    PLP_TRY(build(c, rows, cols));
    PLP_TRY(ensure(c, B));
    if (((uintptr_t)d_imgs % 4 == 0) && (step % 4 == 0) && (frame_stride % 4 == 0)) {
        c->P.img = d_imgs; c->P.img_frame_stride = frame_stride; c->P.img_pitch = (int)step;
    } else {   // the tile loaders read aligned dwords: one aligned copy of odd-pitch frames
        const size_t fs = (size_t)c->P.pitch * rows;
        PLP_HIP(c->aligned.reserve(fs * B));
        for (int f = 0; f < B; ++f)
            PLP_HIP(hipMemcpy2DAsync((uint8_t*)c->aligned.p + f * fs, c->P.pitch, d_imgs + f * frame_stride, step, cols, rows, hipMemcpyDeviceToDevice, st));
        c->P.img = (const uint8_t*)c->aligned.p; c->P.img_frame_stride = fs; c->P.img_pitch = c->P.pitch;
    }
    PLP_HIP(hipMemsetAsync(c->status.p, 0, 16, st));
    if (c->profiling) PLP_HIP(hipMemsetAsync(c->prof.p, 0, 128, st));   // the one-wave kernel writes slots 0..5 only: no stale counts of an earlier several-waves launch
    const bool exact = c->seed_order == PLP_SEED_ORDER_LIBSTDCXX;
    c->lp.seed_exact = exact ? 1 : 0;
    const SeedSortBufs ssb{(uint32_t*)c->seed_ent.p, (uint32_t*)c->seed_ws.p, seed_sort_ws_entries((size_t)(c->P.sw - 1) * (c->P.sh - 1))};
    launch_line_front(st, c->P, c->lp, c->rt, c->t11, c->t5, c->w, d_kl, d_lbd, d_fn, cap, d_counts, B, c->profiling ? c->ev : nullptr,
                      c->side.stream ? &c->side : nullptr, c->grow_waves, exact ? &ssb : nullptr, c->mw_ok, c->grow_on_side);
    PLP_HIP(hipGetLastError());
    if (c->profiling) {
        PLP_HIP(hipEventSynchronize(c->ev[8]));
        for (int i = 0; i < 8; ++i) { float ms = 0; PLP_HIP(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1])); c->stage_ms[i] += ms; }
        float tot = 0; PLP_HIP(hipEventElapsedTime(&tot, c->ev[0], c->ev[8])); c->stage_ms[8] += tot;
        ++c->stage_batches;
    }
    c->last_B = B; c->last_stream = st; c->last_profiled = c->profiling;
    return PLP_OK;
}

}  // namespace

extern "C" {

plp_status plp_line_create(int device, plp_line** out) {
    if (!out) return set_error(PLP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return set_error(PLP_ERR_NO_DEVICE, "no HIP device visible: the line front-end has no CPU fallback");
    if (device < 0 || device >= n) return set_error(PLP_ERR_INVALID_ARG, "device index out of range");
    PLP_HIP(hipSetDevice(device));
    plp_line* c = new plp_line();
    c->device = device;
    // dynamic-LDS limits are per function AND per device: raised here, for this context's device, not once per process
    c->mw_ok = grow_mw_configure() == hipSuccess;
    c->grow_big_ok = grow_configure() == hipSuccess;     // frames whose USED bitmap needs more than 64 KB of LDS (half-resolution image above 516,065 pixels)
    c->seed_sort_ok = seed_sort_configure() == hipSuccess;
    (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_error(PLP_ERR_HIP, "hipStreamCreate failed"); }
    // Experiment (profiles/r04_grow_cu_mask.md): PLP_GROW_CUS=n runs region growing on a stream of its own that may only use n of the CUs
    // (hipExtStreamCreateWithCUMask; the mask's bits go round the XCDs, so the first n bits are n / 8 CUs of each), the rest of the chip stays free
    // of the growers' LDS and registers.  The side stream then serves this purpose (PLP_LINE_SIDE_STREAM is ignored).
    const char* gcu = getenv("PLP_GROW_CUS");
    const int n_gcu = gcu ? atoi(gcu) : 0;
    hipError_t side_err;
    if (n_gcu > 0) {
        uint32_t mask[16] = {0};
        for (int i = 0; i < n_gcu && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
        side_err = hipExtStreamCreateWithCUMask(&c->side.stream, 16, mask);
        c->grow_on_side = side_err == hipSuccess;
    } else side_err = hipStreamCreateWithFlags(&c->side.stream, hipStreamNonBlocking);
    if (side_err != hipSuccess || hipEventCreateWithFlags(&c->side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->side.join, hipEventDisableTiming) != hipSuccess) { c->side.stream = nullptr; c->grow_on_side = false; }   // optional: falls back to one stream
    *out = c;
    return PLP_OK;
}

void plp_line_destroy(plp_line* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->side.stream) { (void)hipStreamSynchronize(c->side.stream); (void)hipStreamDestroy(c->side.stream); }
    if (c->side.fork) (void)hipEventDestroy(c->side.fork);
    if (c->side.join) (void)hipEventDestroy(c->side.join);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

plp_status plp_line_extract_batch_device(plp_line* c, const uint8_t* d_imgs, int32_t B, int32_t rows, int32_t cols, size_t step,
                                         size_t frame_stride, plp_keyline* d_kl, uint8_t* d_lbd, double* d_linefn, int32_t cap,
                                         int32_t* d_counts, void* hip_stream) {
    if (!c || !d_imgs || !d_kl || !d_lbd || !d_linefn || !d_counts) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0 || rows <= 0 || cols <= 0 || cap <= 0 || step < (size_t)cols) return set_error(PLP_ERR_INVALID_ARG, "bad batch geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    return run(c, d_imgs, B, rows, cols, step, frame_stride, d_kl, d_lbd, d_linefn, cap, d_counts, (hipStream_t)hip_stream);
}

plp_status plp_line_last_batch_status(plp_line* c) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B) return PLP_OK;
    PLP_HIP(hipSetDevice(c->device));
    int32_t s[4] = {0, 0, 0, 0};
    PLP_HIP(hipMemcpyAsync(s, c->status.p, 16, hipMemcpyDeviceToHost, c->last_stream));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    if (s[0] & 1) return set_error(PLP_ERR_CAPACITY, "a frame produced more lines than `cap`; output truncated");
    if (s[0] & 4) return set_error(PLP_ERR_OVERFLOW, "more LSD segments than the per-frame capacity");
    if (s[0] & 16) return set_error(PLP_ERR_HIP, "region growing with several waves per frame timed out in a wait (protocol error, please report the frame)");
    if (s[0] & 32) {
        static thread_local char msg[320];
        snprintf(msg, sizeof msg, "the exact seed sort stopped short in frame %d of the batch (reason %d: 1 = a partition's swap count / cut failed its check [m = %d], 2-3, 6-8 = a list or stack of its LDS ran out of space, "
                 "4-5 = a wave waited beyond the limit): the seed order of this batch is not guaranteed; please report the frame", s[2], s[1], s[3]);
        return set_error(PLP_ERR_OVERFLOW, msg);
    }
    return PLP_OK;
}

plp_status plp_line_extract(plp_line* c, const uint8_t* img, int32_t rows, int32_t cols, size_t step, plp_keyline* kl, uint8_t* lbd,
                            double* linefn, int32_t cap, int32_t* n_out) {
    if (!c || !img || !kl || !lbd || !linefn || !n_out) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || cap < 0 || step < (size_t)cols) return set_error(PLP_ERR_INVALID_ARG, "bad image geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int pitch = (cols + 63) / 64 * 64;
    PLP_HIP(c->l0copy.reserve((size_t)pitch * rows));
    if (c->s_cap < kLineCap) {
        PLP_HIP(c->s_kl.reserve(sizeof(plp_keyline) * kLineCap)); PLP_HIP(c->s_lbd.reserve((size_t)32 * kLineCap));
        PLP_HIP(c->s_fn.reserve((size_t)24 * kLineCap)); PLP_HIP(c->s_cnt.reserve(16));
        c->s_cap = kLineCap;
    }
    PLP_HIP(c->pin.reserve((size_t)rows * cols));   // the caller's pageable image never meets a DMA engine (plp_common.hpp HostPinned)
    c->pin.pack(0, img, step, rows, cols);
    PLP_HIP(hipMemcpy2DAsync(c->l0copy.p, pitch, c->pin.p, cols, cols, rows, hipMemcpyHostToDevice, st));
    PLP_TRY(run(c, (const uint8_t*)c->l0copy.p, 1, rows, cols, pitch, (size_t)pitch * rows, (plp_keyline*)c->s_kl.p, (uint8_t*)c->s_lbd.p,
                (double*)c->s_fn.p, kLineCap, (int32_t*)c->s_cnt.p, st));
    int32_t n = 0;
    PLP_HIP(hipMemcpyAsync(&n, c->s_cnt.p, 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    *n_out = n;
    if (n > cap) return set_error(PLP_ERR_CAPACITY, "caller buffers too small");
    if (n > 0) {
        // results come back through the page-locked buffer as well
        const size_t b_kl = sizeof(plp_keyline) * (size_t)n, b_lbd = 32 * (size_t)n, b_fn = 24 * (size_t)n;
        PLP_HIP(c->pin.reserve(b_kl + b_lbd + b_fn));
        uint8_t* hp = static_cast<uint8_t*>(c->pin.p);
        PLP_HIP(hipMemcpyAsync(hp, c->s_kl.p, b_kl, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipMemcpyAsync(hp + b_kl, c->s_lbd.p, b_lbd, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipMemcpyAsync(hp + b_kl + b_lbd, c->s_fn.p, b_fn, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipStreamSynchronize(st));
        memcpy(kl, hp, b_kl); memcpy(lbd, hp + b_kl, b_lbd); memcpy(linefn, hp + b_kl + b_lbd, b_fn);
    }
    int32_t s[4];
    PLP_HIP(hipMemcpy(s, c->status.p, 16, hipMemcpyDeviceToHost));
    if (s[0] & 4) return set_error(PLP_ERR_OVERFLOW, "more LSD segments than the per-frame capacity");
    if (s[0] & 16) return set_error(PLP_ERR_HIP, "region growing with several waves per frame timed out in a wait (protocol error, please report the frame)");
    if (s[0] & 32) {
        static thread_local char msg[320];
        snprintf(msg, sizeof msg, "the exact seed sort stopped short in frame %d of the batch (reason %d: 1 = a partition's swap count / cut failed its check [m = %d], 2-3, 6-8 = a list or stack of its LDS ran out of space, "
                 "4-5 = a wave waited beyond the limit): the seed order of this batch is not guaranteed; please report the frame", s[2], s[1], s[3]);
        return set_error(PLP_ERR_OVERFLOW, msg);
    }
    return PLP_OK;
}

plp_status plp_line_debug_grow_profile(plp_line* c, int64_t* out12) {
    int64_t* out6 = out12;
    if (!c || !out6) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B) return set_error(PLP_ERR_INVALID_ARG, "no batch yet");
    // the growers write their clocks only when profiling is on, and the buffer is cleared only then: without it the bytes are stale or uninitialised
    if (!c->last_profiled) return set_error(PLP_ERR_INVALID_ARG, "the last batch ran with profiling off: call plp_line_set_profiling(ctx, 1) before the batch");
    PLP_HIP(hipSetDevice(c->device));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    long long v[12] = {0};
    PLP_HIP(hipMemcpy(v, c->prof.p, 96, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) out6[i] = v[i];
    return PLP_OK;
}

plp_status plp_line_set_grow_waves(plp_line* c, int32_t waves) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    if (waves < 0 || waves > kMwMaxWaves) return set_error(PLP_ERR_INVALID_ARG, "waves must be 0 (automatic) .. 8");
    std::lock_guard<std::mutex> lk(c->mu);
    c->grow_waves = waves;
    return PLP_OK;
}

plp_status plp_line_set_seed_order(plp_line* c, int32_t order) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    if (order != PLP_SEED_ORDER_STABLE && order != PLP_SEED_ORDER_LIBSTDCXX) return set_error(PLP_ERR_INVALID_ARG, "unknown seed order");
    if (order == PLP_SEED_ORDER_LIBSTDCXX && !c->seed_sort_ok) return set_error(PLP_ERR_UNSUPPORTED, "this device refused the LDS size of the exact seed sort");
    std::lock_guard<std::mutex> lk(c->mu);
    c->seed_order = order;
    //  the exact order's buffers -- the seed array and the sort's scratch: 476 KB per frame of a 640 x 480 batch, INTEGRATION.md -- are NOT freed here any
    // more: hipFree drains the whole device (every stream of every other context of an overlapped step) and a caller that alternates the two orders paid that
    // and a re-allocation of ~0.5 GB per switch.  A caller that leaves the exact order for good gives the memory back with plp_line_trim().
    return PLP_OK;
}

// Give back what the current settings do not need: the exact seed order's buffers while the stable order is selected, the several-waves grower's heap.
// hipFree waits for the device: call it where a pause is acceptable.  The caller's current device is restored.
plp_status plp_line_trim(plp_line* c) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    int prev = -1;
    PLP_HIP(hipGetDevice(&prev));
    PLP_HIP(hipSetDevice(c->device));
    if (c->last_B) (void)hipStreamSynchronize(c->last_stream);
    if (c->seed_order == PLP_SEED_ORDER_STABLE && c->seed_capB > 0) { c->seed_ent.release(); c->seed_ws.release(); c->seed_capB = 0; }
    c->mw_heap.release(); c->mw_capB = 0;
    if (prev >= 0 && prev != c->device) PLP_HIP(hipSetDevice(prev));
    return PLP_OK;
}

plp_status plp_line_get_seed_order(const plp_line* c, int32_t* order) {
    if (!c || !order) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    *order = c->seed_order;
    return PLP_OK;
}

plp_status plp_line_set_profiling(plp_line* c, int32_t enable) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    if (enable && !c->ev[0]) for (auto& e : c->ev) PLP_HIP(hipEventCreate(&e));
    c->profiling = enable != 0;
    for (auto& v : c->stage_ms) v = 0;
    c->stage_batches = 0;
    return PLP_OK;
}

plp_status plp_line_get_stage_times(plp_line* c, double* ms9, int64_t* n_batches) {
    if (!c || !ms9) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    for (int i = 0; i < 9; ++i) ms9[i] = c->stage_ms[i];
    if (n_batches) *n_batches = c->stage_batches;
    return PLP_OK;
}

plp_status plp_line_scaled_size(const plp_line* c, int32_t* rows, int32_t* cols) {
    if (!c || !rows || !cols || !c->rows) return set_error(PLP_ERR_INVALID_ARG, "no frame processed yet");
    *rows = c->P.sh; *cols = c->P.sw;
    return PLP_OK;
}

plp_status plp_line_debug_read(plp_line* c, plp_line_debug_id what, int32_t frame, void* dst, size_t dst_bytes, int64_t* n_out) {
    if (!c || !dst || !n_out) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->last_B || frame < 0 || frame >= c->last_B) return set_error(PLP_ERR_INVALID_ARG, "bad frame");
    PLP_HIP(hipSetDevice(c->device));
    PLP_HIP(hipStreamSynchronize(c->last_stream));
    const LinePlanes& P = c->P;
    const size_t n = (size_t)P.sw * P.sh, nv = (size_t)(P.sw - 1) * (P.sh - 1), full = (size_t)P.W * P.H;
    int32_t cnt = 0;
    switch (what) {
        case PLP_LINE_DBG_SCALED:
            if (dst_bytes < n) return set_error(PLP_ERR_CAPACITY, "dst too small");
            PLP_HIP(hipMemcpy2D(dst, P.sw, P.scaled + (size_t)frame * P.spitch * P.sh, P.spitch, P.sw, P.sh, hipMemcpyDeviceToHost));
            *n_out = (int64_t)n; return PLP_OK;
        case PLP_LINE_DBG_ORDER:
            PLP_HIP(hipMemcpy(&cnt, P.n_order + frame, 4, hipMemcpyDeviceToHost));
            if (dst_bytes < (size_t)cnt * 4) return set_error(PLP_ERR_CAPACITY, "dst too small");
            if (cnt) PLP_HIP(hipMemcpy(dst, P.order + (size_t)frame * nv, (size_t)cnt * 4, hipMemcpyDeviceToHost));
            *n_out = cnt; return PLP_OK;
        case PLP_LINE_DBG_RAW:
            PLP_HIP(hipMemcpy(&cnt, P.n_raw + frame, 4, hipMemcpyDeviceToHost));
            if (dst_bytes < (size_t)cnt * 16) return set_error(PLP_ERR_CAPACITY, "dst too small");
            if (cnt) PLP_HIP(hipMemcpy(dst, P.raw + (size_t)frame * kLineCap, (size_t)cnt * 16, hipMemcpyDeviceToHost));
            *n_out = cnt; return PLP_OK;
        case PLP_LINE_DBG_ALL_KL:
        case PLP_LINE_DBG_ALL_LBD: {
            PLP_HIP(hipMemcpy(&cnt, P.n_all + frame, 4, hipMemcpyDeviceToHost));
            const size_t rec = what == PLP_LINE_DBG_ALL_KL ? sizeof(plp_keyline) : 32;
            if (dst_bytes < (size_t)cnt * rec) return set_error(PLP_ERR_CAPACITY, "dst too small");
            const uint8_t* src = what == PLP_LINE_DBG_ALL_KL ? (const uint8_t*)(P.all_kl + (size_t)frame * kLineCap) : P.all_lbd + (size_t)frame * kLineCap * 32;
            if (cnt) PLP_HIP(hipMemcpy(dst, src, (size_t)cnt * rec, hipMemcpyDeviceToHost));
            *n_out = cnt; return PLP_OK;
        }
        case PLP_LINE_DBG_GROW_STATS:
            if (dst_bytes < 16) return set_error(PLP_ERR_CAPACITY, "dst too small");
            PLP_HIP(hipMemcpy(dst, P.grow_stats + (size_t)frame * 4, 16, hipMemcpyDeviceToHost));
            *n_out = 4; return PLP_OK;
        case PLP_LINE_DBG_SOBEL_DX:
        case PLP_LINE_DBG_SOBEL_DY:
            if (dst_bytes < full * 2) return set_error(PLP_ERR_CAPACITY, "dst too small");
            {   // the device plane interleaves (dx, dy)
                const size_t ent = dxy_frame_entries(P.W, P.H);
                std::vector<int16_t> both(ent * 2);
                PLP_HIP(hipMemcpy(both.data(), P.dxy + (size_t)frame * ent, ent * 4, hipMemcpyDeviceToHost));
                int16_t* d16 = (int16_t*)dst;
                const int tiles8 = (P.W + 7) / 8;
                for (int y = 0; y < P.H; ++y)
                    for (int x = 0; x < P.W; ++x) d16[(size_t)y * P.W + x] = both[2 * (size_t)dxy_index(x, y, tiles8) + (what == PLP_LINE_DBG_SOBEL_DY ? 1 : 0)];
            }
            *n_out = (int64_t)full; return PLP_OK;
    }
    return set_error(PLP_ERR_INVALID_ARG, "unknown debug id");
}

// Host model of the gradient kernel's cos/sin (sincos_ziv.hpp), callable without a GPU: proven[i] = 1 where the rounding test
// succeeds (the kernel uses these values), 0 where the kernel evaluates the general f64 routine.
int32_t plp_model_sincos_host(const float* a, int64_t n, float* c, float* s, uint8_t* proven) {
    int32_t n_proven = 0;
    for (int64_t i = 0; i < n; ++i) { proven[i] = plp::sincos_ziv(a[i], c + i, s + i) ? 1 : 0; n_proven += proven[i]; }
    return n_proven;
}

// Host model of the exact seed sort (seed_sort_model.hpp): std::__introsort_loop on entries whose key is bits 20..29, as the rank-paired
// partitions the kernel runs.  depth_limit < 0: the library's 2 * floor(log2 n).  Callable without a GPU.
int32_t plp_model_seed_introsort_host(uint32_t* entries, int64_t n, int32_t depth_limit, uint32_t skip_key) {
    if (!entries || n < 0 || n > (int64_t)kLsdMaxScaledPixels) return -1;
    plp::seedsort::introsort_loop_model(entries, (int)n, depth_limit, skip_key);
    return 0;
}

// The kernel's introsort loop on caller-made entries (host pointers; one workgroup), with a chosen recursion budget: the tests' way to
// reach every branch (global partitions, LDS window, wave tasks, lanes, heap sort) on arbitrary key distributions.
plp_status plp_seed_introsort_debug(int32_t device, uint32_t* entries, int64_t n, int32_t depth_limit, uint32_t skip_key, int32_t variant, int32_t* n_live) {
    if (!entries || n < 0 || n > (int64_t)kLsdMaxScaledPixels) return set_error(PLP_ERR_INVALID_ARG, "bad entries");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) return set_error(PLP_ERR_NO_DEVICE, "no HIP device visible");
    PLP_HIP(hipSetDevice(device));
    PLP_HIP(seed_sort_configure());
    if (depth_limit < 0) { int lg = 0; while ((2ll << lg) <= n) ++lg; depth_limit = 2 * lg; }
    DevBuf ent, ws, st;
    // PLP_SEED_SORT_DBG_COPIES=N (diagnostic): N workgroups sort N copies at once, workgroup 0's result and clocks are returned -- the phase split with the chip full
    const int copies = std::max(1, std::min(4096, getenv("PLP_SEED_SORT_DBG_COPIES") ? atoi(getenv("PLP_SEED_SORT_DBG_COPIES")) : 1));
    PLP_HIP(ent.reserve((size_t)std::max<int64_t>(n, 1) * 4 * copies)); PLP_HIP(ws.reserve(seed_sort_ws_entries((size_t)n) * 4 * copies)); PLP_HIP(st.reserve(16));
    for (int k = 0; k < copies; ++k) PLP_HIP(hipMemcpy((uint32_t*)ent.p + (size_t)k * n, entries, (size_t)n * 4, hipMemcpyHostToDevice));
    PLP_HIP(hipMemset(st.p, 0, 16));
    DevBuf dbg;
    const char* dflag = getenv("PLP_SEED_SORT_DBG");
    if (dflag) { PLP_HIP(dbg.reserve(4 * (2 + 6 * 4000 + 48))); PLP_HIP(hipMemset(dbg.p, 0, 4 * (2 + 6 * 4000 + 48))); int f = atoi(dflag); PLP_HIP(hipMemcpy(dbg.p, &f, 4, hipMemcpyHostToDevice)); }
    launch_seed_sort_debug(nullptr, (uint32_t*)ent.p, (int)n, depth_limit, skip_key, (uint32_t*)ws.p, (int32_t*)st.p, dflag ? (int*)dbg.p : nullptr, variant, copies);
    PLP_HIP(hipGetLastError());
    PLP_HIP(hipDeviceSynchronize());
    int32_t s = 0;
    PLP_HIP(hipMemcpy(&s, st.p, 4, hipMemcpyDeviceToHost));
    PLP_HIP(hipMemcpy(entries, ent.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (n_live) { uint32_t nl = (uint32_t)n; if (n > 16) PLP_HIP(hipMemcpy(&nl, ws.p, 4, hipMemcpyDeviceToHost)); *n_live = (int32_t)nl; }
    if (dflag && getenv("PLP_SEED_SORT_DBG_FILE")) {
        std::vector<int> h(2 + 6 * 4000 + 48);
        PLP_HIP(hipMemcpy(h.data(), dbg.p, h.size() * 4, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("PLP_SEED_SORT_DBG_FILE"), "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
    }
    if (s & 32) return set_error(PLP_ERR_OVERFLOW, "the exact seed sort stopped short (queue space, or a partner position outside its segment)");
    return PLP_OK;
}

}  // extern "C"
