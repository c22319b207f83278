// TEST INFRASTRUCTURE ONLY (oracle).  ctypes-facing C entry points of the CPU
// restatement; imported by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg only.
#include <chrono>
#include <cstdio>
#include <thread>

#include "orb_oracle.hpp"
#include "lsd_restated.hpp"

using namespace oracle;

namespace {
struct OrbHandle {
    OrbOracle ex;
    std::vector<KeyPoint> kps;
    std::vector<uint8_t> desc;
    explicit OrbHandle(const OrbParams& p) : ex(p) {}
};
Image wrap(const uint8_t* p, int rows, int cols, long step) {
    Image im(rows, cols);
    for (int y = 0; y < rows; ++y) std::memcpy(im.row(y), p + (size_t)y * step, cols);
    return im;
}
}  // namespace

extern "C" {

void* oracle_orb_create(unsigned max_kp, float sf, unsigned levels, unsigned ini_thr, unsigned min_thr,
                        const float* rects, int n_rects) {
    OrbParams p;
    p.max_num_keypts = max_kp; p.scale_factor = sf; p.num_levels = levels;
    p.ini_fast_thr = ini_thr; p.min_fast_thr = min_thr;
    for (int i = 0; i < n_rects; ++i) p.mask_rects.push_back({rects[4 * i], rects[4 * i + 1], rects[4 * i + 2], rects[4 * i + 3]});
    return new OrbHandle(p);
}
void oracle_orb_destroy(void* h) { delete (OrbHandle*)h; }

// returns number of keypoints (or -1 if cap too small); kps = cap x 28 B, desc = cap x 32 B
int oracle_orb_extract(void* h, const uint8_t* img, int rows, int cols, long step, const uint8_t* mask,
                       long mask_step, KeyPoint* kps, uint8_t* desc, int cap) {
    auto* H = (OrbHandle*)h;
    Image im = wrap(img, rows, cols, step);
    Image mk;
    if (mask) mk = wrap(mask, rows, cols, mask_step);
    H->ex.extract(im, mask ? &mk : nullptr, H->kps, H->desc);
    const int n = (int)H->kps.size();
    if (n > cap) return -1;
    if (n) {
        std::memcpy(kps, H->kps.data(), (size_t)n * sizeof(KeyPoint));
        std::memcpy(desc, H->desc.data(), (size_t)n * 32);
    }
    return n;
}
void oracle_orb_tables(void* h, float* sf, float* isf, float* s2, float* is2, unsigned* quota, int* umax) {
    auto& e = ((OrbHandle*)h)->ex;
    const size_t n = e.scale_factors.size();
    std::memcpy(sf, e.scale_factors.data(), n * 4);
    std::memcpy(isf, e.inv_scale_factors.data(), n * 4);
    std::memcpy(s2, e.level_sigma_sq.data(), n * 4);
    std::memcpy(is2, e.inv_level_sigma_sq.data(), n * 4);
    std::memcpy(quota, e.num_keypts_per_level.data(), n * 4);
    std::memcpy(umax, e.u_max.data(), 16 * 4);
}
void oracle_orb_level_size(void* h, int level, int* rows, int* cols) {
    auto& im = ((OrbHandle*)h)->ex.pyramid.at(level);
    *rows = im.rows; *cols = im.cols;
}
void oracle_orb_level_image(void* h, int level, uint8_t* dst) {
    auto& im = ((OrbHandle*)h)->ex.pyramid.at(level);
    std::memcpy(dst, im.data.data(), im.data.size());
}
int oracle_orb_level_blurred(void* h, int level, uint8_t* dst) {  // 0 if level was not blurred
    auto& im = ((OrbHandle*)h)->ex.blurred.at(level);
    if (im.empty()) return 0;
    std::memcpy(dst, im.data.data(), im.data.size());
    return 1;
}
int oracle_orb_num_candidates(void* h, int level) { return (int)((OrbHandle*)h)->ex.candidates.at(level).size(); }
void oracle_orb_candidates(void* h, int level, KeyPoint* dst) {
    auto& v = ((OrbHandle*)h)->ex.candidates.at(level);
    if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(KeyPoint));
}
int oracle_orb_num_level_keypts(void* h, int level) { return (int)((OrbHandle*)h)->ex.level_keypts.at(level).size(); }
void oracle_orb_level_keypts(void* h, int level, KeyPoint* dst) {
    auto& v = ((OrbHandle*)h)->ex.level_keypts.at(level);
    if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(KeyPoint));
}
// quadtree alone: in = n candidates (border-relative x,y,response used), out <= n
int oracle_orb_distribute(void* h, const KeyPoint* in, int n, int min_x, int max_x, int min_y, int max_y,
                          unsigned num_keypts, KeyPoint* out) {
    std::vector<KeyPoint> v(in, in + n);
    auto r = ((OrbHandle*)h)->ex.distribute_via_tree(v, min_x, max_x, min_y, max_y, num_keypts);
    if (!r.empty()) std::memcpy(out, r.data(), r.size() * sizeof(KeyPoint));
    return (int)r.size();
}

// ---- primitives
void oracle_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    Image s = wrap(src, sh, sw, sw);
    Image d = resize_linear_u8(s, dw, dh);
    std::memcpy(dst, d.data.data(), d.data.size());
}
void oracle_gaussian_blur_u8(const uint8_t* src, int h, int w, int ksize, double sigma, uint8_t* dst) {
    Image s = wrap(src, h, w, w);
    Image d = gaussian_blur_u8(s, ksize, sigma);
    std::memcpy(dst, d.data.data(), d.data.size());
}
void oracle_gaussian_taps_q8(int n, double sigma, int* out) {
    auto k = gaussian_taps_q8(n, sigma);
    for (int i = 0; i < n; ++i) out[i] = k[i];
}
// FAST on a w x h ROI; out = (x,y,score) int triples; returns count
int oracle_fast9_16(const uint8_t* base, int step, int w, int h, int threshold, int* out, int cap) {
    std::vector<FastPoint> v;
    fast9_16_nms(base, step, w, h, threshold, v);
    int n = 0;
    for (auto& p : v) { if (n >= cap) break; out[3 * n] = p.x; out[3 * n + 1] = p.y; out[3 * n + 2] = p.score; ++n; }
    return (int)v.size();
}
float oracle_fast_atan2(float y, float x) { return fast_atan2f_deg(y, x); }
// D2: cosf / sinf of the line sources, defined as the rounded f64 libm results (lsd_restated.hpp f_cos / f_sin)
void oracle_f_cos_sin(const float* a, long n, float* c, float* s) { for (long i = 0; i < n; ++i) { c[i] = oracle::f_cos(a[i]); s[i] = oracle::f_sin(a[i]); } }
float oracle_trig_cos(float v) { return trig::cos(v); }
float oracle_trig_sin(float v) { return trig::sin(v); }
void oracle_scale_tables(unsigned n, float sf, float* a, float* b, float* c, float* d) {
    auto v0 = calc_scale_factors(n, sf), v1 = calc_inv_scale_factors(n, sf), v2 = calc_level_sigma_sq(n, sf),
         v3 = calc_inv_level_sigma_sq(n, sf);
    std::memcpy(a, v0.data(), n * 4); std::memcpy(b, v1.data(), n * 4);
    std::memcpy(c, v2.data(), n * 4); std::memcpy(d, v3.data(), n * 4);
}

// ---- CPU baseline timing helper: extract `n_frames` frames with `n_threads` frame-parallel workers.
// Returns wall seconds.  (bench.py cpu_baseline leg only.)
double oracle_orb_time_frames(const uint8_t* frames, int n_frames, int rows, int cols, unsigned max_kp,
                              int n_threads, long* total_kp) {
    std::vector<long> counts(n_threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            OrbParams p;
            p.max_num_keypts = max_kp;
            OrbOracle ex(p);
            std::vector<KeyPoint> k;
            std::vector<uint8_t> d;
            for (int f = t; f < n_frames; f += n_threads) {
                Image im = wrap(frames + (size_t)f * rows * cols, rows, cols, cols);
                ex.extract(im, nullptr, k, d);
                counts[t] += (long)k.size();
            }
        });
    for (auto& x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    long s = 0;
    for (auto c : counts) s += c;
    if (total_kp) *total_kp = s;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"

// ---- CPU baseline of the whole front-end (bench.py cpu_baseline leg only): per frame ORB extract + LSD/LBD
// extract + match_current_and_last_frames[_line] + match_frame_and_landmarks[_line] against the two previous frames,
// the same work bench.py times on the GPU.  Frame-parallel over n_threads workers; returns wall seconds.
namespace oracle { struct LineResult; }
extern "C" {
void* oracle_line_extract(const uint8_t* img, int rows, int cols, long step, int stable_order);
void oracle_line_free(void* h);
int oracle_line_count(void* h, int which);
struct CapiKeyLine { float angle; int class_id, octave; float f[13]; int numOfPixels; };   // 68-byte cv KeyLine record (line_oracle.cpp:35)
void oracle_line_get(void* h, int which, CapiKeyLine* kl, uint8_t* lbd, double* linefn, float* desc_f);
unsigned oracle_match_current_and_last_line(const CapiKeyLine* kl, const uint8_t* lbd, const float* xr_pair, const uint8_t* occupied, int n,
                                            const float* scale_factors_lsd, int num_levels_lsd, const uint8_t* valid, const float* sp,
                                            const float* ep, const float* lxr_sp, const float* lxr_ep, const int* loctave,
                                            const uint8_t* ldesc, const uint8_t* l_has_obs, int m, float margin, int direction, int is_rgbd,
                                            int* line_last);
unsigned oracle_match_frame_and_landmarks_line(const CapiKeyLine* kl, const uint8_t* lbd, const int* kp_octave, const uint8_t* occupied, int n,
                                               const float* scale_factors_lsd, const uint8_t* lm_valid, const float* lm_sp,
                                               const float* lm_ep, const int* lm_level, const uint8_t* lm_desc,
                                               const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio, int* line_landmark);
unsigned oracle_match_frame_and_landmarks(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const float* x_right,
                                          const uint8_t* occupied, int n, const float* scale_factors, const uint8_t* lm_valid,
                                          const float* lm_reproj, const float* lm_x_right, const int* lm_level,
                                          const uint8_t* lm_desc, const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio,
                                          int* kp_landmark);
unsigned oracle_match_current_and_last(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const float* x_right,
                                       const uint8_t* occupied, int n, const float* scale_factors, int num_levels,
                                       const uint8_t* valid, const float* reproj, const float* lx_right, const int* loctave,
                                       const float* langle, const uint8_t* ldesc, const uint8_t* l_has_obs, int m, float margin,
                                       int direction, int check_orientation, int* kp_last);

// two_threads_per_frame != 0: ORB and LSD+LBD of a frame run in two threads that are joined before the matchers, as the reference's
// frame constructor does (data/frame.cc:691-694).  stage_sec3 (nullable): thread-seconds spent in ORB / lines / matchers.
double oracle_front_time_frames2(const uint8_t* frames, int n_frames, int rows, int cols, unsigned max_kp, int n_threads, float shift_x,
                                 int two_threads_per_frame, double* stage_sec3, long* totals3) {
    std::vector<long> kp(n_threads, 0), ln(n_threads, 0), mt(n_threads, 0);
    std::vector<double> s_orb(n_threads, 0), s_line(n_threads, 0), s_match(n_threads, 0);
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const double grid6[6] = {0.0, 0.0, 64.0 / (double)(float)cols, 48.0 / (double)(float)rows, 64, 48};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            OrbParams p;
            p.max_num_keypts = max_kp;
            OrbOracle ex(p);
            const std::vector<float> sf = ex.scale_factors;
            // every worker owns a contiguous block of the sequence and matches inside it (as each GPU rank does)
            const int per = (n_frames + n_threads - 1) / n_threads, f0 = t * per, f1 = std::min(n_frames, f0 + per);
            std::vector<KeyPoint> k[3];
            std::vector<uint8_t> d[3];
            std::vector<CapiKeyLine> kl[3];
            std::vector<uint8_t> lbd[3];
            int have_lines[3] = {-1, -1, -1};
            for (int f = f0; f < f1; ++f) {
                const int cur = f % 3;
                Image im = wrap(frames + (size_t)f * rows * cols, rows, cols, cols);
                void* lh = nullptr;
                auto ta = now();
                if (two_threads_per_frame) {
                    std::thread line_thread([&]() { lh = oracle_line_extract(im.data.data(), rows, cols, cols, 0); });   // std::sort, as the reference
                    ex.extract(im, nullptr, k[cur], d[cur]);
                    s_orb[t] += secs(ta, now());
                    line_thread.join();
                    s_line[t] += secs(ta, now());        // wall time of the pair: the frame waits for the slower of the two
                } else {
                    ex.extract(im, nullptr, k[cur], d[cur]);
                    auto tb = now();
                    s_orb[t] += secs(ta, tb);
                    lh = oracle_line_extract(im.data.data(), rows, cols, cols, 0);
                    s_line[t] += secs(tb, now());
                }
                auto tm = now();
                struct Acc { double& s; std::chrono::steady_clock::time_point t0; ~Acc() { s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc{s_match[t], tm};
                kp[t] += (long)k[cur].size();
                const int nl = oracle_line_count(lh, 0);
                ln[t] += nl;
                kl[cur].assign((size_t)nl, CapiKeyLine{}); lbd[cur].assign((size_t)nl * 32, 0);
                oracle_line_get(lh, 0, kl[cur].data(), lbd[cur].data(), nullptr, nullptr);
                oracle_line_free(lh);
                have_lines[cur] = f;
                const int p1 = (f + 2) % 3, p2 = (f + 1) % 3;   // frames f - 1 and f - 2
                const float sf_lsd[1] = {1.f};
                // match_current_and_last_frames_line: the previous frame's key lines, both end points moved by the pan
                if (f > f0 && have_lines[p1] == f - 1 && nl > 0) {
                    const auto& pk = kl[p1];
                    const int m = (int)pk.size();
                    std::vector<float> sp(2 * (size_t)m), ep(2 * (size_t)m), lxr((size_t)m, -1.f), xrp(2 * (size_t)nl, -1.f);
                    std::vector<int> loct((size_t)m), lout((size_t)nl);
                    std::vector<uint8_t> one((size_t)m, 1), occl((size_t)nl, 0);
                    for (int i = 0; i < m; ++i) {   // f[4..7] = startPointX, startPointY, endPointX, endPointY
                        sp[2 * i] = pk[i].f[4] + shift_x; sp[2 * i + 1] = pk[i].f[5];
                        ep[2 * i] = pk[i].f[6] + shift_x; ep[2 * i + 1] = pk[i].f[7];
                        loct[i] = pk[i].octave;
                    }
                    mt[t] += oracle_match_current_and_last_line(kl[cur].data(), lbd[cur].data(), xrp.data(), occl.data(), nl, sf_lsd, 1,
                                                                one.data(), sp.data(), ep.data(), lxr.data(), lxr.data(), loct.data(),
                                                                lbd[p1].data(), one.data(), m, 20.f, 0, 0, lout.data());
                }
                // match_frame_and_landmarks_line: the key lines of frames f - 2 and f - 1 as local line landmarks (tracking_module.cc:1060)
                if (f - f0 >= 2 && have_lines[p1] == f - 1 && have_lines[p2] == f - 2 && nl > 0) {
                    std::vector<float> sp, ep;
                    std::vector<int> lvl, lout((size_t)nl), kpo((size_t)nl, 0);
                    std::vector<uint8_t> qd;
                    for (int which = 0; which < 2; ++which) {
                        const auto& pk = kl[which ? p1 : p2];
                        const float sx = which ? shift_x : 2 * shift_x;
                        for (size_t i = 0; i < pk.size(); ++i) {
                            sp.push_back(pk[i].f[4] + sx); sp.push_back(pk[i].f[5]);
                            ep.push_back(pk[i].f[6] + sx); ep.push_back(pk[i].f[7]);
                            lvl.push_back(pk[i].octave);
                        }
                        qd.insert(qd.end(), lbd[which ? p1 : p2].begin(), lbd[which ? p1 : p2].end());
                    }
                    const int m = (int)lvl.size();
                    std::vector<uint8_t> one((size_t)std::max(m, 1), 1), occl((size_t)nl, 0);
                    for (int i = 0; i < nl && i < (int)k[cur].size(); ++i) kpo[i] = k[cur][i].octave;   // the key POINT octave read with a line index (projection.cc:187)
                    if (m > 0)
                        mt[t] += oracle_match_frame_and_landmarks_line(kl[cur].data(), lbd[cur].data(), kpo.data(), occl.data(), nl, sf_lsd, one.data(), sp.data(),
                                                                       ep.data(), lvl.data(), qd.data(), one.data(), m, 10.f, 0.8f, lout.data());
                }
                const int n = (int)k[cur].size();
                if (f - f0 < 2 || n == 0) continue;
                std::vector<float> xr(n, -1.f);
                std::vector<uint8_t> occ(n, 0);
                std::vector<int> out(n);
                // queries: previous frame (last-frame matcher), previous two frames (local-landmark matcher)
                std::vector<uint8_t> valid, qd, hobs;
                std::vector<float> rp, qx, qa;
                std::vector<int> ql;
                auto append = [&](int which, float sx) {
                    for (size_t i = 0; i < k[which].size(); ++i) {
                        valid.push_back(1); hobs.push_back(1);
                        rp.push_back(k[which][i].x + sx); rp.push_back(k[which][i].y);
                        qx.push_back(-1.f); qa.push_back(k[which][i].angle); ql.push_back(k[which][i].octave);
                    }
                    qd.insert(qd.end(), d[which].begin(), d[which].end());
                };
                append((f + 2) % 3, shift_x);
                mt[t] += oracle_match_current_and_last(grid6, k[cur].data(), d[cur].data(), xr.data(), occ.data(), n, sf.data(), (int)sf.size(),
                                                       valid.data(), rp.data(), qx.data(), ql.data(), qa.data(), qd.data(), hobs.data(),
                                                       (int)ql.size(), 20.f, 0, 1, out.data());
                // local landmarks: the key points of frame f - 2, then those of frame f - 1 (the order of the device replay, plp_replay_point_queries_device)
                valid.clear(); hobs.clear(); rp.clear(); qx.clear(); qa.clear(); ql.clear(); qd.clear();
                append((f + 1) % 3, 2 * shift_x);
                append((f + 2) % 3, shift_x);
                mt[t] += oracle_match_frame_and_landmarks(grid6, k[cur].data(), d[cur].data(), xr.data(), occ.data(), n, sf.data(), valid.data(),
                                                          rp.data(), qx.data(), ql.data(), qd.data(), hobs.data(), (int)ql.size(), 10.f, 0.8f,
                                                          out.data());
            }
        });
    for (auto& x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    if (totals3) {
        totals3[0] = totals3[1] = totals3[2] = 0;
        for (int t = 0; t < n_threads; ++t) { totals3[0] += kp[t]; totals3[1] += ln[t]; totals3[2] += mt[t]; }
    }
    if (stage_sec3) {
        stage_sec3[0] = stage_sec3[1] = stage_sec3[2] = 0;
        for (int t = 0; t < n_threads; ++t) { stage_sec3[0] += s_orb[t]; stage_sec3[1] += s_line[t]; stage_sec3[2] += s_match[t]; }
    }
    return std::chrono::duration<double>(t1 - t0).count();
}
double oracle_front_time_frames(const uint8_t* frames, int n_frames, int rows, int cols, unsigned max_kp, int n_threads, float shift_x, long* totals3) {
    return oracle_front_time_frames2(frames, n_frames, rows, cols, max_kp, n_threads, shift_x, 0, nullptr, totals3);
}
}
