// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C entry points around the REFERENCE'S OWN translation units
// (src/PLPSLAM/feature/orb_extractor.cc, orb_extractor_node.cc, orb_params.cc, headers util/trigonometric.h,
// match/base.h, match/angle_checker.h), compiled from /root/reference against oracle/ref_shim (see cvshim.hpp for
// what that pins and what it does not).  Built by oracle/ref_build.sh into oracle/_ref/libplpref.so; used by
// tests/test_oracle_vs_ref.py to validate oracle/orb_oracle.cpp and oracle/match_oracle.cpp.
#include <cstring>
#include <vector>

#include "PLPSLAM/feature/orb_extractor.h"
#include "PLPSLAM/match/angle_checker.h"
#include "PLPSLAM/match/base.h"
#include "PLPSLAM/util/trigonometric.h"

using namespace PLPSLAM;

// The reference orders equally full quadtree nodes by their ADDRESS (std::sort over pair<int, orb_extractor_node*>,
// orb_extractor.cc:529), i.e. by whatever the process allocator hands out.  This library gives the reference sources a
// monotonic allocator (addresses only grow, nothing is reused; the arena is rewound at the start of every entry
// point), under which address order = creation order -- the definition the oracle and the HIP path use.
// The library is linked with -Bsymbolic so only ITS OWN allocations go through these operators.
#include <sys/mman.h>
#include <new>
namespace {
char* g_arena = nullptr;
size_t g_off = 0, g_mark = 0;
bool g_marked = false;
constexpr size_t kArena = size_t(4) << 30;
void* bump(size_t n, size_t align) {
    if (!g_arena) {
        g_arena = (char*)mmap(nullptr, kArena, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_arena == (char*)MAP_FAILED) throw std::bad_alloc();
    }
    g_off = (g_off + align - 1) & ~(align - 1);
    if (g_off + n > kArena) throw std::bad_alloc();
    void* p = g_arena + g_off;
    g_off += n ? n : 1;
    return p;
}
struct Rewind { Rewind() { if (!g_marked) { g_mark = g_off; g_marked = true; } g_off = g_mark; } };
}  // namespace
void* operator new(size_t n) { return bump(n, 16); }
void* operator new[](size_t n) { return bump(n, 16); }
void* operator new(size_t n, std::align_val_t a) { return bump(n, (size_t)a); }
void* operator new[](size_t n, std::align_val_t a) { return bump(n, (size_t)a); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, size_t) noexcept {}
void operator delete[](void*, size_t) noexcept {}
void operator delete(void*, std::align_val_t) noexcept {}
void operator delete[](void*, std::align_val_t) noexcept {}
void operator delete(void*, size_t, std::align_val_t) noexcept {}
void operator delete[](void*, size_t, std::align_val_t) noexcept {}

extern "C" {

// feature::orb_extractor(max_kp, scale, levels, ini, min, mask_rects).extract(image, mask)
// mask_rects: n_rects x 4 floats (x_min, x_max, y_min, y_max as fractions); mask: NULL or rows x cols u8
// returns the number of key points; kps (28-byte cv::KeyPoint layout) and desc must hold `cap` entries
int ref_orb_extract(const unsigned char* img, int rows, int cols, int max_kp, float scale, int levels, int ini_thr, int min_thr,
                    const float* mask_rects, int n_rects, const unsigned char* mask, void* kps_out, unsigned char* desc_out, int cap) {
    Rewind rewind;
    std::vector<std::vector<float>> rects;
    for (int i = 0; i < n_rects; ++i) rects.emplace_back(mask_rects + 4 * i, mask_rects + 4 * i + 4);
    feature::orb_extractor ex(max_kp, scale, levels, ini_thr, min_thr, rects);
    cv::Mat image(rows, cols, CV_8UC1, const_cast<unsigned char*>(img));
    cv::Mat mask_m = mask ? cv::Mat(rows, cols, CV_8UC1, const_cast<unsigned char*>(mask)) : cv::Mat();
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    ex.extract(cv::_InputArray(image), cv::_InputArray(mask_m), kps, cv::_OutputArray(desc));
    const int n = (int)kps.size();
    if (n > cap) return -n;
    static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
    std::memcpy(kps_out, kps.data(), (size_t)n * sizeof(cv::KeyPoint));
    for (int i = 0; i < n; ++i) std::memcpy(desc_out + 32 * (size_t)i, desc.ptr<unsigned char>(i), 32);
    return n;
}

// getters of the reference extractor: out = {scale_factors, inv_scale_factors, level_sigma_sq, inv_level_sigma_sq} x levels
void ref_orb_tables(int max_kp, float scale, int levels, float* out) {
    Rewind rewind;
    feature::orb_extractor ex(max_kp, scale, levels, 20, 7);
    const auto a = ex.get_scale_factors(), b = ex.get_inv_scale_factors(), c = ex.get_level_sigma_sq(), d = ex.get_inv_level_sigma_sq();
    for (int i = 0; i < levels; ++i) { out[i] = a[i]; out[levels + i] = b[i]; out[2 * levels + i] = c[i]; out[3 * levels + i] = d[i]; }
}

float ref_cos(float v) { return util::cos(v); }
float ref_sin(float v) { return util::sin(v); }

unsigned ref_hamming32(const unsigned char* a, const unsigned char* b) {
    cv::Mat ma(1, 32, CV_8UC1, const_cast<unsigned char*>(a)), mb(1, 32, CV_8UC1, const_cast<unsigned char*>(b));
    return match::compute_descriptor_distance_32(ma, mb);
}
unsigned ref_hamming64(const unsigned char* a, const unsigned char* b) {
    cv::Mat ma(1, 32, CV_8UC1, const_cast<unsigned char*>(a)), mb(1, 32, CV_8UC1, const_cast<unsigned char*>(b));
    return match::compute_descriptor_distance_64(ma, mb);
}

// match::angle_checker<int>(hist_len, num_bins_thr): append all deltas with ids 0..n-1, return the invalid (or valid) ids
int ref_angle_checker(const float* deltas, int n, int hist_len, int num_bins_thr, int valid, int* out) {
    Rewind rewind;
    match::angle_checker<int> ac(hist_len, num_bins_thr);
    for (int i = 0; i < n; ++i) ac.append_delta_angle(deltas[i], i);
    const auto v = valid ? ac.get_valid_matches() : ac.get_invalid_matches();
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

}  // extern "C"
