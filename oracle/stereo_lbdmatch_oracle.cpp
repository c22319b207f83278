// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of
//   match::stereo::compute                       src/PLPSLAM/match/stereo.cc:45-301   (array form)
//   cv::line_descriptor::BinaryDescriptorMatcher::match  (exact 1-NN by multi-index hashing)
//       src/PLPSLAM/feature/line_descriptor/binary_descriptor_matcher.cpp:197-255, 597-818, 922-937
// PARITY UNPINNED: the reference has no test for either.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "orb_oracle.hpp"

namespace oracle {

static unsigned ham256(const uint8_t* a, const uint8_t* b) {
    unsigned d = 0;
    for (int i = 0; i < 32; ++i) d += (unsigned)__builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

// ---------------------------------------------------------------------------------------- stereo
struct StereoIn {
    const std::vector<Image>* pl; const std::vector<Image>* pr;
    const KeyPoint* kl; int nl; const KeyPoint* kr; int nr;
    const uint8_t* dl; const uint8_t* dr;
    const float* sf; const float* isf;
    float fxb, tb;
};

static bool subpixel(const StereoIn& S, const KeyPoint& kpl, const KeyPoint& kpr, float& best_x_right, float& best_disp, float& best_corr) {
    const float x_right = kpr.x;
    const float isf = S.isf[kpl.octave];
    const int sxl = cv_round(kpl.x * isf), syl = cv_round(kpl.y * isf), sxr = cv_round(x_right * isf);
    constexpr int win = 5, slide = 5;
    const Image& L = (*S.pl)[kpl.octave];
    const Image& R = (*S.pr)[kpl.octave];
    const int ini_x = sxr - slide - win, end_x = sxr + slide + win;
    if (ini_x < 0 || R.cols <= end_x) return false;
    best_corr = (float)UINT_MAX;
    int best_off = 0;
    std::vector<float> corr(2 * slide + 1, -1);
    const float lc = (float)L.at(syl, sxl);
    for (int off = -slide; off <= slide; ++off) {
        const float rc = (float)R.at(syl, sxr + off);
        double s = 0;   // cv::norm(NORM_L1) on CV_32F accumulates in f64; the addends are integers, any order is exact
        for (int dy = -win; dy <= win; ++dy)
            for (int dx = -win; dx <= win; ++dx) {
                const float a = (float)L.at(syl + dy, sxl + dx) - lc, b = (float)R.at(syl + dy, sxr + off + dx) - rc;
                s += std::abs(a - b);
            }
        const float c = (float)s;
        if (c < best_corr) { best_corr = c; best_off = off; }
        corr[slide + off] = c;
    }
    if (best_off == -slide || best_off == slide) return false;
    const float c1 = corr[slide + best_off - 1], c2 = corr[slide + best_off], c3 = corr[slide + best_off + 1];
    const float x_delta = (float)((c1 - c3) / (2.0 * (c1 + c3) - 4.0 * c2));
    if (x_delta < -1.0 || 1.0 < x_delta) return false;
    best_x_right = S.sf[kpl.octave] * (sxr + best_off + x_delta);
    best_disp = kpl.x - best_x_right;
    return true;
}

static void stereo_compute(const StereoIn& S, float* x_right_out, float* depth_out) {
    const float min_disp = 0.0f, max_disp = S.fxb / S.tb;
    const unsigned hamm_thr = (100 + 50) / 2;
    const unsigned rows = (unsigned)(*S.pl)[0].rows;
    std::vector<std::vector<unsigned>> in_row(rows);
    for (int ir = 0; ir < S.nr; ++ir) {   // get_right_keypoint_indices_in_each_row(2.0)
        const float y = S.kr[ir].y, r = 2.0f * S.sf[S.kr[ir].octave];
        const int max_r = cv_ceil((double)(y + r)), min_r = cv_floor(y - r);
        for (int row = min_r; row <= max_r; ++row) in_row.at(row).push_back((unsigned)ir);
    }
    for (int i = 0; i < S.nl; ++i) { x_right_out[i] = -1.0f; depth_out[i] = -1.0f; }
    std::vector<std::pair<int, int>> corr_idx;
    for (int il = 0; il < S.nl; ++il) {
        const KeyPoint& kp = S.kl[il];
        const auto& cand = in_row.at((size_t)kp.y);
        if (cand.empty()) continue;
        const float min_x = kp.x - max_disp, max_x = kp.x - min_disp;
        if (max_x < 0) continue;
        unsigned best_ir = 0, best = hamm_thr;
        for (unsigned ir : cand) {
            const KeyPoint& kr = S.kr[ir];
            if (kr.octave < kp.octave - 1 || kr.octave > kp.octave + 1) continue;
            if (kr.x < min_x || max_x < kr.x) continue;
            const unsigned d = ham256(S.dl + 32 * (size_t)il, S.dr + 32 * (size_t)ir);
            if (d < best) { best_ir = ir; best = d; }
        }
        if (hamm_thr <= best) continue;
        float bx = -1.0f, bd = -1.0f, bc = (float)UINT_MAX;
        if (!subpixel(S, kp, S.kr[best_ir], bx, bd, bc)) continue;
        if (bd < min_disp || max_disp <= bd) continue;
        if (bd <= 0.0f) { bd = 0.01f; bx = kp.x - bd; }
        depth_out[il] = S.fxb / bd;
        x_right_out[il] = bx;
        corr_idx.emplace_back(std::make_pair(bc, il));   // float -> int truncation, as the reference's pair<int,int>
    }
    std::sort(corr_idx.begin(), corr_idx.end());
    const size_t median_i = corr_idx.size() / 2;
    const float median = corr_idx.empty() ? 0.0f : (float)corr_idx[median_i].first;
    const float thr = (float)(2.0 * median);
    for (size_t i = median_i; i < corr_idx.size(); ++i)
        if (thr < corr_idx[i].first) { x_right_out[corr_idx[i].second] = -1; depth_out[corr_idx[i].second] = -1; }
}

// ---------------------------------------------------------------------------------------- MIH 1-NN (K = 1)
// Mihasher(256, 32): m = 32 substrings of b = 8 bits, D = 128, d = 4.  Returns for each query the first-discovered
// train index among those at the minimum Hamming distance, and that distance; -1 / 256 where the reference reads
// uninitialised memory (nothing within distance 128).
static void mih_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int* train_idx, int* dist) {
    const int m = 32, D = 128, d = 4, b = 8;
    std::vector<std::vector<std::vector<uint32_t>>> H(m, std::vector<std::vector<uint32_t>>(256));   // SparseHashtable: bucket = insertion order
    for (int i = 0; i < nt; ++i)
        for (int k = 0; k < m; ++k) H[k][t[32 * (size_t)i + k]].push_back((uint32_t)i);   // split(): chunk k = byte k for b = 8
    std::vector<uint8_t> seen(nt);
    std::vector<uint32_t> numres(257), res(257);
    for (int qi = 0; qi < nq; ++qi) {
        const uint8_t* Q = q + 32 * (size_t)qi;
        std::fill(seen.begin(), seen.end(), 0);
        std::fill(numres.begin(), numres.end(), 0u);
        std::fill(res.begin(), res.end(), 0u);
        uint32_t n = 0;
        const uint32_t maxres = 1;
        int power[16];
        for (int s = 0; s <= d && n < maxres; ++s) {
            for (int k = 0; k < m; ++k) {
                const int curb = b;
                const uint64_t chunk = Q[k];
                uint64_t bitstr = 0;
                for (int i = 0; i < s; ++i) power[i] = i;
                power[s] = curb + 1;
                int bit = s - 1;
                while (true) {
                    if (bit != -1) {
                        bitstr ^= (power[bit] == bit) ? (uint64_t)1 << power[bit] : (uint64_t)3 << (power[bit] - 1);
                        power[bit]++;
                        bit--;
                    } else {
                        const auto& arr = H[k][(chunk ^ bitstr) & 0xff];
                        for (uint32_t index : arr)
                            if (!seen[index]) {
                                seen[index] = 1;
                                const int hammd = (int)ham256(t + 32 * (size_t)index, Q);
                                if (hammd <= D && numres[hammd] < maxres) res[hammd] = index + 1;
                                numres[hammd]++;
                            }
                        while (++bit < s && power[bit] == power[bit + 1] - 1) {
                            bitstr ^= (uint64_t)1 << (power[bit] - 1);
                            power[bit] = bit;
                        }
                        if (bit == s) break;
                    }
                }
                n = n + numres[s * m + k];
                if (n >= maxres) break;
            }
        }
        train_idx[qi] = -1; dist[qi] = 256;
        for (int h = 0; h <= D; ++h)
            if (numres[h] > 0) { train_idx[qi] = (int)res[h] - 1; dist[qi] = h; break; }
    }
}

}  // namespace oracle

using namespace oracle;

extern "C" {

// handles are oracle_orb_create() objects on which extract() ran for the left / right image (their pyramids are used)
struct OrbHandleView { OrbOracle ex; };
void oracle_stereo_compute(void* orb_left, void* orb_right, const KeyPoint* kl, int nl, const KeyPoint* kr, int nr, const uint8_t* dl,
                           const uint8_t* dr, float fxb, float tb, float* x_right_out, float* depth_out) {
    auto& L = ((OrbHandleView*)orb_left)->ex;
    auto& R = ((OrbHandleView*)orb_right)->ex;
    StereoIn S{&L.pyramid, &R.pyramid, kl, nl, kr, nr, dl, dr, L.scale_factors.data(), L.inv_scale_factors.data(), fxb, tb};
    stereo_compute(S, x_right_out, depth_out);
}

void oracle_lbd_match_1nn(const uint8_t* q, int nq, const uint8_t* t, int nt, int* train_idx, int* dist) { mih_match(q, nq, t, nt, train_idx, dist); }

}  // extern "C"
