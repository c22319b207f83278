// TEST INFRASTRUCTURE ONLY (oracle).  See orb_oracle.hpp for scope and pinning status.
#include "orb_oracle.hpp"

#include <cassert>

namespace oracle {

const int8_t kRbriefPattern[1024] = {
#include "rbrief_pattern.inc"
};

// ---------------------------------------------------------------- initialize
// orb_extractor.cc:235-287
void OrbOracle::initialize() {
    scale_factors = calc_scale_factors(p_.num_levels, p_.scale_factor);
    inv_scale_factors = calc_inv_scale_factors(p_.num_levels, p_.scale_factor);
    level_sigma_sq = calc_level_sigma_sq(p_.num_levels, p_.scale_factor);
    inv_level_sigma_sq = calc_inv_level_sigma_sq(p_.num_levels, p_.scale_factor);

    pyramid.resize(p_.num_levels);
    num_keypts_per_level.assign(p_.num_levels, 0);

    double desired = p_.max_num_keypts * (1.0 - 1.0 / p_.scale_factor) /
                     (1.0 - std::pow(1.0 / p_.scale_factor, static_cast<double>(p_.num_levels)));
    unsigned total = 0;
    for (unsigned level = 0; level + 1 < p_.num_levels; ++level) {
        num_keypts_per_level[level] = (unsigned)std::round(desired);
        total += num_keypts_per_level[level];
        desired *= 1.0 / p_.scale_factor;
    }
    num_keypts_per_level[p_.num_levels - 1] = (unsigned)std::max((int)p_.max_num_keypts - (int)total, 0);

    // u_max table, orb_extractor.cc:271-286
    u_max.assign(kHalfPatch + 1, 0);
    const unsigned vmax = (unsigned)std::floor(kHalfPatch * std::sqrt(2.0) / 2 + 1);
    const unsigned vmin = (unsigned)std::ceil(kHalfPatch * std::sqrt(2.0) / 2);
    for (unsigned v = 0; v <= vmax; ++v)
        u_max[v] = (int)std::round(std::sqrt((double)(kHalfPatch * kHalfPatch) - (double)(v * v)));
    for (unsigned v = kHalfPatch, v0 = 0; vmin <= v; --v) {
        while (u_max[v0] == u_max[v0 + 1]) ++v0;
        u_max[v] = (int)v0;
        ++v0;
    }
}

// ---------------------------------------------------------------- extract
void OrbOracle::extract(const Image& image, const Image* image_mask, std::vector<KeyPoint>& keypts,
                        std::vector<uint8_t>& descriptors) {
    if (image.empty()) return;  // :76-79 (outputs untouched)
    build_pyramid(image);

    if (!mask_is_initialized_ && !p_.mask_rects.empty()) {  // :89-93, :297-313
        if (rect_mask.empty()) {
            rect_mask = Image(image.rows, image.cols);
            std::fill(rect_mask.data.begin(), rect_mask.data.end(), (uint8_t)255);
        }
        for (const auto& r : p_.mask_rects) {
            const unsigned x_min = (unsigned)std::round((float)image.cols * r[0]);
            const unsigned x_max = (unsigned)std::round((float)image.cols * r[1]);
            const unsigned y_min = (unsigned)std::round((float)image.rows * r[2]);
            const unsigned y_max = (unsigned)std::round((float)image.rows * r[3]);
            // filled axis-aligned rectangle, corners inclusive, clipped to the image
            for (unsigned y = y_min; y <= y_max && y < (unsigned)image.rows; ++y)
                for (unsigned x = x_min; x <= x_max && x < (unsigned)image.cols; ++x) rect_mask.row(y)[x] = 0;
        }
        mask_is_initialized_ = true;
    }

    const Image* mask = nullptr;
    if (image_mask && !image_mask->empty()) mask = image_mask;
    else if (!rect_mask.empty()) mask = &rect_mask;
    fast_keypoints(mask);

    size_t n = 0;
    for (auto& v : level_keypts) n += v.size();
    descriptors.assign(n * 32, 0);
    keypts.clear();
    keypts.reserve(n);
    blurred.assign(p_.num_levels, Image());

    size_t offset = 0;
    for (unsigned level = 0; level < p_.num_levels; ++level) {
        auto& kl = level_keypts[level];
        if (kl.empty()) continue;
        blurred[level] = gaussian_blur_u8(pyramid[level], 7, 2.0);  // :148-149
        for (size_t i = 0; i < kl.size(); ++i) rbrief(kl[i], blurred[level], &descriptors[(offset + i) * 32]);
        offset += kl.size();
        for (const auto& k : kl) {  // correct_keypoint_scale :695-706
            KeyPoint o = k;
            if (level != 0) {
                const float s = scale_factors[level];
                o.x = k.x * s;
                o.y = k.y * s;
            }
            keypts.push_back(o);
        }
    }
}

// :315-326
void OrbOracle::build_pyramid(const Image& image) {
    pyramid.resize(p_.num_levels);
    pyramid[0] = image;
    for (unsigned level = 1; level < p_.num_levels; ++level) {
        const double scale = scale_factors[level];
        const int w = (int)std::round(image.cols * 1.0 / scale);
        const int h = (int)std::round(image.rows * 1.0 / scale);
        pyramid[level] = resize_linear_u8(pyramid[level - 1], w, h);
    }
}

// :328-466
void OrbOracle::fast_keypoints(const Image* mask) {
    candidates.assign(p_.num_levels, {});
    level_keypts.assign(p_.num_levels, {});
    auto is_in_mask = [&](unsigned y, unsigned x, float sf) {
        // mask.at<uchar>(y * sf, x * sf): float product truncated to int
        return mask->at((int)(y * sf), (int)(x * sf)) == 0;
    };
    constexpr unsigned overlap = 6, cell = 64;

    for (unsigned level = 0; level < p_.num_levels; ++level) {
        const float sf = scale_factors[level];
        const Image& img = pyramid[level];
        constexpr unsigned min_bx = kBorder, min_by = kBorder;
        const unsigned max_bx = img.cols - kBorder, max_by = img.rows - kBorder;
        const unsigned width = max_bx - min_bx, height = max_by - min_by;
        const unsigned num_cols = (unsigned)std::ceil(width / cell) + 1;   // integer quotient (quirk)
        const unsigned num_rows = (unsigned)std::ceil(height / cell) + 1;

        std::vector<KeyPoint>& todo = candidates[level];
        std::vector<FastPoint> in_cell;
        for (unsigned i = 0; i < num_rows; ++i) {
            const unsigned min_y = min_by + i * cell;
            if (max_by - overlap <= min_y) continue;
            unsigned max_y = min_y + cell + overlap;
            if (max_by < max_y) max_y = max_by;
            for (unsigned j = 0; j < num_cols; ++j) {
                const unsigned min_x = min_bx + j * cell;
                if (max_bx - overlap <= min_x) continue;
                unsigned max_x = min_x + cell + overlap;
                if (max_bx < max_x) max_x = max_bx;
                if (mask) {
                    if (is_in_mask(min_y, min_x, sf) || is_in_mask(max_y, min_x, sf) ||
                        is_in_mask(min_y, max_x, sf) || is_in_mask(max_y, max_x, sf))
                        continue;
                }
                const uint8_t* base = img.row(min_y) + min_x;
                fast9_16_nms(base, img.cols, (int)(max_x - min_x), (int)(max_y - min_y), (int)p_.ini_fast_thr, in_cell);
                if (in_cell.empty())
                    fast9_16_nms(base, img.cols, (int)(max_x - min_x), (int)(max_y - min_y), (int)p_.min_fast_thr, in_cell);
                if (in_cell.empty()) continue;
                for (const auto& fp : in_cell) {
                    KeyPoint k{(float)fp.x, (float)fp.y, 7.f, -1.f, (float)fp.score, 0, -1};
                    k.x += j * cell;
                    k.y += i * cell;
                    if (mask && is_in_mask((unsigned)(min_by + k.y), (unsigned)(min_bx + k.x), sf)) continue;
                    todo.push_back(k);
                }
            }
        }

        auto& out = level_keypts[level];
        out = distribute_via_tree(todo, min_bx, max_bx, min_by, max_by, num_keypts_per_level[level]);
        const unsigned scaled_patch_size = (unsigned)(kPatch * scale_factors[level]);
        for (auto& k : out) {
            k.x += min_bx;
            k.y += min_by;
            k.octave = (int)level;
            k.size = (float)scaled_patch_size;
        }
    }
    for (unsigned level = 0; level < p_.num_levels; ++level)
        for (auto& k : level_keypts[level]) k.angle = ic_angle(pyramid[level], k.x, k.y);
}

// ---------------------------------------------------------------- quadtree
namespace {
struct Node {
    std::vector<KeyPoint> kps;
    int bx = 0, by = 0, ex = 0, ey = 0;
    std::list<Node>::iterator self;
    bool is_leaf = false;
    uint64_t seq = 0;  // creation order == "pointer value" of the list node
};
using NodeList = std::list<Node>;
struct PoolEntry {
    int count;
    uint64_t seq;
    Node* node;
    bool operator<(const PoolEntry& o) const { return count != o.count ? count < o.count : seq < o.seq; }
};

// orb_extractor_node.cc:31-80
std::array<Node, 4> divide(const Node& n) {
    const unsigned half_x = (unsigned)cv_ceil((n.ex - n.bx) / 2.0);
    const unsigned half_y = (unsigned)cv_ceil((n.ey - n.by) / 2.0);
    std::array<Node, 4> c;
    const int cx = (int)(n.bx + half_x), cy = (int)(n.by + half_y);
    c[0].bx = n.bx; c[0].by = n.by; c[0].ex = cx;   c[0].ey = cy;
    c[1].bx = cx;   c[1].by = n.by; c[1].ex = n.ex; c[1].ey = cy;
    c[2].bx = n.bx; c[2].by = cy;   c[2].ex = cx;   c[2].ey = n.ey;
    c[3].bx = cx;   c[3].by = cy;   c[3].ex = n.ex; c[3].ey = n.ey;
    for (const auto& k : n.kps) {
        unsigned idx = 0;
        if ((float)(n.bx + half_x) <= k.x) idx += 1;
        if ((float)(n.by + half_y) <= k.y) idx += 2;
        c[idx].kps.push_back(k);
    }
    return c;
}
}  // namespace

std::vector<KeyPoint> OrbOracle::distribute_via_tree(const std::vector<KeyPoint>& todo, int min_x, int max_x,
                                                     int min_y, int max_y, unsigned num_keypts) const {
    uint64_t seq = 0;
    NodeList nodes;
    // initialize_nodes :557-637
    {
        const double ratio = (double)(max_x - min_x) / (max_y - min_y);
        double delta_x, delta_y;
        unsigned nx, ny;
        if (ratio > 1) {
            nx = (unsigned)std::round(ratio); ny = 1;
            delta_x = (double)(max_x - min_x) / nx;
            delta_y = max_y - min_y;
        } else {
            nx = 1; ny = (unsigned)std::round(1 / ratio);
            delta_x = max_x - min_y;  // quirk kept (:580)
            delta_y = (double)(max_y - min_y) / ny;
        }
        const unsigned n_init = nx * ny;
        std::vector<Node*> init(n_init);
        for (unsigned i = 0; i < n_init; ++i) {
            const unsigned ix = i % nx, iy = i / nx;
            Node n;
            n.bx = (int)(delta_x * ix); n.by = (int)(delta_y * iy);
            n.ex = (int)(delta_x * (ix + 1)); n.ey = (int)(delta_y * (iy + 1));
            n.seq = seq++;
            nodes.push_back(n);
            init[i] = &nodes.back();
        }
        for (const auto& k : todo) {
            const unsigned ix = (unsigned)(k.x / delta_x);
            const unsigned iy = (unsigned)(k.y / delta_y);
            init.at(ix + iy * nx)->kps.push_back(k);
        }
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->kps.empty()) { it = nodes.erase(it); continue; }
            it->is_leaf = (it->kps.size() == 1);
            ++it;
        }
    }

    std::vector<PoolEntry> pool;
    // assign_child_nodes :639-657 (children are never flagged leaf)
    auto assign = [&](std::array<Node, 4>& children, std::vector<PoolEntry>& dst) {
        for (auto& c : children) {
            if (c.kps.empty()) continue;
            c.seq = seq++;
            nodes.push_front(c);
            if (c.kps.size() == 1) continue;
            dst.push_back({(int)c.kps.size(), c.seq, &nodes.front()});
            nodes.front().self = nodes.begin();
        }
    };

    bool is_filled = false;
    while (true) {  // :482-518
        const size_t prev_size = nodes.size();
        pool.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->is_leaf) { ++it; continue; }
            auto ch = divide(*it);
            assign(ch, pool);
            it = nodes.erase(it);
        }
        if (num_keypts <= nodes.size() || nodes.size() == prev_size) { is_filled = true; break; }
        if (num_keypts < nodes.size() + pool.size()) { is_filled = false; break; }
    }
    while (!is_filled) {  // :520-552
        const size_t prev_size = nodes.size();
        std::vector<PoolEntry> prev_pool = pool;
        pool.clear();
        std::sort(prev_pool.rbegin(), prev_pool.rend());
        for (const auto& e : prev_pool) {
            auto ch = divide(*e.node);
            assign(ch, pool);
            nodes.erase(e.node->self);
            if (num_keypts <= nodes.size()) { is_filled = true; break; }
        }
        if (is_filled || num_keypts <= nodes.size() || nodes.size() == prev_size) { is_filled = true; break; }
    }

    // find_keypoints_with_max_response :659-685 (first maximum in node order)
    std::vector<KeyPoint> result;
    result.reserve(nodes.size());
    for (auto& n : nodes) {
        KeyPoint best = n.kps[0];
        double max_r = best.response;
        for (size_t k = 1; k < n.kps.size(); ++k)
            if (n.kps[k].response > max_r) { best = n.kps[k]; max_r = n.kps[k].response; }
        result.push_back(best);
    }
    return result;
}

// ---------------------------------------------------------------- orientation :708-735
float OrbOracle::ic_angle(const Image& img, float px, float py) const {
    int m01 = 0, m10 = 0;
    const int cx = cv_round(px), cy = cv_round(py);
    const int step = img.cols;
    const uint8_t* center = img.row(cy) + cx;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        const int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * step], vm = center[u - v * step];
            v_sum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * v_sum;
    }
    return fast_atan2f_deg((float)m01, (float)m10);
}

// ---------------------------------------------------------------- rBRIEF :747-807 (scalar GET_VALUE build)
void OrbOracle::rbrief(const KeyPoint& kp, const Image& img, uint8_t* desc) const {
    const float angle = (float)(kp.angle * M_PI / 180.0);
    const float c = trig::cos(angle), s = trig::sin(angle);
    const int step = img.cols;
    const uint8_t* center = img.row(cv_round(kp.y)) + cv_round(kp.x);
    auto tap = [&](int idx) -> int {
        const float x = (float)kRbriefPattern[idx], y = (float)kRbriefPattern[idx + 1];
        // built with -ffp-contract=off: separate f32 multiplies and adds, as the scalar reference build
        const int r = cv_round(x * s + y * c);
        const int q = cv_round(x * c - y * s);
        return center[r * step + q];
    };
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int b = 0; b < 8; ++b) {
            const int idx = i * 32 + b * 4;
            val |= (tap(idx) < tap(idx + 2)) << b;
        }
        desc[i] = (uint8_t)val;
    }
}

}  // namespace oracle
