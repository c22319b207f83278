// Host driver + C ABI of the array-form Hamming matchers (include/plp_front.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "libstdcxx_sort.hpp"
#include "match_device.hpp"
#include "plp_common.hpp"

using namespace plp;

struct plp_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf klist, klist2, kcount, claim, full_list, sorted, sorted_xr, row_start, dbg;  // scratch of the device path
    DevBuf stage;                            // one slab for the host-pointer path
    HostPinned pin;                          // page-locked staging of host images (post-extract depth)
    std::mutex mu;
};

namespace {

// An EMPTY side -- a frame without key points / key lines (n_cap = 0) or an empty set of landmarks / queries (m_cap = 0): nothing can match, the reference's loops do not
// run (projection.cc, robust.cc, fuse.cc: every matcher starts from `unsigned int num_matches = 0` and iterates over the landmarks).  The result is defined without a
// kernel: every out_match slot -1, every out_num 0 (fuse modes: every out_query_best -1).  Returns 1 when the call is such a call and well-formed, 0 otherwise.
int empty_side(const plp_match_args* a) {
    if (!a || a->B <= 0 || a->n_cap < 0 || a->m_cap < 0 || (a->n_cap > 0 && a->m_cap > 0)) return 0;
    if (a->mode < PLP_MATCH_MODE_LANDMARKS || a->mode > PLP_MATCH_MODE_TRIANGULATION) return 0;
    const bool fuse = a->mode == PLP_MATCH_MODE_FUSE || a->mode == PLP_MATCH_MODE_FUSE_LINE;
    if (fuse ? (a->m_cap > 0 && !a->out_query_best) : (!a->out_num || (a->n_cap > 0 && !a->out_match))) return 0;
    return 1;
}

plp_status check_args(const plp_match_args* a) {
    if (!a) return set_error(PLP_ERR_INVALID_ARG, "args is NULL");
    if (a->B <= 0 || a->n_cap <= 0 || a->m_cap <= 0) return set_error(PLP_ERR_INVALID_ARG, "B must be positive, n_cap and m_cap non-negative (with the outputs an empty side needs)");
    if (a->n_cap > 8192) return set_error(PLP_ERR_UNSUPPORTED, "more than 8192 key points per frame");
    if (!a->t_desc || !a->q_desc) return set_error(PLP_ERR_INVALID_ARG, "descriptor arrays are required");
    if (a->mode != PLP_MATCH_MODE_FUSE && a->mode != PLP_MATCH_MODE_FUSE_LINE && (!a->out_match || !a->out_num)) return set_error(PLP_ERR_INVALID_ARG, "output arrays are required");
    if (a->mode == PLP_MATCH_MODE_BRUTE_FORCE) {
        if (a->check_orientation && (!a->t_angle || !a->q_angle)) return set_error(PLP_ERR_INVALID_ARG, "check_orientation needs t_angle and q_angle");
    } else if (a->mode == PLP_MATCH_MODE_LANDMARKS || a->mode == PLP_MATCH_MODE_LAST_FRAME) {
        if (!a->t_kps || !a->q_reproj || !a->q_level || !a->scale_factors) return set_error(PLP_ERR_INVALID_ARG, "t_kps, q_reproj, q_level, scale_factors are required");
        if (a->num_levels <= 0 || a->num_levels > 16) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
        if (a->grid.cols <= 0 || a->grid.rows <= 0 || a->grid.cols * a->grid.rows > 4096) return set_error(PLP_ERR_INVALID_ARG, "grid must have 1..4096 cells");
        if (a->mode == PLP_MATCH_MODE_LAST_FRAME && a->check_orientation && !a->q_angle) return set_error(PLP_ERR_INVALID_ARG, "check_orientation needs q_angle");
    } else if (a->mode == PLP_MATCH_MODE_LANDMARKS_LINE || a->mode == PLP_MATCH_MODE_LAST_FRAME_LINE) {
        if (!a->t_kl || !a->q_reproj || !a->q_reproj2 || !a->q_level || !a->scale_factors) return set_error(PLP_ERR_INVALID_ARG, "t_kl, q_reproj, q_reproj2, q_level, scale_factors are required");
        if (a->mode == PLP_MATCH_MODE_LANDMARKS_LINE && !a->t_kp_octave) return set_error(PLP_ERR_INVALID_ARG, "t_kp_octave is required");
        if (a->num_levels <= 0 || a->num_levels > 16) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
    } else if (a->mode == PLP_MATCH_MODE_BOW) {
        if (!a->q_group || !a->t_group) return set_error(PLP_ERR_INVALID_ARG, "q_group and t_group are required");
        if (a->check_orientation && (!a->t_angle || !a->q_angle)) return set_error(PLP_ERR_INVALID_ARG, "check_orientation needs t_angle and q_angle");
    } else if (a->mode == PLP_MATCH_MODE_FUSE) {
        if (!a->t_kps || !a->q_reproj_d || !a->q_level || !a->scale_factors || !a->inv_level_sigma_sq || !a->out_query_best) return set_error(PLP_ERR_INVALID_ARG, "t_kps, q_reproj_d, q_level, scale_factors, inv_level_sigma_sq, out_query_best are required");
        if (a->num_levels <= 0 || a->num_levels > 16) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
        if (a->grid.cols <= 0 || a->grid.rows <= 0 || a->grid.cols * a->grid.rows > 4096) return set_error(PLP_ERR_INVALID_ARG, "grid must have 1..4096 cells");
    } else if (a->mode == PLP_MATCH_MODE_FUSE_LINE) {
        if (!a->t_kl || !a->q_reproj_d || !a->q_reproj2_d || !a->q_level || !a->scale_factors || !a->inv_level_sigma_sq || !a->out_query_best) return set_error(PLP_ERR_INVALID_ARG, "t_kl, q_reproj_d, q_reproj2_d, q_level, scale_factors, inv_level_sigma_sq, out_query_best are required");
        if (a->num_levels <= 0 || a->num_levels > 16) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
    } else if (a->mode == PLP_MATCH_MODE_TRIANGULATION) {
        if (!a->q_group || !a->t_group || !a->q_bearing || !a->t_bearing || !a->epipolar || !a->q_level || !a->scale_factors) return set_error(PLP_ERR_INVALID_ARG, "q_group, t_group, q_bearing, t_bearing, epipolar, q_level, scale_factors are required");
        if (a->num_levels <= 0 || a->num_levels > 16) return set_error(PLP_ERR_INVALID_ARG, "num_levels must be in [1,16]");
        if (a->check_orientation && (!a->t_angle || !a->q_angle)) return set_error(PLP_ERR_INVALID_ARG, "check_orientation needs t_angle and q_angle");
    } else return set_error(PLP_ERR_INVALID_ARG, "unknown mode");
    if (a->hamm_dist_thr < 0 || a->hamm_dist_thr > 256 || a->level_window < 0 || a->level_window > 2) return set_error(PLP_ERR_INVALID_ARG, "hamm_dist_thr / level_window out of range");
    return PLP_OK;
}

plp_status run_device(plp_matcher* c, const plp_match_args* a, hipStream_t st) {
    PLP_HIP(hipSetDevice(c->device));
    const size_t qn = (size_t)a->B * a->m_cap;
    PLP_HIP(c->klist.reserve(qn * kMatchK * 4));
    PLP_HIP(c->klist2.reserve(qn * kMatchK * 4));
    PLP_HIP(c->kcount.reserve(qn * 4));
    PLP_HIP(c->claim.reserve(qn * 4));
    PLP_HIP(c->full_list.reserve(qn * 4));
    const size_t tn_ = (size_t)a->B * a->n_cap;
    PLP_HIP(c->sorted.reserve(tn_ * sizeof(StagedTarget)));
    PLP_HIP(c->sorted_xr.reserve(tn_ * 4));
    PLP_HIP(c->row_start.reserve((size_t)a->B * 4104 * 2));
    if (!c->dbg.p) { PLP_HIP(c->dbg.reserve(16)); PLP_HIP(hipMemsetAsync(c->dbg.p, 0, 16, st)); }
    MatchProblem P{};
    P.mode = a->mode; P.n_cap = a->n_cap; P.m_cap = a->m_cap; P.q_desc_stride = a->q_desc_stride > 0 ? a->q_desc_stride : a->m_cap;
    P.t_kps = (a->mode == PLP_MATCH_MODE_LANDMARKS || a->mode == PLP_MATCH_MODE_LAST_FRAME || a->mode == PLP_MATCH_MODE_FUSE) ? a->t_kps : nullptr;
    P.q_group = a->q_group; P.t_group = a->t_group; P.q_reproj_d = a->q_reproj_d; P.out_query_best = a->out_query_best;
    P.hamm_dist_thr = a->hamm_dist_thr; P.level_window = a->level_window; P.flags = a->flags;
    P.q_reproj2_d = a->q_reproj2_d; P.q_bearing = a->q_bearing; P.t_bearing = a->t_bearing; P.epipolar = a->epipolar;
    for (int i = 0; i < 16; ++i) P.inv_level_sigma_sq[i] = (a->inv_level_sigma_sq && i < a->num_levels) ? a->inv_level_sigma_sq[i] : 1.0f;
    P.t_desc = a->t_desc; P.t_x_right = a->t_x_right; P.t_occupied = a->t_occupied; P.t_angle = a->t_angle; P.t_counts = a->t_counts;
    P.q_valid = a->q_valid; P.q_reproj = a->q_reproj; P.q_x_right = a->q_x_right; P.q_level = a->q_level; P.q_angle = a->q_angle;
    P.q_desc = a->q_desc; P.q_has_obs = a->q_has_obs; P.q_counts = a->q_counts;
    P.t_kl = a->t_kl; P.t_kp_octave = a->t_kp_octave; P.t_x_right2 = a->t_x_right2; P.q_reproj2 = a->q_reproj2; P.q_x_right2 = a->q_x_right2;
    P.is_rgbd = a->is_rgbd; P.num_levels_lsd = a->num_levels_lsd;
    P.margin = a->margin; P.lowe_ratio = a->lowe_ratio; P.direction = a->direction; P.check_orientation = a->check_orientation;
    P.num_levels = a->num_levels;
    for (int i = 0; i < 16; ++i) P.scale_factors[i] = (a->scale_factors && i < a->num_levels) ? a->scale_factors[i] : 1.0f;
    P.grid_min_x = a->grid.min_x; P.grid_min_y = a->grid.min_y; P.inv_cell_w = a->grid.inv_cell_width; P.inv_cell_h = a->grid.inv_cell_height;
    P.grid_cols = a->grid.cols; P.grid_rows = a->grid.rows;
    P.klist = (uint32_t*)c->klist.p; P.klist2 = (uint32_t*)c->klist2.p; P.kcount = (int32_t*)c->kcount.p; P.claim = (int32_t*)c->claim.p; P.full_list = (int32_t*)c->full_list.p;
    P.sorted = (StagedTarget*)c->sorted.p; P.sorted_xr = (float*)c->sorted_xr.p; P.cell_start = (uint16_t*)c->row_start.p; P.dbg = (int32_t*)c->dbg.p;
    P.out_match = a->out_match; P.out_num = a->out_num;
    P.lds_targets = a->t_count_hint > 0 ? std::min(a->t_count_hint, a->n_cap) : a->n_cap;
    launch_match(st, P, a->B);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

}  // namespace

extern "C" {

plp_status plp_matcher_create(int device, plp_matcher** out) {
    if (!out) return set_error(PLP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return set_error(PLP_ERR_NO_DEVICE, "no HIP device visible: the matchers have no CPU fallback");
    if (device < 0 || device >= n) return set_error(PLP_ERR_INVALID_ARG, "device index out of range");
    PLP_HIP(hipSetDevice(device));
    PLP_HIP(configure_match_kernels());
    plp_matcher* c = new plp_matcher();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_error(PLP_ERR_HIP, "hipStreamCreate failed"); }
    *out = c;
    return PLP_OK;
}

void plp_matcher_destroy(plp_matcher* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

plp_status plp_match_device(plp_matcher* c, const plp_match_args* a, void* hip_stream) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    if (empty_side(a)) {
        std::lock_guard<std::mutex> lk(c->mu);
        PLP_HIP(hipSetDevice(c->device));
        hipStream_t st = (hipStream_t)hip_stream;
        if (a->mode == PLP_MATCH_MODE_FUSE || a->mode == PLP_MATCH_MODE_FUSE_LINE) {
            if (a->m_cap > 0) PLP_HIP(hipMemsetAsync(a->out_query_best, 0xFF, (size_t)a->B * a->m_cap * 4, st));
            return PLP_OK;
        }
        if (a->n_cap > 0) PLP_HIP(hipMemsetAsync(a->out_match, 0xFF, (size_t)a->B * a->n_cap * 4, st));
        PLP_HIP(hipMemsetAsync(a->out_num, 0, (size_t)a->B * 4, st));
        return PLP_OK;
    }
    PLP_TRY(check_args(a));
    std::lock_guard<std::mutex> lk(c->mu);
    return run_device(c, a, (hipStream_t)hip_stream);
}

plp_status plp_match_host(plp_matcher* c, const plp_match_args* a) {
    if (!c) return set_error(PLP_ERR_INVALID_ARG, "ctx is NULL");
    if (empty_side(a)) {
        if (a->mode == PLP_MATCH_MODE_FUSE || a->mode == PLP_MATCH_MODE_FUSE_LINE) { std::fill_n(a->out_query_best, (size_t)a->B * a->m_cap, -1); return PLP_OK; }
        std::fill_n(a->out_match, (size_t)a->B * a->n_cap, -1);
        std::fill_n(a->out_num, (size_t)a->B, 0);
        return PLP_OK;
    }
    PLP_TRY(check_args(a));
    if (a->q_desc_stride != 0 && a->q_desc_stride != a->m_cap) return set_error(PLP_ERR_UNSUPPORTED, "q_desc_stride is a device-path option (overlapping query windows of a batched replay)");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t tn = (size_t)a->B * a->n_cap, qn = (size_t)a->B * a->m_cap;
    // slab layout (256-byte aligned pieces)
    struct Piece { const void* src; size_t bytes; size_t off; };
    std::vector<Piece> in;
    size_t off = 0;
    auto add = [&](const void* src, size_t bytes) -> size_t {
        const size_t o = off;
        in.push_back({src, bytes, o});
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_tk = add(a->t_kps, a->t_kps ? tn * sizeof(plp_keypoint) : 0), o_td = add(a->t_desc, tn * 32);
    const size_t o_tx = add(a->t_x_right, a->t_x_right ? tn * 4 : 0), o_to = add(a->t_occupied, a->t_occupied ? tn : 0);
    const size_t o_ta = add(a->t_angle, a->t_angle ? tn * 4 : 0), o_tc = add(a->t_counts, a->t_counts ? (size_t)a->B * 4 : 0);
    const size_t o_qv = add(a->q_valid, a->q_valid ? qn : 0), o_qr = add(a->q_reproj, a->q_reproj ? qn * 8 : 0);
    const size_t o_qx = add(a->q_x_right, a->q_x_right ? qn * 4 : 0), o_ql = add(a->q_level, a->q_level ? qn * 4 : 0);
    const size_t o_qa = add(a->q_angle, a->q_angle ? qn * 4 : 0), o_qd = add(a->q_desc, qn * 32);
    const size_t o_qh = add(a->q_has_obs, a->q_has_obs ? qn : 0), o_qc = add(a->q_counts, a->q_counts ? (size_t)a->B * 4 : 0);
    const size_t o_kl = add(a->t_kl, a->t_kl ? tn * sizeof(plp_keyline) : 0), o_ko = add(a->t_kp_octave, a->t_kp_octave ? tn * 4 : 0);
    const size_t o_tx2 = add(a->t_x_right2, a->t_x_right2 ? tn * 4 : 0), o_qr2 = add(a->q_reproj2, a->q_reproj2 ? qn * 8 : 0);
    const size_t o_qx2 = add(a->q_x_right2, a->q_x_right2 ? qn * 4 : 0);
    const size_t o_qg = add(a->q_group, a->q_group ? qn * 4 : 0), o_tg = add(a->t_group, a->t_group ? tn * 4 : 0);
    const size_t o_qrd = add(a->q_reproj_d, a->q_reproj_d ? qn * 16 : 0);
    const size_t o_qrd2 = add(a->q_reproj2_d, a->q_reproj2_d ? qn * 16 : 0);
    const size_t o_qb = add(a->q_bearing, a->q_bearing ? qn * 24 : 0), o_tb = add(a->t_bearing, a->t_bearing ? tn * 24 : 0);
    const size_t o_ep = add(a->epipolar, a->epipolar ? (size_t)a->B * 96 : 0);
    const size_t o_oq = off; off += (qn * 4 + 255) / 256 * 256;
    const size_t o_om = off; off += (tn * 4 + 255) / 256 * 256;
    const size_t o_on = off; off += ((size_t)a->B * 4 + 255) / 256 * 256;
    PLP_HIP(c->stage.reserve(off));
    uint8_t* base = (uint8_t*)c->stage.p;
    for (const Piece& p : in)
        if (p.src && p.bytes) PLP_HIP(hipMemcpyAsync(base + p.off, p.src, p.bytes, hipMemcpyHostToDevice, st));
    plp_match_args d = *a;
    auto dp = [&](const void* src, size_t o) -> const void* { return src ? base + o : nullptr; };
    d.t_kps = (const plp_keypoint*)dp(a->t_kps, o_tk); d.t_desc = (const uint8_t*)dp(a->t_desc, o_td);
    d.t_x_right = (const float*)dp(a->t_x_right, o_tx); d.t_occupied = (const uint8_t*)dp(a->t_occupied, o_to);
    d.t_angle = (const float*)dp(a->t_angle, o_ta); d.t_counts = (const int32_t*)dp(a->t_counts, o_tc);
    d.q_valid = (const uint8_t*)dp(a->q_valid, o_qv); d.q_reproj = (const float*)dp(a->q_reproj, o_qr);
    d.q_x_right = (const float*)dp(a->q_x_right, o_qx); d.q_level = (const int32_t*)dp(a->q_level, o_ql);
    d.q_angle = (const float*)dp(a->q_angle, o_qa); d.q_desc = (const uint8_t*)dp(a->q_desc, o_qd);
    d.q_has_obs = (const uint8_t*)dp(a->q_has_obs, o_qh); d.q_counts = (const int32_t*)dp(a->q_counts, o_qc);
    d.t_kl = (const plp_keyline*)dp(a->t_kl, o_kl); d.t_kp_octave = (const int32_t*)dp(a->t_kp_octave, o_ko);
    d.t_x_right2 = (const float*)dp(a->t_x_right2, o_tx2); d.q_reproj2 = (const float*)dp(a->q_reproj2, o_qr2);
    d.q_x_right2 = (const float*)dp(a->q_x_right2, o_qx2);
    d.q_group = (const int32_t*)dp(a->q_group, o_qg); d.t_group = (const int32_t*)dp(a->t_group, o_tg);
    d.q_reproj_d = (const double*)dp(a->q_reproj_d, o_qrd);
    d.q_reproj2_d = (const double*)dp(a->q_reproj2_d, o_qrd2); d.q_bearing = (const double*)dp(a->q_bearing, o_qb);
    d.t_bearing = (const double*)dp(a->t_bearing, o_tb); d.epipolar = (const double*)dp(a->epipolar, o_ep);
    d.out_query_best = a->out_query_best ? (int32_t*)(base + o_oq) : nullptr;
    d.out_match = (int32_t*)(base + o_om); d.out_num = (int32_t*)(base + o_on);
    PLP_TRY(run_device(c, &d, st));
    if (a->mode == PLP_MATCH_MODE_FUSE || a->mode == PLP_MATCH_MODE_FUSE_LINE) {
        PLP_HIP(hipMemcpyAsync(a->out_query_best, base + o_oq, qn * 4, hipMemcpyDeviceToHost, st));
        PLP_HIP(hipStreamSynchronize(st));
        return PLP_OK;
    }
    PLP_HIP(hipMemcpyAsync(a->out_match, base + o_om, tn * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipMemcpyAsync(a->out_num, base + o_on, (size_t)a->B * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    return PLP_OK;
}

// order in which Mihasher::query enumerates the bit-flip patterns with s ones inside a b-bit substring
// (binary_descriptor_matcher.cpp:671-735): rank[pattern] = position inside its popcount class
static MihRanks mih_ranks() {
    MihRanks R;
    for (int i = 0; i < 256; ++i) R.rank[i] = 255;
    const int curb = 8;
    for (int s = 0; s <= 8; ++s) {
        int power[16];
        unsigned long long bitstr = 0;
        for (int i = 0; i < s; ++i) power[i] = i;
        power[s] = curb + 1;
        int bit = s - 1, pos = 0;
        while (true) {
            if (bit != -1) {
                bitstr ^= (power[bit] == bit) ? 1ull << power[bit] : 3ull << (power[bit] - 1);
                power[bit]++;
                bit--;
            } else {
                if (bitstr < 256 && R.rank[bitstr] == 255) R.rank[bitstr] = (uint8_t)std::min(pos, 254);
                ++pos;
                while (++bit < s && power[bit] == power[bit + 1] - 1) {
                    bitstr ^= 1ull << (power[bit] - 1);
                    power[bit] = bit;
                }
                if (bit == s) break;
            }
        }
    }
    return R;
}

plp_status plp_lbd_match_1nn_device(plp_matcher* c, const uint8_t* d_q, const int32_t* d_q_counts, int32_t nq_cap, const uint8_t* d_t,
                                    const int32_t* d_t_counts, int32_t nt_cap, int32_t B, int32_t* d_train_idx, int32_t* d_dist, void* hip_stream) {
    if (!c || !d_q || !d_t || !d_train_idx || !d_dist) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (nq_cap <= 0 || nt_cap <= 0 || B <= 0 || nt_cap > 65535) return set_error(PLP_ERR_INVALID_ARG, "bad sizes");
    PLP_HIP(hipSetDevice(c->device));
    static const MihRanks R = mih_ranks();
    launch_lbd_match_1nn((hipStream_t)hip_stream, d_q, d_q_counts, nq_cap, d_t, d_t_counts, nt_cap, R, d_train_idx, d_dist, B);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_lbd_match_1nn_host(plp_matcher* c, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t* train_idx, int32_t* dist) {
    if (!c || !q || !t || !train_idx || !dist) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (nq <= 0 || nt <= 0) return PLP_OK;   // the reference prints an error and returns with `matches` untouched (:201-205)
    if (nt > 65535) return set_error(PLP_ERR_UNSUPPORTED, "more than 65535 train descriptors (16-bit index in the tie-break key)");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    const size_t bq = ((size_t)nq * 32 + 255) / 256 * 256, bt = ((size_t)nt * 32 + 255) / 256 * 256, bo = ((size_t)nq * 4 + 255) / 256 * 256;
    PLP_HIP(c->stage.reserve(bq + bt + 2 * bo));
    uint8_t* base = (uint8_t*)c->stage.p;
    PLP_HIP(hipMemcpyAsync(base, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    PLP_HIP(hipMemcpyAsync(base + bq, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    static const MihRanks R = mih_ranks();
    launch_lbd_match_1nn(c->stream, base, nullptr, nq, base + bq, nullptr, nt, R, (int32_t*)(base + bq + bt), (int32_t*)(base + bq + bt + bo), 1);
    PLP_HIP(hipGetLastError());
    PLP_HIP(hipMemcpyAsync(train_idx, base + bq + bt, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    PLP_HIP(hipMemcpyAsync(dist, base + bq + bt + bo, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    PLP_HIP(hipStreamSynchronize(c->stream));
    return PLP_OK;
}

plp_status plp_match_area_host(plp_matcher* c, const plp_keypoint* kps_1, const uint8_t* desc_1, int32_t n1, const plp_keypoint* kps_2,
                               const uint8_t* desc_2, int32_t n2, const plp_match_grid* grid, float* prev_matched_pts, int32_t margin,
                               float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches) {
    if (!c || !grid || !matched_2_in_1 || !num_matches) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    *num_matches = 0;
    if (n1 <= 0) return PLP_OK;
    if (!kps_1 || !desc_1 || !prev_matched_pts || n2 < 0 || (n2 > 0 && (!kps_2 || !desc_2)) || n2 > 65535) return set_error(PLP_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t n2c = std::max(n2, 1);
    const size_t o_k1 = 0, o_d1 = o_k1 + al((size_t)n1 * sizeof(plp_keypoint)), o_k2 = o_d1 + al((size_t)n1 * 32), o_d2 = o_k2 + al(n2c * sizeof(plp_keypoint));
    const size_t o_pp = o_d2 + al(n2c * 32), o_m = o_pp + al((size_t)n1 * 8), o_n = o_m + al((size_t)n1 * 4), o_s = o_n + 256, total = o_s + al(n2c * 8);
    PLP_HIP(c->stage.reserve(total));
    uint8_t* base = (uint8_t*)c->stage.p;
    PLP_HIP(hipMemcpyAsync(base + o_k1, kps_1, (size_t)n1 * sizeof(plp_keypoint), hipMemcpyHostToDevice, st));
    PLP_HIP(hipMemcpyAsync(base + o_d1, desc_1, (size_t)n1 * 32, hipMemcpyHostToDevice, st));
    if (n2) {
        PLP_HIP(hipMemcpyAsync(base + o_k2, kps_2, (size_t)n2 * sizeof(plp_keypoint), hipMemcpyHostToDevice, st));
        PLP_HIP(hipMemcpyAsync(base + o_d2, desc_2, (size_t)n2 * 32, hipMemcpyHostToDevice, st));
    }
    PLP_HIP(hipMemcpyAsync(base + o_pp, prev_matched_pts, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    AreaArgs A{};
    A.kps1 = (const plp_keypoint*)(base + o_k1); A.desc1 = base + o_d1; A.kps2 = (const plp_keypoint*)(base + o_k2); A.desc2 = base + o_d2;
    A.n1 = n1; A.n2 = n2;
    A.grid_min_x = grid->min_x; A.grid_min_y = grid->min_y; A.inv_cell_w = grid->inv_cell_width; A.inv_cell_h = grid->inv_cell_height;
    A.grid_cols = grid->cols; A.grid_rows = grid->rows;
    A.prev_pts = (float*)(base + o_pp); A.margin = (float)margin; A.lowe_ratio = lowe_ratio; A.check_orientation = check_orientation;
    A.matched_2_in_1 = (int32_t*)(base + o_m); A.num_matches = (int32_t*)(base + o_n); A.scratch = (uint32_t*)(base + o_s);
    launch_match_area(st, A);
    PLP_HIP(hipGetLastError());
    PLP_HIP(hipMemcpyAsync(matched_2_in_1, base + o_m, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipMemcpyAsync(prev_matched_pts, base + o_pp, (size_t)n1 * 8, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipMemcpyAsync(num_matches, base + o_n, 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    return PLP_OK;
}

namespace {
PostArgs post_args(const plp_camera* cam) {
    PostArgs A{};
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy;
    // cv_cam_matrix_ / cv_dist_params_ are cv::Mat_<float> (camera/perspective.cc:47-48)
    A.fx_f = (double)(float)cam->fx; A.fy_f = (double)(float)cam->fy; A.cx_f = (double)(float)cam->cx; A.cy_f = (double)(float)cam->cy;
    const double k[5] = {cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
    for (int i = 0; i < 5; ++i) A.k[i] = (double)(float)k[i];
    A.fxb = cam->focal_x_baseline;
    return A;
}
}  // namespace

plp_status plp_post_extract_device(plp_matcher* c, const plp_camera* cam, const plp_keypoint* d_kps, const int32_t* d_counts, int32_t cap,
                                   int32_t B, const float* d_depth, int32_t rows, int32_t cols, size_t depth_step, size_t depth_frame_stride,
                                   plp_keypoint* d_undist, double* d_bearings, float* d_x_right, float* d_depths,
                                   const plp_keyline* d_kl, const int32_t* d_kl_counts, int32_t kl_cap, float* d_kl_depths,
                                   float* d_kl_x_right, void* hip_stream) {
    if (!c || !cam) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (B <= 0 || (d_kps && cap <= 0) || (d_kl && kl_cap <= 0)) return set_error(PLP_ERR_INVALID_ARG, "bad sizes");
    if (!d_kps && !d_kl) return PLP_OK;
    if (d_kps && !d_undist) return set_error(PLP_ERR_INVALID_ARG, "d_undist is required with d_kps");
    if (d_depth && (rows <= 0 || cols <= 0 || depth_step < (size_t)cols * 4)) return set_error(PLP_ERR_INVALID_ARG, "bad depth geometry");
    if (d_kl && (!d_depth || !d_kl_depths || !d_kl_x_right)) return set_error(PLP_ERR_INVALID_ARG, "key lines need depth, d_kl_depths, d_kl_x_right");
    if (!(cam->fx != 0) || !(cam->fy != 0)) return set_error(PLP_ERR_INVALID_ARG, "fx, fy must be non-zero");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    PostArgs A = post_args(cam);
    A.kps = d_kps; A.counts = d_counts; A.cap = d_kps ? cap : 0;
    A.depth = d_depth; A.depth_step = depth_step; A.depth_frame_stride = depth_frame_stride;
    A.undist = d_undist; A.bearings = d_bearings; A.x_right = d_x_right; A.depths = d_depths;
    A.kl = d_kl; A.kl_counts = d_kl_counts; A.kl_cap = d_kl ? kl_cap : 0; A.kl_depths = d_kl_depths; A.kl_x_right = d_kl_x_right;
    launch_post_extract((hipStream_t)hip_stream, A, B);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_post_extract_host(plp_matcher* c, const plp_camera* cam, const plp_keypoint* kps, int32_t n, const float* depth, int32_t rows,
                                 int32_t cols, size_t depth_step, plp_keypoint* undist, double* bearings, float* x_right, float* depths,
                                 const plp_keyline* kl, int32_t n_kl, float* kl_depths, float* kl_x_right) {
    if (!c || !cam) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (n < 0 || n_kl < 0) return set_error(PLP_ERR_INVALID_ARG, "negative count");
    if (n == 0 && n_kl == 0) return PLP_OK;
    if (n > 0 && (!kps || !undist)) return set_error(PLP_ERR_INVALID_ARG, "kps and undist are required");
    if (n_kl > 0 && (!kl || !depth || !kl_depths || !kl_x_right)) return set_error(PLP_ERR_INVALID_ARG, "key lines need depth and both outputs");
    if (depth && (rows <= 0 || cols <= 0 || depth_step < (size_t)cols * 4)) return set_error(PLP_ERR_INVALID_ARG, "bad depth geometry");
    hipStream_t st;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        PLP_HIP(hipSetDevice(c->device));
        st = c->stream;
        auto al = [](size_t v) { return (v + 255) / 256 * 256; };
        const size_t nk = std::max(n, 1), nl = std::max(n_kl, 1);
        const size_t o_k = 0, o_u = o_k + al(nk * sizeof(plp_keypoint)), o_b = o_u + al(nk * sizeof(plp_keypoint)), o_x = o_b + al(nk * 24),
                     o_d = o_x + al(nk * 4), o_l = o_d + al(nk * 4), o_ld = o_l + al(nl * sizeof(plp_keyline)), o_lx = o_ld + al(nl * 8),
                     o_img = o_lx + al(nl * 8), total = o_img + (depth ? al((size_t)rows * cols * 4) : 0);
        PLP_HIP(c->stage.reserve(total));
        uint8_t* base = (uint8_t*)c->stage.p;
        if (n) PLP_HIP(hipMemcpyAsync(base + o_k, kps, (size_t)n * sizeof(plp_keypoint), hipMemcpyHostToDevice, st));
        if (n_kl) {
            PLP_HIP(hipMemcpyAsync(base + o_l, kl, (size_t)n_kl * sizeof(plp_keyline), hipMemcpyHostToDevice, st));
            PLP_HIP(hipMemcpyAsync(base + o_ld, kl_depths, (size_t)n_kl * 8, hipMemcpyHostToDevice, st));     // skipped lines keep the caller's values
            PLP_HIP(hipMemcpyAsync(base + o_lx, kl_x_right, (size_t)n_kl * 8, hipMemcpyHostToDevice, st));
        }
        if (depth) {   // the depth image through a page-locked buffer with slack (plp_common.hpp HostPinned): no 2-D copy reads the caller's pageable memory
            PLP_HIP(c->pin.reserve((size_t)rows * cols * 4));
            c->pin.pack(0, reinterpret_cast<const uint8_t*>(depth), depth_step, rows, cols * 4);
            PLP_HIP(hipMemcpyAsync(base + o_img, c->pin.p, (size_t)rows * cols * 4, hipMemcpyHostToDevice, st));
        }
        PostArgs A = post_args(cam);
        A.kps = n ? (const plp_keypoint*)(base + o_k) : nullptr; A.counts = nullptr; A.cap = n;
        A.depth = depth ? (const float*)(base + o_img) : nullptr; A.depth_step = (size_t)cols * 4; A.depth_frame_stride = 0;
        A.undist = (plp_keypoint*)(base + o_u); A.bearings = bearings ? (double*)(base + o_b) : nullptr;
        A.x_right = (depth && x_right && depths) ? (float*)(base + o_x) : nullptr; A.depths = A.x_right ? (float*)(base + o_d) : nullptr;
        A.kl = n_kl ? (const plp_keyline*)(base + o_l) : nullptr; A.kl_counts = nullptr; A.kl_cap = n_kl;
        A.kl_depths = (float*)(base + o_ld); A.kl_x_right = (float*)(base + o_lx);
        launch_post_extract(st, A, 1);
        PLP_HIP(hipGetLastError());
        if (n) {
            PLP_HIP(hipMemcpyAsync(undist, base + o_u, (size_t)n * sizeof(plp_keypoint), hipMemcpyDeviceToHost, st));
            if (bearings) PLP_HIP(hipMemcpyAsync(bearings, base + o_b, (size_t)n * 24, hipMemcpyDeviceToHost, st));
            if (A.x_right) {
                PLP_HIP(hipMemcpyAsync(x_right, base + o_x, (size_t)n * 4, hipMemcpyDeviceToHost, st));
                PLP_HIP(hipMemcpyAsync(depths, base + o_d, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            }
        }
        if (n_kl) {
            PLP_HIP(hipMemcpyAsync(kl_depths, base + o_ld, (size_t)n_kl * 8, hipMemcpyDeviceToHost, st));
            PLP_HIP(hipMemcpyAsync(kl_x_right, base + o_lx, (size_t)n_kl * 8, hipMemcpyDeviceToHost, st));
        }
        PLP_HIP(hipStreamSynchronize(st));
    }
    return PLP_OK;
}

plp_status plp_convert_to_grayscale_device(plp_matcher* c, const uint8_t* d_src, int32_t rows, int32_t cols, size_t src_step,
                                           size_t src_frame_stride, int32_t channels, int32_t color_order, int32_t B, uint8_t* d_gray,
                                           size_t gray_step, size_t gray_frame_stride, void* hip_stream) {
    if (!c || !d_src || !d_gray) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || B <= 0 || (channels != 3 && channels != 4) || (color_order != 0 && color_order != 1) ||
        src_step < (size_t)cols * channels || gray_step < (size_t)cols) return set_error(PLP_ERR_INVALID_ARG, "bad geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_to_gray((hipStream_t)hip_stream, d_src, rows, cols, src_step, src_frame_stride, channels, color_order, B, d_gray,
                   gray_step, gray_frame_stride);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_convert_to_true_depth_device(plp_matcher* c, const void* d_src, int32_t src_is_u16, int32_t rows, int32_t cols, size_t src_step,
                                            size_t src_frame_stride, double depthmap_factor, int32_t B, float* d_dst, size_t dst_step,
                                            size_t dst_frame_stride, void* hip_stream) {
    if (!c || !d_src || !d_dst) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || B <= 0 || src_step < (size_t)cols * (src_is_u16 ? 2 : 4) || dst_step < (size_t)cols * 4)
        return set_error(PLP_ERR_INVALID_ARG, "bad geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_to_depth((hipStream_t)hip_stream, d_src, src_is_u16 != 0, rows, cols, src_step, src_frame_stride,
                    (float)(1.0 / depthmap_factor), B, d_dst, dst_step, dst_frame_stride);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

static plp_status rectify_map_impl(plp_matcher* c, const double* K, const double* D, int32_t n_dist, const double* R, const plp_camera* rect_cam,
                                   int32_t rows, int32_t cols, float* d_map_x, float* d_map_y, size_t map_step, void* hip_stream, bool fisheye) {
    if (!c || !K || !R || !rect_cam || !d_map_x || !d_map_y || (n_dist > 0 && !D)) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || map_step < (size_t)cols * 4 || (map_step & 3)) return set_error(PLP_ERR_INVALID_ARG, "bad geometry");
    if (fisheye ? n_dist != 4 : (n_dist != 0 && n_dist != 4 && n_dist != 5 && n_dist != 8 && n_dist != 12))
        return set_error(PLP_ERR_INVALID_ARG, fisheye ? "the fisheye model takes 4 distortion coefficients" : "distortion vector must have 0, 4, 5, 8 or 12 entries");
    RectifyArgs A{};
    // iR = (K_rect * R)^-1, K_rect float-rounded; closed-form 3x3 inverse in the order cv::Matx evaluates it
    const double Ar[9] = {(double)(float)rect_cam->fx, 0, (double)(float)rect_cam->cx, 0, (double)(float)rect_cam->fy, (double)(float)rect_cam->cy, 0, 0, 1};
    double m[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Ar[i * 3 + k] * R[k * 3 + j];
            m[i * 3 + j] = s;
        }
    double det = m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[6] * m[5]) + m[2] * (m[3] * m[7] - m[6] * m[4]);
    if (det == 0) return set_error(PLP_ERR_INVALID_ARG, "K_rect * R is singular");
    det = 1 / det;
    A.ir[0] = (m[4] * m[8] - m[5] * m[7]) * det; A.ir[1] = (m[2] * m[7] - m[1] * m[8]) * det; A.ir[2] = (m[1] * m[5] - m[2] * m[4]) * det;
    A.ir[3] = (m[5] * m[6] - m[3] * m[8]) * det; A.ir[4] = (m[0] * m[8] - m[2] * m[6]) * det; A.ir[5] = (m[2] * m[3] - m[0] * m[5]) * det;
    A.ir[6] = (m[3] * m[7] - m[4] * m[6]) * det; A.ir[7] = (m[1] * m[6] - m[0] * m[7]) * det; A.ir[8] = (m[0] * m[4] - m[1] * m[3]) * det;
    for (int i = 0; i < n_dist; ++i) A.d[i] = D[i];
    A.fx = K[0]; A.fy = K[4]; A.u0 = K[2]; A.v0 = K[5];
    A.rows = rows; A.cols = cols; A.map_x = d_map_x; A.map_y = d_map_y; A.map_step = map_step;
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_rectify_map((hipStream_t)hip_stream, A, fisheye);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_rectify_map_device(plp_matcher* c, const double* K, const double* D, int32_t n_dist, const double* R, const plp_camera* rect_cam,
                                  int32_t rows, int32_t cols, float* d_map_x, float* d_map_y, size_t map_step, void* hip_stream) {
    return rectify_map_impl(c, K, D, n_dist, R, rect_cam, rows, cols, d_map_x, d_map_y, map_step, hip_stream, false);
}

plp_status plp_rectify_map_fisheye_device(plp_matcher* c, const double* K, const double* D4, const double* R, const plp_camera* rect_cam, int32_t rows,
                                          int32_t cols, float* d_map_x, float* d_map_y, size_t map_step, void* hip_stream) {
    return rectify_map_impl(c, K, D4, 4, R, rect_cam, rows, cols, d_map_x, d_map_y, map_step, hip_stream, true);
}

plp_status plp_remap_linear_device(plp_matcher* c, const uint8_t* d_src, int32_t rows, int32_t cols, size_t src_step, size_t src_frame_stride,
                                   const float* d_map_x, const float* d_map_y, size_t map_step, int32_t dst_rows, int32_t dst_cols, int32_t B,
                                   uint8_t* d_dst, size_t dst_step, size_t dst_frame_stride, void* hip_stream) {
    if (!c || !d_src || !d_map_x || !d_map_y || !d_dst) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || dst_rows <= 0 || dst_cols <= 0 || B <= 0 || rows > 32767 || cols > 32767 || src_step < (size_t)cols ||
        dst_step < (size_t)dst_cols || map_step < (size_t)dst_cols * 4 || (map_step & 3))
        return set_error(PLP_ERR_INVALID_ARG, "bad geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_remap_linear((hipStream_t)hip_stream, d_src, rows, cols, src_step, src_frame_stride, d_map_x, d_map_y, map_step, dst_rows, dst_cols, B, d_dst,
                        dst_step, dst_frame_stride);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_color_vote_device(plp_matcher* c, const uint8_t* d_mask, int32_t rows, int32_t cols, size_t mask_step,
                                 size_t mask_frame_stride, const plp_keypoint* d_undist, const uint8_t* d_valid, const int32_t* d_counts,
                                 int32_t cap, int32_t B, int32_t check_3x3_window, int32_t* d_labels, void* hip_stream) {
    if (!c || !d_mask || !d_undist || !d_labels) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (rows <= 0 || cols <= 0 || cap <= 0 || B <= 0 || mask_step < (size_t)cols * 3) return set_error(PLP_ERR_INVALID_ARG, "bad geometry");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_color_vote((hipStream_t)hip_stream, d_mask, rows, cols, mask_step, mask_frame_stride, d_undist, d_valid, d_counts, cap, B,
                      check_3x3_window != 0, d_labels);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_landmark_descriptor_device(plp_matcher* c, const uint8_t* d_descs, const int32_t* d_offsets, int32_t L, int32_t* d_best_idx,
                                          void* hip_stream) {
    if (!c || !d_offsets || !d_best_idx || L < 0) return set_error(PLP_ERR_INVALID_ARG, "bad argument");
    if (L == 0) return PLP_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    launch_landmark_descriptor((hipStream_t)hip_stream, d_descs, d_offsets, L, d_best_idx);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_landmark_descriptor_host(plp_matcher* c, const uint8_t* descs, const int32_t* offsets, int32_t L, int32_t* best_idx) {
    if (!c || !offsets || !best_idx || L < 0) return set_error(PLP_ERR_INVALID_ARG, "bad argument");
    if (L == 0) return PLP_OK;
    const int64_t total = offsets[L];
    if (total < 0 || (total > 0 && !descs)) return set_error(PLP_ERR_INVALID_ARG, "bad offsets / descs");
    for (int l = 0; l < L; ++l)
        if (offsets[l + 1] < offsets[l] || offsets[l + 1] - offsets[l] > 1024) return set_error(PLP_ERR_UNSUPPORTED, "offsets must ascend, at most 1024 rows per landmark");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_d = 0, o_o = al((size_t)std::max<int64_t>(total, 1) * 32), o_b = o_o + al((size_t)(L + 1) * 4), tot = o_b + al((size_t)L * 4);
    PLP_HIP(c->stage.reserve(tot));
    uint8_t* base = (uint8_t*)c->stage.p;
    if (total) PLP_HIP(hipMemcpyAsync(base + o_d, descs, (size_t)total * 32, hipMemcpyHostToDevice, st));
    PLP_HIP(hipMemcpyAsync(base + o_o, offsets, (size_t)(L + 1) * 4, hipMemcpyHostToDevice, st));
    launch_landmark_descriptor(st, base + o_d, (const int32_t*)(base + o_o), L, (int32_t*)(base + o_b));
    PLP_HIP(hipGetLastError());
    PLP_HIP(hipMemcpyAsync(best_idx, base + o_b, (size_t)L * 4, hipMemcpyDeviceToHost, st));
    PLP_HIP(hipStreamSynchronize(st));
    return PLP_OK;
}

plp_status plp_match_debug_counters(plp_matcher* c, int64_t* out4) {
    if (!c || !out4) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    int32_t v[4] = {0, 0, 0, 0};
    if (c->dbg.p) PLP_HIP(hipMemcpy(v, c->dbg.p, 16, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out4[i] = v[i];
    return PLP_OK;
}

plp_status plp_hamming_matrix_device(plp_matcher* c, const uint8_t* d_q, int32_t nq, const uint8_t* d_t, int32_t nt, uint16_t* d_dist,
                                     void* hip_stream) {
    if (!c || !d_q || !d_t || !d_dist) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (nq <= 0 || nt <= 0) return PLP_OK;
    PLP_HIP(hipSetDevice(c->device));
    launch_hamming_matrix((hipStream_t)hip_stream, d_q, nq, d_t, nt, d_dist);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_hamming_matrix_host(plp_matcher* c, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, uint16_t* dist) {
    if (!c || !q || !t || !dist) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (nq <= 0 || nt <= 0) return PLP_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    PLP_HIP(hipSetDevice(c->device));
    const size_t bq = ((size_t)nq * 32 + 255) / 256 * 256, bt = ((size_t)nt * 32 + 255) / 256 * 256;
    PLP_HIP(c->stage.reserve(bq + bt + (size_t)nq * nt * 2));
    uint8_t* base = (uint8_t*)c->stage.p;
    PLP_HIP(hipMemcpyAsync(base, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    PLP_HIP(hipMemcpyAsync(base + bq, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    launch_hamming_matrix(c->stream, base, nq, base + bq, nt, (uint16_t*)(base + bq + bt));
    PLP_HIP(hipGetLastError());
    PLP_HIP(hipMemcpyAsync(dist, base + bq + bt, (size_t)nq * nt * 2, hipMemcpyDeviceToHost, c->stream));
    PLP_HIP(hipStreamSynchronize(c->stream));
    return PLP_OK;
}

// Host model of the bin ranking inside the matchers' orientation check (csrc/libstdcxx_sort.hpp), callable without a GPU: the
// indices 0..n-1 (n <= 64) as std::sort orders them by bin size, descending.  depth_limit < 0: the library's own recursion budget.
int32_t plp_model_index_sort_host(const int32_t* sizes, int32_t n, int32_t depth_limit, uint32_t* idx) {
    if (!sizes || !idx || n < 0 || n > 64) return -1;
    int ws[48];
    plp::libstdcxx::index_sort_by_size(sizes, n, idx, ws, depth_limit);
    return n;
}

}  // extern "C"
