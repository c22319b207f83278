// Diagnostic (CPU only, test infrastructure: it includes the oracle's LSD restatement): region growing of replay frames in the ROUND STRUCTURE of k_lsd_grow
// (7 region points x 3 x 3 neighbours per round) -- rounds, queue lengths, acceptances per round, region sizes, how often the next batch is known a round early
// (VERDICT r04 item 1b), how often the set accepted under the start angle equals the sequential result (bulk acceptance), lone seeds.  profiles/r05_lsd_grow.md quotes it.
//   python -c "import importlib,sys; sys.path.insert(0,'.'); importlib.import_module('structure-plp-slam_amd.synth').replay(1234, 8).tofile('/tmp/frames.bin')"
//   g++ -O2 -std=c++17 -Ioracle -o /tmp/grow_round_stats tools/experiments/grow_round_stats.cpp && /tmp/grow_round_stats /tmp/frames.bin
#define private public
#include "lsd_restated.hpp"
#undef private
#include <cstdio>
#include <map>
using namespace oracle;
struct Stats {
    long rounds = 0, acc = 0, regions = 0, cand = 0;
    long rounds_by_q[64] = {0};      // queue length (nreg - i) at round start, capped 63
    long acc_hist[64] = {0};
    long rounds_prefetchable = 0;    // rounds whose whole batch was known one round earlier
    long rounds_partial = 0;
    std::map<int,long> reg_rounds;   // region size class -> rounds
    std::map<int,long> reg_count, reg_pix;
    long first_rounds = 0, first_round_noacc = 0;
    long dup_cand = 0;
    long rounds_small_S = 0;
    long bulk_ok[16] = {0}, bulk_bad[16] = {0};
    long seeds_static_lone = 0, seeds_lone_dynamic = 0, seeds_total = 0, seeds_static_nonlone_but_single = 0;
};
static Stats st;
static int g_slots = 7;                          // region points per round (7 in k_lsd_grow; 3 when two frames share a wave: 27 of a half-wave's 32 lanes)
static std::vector<std::vector<int>> g_round_acc;   // per frame: acceptances of every round, in order (for the two-frames-per-wave projection)
// batched simulation of region_grow: identical result to the sequential one (the kernel's round structure)
void grow_sim(Lsd& L, int sx, int sy, std::vector<Lsd::RegionPoint>& reg, double& reg_angle, double prec) {
    const int w = L.w_, h = L.h_;
    reg.clear();
    reg_angle = L.angles_[(size_t)sy * w + sx];
    reg.push_back({sx, sy, reg_angle, L.modgrad_[(size_t)sy * w + sx]});
    float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
    L.used_[(size_t)sy * w + sx] = 1;
    long my_rounds = 0;
    size_t prev_known = 1;   // nreg before the previous round
    for (size_t i = 0; i < reg.size();) {
        const size_t n0 = reg.size();
        const size_t nb = std::min<size_t>((size_t)g_slots, n0 - i);
        ++st.rounds; ++my_rounds;
        st.rounds_by_q[std::min<size_t>(63, n0 - i)]++;
        if (i > 0) { if (i + nb <= prev_known) ++st.rounds_prefetchable; else if (i < prev_known) ++st.rounds_partial; }
        // candidates at fetch time
        std::vector<int> seen;
        for (size_t s = 0; s < nb; ++s) {
            const int px = reg[i + s].x, py = reg[i + s].y;
            for (int yy = py - 1; yy <= py + 1; ++yy) for (int xx = px - 1; xx <= px + 1; ++xx) {
                if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
                if (L.used_[(size_t)yy * w + xx] == 1 || L.angles_[(size_t)yy*w+xx] == Lsd::NOTDEF) continue;
                ++st.cand;
                const int p = yy * w + xx;
                if (std::find(seen.begin(), seen.end(), p) != seen.end()) ++st.dup_cand; else seen.push_back(p);
            }
        }
        // speculation: the candidates aligned with the angle at the START of the round (one per pixel)
        std::vector<int> spec;
        for (int p : seen) if (L.is_aligned(p % w, p / w, reg_angle, prec)) spec.push_back(p);
        std::sort(spec.begin(), spec.end());
        const size_t n_start = reg.size();
        int a = 0;
        for (size_t s = 0; s < nb; ++s) {
            const int px = reg[i + s].x, py = reg[i + s].y;
            const int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1);
            const int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    uint8_t& is_used = L.used_[(size_t)yy * w + xx];
                    if (is_used != 1 && L.is_aligned(xx, yy, reg_angle, prec)) {
                        const double angle = L.angles_[(size_t)yy * w + xx];
                        is_used = 1;
                        reg.push_back({xx, yy, angle, L.modgrad_[(size_t)yy * w + xx]});
                        sumdx += f_cos(float(angle)); sumdy += f_sin(float(angle));
                        reg_angle = fast_atan2f_deg(sumdy, sumdx) * Lsd::DEG_TO_RADS;
                        ++a;
                    }
                }
        }
        {
            std::vector<int> got;
            for (size_t k = n_start; k < reg.size(); ++k) got.push_back(reg[k].y * w + reg[k].x);
            std::sort(got.begin(), got.end());
            const int m = std::min<int>(15, (int)std::max(got.size(), spec.size()));
            if (got == spec) st.bulk_ok[m]++; else st.bulk_bad[m]++;
        }
        st.acc += a; st.acc_hist[std::min(63, a)]++;
        if (!g_round_acc.empty()) g_round_acc.back().push_back(a);
        if (i == 0) { ++st.first_rounds; if (a == 0) ++st.first_round_noacc; }
        prev_known = n0;
        i += nb;
    }
    ++st.regions;
    int cls = reg.size() <= 1 ? 1 : reg.size() <= 2 ? 2 : reg.size() <= 4 ? 4 : reg.size() <= 8 ? 8 : reg.size() <= 16 ? 16 : reg.size() <= 32 ? 32 : reg.size() <= 64 ? 64 : reg.size() <= 128 ? 128 : reg.size() <= 256 ? 256 : 100000;
    st.reg_rounds[cls] += my_rounds; st.reg_count[cls]++; st.reg_pix[cls] += reg.size();
}
int main(int argc, char** argv) {
    const int H = 480, W = 640, NF = 8;
    if (argc > 2) g_slots = atoi(argv[2]);
    std::vector<uint8_t> buf((size_t)H * W * NF);
    FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/frames.bin", "rb"); if (!f) { printf("usage: grow_round_stats frames.bin (8 frames of 480 x 640 bytes)\n"); return 2; } if (fread(buf.data(), 1, buf.size(), f) != buf.size()) return 1; fclose(f);
    long total_lines = 0;
    for (int fr = 0; fr < NF; ++fr) {
        g_round_acc.emplace_back();
        Image img(H, W); std::copy(buf.begin() + (size_t)fr * H * W, buf.begin() + (size_t)(fr + 1) * H * W, img.data.begin());
        LsdOptions o; Lsd L(o, false);
        const double prec = M_PI * o.ang_th / 180, p = o.ang_th / 180, rho = o.quant / std::sin(prec);
        const double sigma = o.sigma_scale / o.scale;
        const unsigned hh = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3 * std::log(10.0))));
        Image g = gaussian_blur_u8(img, 1 + 2 * (int)hh, sigma);
        L.scaled = resize_linear_exact_u8(g, o.scale, o.scale);
        L.ll_angle(rho, (unsigned)o.n_bins);
        const double LOG_NT = 5 * (std::log10(double(L.w_)) + std::log10(double(L.h_))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        L.used_.assign((size_t)L.w_ * L.h_, 0);
        std::vector<Lsd::RegionPoint> reg;
        for (const auto& op : L.ordered_) {
            if (L.used_[(size_t)op.y * L.w_ + op.x] == 0 && L.angles_[(size_t)op.y * L.w_ + op.x] != Lsd::NOTDEF) {
                double reg_angle;
                {   // static alignment of the 8 neighbours with the seed's own angle (defined pixels only)
                    int n_al = 0, n_al_unused = 0;
                    const double a0 = L.angles_[(size_t)op.y * L.w_ + op.x];
                    for (int yy = op.y - 1; yy <= op.y + 1; ++yy) for (int xx = op.x - 1; xx <= op.x + 1; ++xx) {
                        if ((xx == op.x && yy == op.y) || xx < 0 || yy < 0 || xx >= L.w_ || yy >= L.h_) continue;
                        if (L.is_aligned(xx, yy, a0, prec)) { ++n_al; if (L.used_[(size_t)yy * L.w_ + xx] != 1) ++n_al_unused; }
                    }
                    ++st.seeds_total;
                    if (n_al == 0) ++st.seeds_static_lone; else if (n_al_unused == 0) ++st.seeds_lone_dynamic;
                }
                grow_sim(L, op.x, op.y, reg, reg_angle, prec);
                if (reg.size() < min_reg_size) continue;
                Lsd::Rect rec;
                L.region2rect(reg, reg_angle, prec, p, rec);
                if (o.refine > 0 && !L.refine(reg, reg_angle, prec, p, rec, o.density_th)) continue;
                ++total_lines;
            }
        }
        if (fr == 0) printf("min_reg_size %zu\n", min_reg_size);
    }
    printf("per frame: regions %.0f rounds %.0f accepted %.0f cand %.0f dupcand %.0f lines %.1f\n", st.regions / 8.0, st.rounds / 8.0, st.acc / 8.0, st.cand / 8.0, st.dup_cand / 8.0, total_lines / 8.0);
    printf("first rounds %.0f of which no acceptance %.0f; prefetchable rounds (full batch known a round earlier) %.0f partial %.0f\n", st.first_rounds / 8.0, st.first_round_noacc / 8.0, st.rounds_prefetchable / 8.0, st.rounds_partial / 8.0);
    printf("queue length at round start: "); for (int q = 1; q < 64; ++q) if (st.rounds_by_q[q]) printf("%d:%.0f ", q, st.rounds_by_q[q] / 8.0); printf("\n");
    printf("acceptances per round: "); for (int q = 0; q < 64; ++q) if (st.acc_hist[q]) printf("%d:%.0f ", q, st.acc_hist[q] / 8.0); printf("\n");
    printf("speculation (set accepted under the start angle == sequential set), by max(set sizes): ");
    for (int m = 0; m < 16; ++m) if (st.bulk_ok[m] + st.bulk_bad[m]) printf("%d: %.0f ok %.0f bad | ", m, st.bulk_ok[m] / 8.0, st.bulk_bad[m] / 8.0);
    printf("\n");
    printf("seeds grown %.0f: no neighbour aligned with the seed's own angle (static) %.0f, aligned neighbours all USED at its turn %.0f\n", st.seeds_total / 8.0, st.seeds_static_lone / 8.0, st.seeds_lone_dynamic / 8.0);
    {   // Projection of "two frames per wave" (VERDICT r05 item 4a).  Cost model from the kernel's own clocks (profiles/r05_lsd_grow.md, cycles per round: ring read 340,
        // USED test 320, record gather 390, appends 230 = 1280 fixed; acceptance loop 730 at 2.35 acceptances = 100 + 268 per acceptance).  A PAIR of frames in one
        // instruction stream, lanes 0..31 / 32..63, three points x nine lanes each, rounds in lockstep: the fixed phases once per round of the pair (+10 % for the half-wave
        // bookkeeping), the acceptance loop as two interleaved chains: 1.3 x 268 per iteration, max(a_A, a_B) iterations.
        double one = 0, pair = 0; long r_one = 0, r_pair = 0;
        for (auto& f : g_round_acc) { for (int a : f) one += 1280 + 100 + 268.0 * a; r_one += (long)f.size(); }
        for (size_t k = 0; k + 1 < g_round_acc.size(); k += 2) {
            const auto &A = g_round_acc[k], &B = g_round_acc[k + 1];
            const size_t n = std::max(A.size(), B.size());
            for (size_t i = 0; i < n; ++i) { const int a = i < A.size() ? A[i] : 0, b = i < B.size() ? B[i] : 0; pair += 1280 * 1.10 + 100 + 268.0 * 1.3 * std::max(a, b); }
            r_pair += (long)n;
        }
        printf("slots %d: rounds per frame %.0f; modelled round cycles per frame, one frame per wave: %.2f M; as pairs in lockstep: %ld rounds per pair, %.2f M cycles per PAIR = %.2f M per frame\n",
               g_slots, r_one / 8.0, one / 8.0 / 1e6, r_pair / 4, pair / 4.0 / 1e6, pair / 8.0 / 1e6);
    }
    for (auto& kv : st.reg_count) printf("regions <= %d px: %.0f regions, %.0f px, %.0f rounds\n", kv.first, kv.second / 8.0, st.reg_pix[kv.first] / 8.0, st.reg_rounds[kv.first] / 8.0);
    return 0;
}
