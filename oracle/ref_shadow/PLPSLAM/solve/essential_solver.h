// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/solve/essential_solver.h.  The RANSAC solver that
// robust::match_frame_and_keyframe runs AFTER brute_force_match (match/robust.cc:232-252) is host-side geometry outside the
// pinned path: here every brute-force match is reported as an inlier, so the function returns brute_force_match's result.
#ifndef PLPSLAM_SOLVE_ESSENTIAL_SOLVER_H
#define PLPSLAM_SOLVE_ESSENTIAL_SOLVER_H
#include <utility>
#include <vector>
#include "PLPSLAM/type.h"
namespace PLPSLAM {
namespace solve {
class essential_solver {
public:
    essential_solver(const eigen_alloc_vector<Vec3_t>&, const eigen_alloc_vector<Vec3_t>&, const std::vector<std::pair<int, int>>& matches_12)
        : n_(matches_12.size()) {}
    void find_via_ransac(const unsigned int, const bool = true) {}
    bool solution_is_valid() const { return true; }
    std::vector<bool> get_inlier_matches() const { return std::vector<bool>(n_, true); }
private:
    size_t n_;
};
}  // namespace solve
}  // namespace PLPSLAM
#endif
