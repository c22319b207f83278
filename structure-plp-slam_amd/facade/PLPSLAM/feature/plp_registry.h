// Shared by the facade headers: which device context stands behind a feature::orb_extractor's public image_pyramid_.
// The reference hands `extractor->image_pyramid_` (a std::vector<cv::Mat>&) to match::stereo (data/frame.cc:277-281); the
// facade stereo matcher reads the pyramids where they already are (HBM), so it needs the extractor's context, and the one
// thing the reference passes along is that vector's address.
#ifndef PLPSLAM_FEATURE_PLP_REGISTRY_H
#define PLPSLAM_FEATURE_PLP_REGISTRY_H

#include <map>
#include <mutex>

#include "plp_front.h"

namespace PLPSLAM {
namespace feature {
namespace plp_registry {

struct table {
    std::mutex mu;
    std::map<const void*, plp_orb*> by_pyramid;
};
inline table& instance() {
    static table t;
    return t;
}
inline void add(const void* pyramid_vector, plp_orb* ctx) {
    table& t = instance();
    std::lock_guard<std::mutex> lk(t.mu);
    t.by_pyramid[pyramid_vector] = ctx;
}
inline void remove(const void* pyramid_vector) {
    table& t = instance();
    std::lock_guard<std::mutex> lk(t.mu);
    t.by_pyramid.erase(pyramid_vector);
}
inline plp_orb* find(const void* pyramid_vector) {
    table& t = instance();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.by_pyramid.find(pyramid_vector);
    return it == t.by_pyramid.end() ? nullptr : it->second;
}

}  // namespace plp_registry
}  // namespace feature
}  // namespace PLPSLAM

#endif  // PLPSLAM_FEATURE_PLP_REGISTRY_H
