// Drop-in replacement of the reference TRANSLATION UNIT src/PLPSLAM/feature/line_descriptor/binary_descriptor_matcher.cpp:
// compiled against the reference's own, unchanged header (feature/line_descriptor/descriptor_custom.hpp:1000-1316), it defines
// the members of cv::line_descriptor::BinaryDescriptorMatcher that the reference calls
//     createBinaryDescriptorMatcher()                                   data/frame.cc:392-393,497-498, mapping_module.cc:492-493,
//     match(queryDescriptors, trainDescriptors, matches, mask) const    module/initializer.cc:594-595 (and the ->match right after)
// over the C ABI (plp_lbd_match_1nn_host: exact 1-NN in Hamming space with the multi-index-hashing discovery order of
// binary_descriptor_matcher.cpp:597-818, on the device).  With this file and the header-only facades of feature/ and match/ the
// three line_descriptor sources LSDDetector_custom.cpp, binary_descriptor_custom.cpp and binary_descriptor_matcher.cpp leave the
// build; draw_custom.cpp stays.  The data-set interface (add / train / match against the stored set,
// knnMatch, radiusMatch) has no caller in the reference and throws here instead of silently doing something else.
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "PLPSLAM/feature/line_descriptor/line_descriptor_custom.hpp"
#include "plp_front.h"

namespace cv {
namespace line_descriptor {

namespace {
plp_matcher* lbd_matcher() {            // one context per process, created at first use
    static plp_matcher* ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = std::getenv("PLP_DEVICE");
        if (plp_matcher_create(e ? std::atoi(e) : 0, &ctx) != PLP_OK)
            throw std::runtime_error(std::string("plp_front: ") + plp_last_error());
    });
    return ctx;
}
[[noreturn]] void no_caller(const char* what) {
    throw std::logic_error(std::string("cv::line_descriptor::BinaryDescriptorMatcher::") + what + " has no caller in Structure-PLP-SLAM and is not provided by plp_front");
}
}  // namespace

BinaryDescriptorMatcher::BinaryDescriptorMatcher() {
    dataset = 0;
    nextAddedIndex = 0;
    numImages = 0;
    descrInDS = 0;
}

Ptr<BinaryDescriptorMatcher> BinaryDescriptorMatcher::createBinaryDescriptorMatcher() { return Ptr<BinaryDescriptorMatcher>(new BinaryDescriptorMatcher()); }

void BinaryDescriptorMatcher::clear() {
    descriptorsMat.release();
    indexesMap.clear();
    nextAddedIndex = 0;
    numImages = 0;
    descrInDS = 0;
}

void BinaryDescriptorMatcher::match(const Mat& queryDescriptors, const Mat& trainDescriptors, std::vector<DMatch>& matches, const Mat& mask) const {
    if (queryDescriptors.rows == 0 || trainDescriptors.rows == 0) {            // :200-205
        std::cout << "Error: descriptors matrices cannot be void" << std::endl;
        return;
    }
    if (!mask.empty() && (mask.rows != queryDescriptors.rows && mask.cols != 1)) {   // :207-213
        std::cout << "Error: input mask should have " << queryDescriptors.rows << " rows and 1 column. "
                  << "Program will be terminated" << std::endl;
        return;
    }
    if (queryDescriptors.cols != 32 || trainDescriptors.cols != 32 || queryDescriptors.type() != CV_8UC1 || trainDescriptors.type() != CV_8UC1)
        throw std::runtime_error("BinaryDescriptorMatcher::match: 32-byte CV_8U descriptors expected");
    const Mat q = queryDescriptors.isContinuous() ? queryDescriptors : queryDescriptors.clone();
    const Mat t = trainDescriptors.isContinuous() ? trainDescriptors : trainDescriptors.clone();
    std::vector<int32_t> idx(static_cast<size_t>(q.rows), -1), dist(static_cast<size_t>(q.rows), 256);
    if (plp_lbd_match_1nn_host(lbd_matcher(), q.data, q.rows, t.data, t.rows, idx.data(), dist.data()) != PLP_OK)
        throw std::runtime_error(std::string("plp_front: ") + plp_last_error());
    for (int counter = 0; counter < q.rows; counter++) {                        // :229-250
        if (mask.empty() || mask.at<uchar>(counter) != 0) {
            DMatch dm;
            dm.queryIdx = counter;
            // nothing within the search reach: the reference reads uninitialised memory here (:236-243); this returns (-1, 256),
            // which every caller's `distance < 30 / 50` test rejects
            dm.trainIdx = idx[static_cast<size_t>(counter)];
            dm.imgIdx = 0;
            dm.distance = static_cast<float>(dist[static_cast<size_t>(counter)]);
            matches.push_back(dm);
        }
    }
}

void BinaryDescriptorMatcher::match(const Mat&, std::vector<DMatch>&, const std::vector<Mat>&) { no_caller("match(query, matches, masks)"); }
void BinaryDescriptorMatcher::knnMatch(const Mat&, const Mat&, std::vector<std::vector<DMatch>>&, int, const Mat&, bool) const { no_caller("knnMatch"); }
void BinaryDescriptorMatcher::knnMatch(const Mat&, std::vector<std::vector<DMatch>>&, int, const std::vector<Mat>&, bool) { no_caller("knnMatch"); }
void BinaryDescriptorMatcher::radiusMatch(const Mat&, const Mat&, std::vector<std::vector<DMatch>>&, float, const Mat&, bool) const { no_caller("radiusMatch"); }
void BinaryDescriptorMatcher::radiusMatch(const Mat&, std::vector<std::vector<DMatch>>&, float, const std::vector<Mat>&, bool) { no_caller("radiusMatch"); }
void BinaryDescriptorMatcher::add(const std::vector<Mat>&) { no_caller("add"); }
void BinaryDescriptorMatcher::train() { no_caller("train"); }

}  // namespace line_descriptor
}  // namespace cv
