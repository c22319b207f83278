// TEST INFRASTRUCTURE ONLY.  Stand-in for the reference's line_descriptor_custom.hpp: cv::line_descriptor::KeyLine with the
// field order of descriptor_custom.hpp:139-174 (68 bytes), which is all the line-extractor facade needs from it.
#pragma once
#include <opencv2/core.hpp>
namespace cv {
namespace line_descriptor {
struct KeyLine {
    float angle = 0;
    int class_id = -1;
    int octave = 0;
    Point2f pt;
    float response = 0, size = 0;
    float startPointX = 0, startPointY = 0, endPointX = 0, endPointY = 0;
    float sPointInOctaveX = 0, sPointInOctaveY = 0, ePointInOctaveX = 0, ePointInOctaveY = 0;
    float lineLength = 0;
    int numOfPixels = 0;
};
}  // namespace line_descriptor
}  // namespace cv
