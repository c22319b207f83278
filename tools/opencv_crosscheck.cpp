// opencv_crosscheck: runs the OpenCV calls the reference makes on this path, on the committed fixture frames, with a REAL
// OpenCV (3.4.16 is what oracle/cv_restated.hpp and oracle/lsd_restated.hpp restate; any 3.4.x / 4.x build shows where they
// differ), and dumps the results as .npy files.  tools/opencv_crosscheck_pack.py packs the directory into
// tests/golden/opencv_crosscheck.npz, and tests/test_oracle_opencv_crosscheck.py then holds the restatements to it -- this is
// the route to pinning the third-party half of the path, which cannot be done in the build image (no OpenCV there).
//
//   g++ -O2 -std=c++17 tools/opencv_crosscheck.cpp -o opencv_crosscheck $(pkg-config --cflags --libs opencv)      # or opencv4
//   ./opencv_crosscheck tests/golden out_dir && python tools/opencv_crosscheck_pack.py out_dir
//
// Call sites mirrored (reference file:line):
//   cv::resize INTER_LINEAR                 feature/orb_extractor.cc:324        (pyramid, sizes of :321-323)
//   cv::FAST(.., 20 / 7, true)              feature/orb_extractor.cc:404,410
//   cv::GaussianBlur 7x7 s2 REFLECT_101     feature/orb_extractor.cc:149;  5x5 s1: binary_descriptor_custom.cpp:355;  11x11 s1.2: lsd.cpp
//   cv::fastAtan2                           feature/orb_extractor.cc:734
//   cv::createLineSegmentDetector->detect   feature/line_descriptor/LSDDetector_custom.cpp:241-257 (options of line_extractor.cc:113-121)
//   cv::resize INTER_LINEAR_EXACT x0.5      inside lsd.cpp (>= 3.4.1)
//   cv::Sobel CV_16S 3x3                    binary_descriptor_custom.cpp:392-393
//   cv::LineIterator::count                 LSDDetector_custom.cpp:292-293
//   cv::remap INTER_LINEAR (identity map)   feature/line_extractor.cc:65-84,103
//   cv::initUndistortRectifyMap + remap     util/stereo_rectifier.cc:61-84
//   cv::undistortPoints                     camera/perspective.cc:148
//   cv::cvtColor RGB2GRAY/BGR2GRAY, convertTo CV_32F   util/image_converter.cc:45-50,77-80
//   cv::norm(NORM_L1) on CV_32F patches     match/stereo.cc:257-270
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>

static std::string g_out;

// NumPy .npy, format 1.0
static void save_npy(const std::string& name, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) hdr += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    hdr += "), }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(g_out + "/" + name + ".npy", std::ios::binary);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write((const char*)magic, 8);
    const uint16_t hl = (uint16_t)hdr.size();
    f.write((const char*)&hl, 2);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    f.write((const char*)data, (std::streamsize)bytes);
}
static void save_mat(const std::string& name, const cv::Mat& m_) {
    cv::Mat m = m_.isContinuous() ? m_ : m_.clone();
    const char* d = m.depth() == CV_8U ? "|u1" : m.depth() == CV_16S ? "<i2" : m.depth() == CV_32F ? "<f4" : m.depth() == CV_64F ? "<f8" : m.depth() == CV_16U ? "<u2" : "<i4";
    std::vector<size_t> shape = {(size_t)m.rows, (size_t)m.cols};
    if (m.channels() > 1) shape.push_back((size_t)m.channels());
    save_npy(name, d, shape, m.data, m.total() * m.elemSize());
}
template <typename T> static void save_vec(const std::string& name, const char* descr, const std::vector<T>& v, size_t cols) {
    save_npy(name, descr, {v.size() / cols, cols}, v.data(), v.size() * sizeof(T));
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <tests/golden> <out_dir>\n", argv[0]); return 2; }
    const std::string in = argv[1];
    g_out = argv[2];
    {
        const std::string v = CV_VERSION;
        std::vector<uint8_t> b(v.begin(), v.end());
        save_npy("opencv_version", "|u1", {b.size()}, b.data(), b.size());
    }
    const char* names[] = {"equirect1_640x480", "equirect1_crop_640x480", "equirect2_640x480", "equirect2_crop_640x480"};
    for (const char* nm : names) {
        const cv::Mat img = cv::imread(in + "/" + nm + ".png", cv::IMREAD_GRAYSCALE);
        if (img.empty()) { std::fprintf(stderr, "cannot read %s\n", nm); return 1; }
        const std::string p = std::string(nm) + "__";
        // ---- pyramid: level l = resize(level l-1, Size(round(W / s_l), round(H / s_l)), INTER_LINEAR), s_l an f32 running product of 1.2f
        std::vector<cv::Mat> pyr = {img};
        float sf = 1.f;
        for (int l = 1; l < 8; ++l) {
            sf *= 1.2f;
            const cv::Size sz(cvRound(img.cols * 1.0 / sf), cvRound(img.rows * 1.0 / sf));
            cv::Mat lvl;
            cv::resize(pyr[l - 1], lvl, sz, 0, 0, cv::INTER_LINEAR);
            pyr.push_back(lvl);
            save_mat(p + "pyr" + std::to_string(l), lvl);
        }
        // ---- FAST on the whole level 0 and level 3, thresholds 20 and 7
        for (int l : {0, 3})
            for (int thr : {20, 7}) {
                std::vector<cv::KeyPoint> kps;
                cv::FAST(pyr[l], kps, thr, true);
                std::vector<int32_t> v;
                for (const auto& k : kps) { v.push_back((int32_t)k.pt.x); v.push_back((int32_t)k.pt.y); v.push_back((int32_t)k.response); }
                if (v.empty()) v.assign(3, -1);
                save_vec(p + "fast_l" + std::to_string(l) + "_t" + std::to_string(thr), "<i4", v, 3);
            }
        // ---- Gaussian blurs
        cv::Mat b7, b5, b11;
        cv::GaussianBlur(img, b7, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        cv::GaussianBlur(img, b5, cv::Size(5, 5), 1);
        cv::GaussianBlur(img, b11, cv::Size(11, 11), 0.6 / 0.5);
        save_mat(p + "blur7", b7); save_mat(p + "blur5", b5); save_mat(p + "blur11", b11);
        cv::Mat b7l3;
        cv::GaussianBlur(pyr[3], b7l3, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        save_mat(p + "blur7_l3", b7l3);
        // ---- LSD's internal down-scaling
#if CV_VERSION_MAJOR > 3 || (CV_VERSION_MAJOR == 3 && (CV_VERSION_MINOR > 4 || (CV_VERSION_MINOR == 4 && CV_VERSION_REVISION >= 1)))
        cv::Mat half;
        cv::resize(b11, half, cv::Size(), 0.5, 0.5, cv::INTER_LINEAR_EXACT);
        save_mat(p + "lsd_scaled", half);
#endif
        // ---- LSD (needs a build that still has it: 3.4.x < 3.4.6, >= 3.4.16 / 4.5.4, or a restored lsd.cpp as the reference's README says)
        try {
            cv::Ptr<cv::LineSegmentDetector> ls = cv::createLineSegmentDetector(1, 0.5, 0.6, 2.0, 22.5, 1.0, 0.6, 1024);
            std::vector<cv::Vec4f> lines;
            ls->detect(img, lines);
            std::vector<float> v;
            for (const auto& s : lines) for (int k = 0; k < 4; ++k) v.push_back(s[k]);
            if (v.empty()) v.assign(4, -1.f);
            save_vec(p + "lsd_lines", "<f4", v, 4);
            std::vector<int32_t> cnt;
            for (const auto& s : lines) cnt.push_back(cv::LineIterator(img, cv::Point2f(s[0], s[1]), cv::Point2f(s[2], s[3])).count);
            if (cnt.empty()) cnt.push_back(-1);
            save_vec(p + "lsd_line_pixels", "<i4", cnt, 1);
        } catch (const cv::Exception& e) { std::fprintf(stderr, "LSD not available in this build: %s\n", e.what()); }
        // ---- Sobel on the 5x5-blurred image
        cv::Mat dx, dy;
        cv::Sobel(b5, dx, CV_16SC1, 1, 0, 3);
        cv::Sobel(b5, dy, CV_16SC1, 0, 1, 3);
        save_mat(p + "sobel_dx", dx); save_mat(p + "sobel_dy", dy);
        // ---- remap: identity map of line_extractor.cc (K K^-1 (u, v, 1) in f64 -> f32) and a rectification map (EuRoC-like calibration)
        {
            const double fx = 520.9, fy = 521.0, cx = 325.1, cy = 249.7;
            cv::Mat mx(img.rows, img.cols, CV_32F), my(img.rows, img.cols, CV_32F);
            for (int v = 0; v < img.rows; ++v)
                for (int u = 0; u < img.cols; ++u) {
                    const double x = (u - cx) / fx, y = (v - cy) / fy;
                    mx.at<float>(v, u) = (float)(fx * x + cx);
                    my.at<float>(v, u) = (float)(fy * y + cy);
                }
            cv::Mat m1, m2, out;
            cv::convertMaps(mx, my, m1, m2, CV_32FC1, false);
            cv::remap(img, out, m1, m2, cv::INTER_LINEAR);
            save_mat(p + "remap_identity", out);
            const cv::Mat K = (cv::Mat_<double>(3, 3) << 458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1);
            const cv::Mat D = (cv::Mat_<double>(1, 5) << -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0);
            const cv::Mat R = (cv::Mat_<double>(3, 3) << 0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847,
                               0.007055629199258132, -0.008089410156878961, -0.007044357138835809, 0.9999424675829176);
            const cv::Mat Kr = (cv::Mat_<float>(3, 3) << 435.2046959714599f, 0, 367.4517211914062f, 0, 435.2046959714599f, 252.2008514404297f, 0, 0, 1);
            cv::Mat rx, ry, rect;
            cv::initUndistortRectifyMap(K, D, R, Kr, img.size(), CV_32F, rx, ry);
            cv::remap(img, rect, rx, ry, cv::INTER_LINEAR);
            save_mat(p + "rectify_map_x", rx); save_mat(p + "rectify_map_y", ry); save_mat(p + "rectified", rect);
        }
    }
    // ---- fastAtan2 on a grid
    {
        std::vector<float> v;
        for (int y = -300; y <= 300; y += 7)
            for (int x = -300; x <= 300; x += 11) { v.push_back((float)y); v.push_back((float)x); v.push_back(cv::fastAtan2((float)y, (float)x)); }
        for (float y : {1e-3f, -2.5f, 1234567.f, 0.f})
            for (float x : {0.f, 3e-7f, -17.25f, 2.9e6f}) { v.push_back(y); v.push_back(x); v.push_back(cv::fastAtan2(y, x)); }
        save_vec("fast_atan2", "<f4", v, 3);
    }
    // ---- undistortPoints (TUM fr1 distortion), as camera/perspective.cc:130-162 calls it
    {
        const cv::Mat K = (cv::Mat_<float>(3, 3) << 517.306408f, 0, 318.643040f, 0, 516.469215f, 255.313989f, 0, 0, 1);
        const cv::Mat D = (cv::Mat_<float>(5, 1) << 0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f);
        std::vector<float> pts;
        for (int y = 0; y < 480; y += 37) for (int x = 0; x < 640; x += 41) { pts.push_back((float)x + 0.25f); pts.push_back((float)y + 0.5f); }
        cv::Mat m((int)pts.size() / 2, 2, CV_32F, pts.data());
        cv::Mat mc = m.reshape(2).clone();
        cv::undistortPoints(mc, mc, K, D, cv::Mat(), K, cv::TermCriteria(cv::TermCriteria::EPS | cv::TermCriteria::MAX_ITER, 20, 1e-6));
        cv::Mat out = mc.reshape(1);
        save_vec("undistort_in", "<f4", pts, 2);
        save_mat("undistort_out", out);
    }
    // ---- colour / depth conversion
    {
        cv::Mat col(48, 64, CV_8UC3);
        for (int y = 0; y < col.rows; ++y) for (int x = 0; x < col.cols; ++x) col.at<cv::Vec3b>(y, x) = cv::Vec3b((uchar)(x * 4 + y), (uchar)(255 - x * 3), (uchar)(y * 5 + x * 2));
        cv::Mat g1, g2;
        cv::cvtColor(col, g1, cv::COLOR_RGB2GRAY); cv::cvtColor(col, g2, cv::COLOR_BGR2GRAY);
        save_mat("color_src", col); save_mat("gray_rgb", g1); save_mat("gray_bgr", g2);
        cv::Mat d16(48, 64, CV_16UC1), d32;
        for (int y = 0; y < d16.rows; ++y) for (int x = 0; x < d16.cols; ++x) d16.at<uint16_t>(y, x) = (uint16_t)(x * 997 + y * 131);
        d16.convertTo(d32, CV_32F, 1.0 / 5208.0);
        save_mat("depth_u16", d16); save_mat("depth_f32", d32);
    }
    std::printf("wrote %s (OpenCV %s)\n", g_out.c_str(), CV_VERSION);
    return 0;
}
