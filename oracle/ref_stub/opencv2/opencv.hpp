// TEST INFRASTRUCTURE ONLY (oracle build).  Minimal stand-in for <opencv2/opencv.hpp>.
//
// The reference's cpp/volumetric headers use OpenCV for exactly one thing on this path: the
// cv::Mat *type* in the signatures of carve()/image_utils.h (element access + depth constants).
// No OpenCV algorithm is called.  OpenCV is not installed in this image, so the oracle build
// (oracle/Makefile) puts this directory on the include path instead.  Nothing in the product
// (pyslam_amd/) includes this file.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH(flags) ((flags) & 7)

namespace cv {

class Mat {
  public:
    int rows = 0;
    int cols = 0;

    Mat() = default;
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
        owned_ = std::make_shared<std::vector<uint8_t>>(size_t(r) * size_t(c) * elem_size(), 0);
        data_ = owned_->data();
    }
    // Wraps caller memory (row-major, densely packed); caller keeps it alive.
    Mat(int r, int c, int type, void *external) : rows(r), cols(c), type_(type) {
        data_ = static_cast<uint8_t *>(external);
    }

    bool empty() const { return rows == 0 || cols == 0 || data_ == nullptr; }
    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return 1; }

    size_t elem_size() const {
        static const size_t sz[7] = {1, 1, 2, 2, 4, 4, 8};
        return sz[CV_MAT_DEPTH(type_)];
    }

    template <typename T> T *ptr(int r = 0) {
        return reinterpret_cast<T *>(data_ + size_t(r) * size_t(cols) * elem_size());
    }
    template <typename T> const T *ptr(int r = 0) const {
        return reinterpret_cast<const T *>(data_ + size_t(r) * size_t(cols) * elem_size());
    }
    template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T &at(int r, int c) const { return ptr<T>(r)[c]; }

    // Element-wise numeric conversion between the scalar depths above.
    void convertTo(Mat &dst, int rtype) const {
        dst = Mat(rows, cols, rtype);
        const size_t n = size_t(rows) * size_t(cols);
        for (size_t i = 0; i < n; ++i) {
            dst.store(i, load(i));
        }
    }
    template <typename S> void setTo(S value) {
        const size_t n = size_t(rows) * size_t(cols);
        for (size_t i = 0; i < n; ++i) {
            store(i, static_cast<double>(value));
        }
    }

  private:
    double load(size_t i) const {
        switch (CV_MAT_DEPTH(type_)) {
        case CV_8U: return reinterpret_cast<const uint8_t *>(data_)[i];
        case CV_8S: return reinterpret_cast<const int8_t *>(data_)[i];
        case CV_16U: return reinterpret_cast<const uint16_t *>(data_)[i];
        case CV_16S: return reinterpret_cast<const int16_t *>(data_)[i];
        case CV_32S: return reinterpret_cast<const int32_t *>(data_)[i];
        case CV_32F: return reinterpret_cast<const float *>(data_)[i];
        default: return reinterpret_cast<const double *>(data_)[i];
        }
    }
    void store(size_t i, double v) {
        switch (CV_MAT_DEPTH(type_)) {
        case CV_8U: reinterpret_cast<uint8_t *>(data_)[i] = static_cast<uint8_t>(v); break;
        case CV_8S: reinterpret_cast<int8_t *>(data_)[i] = static_cast<int8_t>(v); break;
        case CV_16U: reinterpret_cast<uint16_t *>(data_)[i] = static_cast<uint16_t>(v); break;
        case CV_16S: reinterpret_cast<int16_t *>(data_)[i] = static_cast<int16_t>(v); break;
        case CV_32S: reinterpret_cast<int32_t *>(data_)[i] = static_cast<int32_t>(v); break;
        case CV_32F: reinterpret_cast<float *>(data_)[i] = static_cast<float>(v); break;
        default: reinterpret_cast<double *>(data_)[i] = v; break;
        }
    }

    int type_ = CV_32F;
    uint8_t *data_ = nullptr;
    std::shared_ptr<std::vector<uint8_t>> owned_;
};

} // namespace cv
