// Minimal stand-in for cv::Mat as ro-map_amd/compat/ uses it (data, channels, continuity, clone, typed row pointer) -- TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstring>
#include <memory>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24
#define CV_32FC1 5
namespace cv {
class Mat {
public:
    Mat() = default;
    Mat(int rows_, int cols_, int type_) : rows(rows_), cols(cols_), type(type_) {
        store = std::make_shared<std::vector<unsigned char>>((size_t)rows * cols * elemSize()); data = store->data();
    }
    int channels() const { return (type >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * ((type & 7) == 5 ? 4 : 1); }
    bool isContinuous() const { return true; }
    bool empty() const { return data == nullptr; }
    Mat clone() const { Mat m(rows, cols, type); if (data) std::memcpy(m.data, data, (size_t)rows * cols * elemSize()); return m; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * cols * elemSize()); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * cols * elemSize()); }
    unsigned char* data = nullptr; int rows = 0, cols = 0, type = 0;
private:
    std::shared_ptr<std::vector<unsigned char>> store;
};
}  // namespace cv
