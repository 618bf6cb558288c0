// Grid topology and small helpers of the host side.
// Mirrors diffusion_stereo/src/utils.rs (neighbour slots: 0 = left, 1 = top, 2 = right,
// 3 = bottom).  Where the reference panics, these throw std::invalid_argument.
#pragma once
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace stereo {

using Matrix = std::vector<std::vector<std::size_t>>;

// utils.rs:42-67
inline bool neighbor_exists(std::size_t pixel_i, std::size_t pixel_j, std::size_t neighbor,
                            std::size_t height, std::size_t width)
{
    switch (neighbor) {
    case 0: return pixel_j > 0;
    case 1: return pixel_i > 0;
    case 2: return pixel_j + 1 < width;
    case 3: return pixel_i + 1 < height;
    default: throw std::invalid_argument("Non-existent neighbor index");
    }
}

// utils.rs:75-83: (row, column, slot under which the neighbour sees this pixel)
inline std::tuple<std::size_t, std::size_t, std::size_t> neighbor_index(std::size_t i, std::size_t j,
                                                                       std::size_t neighbor)
{
    switch (neighbor) {
    case 0: return {i, j - 1, 2};
    case 1: return {i - 1, j, 3};
    case 2: return {i, j + 1, 0};
    case 3: return {i + 1, j, 1};
    default: throw std::invalid_argument("Non-existent neighbor index");
    }
}

// utils.rs:93-101
inline std::size_t number_of_neighbors(std::size_t i, std::size_t j, std::size_t height, std::size_t width)
{
    std::size_t count = 0;
    for (std::size_t n = 0; n < 4; ++n)
        if (neighbor_exists(i, j, n, height, width)) ++count;
    return count;
}

// utils.rs:109-114
inline bool approx_equal(double x, double y, double epsilon) { return !(std::fabs(x - y) > epsilon); }

// utils.rs:137-163: keeps the first element of every run of values within `tolerance` of
// that first element -- and always the last element of the (sorted) array.
inline std::vector<double> dedup_f64(const std::vector<double> &array, double tolerance)
{
    std::vector<double> unique_values;
    if (array.empty()) return unique_values;
    std::size_t i = 0, last_kept = 0;
    while (i < array.size()) {
        unique_values.push_back(array[i]);
        last_kept = i;
        std::size_t j = i + 1;
        while (j < array.size() && approx_equal(array[i], array[j], tolerance)) ++j;
        i = j;
    }
    if (last_kept != array.size() - 1) unique_values.push_back(array.back());
    return unique_values;
}

// utils.rs:121-130
inline bool check_disparity_map(const Matrix &disparity_map)
{
    for (const auto &row : disparity_map)
        for (std::size_t j = 0; j + 1 < row.size(); ++j)
            if (j < row[j] || row[j + 1] > row[j] + 1) return false;
    return true;
}

}  // namespace stereo
