// ASCII PGM ("P2") input/output with the reference's file contract
// (diffusion_stereo/src/pgm_handler.rs:36-74,122-133).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "utils.hpp"

namespace stereo {

struct PgmImage {
    std::vector<std::vector<std::uint32_t>> matrix;  // [height][width]
    std::size_t width = 0, height = 0;
    std::uint32_t max_intensity = 0;
};

// Throws std::runtime_error where the reference panics (missing file, malformed
// number, value count != width * height).
PgmImage pgm_reader(const std::string &path);

// "P2\nW H\nmax\n" followed by one space-separated row per line.
void pgm_writer(const Matrix &matrix, const std::string &path, std::size_t max_intensity);

}  // namespace stereo
