#include "pgm.hpp"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace stereo {

PgmImage pgm_reader(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string line;
    PgmImage img;
    if (!std::getline(f, line)) throw std::runtime_error("missing magic line in " + path);  // not checked
    if (!std::getline(f, line)) throw std::runtime_error("missing size line in " + path);
    {
        std::istringstream s(line);
        if (!(s >> img.width >> img.height)) throw std::runtime_error("bad size line in " + path);
    }
    if (!std::getline(f, line)) throw std::runtime_error("missing maxval line in " + path);
    {
        std::istringstream s(line);
        if (!(s >> img.max_intensity)) throw std::runtime_error("bad maxval line in " + path);
    }
    std::vector<std::int32_t> values;
    std::string token;
    while (f >> token) {
        std::size_t used = 0;
        long v;
        try {
            v = std::stol(token, &used);
        } catch (const std::exception &) {
            throw std::runtime_error("bad pixel value '" + token + "' in " + path);
        }
        if (used != token.size()) throw std::runtime_error("bad pixel value '" + token + "' in " + path);
        values.push_back(static_cast<std::int32_t>(v));
    }
    if (values.size() != img.width * img.height)
        throw std::runtime_error("pixel count does not match width * height in " + path);
    img.matrix.assign(img.height, std::vector<std::uint32_t>(img.width));
    for (std::size_t i = 0; i < img.height; ++i)
        for (std::size_t j = 0; j < img.width; ++j)
            img.matrix[i][j] = static_cast<std::uint32_t>(values[i * img.width + j]);
    return img;
}

void pgm_writer(const Matrix &matrix, const std::string &path, std::size_t max_intensity)
{
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Unable to create file " + path);
    f << "P2\n" << (matrix.empty() ? 0 : matrix[0].size()) << " " << matrix.size() << "\n"
      << max_intensity << "\n";
    for (const auto &row : matrix) {
        for (std::size_t j = 0; j < row.size(); ++j) f << (j ? " " : "") << row[j];
        f << "\n";
    }
    if (!f) throw std::runtime_error("Unable to write " + path);
}

}  // namespace stereo
