#include "epsilon_search.hpp"

#include "diffstereo.h"

#include <algorithm>
#include <cstdio>
#include <stdexcept>

namespace stereo {

std::vector<double> create_array_of_epsilons(CrossingOutGraph &crossing_out_graph, double tolerance)
{
    DiffusionGraph &g = crossing_out_graph.diffusion_graph;
    if (crossing_out_graph.on_device()) {
        // The gaps come from the GPU already sorted and without exact duplicates; the library's
        // ds_dedup_gaps is dedup_f64 for that form (it knows whether the largest gap was repeated,
        // which decides whether the reference's dedup pushes the last element once more).
        bool largest_repeated = false;
        const std::vector<double> gaps = g.epsilon_gaps_on_device(largest_repeated);
        std::vector<double> unique_values(gaps.size() + 1);
        std::size_t kept = 0;
        if (ds_dedup_gaps(gaps.data(), gaps.size(), tolerance, largest_repeated ? 1 : 0, unique_values.data(), &kept) != 0)
            throw std::runtime_error("ds_dedup_gaps failed");
        unique_values.resize(kept);
        return unique_values;
    }
    const std::size_t height = g.height(), width = g.width(), max_disparity = g.max_disparity;
    std::vector<double> array;
    for (std::size_t i = 0; i < height; ++i)
        for (std::size_t j = 0; j < width; ++j) {
            const double min_vertex = g.min_penalty_vertex(i, j).second;
            for (std::size_t d = 0; d < max_disparity && d <= j; ++d) {
                const double vertex = g.vertex_penalty_with_potentials(i, j, d);
                if (vertex < min_vertex) throw std::runtime_error("vertex penalty below the pixel's minimum");
                array.push_back(vertex - min_vertex);
                for (std::size_t n = 0; n < 4; ++n) {
                    if (!neighbor_exists(i, j, n, height, width)) continue;
                    const double min_edge = g.min_penalty_edge(i, j, n);
                    for (std::size_t n_d = 0; n_d < max_disparity; ++n_d) {
                        if (!g.edge_exists(i, j, n, d, n_d)) continue;
                        const double edge = g.edge_penalty_with_potential(i, j, n, d, n_d);
                        if (edge < min_edge) throw std::runtime_error("edge penalty below the pair's minimum");
                        array.push_back(edge - min_edge);
                    }
                }
            }
        }
    std::sort(array.begin(), array.end());
    return dedup_f64(array, tolerance);
}

double epsilon_search(CrossingOutGraph &crossing_out_graph, const std::vector<double> &array)
{
    if (array.empty()) throw std::invalid_argument("epsilon_search: empty array");
    std::size_t first = 0, last = array.size() - 1, median = (first + last) / 2;
    while (first < last) {
        if (crossing_out_graph.consistent_with_epsilon(array[median])) last = median;
        else first = median + 1;
        median = (first + last) / 2;
    }
    return array[median];
}

double epsilon_search_streamed(CrossingOutGraph &crossing_out_graph)
{
    if (!crossing_out_graph.on_device()) throw std::invalid_argument("epsilon_search_streamed needs a graph in device mode");
    return crossing_out_graph.diffusion_graph.epsilon_search_on_device();
}

}  // namespace stereo
