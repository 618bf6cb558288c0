#include "epsilon_search.hpp"

#include <algorithm>
#include <cstdio>
#include <stdexcept>

namespace stereo {

std::vector<double> create_array_of_epsilons(CrossingOutGraph &crossing_out_graph, double tolerance)
{
    DiffusionGraph &g = crossing_out_graph.diffusion_graph;
    if (crossing_out_graph.on_device()) {
        // The gaps come from the GPU already sorted and without exact duplicates.  dedup_f64
        // repeats the array's last element when its run is longer than one value: a run of
        // exact duplicates of the largest gap is no longer visible here, the flag stands for it.
        bool largest_repeated = false;
        const std::vector<double> gaps = g.epsilon_gaps_on_device(largest_repeated);
        std::vector<double> unique_values = dedup_f64(gaps, tolerance);
        if (!gaps.empty() && largest_repeated) {
            std::size_t run_start = gaps.size() - 1;  // first element of the last kept run
            {
                std::size_t i = 0;
                while (i < gaps.size()) {
                    run_start = i;
                    std::size_t j = i + 1;
                    while (j < gaps.size() && approx_equal(gaps[i], gaps[j], tolerance)) ++j;
                    i = j;
                }
            }
            if (run_start == gaps.size() - 1) unique_values.push_back(gaps.back());
        }
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

}  // namespace stereo
