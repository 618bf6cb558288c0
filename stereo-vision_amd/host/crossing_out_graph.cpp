#include "crossing_out_graph.hpp"

#include <cstdio>
#include <stdexcept>
#include <string>

namespace stereo {

CrossingOutGraph CrossingOutGraph::initialize(DiffusionGraph g, bool on_device)
{
    CrossingOutGraph c(std::move(g));
    c.on_device_ = on_device;
    if (c.diffusion_graph.banded() && !on_device)
        throw std::logic_error("a band keeps its masks on the device (initialize(..., true))");
    c.H_ = c.diffusion_graph.rows();  // a band's masks cover its own rows
    c.W_ = c.diffusion_graph.width();
    c.D_ = c.diffusion_graph.max_disparity;
    c.vertices_.assign(c.H_ * c.W_ * c.D_, 1);
    if (!on_device) c.edges_.assign(c.H_ * c.W_ * c.D_ * 4 * c.D_, 1);
    return c;
}

void CrossingOutGraph::need_host_masks(const char *what) const
{
    if (on_device_) throw std::logic_error(std::string(what) + " needs the host edge mask (initialize(..., false))");
}

void CrossingOutGraph::initialize_with_epsilon(double epsilon)
{
    need_host_masks("initialize_with_epsilon");
    initialize_vertices(epsilon);
    initialize_edges(epsilon);
}

void CrossingOutGraph::initialize_vertices(double epsilon)
{
    DiffusionGraph &g = diffusion_graph;
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j) {
            const double mn = g.min_penalty_vertex(i, j).second;
            for (std::size_t d = 0; d < D_; ++d)
                set_vertex(i, j, d, j >= d && g.vertex_penalty_with_potentials(i, j, d) <= mn + epsilon);
        }
}

void CrossingOutGraph::initialize_edges(double epsilon)
{
    // The reference recomputes min_penalty_edge(i, j, n) for every d; it does not depend
    // on d, so it is evaluated once per (pixel, slot) here with the same result.
    DiffusionGraph &g = diffusion_graph;
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j)
            for (std::size_t n = 0; n < 4; ++n) {
                const bool ex = neighbor_exists(i, j, n, H_, W_);
                const double mn = ex ? g.min_penalty_edge(i, j, n) : 0.;
                for (std::size_t d = 0; d < D_; ++d)
                    for (std::size_t n_d = 0; n_d < D_; ++n_d)
                        set_edge(i, j, d, n, n_d,
                                 ex && g.edge_exists(i, j, n, d, n_d) &&
                                     g.edge_penalty_with_potential(i, j, n, d, n_d) <= mn + epsilon);
            }
}

void CrossingOutGraph::crossing_out()
{
    need_host_masks("crossing_out");
    const DiffusionGraph &g = diffusion_graph;
    bool change_indicator = true;
    while (change_indicator) {
        change_indicator = false;
        for (std::size_t i = 0; i < H_; ++i)
            for (std::size_t j = 0; j < W_; ++j)
                for (std::size_t d = 0; d < D_ && d <= j; ++d) {
                    if (!vertex(i, j, d)) {
                        for (std::size_t n = 0; n < 4; ++n) {
                            if (!neighbor_exists(i, j, n, H_, W_)) continue;
                            auto [n_i, n_j, n_index] = neighbor_index(i, j, n);
                            for (std::size_t n_d = 0; n_d < D_; ++n_d)
                                if (g.edge_exists(i, j, n, d, n_d) && edge(i, j, d, n, n_d)) {
                                    set_edge(i, j, d, n, n_d, false);
                                    set_edge(n_i, n_j, n_d, n_index, d, false);
                                    change_indicator = true;
                                }
                        }
                    }
                    for (std::size_t n = 0; n < 4; ++n) {
                        if (!neighbor_exists(i, j, n, H_, W_)) continue;
                        bool edge_exist = false;
                        for (std::size_t n_d = 0; n_d < D_; ++n_d)
                            if (g.edge_exists(i, j, n, d, n_d) && edge(i, j, d, n, n_d)) {
                                edge_exist = true;
                                break;
                            }
                        if (!edge_exist && vertex(i, j, d)) {
                            set_vertex(i, j, d, false);
                            change_indicator = true;
                        }
                    }
                }
    }
}

bool CrossingOutGraph::is_not_empty() const
{
    // after the device closure every alive vertex has an alive edge in every direction,
    // so edges_exist() is implied by vertices_exist()
    return vertices_exist() && (on_device_ || edges_exist());
}

bool CrossingOutGraph::consistent_with_epsilon(double epsilon)
{
    if (on_device_) {
        device_masks_ = true;
        return diffusion_graph.crossing_out_on_device(epsilon, vertices_);
    }
    initialize_with_epsilon(epsilon);
    crossing_out();
    return is_not_empty();
}

bool CrossingOutGraph::vertices_exist() const
{
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j) {
            bool found = false;
            for (std::size_t d = 0; d < D_ && d <= j; ++d)
                if (vertex(i, j, d)) found = true;
            if (!found) return false;
        }
    return true;
}

bool CrossingOutGraph::edges_exist() const
{
    need_host_masks("edges_exist");
    const DiffusionGraph &g = diffusion_graph;
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j)
            for (std::size_t n = 0; n < 4; ++n) {
                if (!neighbor_exists(i, j, n, H_, W_)) continue;
                bool found = false;
                for (std::size_t d = 0; d < D_ && !found; ++d)
                    for (std::size_t n_d = 0; n_d < D_; ++n_d)
                        if (g.edge_exists(i, j, n, d, n_d) && edge(i, j, d, n, n_d)) {
                            found = true;
                            break;
                        }
                if (!found) return false;
            }
    return true;
}

std::size_t CrossingOutGraph::diffusion_while_not_consistent(double epsilon, std::size_t batch_size,
                                                             std::size_t max_batches, bool verbose)
{
    std::size_t i = 1;
    for (std::size_t b = 0; b < max_batches; ++b) {
        diffusion_graph.diffusion(i, batch_size, verbose);
        if (verbose) {
            std::printf("%zu iterations of diffusion ended\n", i + batch_size - 1);
            std::printf("Initializing crossing out graph with epsilon ...\n");
        }
        if (on_device_) {
            int passes = 0;
            const bool ok = diffusion_graph.crossing_out_on_device(epsilon, vertices_, &passes);
            device_masks_ = true;
            if (verbose) std::printf("Crossing out is done (%d passes on the device)\n", passes);
            if (ok) {
                if (verbose) std::printf("Graph is consistent. Diffusion is done\n");
                return i + batch_size - 1;
            }
            if (verbose) std::printf("Need more diffusion iterations\n");
            i += batch_size;
            continue;
        }
        initialize_with_epsilon(epsilon);
        if (is_not_empty()) {
            if (verbose) {
                std::printf("After initialization with epsilon, resulting graph is not empty\n");
                std::printf("Crossing out ...\n");
            }
            crossing_out();
            if (verbose) std::printf("Crossing out is done\n");
            if (is_not_empty()) {
                if (verbose) std::printf("Graph is consistent. Diffusion is done\n");
                return i + batch_size - 1;
            }
            if (verbose) std::printf("Need more diffusion iterations\n");
        }
        i += batch_size;
    }
    throw std::runtime_error("graph is not consistent after " + std::to_string(max_batches) + " batches");
}

std::size_t CrossingOutGraph::min_vertex_between_existing(std::size_t i, std::size_t j)
{
    double min_vertex = std::numeric_limits<double>::infinity();
    std::size_t disparity = 0;
    bool found = false;
    for (std::size_t d = 0; d < D_ && d <= j; ++d)
        if (vertex(i, j, d)) {
            const double v = diffusion_graph.vertex_penalty_with_potentials(i, j, d);
            if (v < min_vertex) { min_vertex = v; disparity = d; found = true; }
        }
    if (!found)
        throw std::runtime_error("Disparity for pixel (" + std::to_string(i) + ", " + std::to_string(j) +
                                 ") wasn't found");
    return disparity;
}

std::size_t CrossingOutGraph::top_vertex_between_existing(std::size_t i, std::size_t j) const
{
    for (std::size_t d = 0; d < D_ && d <= j; ++d)
        if (vertex(i, j, d)) return d;
    throw std::runtime_error("Disparity for pixel (" + std::to_string(i) + ", " + std::to_string(j) +
                             ") wasn't found");
}

Matrix CrossingOutGraph::simple_best_labeling() const
{
    Matrix m(H_, std::vector<std::size_t>(W_, 0));
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j) m[i][j] = top_vertex_between_existing(i, j);
    return m;
}

void CrossingOutGraph::cross_vertex(std::size_t i, std::size_t j, std::size_t disparity)
{
    if (disparity >= D_ || !vertex(i, j, disparity))
        throw std::runtime_error("Choosen vertex is crossed out");
    for (std::size_t d = 0; d < D_; ++d)
        if (vertex(i, j, d) && d != disparity) set_vertex(i, j, d, false);
}

Matrix CrossingOutGraph::find_best_labeling()
{
    if (diffusion_graph.banded()) throw std::logic_error("find_best_labeling walks the whole image: one process only");
    if (on_device_) {
        // The masks live on the GPU since the last consistent_with_epsilon /
        // diffusion_while_not_consistent.  Without one, every vertex and edge is alive, as
        // initialize() leaves them: an epsilon no gap reaches gives the device that state.
        if (!device_masks_) diffusion_graph.crossing_out_on_device(1e30, vertices_);
        device_masks_ = true;
        bool not_empty = false;
        Matrix m = diffusion_graph.find_best_labeling_on_device(not_empty);
        if (!not_empty) throw std::runtime_error("Graph is empty");
        for (std::size_t i = 0; i < H_; ++i)
            for (std::size_t j = 0; j < W_; ++j)
                for (std::size_t d = 0; d < D_; ++d) set_vertex(i, j, d, d == m[i][j]);
        return m;
    }
    Matrix m(H_, std::vector<std::size_t>(W_, 0));
    for (std::size_t i = 0; i < H_; ++i)
        for (std::size_t j = 0; j < W_; ++j) {
            m[i][j] = min_vertex_between_existing(i, j);
            if (m[i][j] > j) throw std::runtime_error("d > j");
            cross_vertex(i, j, m[i][j]);
            crossing_out();
            if (!is_not_empty()) throw std::runtime_error("Graph is empty");
        }
    return m;
}

}  // namespace stereo
