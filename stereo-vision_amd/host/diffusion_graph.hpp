// Host-side DiffusionGraph: the reference's public interface
// (diffusion_stereo/src/diffusion_graph.rs:39-539) over the device library.
//
// State lives on the GPU (include/diffstereo.h).  The sweeps, the energy and the
// per-pixel argmin run there; the fine-grained accessors the crossing-out code needs
// (vertex / edge penalties for one (i, j, n, d)) are evaluated on a host copy of phi
// that is fetched lazily and dropped whenever the device state changes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "diffstereo.h"
#include "utils.hpp"

namespace stereo {

class DiffusionGraph {
public:
    // diffusion_graph.rs:65-88.  Images hold 0..255; 1 <= max_disparity <= 256.
    // dtype: DS_F64 (the reference's precision) or DS_F32.
    static DiffusionGraph initialize(const std::vector<std::vector<std::uint32_t>> &left_image,
                                     const std::vector<std::vector<std::uint32_t>> &right_image,
                                     std::size_t max_disparity, double smoothing_term, int dtype = DS_F64);
    // The same for the band [row_begin, row_end) of the image on HIP device `device`: one of
    // several processes, one per GPU (the reference has one process; ds_create_band).  Only the
    // device operations work on a band -- diffusion, energy, crossing_out_on_device -- and they
    // run over all bands once attach() has said how the halo rows travel.
    static DiffusionGraph initialize_band(const std::vector<std::vector<std::uint32_t>> &left_image,
                                          const std::vector<std::vector<std::uint32_t>> &right_image,
                                          std::size_t max_disparity, double smoothing_term, int dtype,
                                          std::size_t row_begin, std::size_t row_end, int device);
    void attach(ds_comm *comm, int up_rank, int down_rank);              // the library's RCCL transport
    void attach(const ds_transport &transport, int up_rank, int down_rank);
    bool banded() const { return banded_; }
    std::size_t rows() const { return rows_; }            // rows this process owns (= height() unless banded)
    std::size_t row_begin() const { return row_begin_; }
    DiffusionGraph(DiffusionGraph &&o) noexcept;
    DiffusionGraph &operator=(DiffusionGraph &&o) noexcept;
    DiffusionGraph(const DiffusionGraph &) = delete;
    DiffusionGraph &operator=(const DiffusionGraph &) = delete;
    ~DiffusionGraph();

    std::size_t height() const { return H_; }
    std::size_t width() const { return W_; }
    std::size_t max_disparity;   // public fields of the reference struct
    double smoothing_term;
    std::vector<std::vector<std::uint32_t>> left_image, right_image;

    // potentials[i][j][n][d] (diffusion_graph.rs:43): read / write through the host copy
    double potential(std::size_t i, std::size_t j, std::size_t n, std::size_t d);
    void set_potential(std::size_t i, std::size_t j, std::size_t n, std::size_t d, double v);

    double sum_of_potentials(std::size_t i, std::size_t j, std::size_t d);                 // :99-107
    double vertex_penalty_with_potentials(std::size_t i, std::size_t j, std::size_t d);    // :118-122
    double edge_penalty_with_potential(std::size_t i, std::size_t j, std::size_t n, std::size_t d,
                                       std::size_t n_d);                                   // :135-141
    bool edge_exists(std::size_t i, std::size_t j, std::size_t n, std::size_t d, std::size_t n_d) const;  // :152-169
    double penalty(const Matrix &disparity_map);                                           // :177-201
    double min_edge_between_neighbors(std::size_t i, std::size_t j, std::size_t n, std::size_t d);  // :212-224
    std::pair<std::size_t, double> min_penalty_vertex(std::size_t i, std::size_t j);       // :456-470
    double min_penalty_edge(std::size_t i, std::size_t j, std::size_t n);                  // :480-494

    // Device operations.
    // false: red-black order (fast); true: the reference's raster order, exactly (one launch
    // per anti-diagonal: slow, for checking results against the sequential implementation)
    bool raster_order = false;
    void diffusion_act();                                                   // :264-293
    void diffusion(std::size_t first_iteration, std::size_t number_of_iterations,
                   bool verbose = true);                                    // :349-368
    double energy();                                                        // :502-518
    Matrix depth_map();          // :376-390 without the file write: argmin_d f_t(d)
    Matrix belief_depth_map();   // argmin_d of f_t after the pixel's vertex step (:302-317)
    void build_depth_map(std::size_t iteration);                            // :376-390
    // CrossingOutGraph::initialize_with_epsilon + crossing_out on the current phi
    // (crossing_out_graph.rs:67-183) on the device; vertices[i][j][d] is the resulting
    // vertex mask.  Returns is_not_empty() (:187-193).
    bool crossing_out_on_device(double epsilon, std::vector<std::uint8_t> &vertices, int *passes = nullptr);
    // find_best_labeling (crossing_out_graph.rs:349-364) on the masks the last
    // crossing_out_on_device left on the GPU (ds_find_best_labeling).
    Matrix find_best_labeling_on_device(bool &not_empty);
    // All vertex and edge penalty gaps of the graph (epsilon_search.rs:39-74), computed and
    // sorted on the GPU, exact duplicates removed; largest_repeated: the largest one occurred
    // more than once (ds_epsilon_gaps).
    std::vector<double> epsilon_gaps_on_device(bool &largest_repeated);
    // epsilon_search over the array of all gaps with tolerance 0 (epsilon_search.rs:39-112) without
    // building the array (ds_epsilon_search): for images of any size.
    double epsilon_search_on_device();

private:
    DiffusionGraph() = default;
    void check(int rc) const;
    int iterate(int n);  // n iterations in the chosen order, over all bands if banded
    void pull();   // device -> host copy of phi, if stale
    void push();   // host copy -> device, if modified
    std::size_t idx(std::size_t i, std::size_t j, std::size_t n, std::size_t d) const
    {
        return ((i * W_ + j) * 4 + n) * max_disparity + d;
    }
    double q(std::size_t i, std::size_t j, std::size_t d) const
    {
        const double a = left_image[i][j], b = right_image[i][j - d];
        return a > b ? a - b : b - a;
    }

    ds_handle *h_ = nullptr;
    int dtype_ = DS_F64;
    std::size_t H_ = 0, W_ = 0;
    std::size_t row_begin_ = 0, rows_ = 0;
    bool banded_ = false;
    void whole_image_only(const char *what) const
    {
        if (banded_) throw std::logic_error(std::string(what) + " needs the whole image in one process");
    }
    std::vector<double> phi_;  // host copy, reference layout
    bool host_valid_ = false, host_dirty_ = false;
};

}  // namespace stereo
