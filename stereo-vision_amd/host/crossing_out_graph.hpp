// Host-side CrossingOutGraph: epsilon-consistent sub-graph and labeling
// (diffusion_stereo/src/crossing_out_graph.rs:33-365).  The diffusion batches run on
// the GPU through DiffusionGraph.  Two modes:
//   host masks   - vertex and edge masks on the host, as in the reference; the edge mask
//                  needs O(H*W*4*D^2) bytes, so this is for small images;
//   device       - initialize_with_epsilon + crossing_out run on the GPU (ds_crossing_out),
//                  only the resulting vertex mask comes back; find_best_labeling continues
//                  on the GPU's masks (ds_find_best_labeling); any image size.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "diffusion_graph.hpp"

namespace stereo {

class CrossingOutGraph {
public:
    // crossing_out_graph.rs:51-59; both masks start all-true (main.rs:49-50).
    // on_device: see above; the edge mask is then never allocated.
    static CrossingOutGraph initialize(DiffusionGraph diffusion_graph, bool on_device = false);
    bool on_device() const { return on_device_; }

    DiffusionGraph diffusion_graph;

    bool vertex(std::size_t i, std::size_t j, std::size_t d) const { return vertices_[vi(i, j, d)] != 0; }
    void set_vertex(std::size_t i, std::size_t j, std::size_t d, bool v) { vertices_[vi(i, j, d)] = v; }
    bool edge(std::size_t i, std::size_t j, std::size_t d, std::size_t n, std::size_t n_d) const
    {
        return edges_[ei(i, j, d, n, n_d)] != 0;
    }
    void set_edge(std::size_t i, std::size_t j, std::size_t d, std::size_t n, std::size_t n_d, bool v)
    {
        edges_[ei(i, j, d, n, n_d)] = v;
    }

    void initialize_with_epsilon(double epsilon);   // :67-70
    void initialize_vertices(double epsilon);       // :78-93
    void initialize_edges(double epsilon);          // :102-130
    void crossing_out();                            // :135-183
    bool is_not_empty() const;                      // :187-193
    // initialize_with_epsilon + crossing_out + is_not_empty in one call, in either mode
    // (the sequence epsilon_search.rs:100-102 and crossing_out_graph.rs:258-264 run)
    bool consistent_with_epsilon(double epsilon);
    bool vertices_exist() const;                    // :196-211
    bool edges_exist() const;                       // :214-246
    // :254-275.  Returns the number of iterations run; throws std::runtime_error if
    // the graph is still not consistent after max_batches batches (the reference
    // would loop forever).
    std::size_t diffusion_while_not_consistent(double epsilon, std::size_t batch_size,
                                               std::size_t max_batches = 1000, bool verbose = true);
    std::size_t min_vertex_between_existing(std::size_t i, std::size_t j);   // :284-300
    std::size_t top_vertex_between_existing(std::size_t i, std::size_t j) const;  // :307-315
    Matrix simple_best_labeling() const;                                     // :318-327
    void cross_vertex(std::size_t i, std::size_t j, std::size_t disparity);  // :336-345
    Matrix find_best_labeling();                                             // :349-364

private:
    explicit CrossingOutGraph(DiffusionGraph g) : diffusion_graph(std::move(g)) {}
    std::size_t vi(std::size_t i, std::size_t j, std::size_t d) const { return (i * W_ + j) * D_ + d; }
    std::size_t ei(std::size_t i, std::size_t j, std::size_t d, std::size_t n, std::size_t n_d) const
    {
        return (((i * W_ + j) * D_ + d) * 4 + n) * D_ + n_d;
    }
    void need_host_masks(const char *what) const;
    std::size_t H_ = 0, W_ = 0, D_ = 0;
    bool on_device_ = false;
    bool device_masks_ = false;  // device mode: the GPU holds masks of some epsilon
    std::vector<std::uint8_t> vertices_, edges_;
};

}  // namespace stereo
