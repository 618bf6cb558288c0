// Epsilon search (diffusion_stereo/src/epsilon_search.rs:39-112): the smallest epsilon, out of
// all penalty gaps in the graph, whose epsilon-consistent sub-graph is non-empty.
//
// The array of gaps has one entry per vertex and per existing edge (O(H*W*4*D^2) values), so
// building it is for small images, as in the reference (which leaves the search disabled in
// its main).  Both the array and the search follow the graph's mode: for a CrossingOutGraph in
// device mode the gaps are computed and sorted on the GPU (ds_epsilon_gaps) and the search runs
// the GPU crossing-out; otherwise host loops and host masks.
#pragma once
#include <vector>

#include "crossing_out_graph.hpp"

namespace stereo {

// :39-86.  Sorted ascending, then dedup_f64(tolerance).  Throws std::runtime_error where the
// reference's assert_ge! fires (a penalty below its minimum).
std::vector<double> create_array_of_epsilons(CrossingOutGraph &crossing_out_graph, double tolerance);

// :94-112.  Binary search over `array` (non-empty, sorted ascending).
double epsilon_search(CrossingOutGraph &crossing_out_graph, const std::vector<double> &array);

// Both steps in one for a graph in device mode, without the array (ds_epsilon_search): what
// epsilon_search(g, create_array_of_epsilons(g, 0.)) returns, for images of any size.
double epsilon_search_streamed(CrossingOutGraph &crossing_out_graph);

}  // namespace stereo
