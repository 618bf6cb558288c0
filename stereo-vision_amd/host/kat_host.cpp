// Known-answer checks of the host-side classes against the reference's own unit tests
// (file:line cited per block).  Needs a GPU: DiffusionGraph keeps its state on the
// device.  Exit status 0 = all passed.
#include <cmath>
#include <cstdio>
#include <limits>

#include "crossing_out_graph.hpp"
#include "epsilon_search.hpp"
#include "diffusion_graph.hpp"

using namespace stereo;
using Img = std::vector<std::vector<std::uint32_t>>;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static bool close(double a, double b, double eps = 1e-6) { return approx_equal(a, b, eps); }

int main()
{
    const double inf = std::numeric_limits<double>::infinity();
    {   // diffusion_stereo/src/diffusion_graph.rs:560-578 (penalty)
        DiffusionGraph g = DiffusionGraph::initialize(Img{{1, 1}, {0, 0}}, Img{{1, 0}, {0, 0}}, 2, 1.);
        CHECK(g.penalty(Matrix{{0, 1}, {0, 1}}) == 2.);
        g.set_potential(0, 0, 2, 0, 1.);
        CHECK(g.penalty(Matrix{{0, 1}, {0, 1}}) == 2.);
        CHECK(g.penalty(Matrix{{1, 1}, {1, 1}}) == inf);
    }
    {   // diffusion_graph.rs:618-668 (edge penalties, edge minima)
        DiffusionGraph g = DiffusionGraph::initialize(Img{{244, 172}}, Img{{168, 83}}, 2, 1.);
        g.set_potential(0, 0, 2, 0, 0.8);
        g.set_potential(0, 1, 0, 1, 0.1);
        CHECK(close(g.edge_penalty_with_potential(0, 0, 2, 0, 0), 0.8));
        CHECK(close(g.edge_penalty_with_potential(0, 1, 0, 1, 0), 1.9));
        CHECK(close(g.min_edge_between_neighbors(0, 0, 2, 0), 0.8));
        g.set_potential(0, 1, 0, 0, 2.);
        CHECK(close(g.min_edge_between_neighbors(0, 0, 2, 0), 1.9));
        CHECK(g.min_edge_between_neighbors(0, 0, 2, 1) == inf);
    }
    {   // diffusion_graph.rs:636-652 (edge_exists)
        DiffusionGraph g = DiffusionGraph::initialize(Img{{244, 172, 168, 192}, {83, 248, 38, 204}},
                                                      Img{{218, 138, 65, 18}, {225, 7, 114, 127}}, 4, 1.);
        CHECK(g.edge_exists(0, 0, 2, 0, 1));
        CHECK(g.edge_exists(0, 1, 0, 1, 0));
        CHECK(!g.edge_exists(0, 0, 2, 1, 0));
        CHECK(!g.edge_exists(0, 1, 0, 2, 1));
        CHECK(!g.edge_exists(0, 0, 0, 0, 0));
        CHECK(!g.edge_exists(0, 2, 2, 0, 3));
    }
    {   // diffusion_graph.rs:748-760 (one whole iteration drives both messages to 0)
        DiffusionGraph g = DiffusionGraph::initialize(Img{{1}, {0}}, Img{{1}, {0}}, 2, 1.);
        g.set_potential(0, 0, 3, 0, -0.3);
        g.set_potential(1, 0, 1, 0, 0.1);
        g.diffusion_act();
        CHECK(close(g.potential(0, 0, 3, 0), 0.));
        CHECK(close(g.potential(1, 0, 1, 0), 0.));
    }
    {   // diffusion_graph.rs:763-806 (vertex / edge minima on the host, energy on the device)
        DiffusionGraph g = DiffusionGraph::initialize(Img{{1, 1}, {0, 0}}, Img{{1, 0}, {0, 0}}, 2, 1.);
        g.set_potential(0, 0, 2, 0, 0.6);
        g.set_potential(0, 0, 3, 0, 358.);
        g.set_potential(0, 1, 0, 0, -13.7);
        g.set_potential(0, 1, 0, 1, 80.);
        g.set_potential(0, 1, 3, 1, -1E9);
        CHECK(g.min_penalty_vertex(0, 0).first == 0 && g.min_penalty_vertex(0, 0).second == -358.6);
        CHECK(g.min_penalty_vertex(0, 1).first == 0 && g.min_penalty_vertex(0, 1).second == 14.7);
        CHECK(g.min_penalty_edge(0, 0, 2) == -13.1);
        CHECK(g.min_penalty_edge(0, 1, 0) == -13.1);
        CHECK(close(g.energy(), -999999999.0, 1e-3));
        const Matrix m = g.depth_map();
        CHECK(m[0][0] == 0 && m[0][1] == 0);
    }
    {   // diffusion_stereo/src/crossing_out_graph.rs:402-428 (crossing_out)
        CrossingOutGraph c = CrossingOutGraph::initialize(
            DiffusionGraph::initialize(Img{{1, 1}, {1, 0}}, Img{{1, 0}, {0, 0}}, 2, 1.));
        c.set_edge(0, 0, 0, 2, 0, false);
        c.set_edge(0, 1, 0, 0, 0, false);
        c.set_edge(0, 1, 1, 3, 1, false);
        c.set_edge(1, 1, 1, 1, 1, false);
        c.crossing_out();
        CHECK(c.vertex(0, 0, 0) && c.vertex(0, 1, 1) && c.vertex(1, 0, 0) && c.vertex(1, 1, 0));
        CHECK(!c.vertex(0, 1, 0) && !c.vertex(1, 1, 1));
        CHECK(c.edge(0, 0, 0, 2, 1) && c.edge(0, 0, 0, 3, 0) && c.edge(1, 0, 0, 2, 0));
        CHECK(!c.edge(1, 0, 0, 2, 1) && c.edge(0, 1, 1, 3, 0) && !c.edge(0, 1, 1, 3, 1));
    }
    {   // crossing_out_graph.rs:449-491 (initialize_edges / initialize_with_epsilon)
        CrossingOutGraph c = CrossingOutGraph::initialize(
            DiffusionGraph::initialize(Img{{1, 1, 0}}, Img{{1, 0, 0}}, 2, 1.));
        DiffusionGraph &g = c.diffusion_graph;
        g.set_potential(0, 0, 2, 0, 0.6);
        g.set_potential(0, 1, 0, 0, -13.7);
        g.set_potential(0, 1, 0, 1, 80.);
        g.set_potential(0, 1, 2, 0, 358.);
        g.set_potential(0, 1, 2, 1, -1E9);
        g.set_potential(0, 2, 0, 0, -0.3);
        g.set_potential(0, 2, 0, 1, 0.1);
        c.initialize_edges(0.1);
        CHECK(c.edge(0, 0, 0, 2, 0) && !c.edge(0, 0, 0, 2, 1));
        CHECK(!c.edge(0, 1, 0, 2, 0) && !c.edge(0, 1, 0, 2, 1) && !c.edge(0, 1, 1, 2, 0) && c.edge(0, 1, 1, 2, 1));
    }
    for (bool on_device : {false, true}) {   // crossing_out_graph.rs:643-663 (find_best_labeling -> [0, 0, 1])
        CrossingOutGraph c = CrossingOutGraph::initialize(
            DiffusionGraph::initialize(Img{{1, 1, 0}}, Img{{1, 0, 0}}, 3, 1.), on_device);
        DiffusionGraph &g = c.diffusion_graph;
        g.set_potential(0, 0, 2, 0, 0.6);
        g.set_potential(0, 1, 0, 0, -13.7);
        g.set_potential(0, 1, 0, 1, 80.);
        g.set_potential(0, 1, 2, 0, 358.);
        g.set_potential(0, 1, 2, 1, -1E9);
        g.set_potential(0, 2, 0, 0, -0.3);
        g.set_potential(0, 2, 0, 1, 0.1);
        g.set_potential(0, 2, 0, 2, 0.8);
        const Matrix m = c.find_best_labeling();
        CHECK(m[0][0] == 0 && m[0][1] == 0 && m[0][2] == 1);
    }
    {   // crossing_out_graph.rs:543-562 (is_not_empty)
        CrossingOutGraph c = CrossingOutGraph::initialize(
            DiffusionGraph::initialize(Img{{1, 1}}, Img{{1, 0}}, 2, 1.));
        c.set_vertex(0, 1, 0, false);
        c.set_edge(0, 0, 0, 2, 1, false);
        c.set_edge(0, 1, 1, 0, 0, false);
        CHECK(c.is_not_empty());
        c.set_vertex(0, 1, 0, true);
        c.set_vertex(0, 0, 0, false);
        CHECK(!c.is_not_empty());
    }
    {   // device crossing-out against the host masks on the same phi, and the whole
        // diffusion_while_not_consistent loop (crossing_out_graph.rs:254-275) in both modes
        const Img left{{3, 9, 4, 4, 7, 1}, {2, 8, 5, 3, 7, 2}, {3, 9, 5, 4, 6, 1}, {1, 2, 3, 4, 5, 6}};
        const Img right{{9, 4, 4, 7, 1, 0}, {8, 5, 3, 7, 2, 0}, {9, 5, 4, 6, 1, 0}, {2, 3, 4, 5, 6, 0}};
        for (double eps : {0.01, 2.0, 25.0}) {
            CrossingOutGraph host = CrossingOutGraph::initialize(DiffusionGraph::initialize(left, right, 4, 1.2));
            host.diffusion_graph.diffusion(1, 7, false);
            host.initialize_with_epsilon(eps);
            host.crossing_out();
            std::vector<std::uint8_t> dev;
            const bool ok = host.diffusion_graph.crossing_out_on_device(eps, dev);
            CHECK(ok == host.is_not_empty());
            bool same = true;
            for (std::size_t i = 0; i < 4; ++i)
                for (std::size_t j = 0; j < 6; ++j)
                    for (std::size_t d = 0; d < 4; ++d) same = same && (dev[(i * 6 + j) * 4 + d] != 0) == host.vertex(i, j, d);
            CHECK(same);
        }
        CrossingOutGraph a = CrossingOutGraph::initialize(DiffusionGraph::initialize(left, right, 4, 1.2), false);
        CrossingOutGraph b = CrossingOutGraph::initialize(DiffusionGraph::initialize(left, right, 4, 1.2), true);
        const double eps = 1. / 240.;
        const std::size_t na = a.diffusion_while_not_consistent(eps, 50, 100, false);
        const std::size_t nb = b.diffusion_while_not_consistent(eps, 50, 100, false);
        CHECK(na == nb);
        CHECK(a.simple_best_labeling() == b.simple_best_labeling());
        CHECK(b.is_not_empty());
    }
    for (bool on_device : {false, true}) {   // diffusion_stereo/src/epsilon_search.rs:115-170
        {   // :115-128 (test_epsilon_search_zero)
            CrossingOutGraph c = CrossingOutGraph::initialize(
                DiffusionGraph::initialize(Img{{0, 0}}, Img{{0, 0}}, 2, 1.), on_device);
            const std::vector<double> array = create_array_of_epsilons(c, 0.01);
            CHECK(epsilon_search(c, array) == 0.);
            if (on_device) CHECK(epsilon_search_streamed(c) == 0.);   // the same search without the array
        }
        {   // :131-145 (test_epsilon_search)
            CrossingOutGraph c = CrossingOutGraph::initialize(
                DiffusionGraph::initialize(Img{{1, 1}}, Img{{1, 0}}, 2, 1.), on_device);
            c.diffusion_graph.set_potential(0, 0, 2, 0, 9.4);
            c.diffusion_graph.set_potential(0, 1, 0, 0, -1.8);
            c.diffusion_graph.set_potential(0, 1, 0, 1, 6.7);
            CHECK(epsilon_search(c, {0., 1., 2., 3., 4., 5., 6., 7., 8., 9.}) == 9.);
        }
        {   // :148-170 (test_create_array_of_epsilons)
            CrossingOutGraph c = CrossingOutGraph::initialize(
                DiffusionGraph::initialize(Img{{1, 1}}, Img{{1, 0}}, 2, 1.), on_device);
            c.diffusion_graph.set_potential(0, 0, 2, 0, 0.6);
            c.diffusion_graph.set_potential(0, 1, 0, 0, -0.3);
            c.diffusion_graph.set_potential(0, 1, 0, 1, 0.1);
            std::vector<double> array = create_array_of_epsilons(c, 0.);
            CHECK(array.size() == 3 && array[0] == 0. && close(array[1], 1.4, 1e-6) && close(array[2], 1.4, 1e-6));
            array = create_array_of_epsilons(c, 1.5);
            CHECK(array.size() == 2 && array[0] == 0. && close(array[1], 1.4, 1e-6));
            // over that graph's own array (tolerance 0) both searches agree
            if (on_device) CHECK(epsilon_search_streamed(c) == epsilon_search(c, create_array_of_epsilons(c, 0.)));
        }
    }
    {   // utils.rs:250-268 (dedup_f64)
        CHECK((dedup_f64({1., 1.5, 2., 3.}, 1.) == std::vector<double>{1., 3.}));
        CHECK((dedup_f64({0.}, 0.) == std::vector<double>{0.}));
        CHECK((dedup_f64({1., 1.5, 2., 3.}, 0.) == std::vector<double>{1., 1.5, 2., 3.}));
    }
    {   // diffusion_stereo/src/utils.rs:214-247
        CHECK(number_of_neighbors(0, 1, 2, 3) == 3 && number_of_neighbors(1, 1, 3, 3) == 4);
        CHECK(check_disparity_map(Matrix{{0, 0}, {0, 1}}) && !check_disparity_map(Matrix{{0, 2, 1}}));
    }
    std::printf(failures ? "%d host KAT failure(s)\n" : "host KATs ok\n", failures);
    return failures ? 1 : 0;
}
