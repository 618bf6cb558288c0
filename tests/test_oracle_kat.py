"""Known-answer tests of the reference, run against the CPU oracle.

Each test is the oracle-side twin of one #[test] in the reference (file:line in
the docstring); names, inputs and expected values are the reference's.  They
are what pins oracle/diffusion_oracle.c.  No GPU needed.
"""
import math

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle.pyoracle import OracleGraph

INF = float("inf")


def approx(a, b, eps=1e-6):
    return po.approx_equal(a, b, eps)


# ---------------------------------------------------------------- utils.rs

def test_neighbor_exists():
    """diffusion_stereo/src/utils.rs:171-180"""
    assert po.neighbor_exists(1, 1, 0, 2, 2)
    assert not po.neighbor_exists(0, 0, 0, 2, 2)
    assert po.neighbor_exists(1, 0, 1, 2, 2)
    assert not po.neighbor_exists(0, 0, 1, 2, 2)
    assert po.neighbor_exists(0, 0, 2, 2, 2)
    assert not po.neighbor_exists(0, 1, 2, 2, 2)
    assert po.neighbor_exists(0, 0, 3, 2, 2)
    assert not po.neighbor_exists(1, 0, 3, 2, 2)


def test_edge_exists_panic():
    """utils.rs:183-186 (#[should_panic])"""
    with pytest.raises(ValueError):
        po.neighbor_exists(1, 1, 6, 2, 2)


def test_neighbor_index():
    """utils.rs:189-205"""
    assert po.neighbor_index(0, 1, 0) == (0, 0, 2)
    assert po.neighbor_index(1, 0, 1) == (0, 0, 3)
    assert po.neighbor_index(0, 0, 2) == (0, 1, 0)
    assert po.neighbor_index(0, 0, 3) == (1, 0, 1)


def test_neighbor_index_panic():
    """utils.rs:208-211 (#[should_panic])"""
    with pytest.raises(ValueError):
        po.neighbor_index(0, 0, 4)


def test_number_of_neighbors():
    """utils.rs:214-223 (arguments: i, j, height, width)"""
    assert po.number_of_neighbors(0, 0, 0, 0) == 0
    assert po.number_of_neighbors(0, 0, 1, 0) == 0
    assert po.number_of_neighbors(0, 0, 1, 1) == 0
    assert po.number_of_neighbors(0, 0, 1, 2) == 1
    assert po.number_of_neighbors(0, 0, 2, 2) == 2
    assert po.number_of_neighbors(0, 1, 1, 3) == 2
    assert po.number_of_neighbors(0, 1, 2, 3) == 3
    assert po.number_of_neighbors(1, 1, 3, 3) == 4


def test_approx_equal():
    """utils.rs:226-232"""
    assert po.approx_equal(0., 0., 0.)
    assert po.approx_equal(1., 0., 1.)
    assert po.approx_equal(-0.1, -0.2, 0.3)
    assert not po.approx_equal(1., 0., 0.5)
    assert not po.approx_equal(-1., 1., 1.)


def test_check_disparity_map():
    """utils.rs:235-247"""
    assert po.check_disparity_map([[0]])
    assert not po.check_disparity_map([[1, 1], [1, 1]])
    assert not po.check_disparity_map([[0, 2, 1]])
    assert po.check_disparity_map([[0, 0], [0, 1]])


def test_dedup_f64():
    """utils.rs:250-254"""
    assert po.dedup_f64([1., 1.5, 2., 3.], 1.) == [1., 3.]


def test_dedup_f64_one_element():
    """utils.rs:257-261"""
    assert po.dedup_f64([0.], 0.) == [0.]


def test_dedup_f64_no_remove():
    """utils.rs:264-268"""
    assert po.dedup_f64([1., 1.5, 2., 3.], 0.) == [1., 1.5, 2., 3.]


# ------------------------------------------------------- diffusion_graph.rs

def test_penalty_one_pixel():
    """diffusion_stereo/src/diffusion_graph.rs:542-548"""
    g = OracleGraph([[0]], [[0]], 1, 1.)
    assert g.penalty([[0]]) == 0.


def test_penalty_four_pixels_inf():
    """diffusion_graph.rs:551-557"""
    g = OracleGraph([[1, 1], [0, 0]], [[1, 0], [0, 0]], 2, 1.)
    assert g.penalty([[1, 1], [1, 1]]) == INF


def test_penalty_four_pixels():
    """diffusion_graph.rs:560-569"""
    g = OracleGraph([[1, 1], [0, 0]], [[1, 0], [0, 0]], 2, 1.)
    assert g.penalty([[0, 1], [0, 1]]) == 2.
    g.potentials[0][0][2][0] = 1.
    assert g.penalty([[0, 1], [0, 1]]) == 2.


def test_penalty():
    """diffusion_graph.rs:572-578"""
    g = OracleGraph([[1, 1]], [[1, 0]], 2, 1.)
    assert g.penalty([[0, 2]]) == INF


def test_sum_of_potentials():
    """diffusion_graph.rs:581-596"""
    g = OracleGraph([[1, 9], [5, 6]], [[6, 3], [6, 6]], 2, 1.)
    g.potentials[0][0][2][0] = 9.4
    g.potentials[0][0][3][0] = -1.8
    g.potentials[0][1][0][0] = 6.7
    g.potentials[0][1][0][1] = -1.4
    g.potentials[1][0][1][0] = 4.1
    assert approx(7.6, g.sum_of_potentials(0, 0, 0))
    assert approx(6.7, g.sum_of_potentials(0, 1, 0))
    assert approx(-1.4, g.sum_of_potentials(0, 1, 1))
    assert approx(4.1, g.sum_of_potentials(1, 0, 0))
    assert approx(0., g.sum_of_potentials(1, 1, 0))
    assert approx(0., g.sum_of_potentials(1, 1, 1))


def test_vertex_penalty_with_potentials():
    """diffusion_graph.rs:599-615"""
    g = OracleGraph([[166, 26, 215], [52, 66, 27], [113, 33, 214]],
                    [[203, 179, 158], [123, 160, 222], [90, 139, 127]], 2, 1.)
    g.potentials[1][1][0][0] = 0.
    g.potentials[1][1][1][0] = 0.
    g.potentials[1][1][2][0] = 1.
    g.potentials[1][1][3][0] = 0.1
    g.potentials[1][1][0][1] = 0.9
    g.potentials[1][1][1][1] = 0.6
    g.potentials[1][1][2][1] = 1.
    g.potentials[1][1][3][1] = 0.6
    assert approx(92.9, g.vertex_penalty_with_potentials(1, 1, 0))
    assert approx(53.9, g.vertex_penalty_with_potentials(1, 1, 1))


def _graph_1x2():
    g = OracleGraph([[244, 172]], [[168, 83]], 2, 1.)
    g.potentials[0][0][2][0] = 0.8
    g.potentials[0][1][0][0] = 0.
    g.potentials[0][1][0][1] = 0.1
    return g


def test_edge_penalty_with_potentials():
    """diffusion_graph.rs:618-633"""
    g = _graph_1x2()
    assert approx(0.8, g.edge_penalty_with_potential(0, 0, 2, 0, 0))
    assert approx(0.8, g.edge_penalty_with_potential(0, 1, 0, 0, 0))
    assert approx(1.9, g.edge_penalty_with_potential(0, 0, 2, 0, 1))
    assert approx(1.9, g.edge_penalty_with_potential(0, 1, 0, 1, 0))


def test_edge_exists():
    """diffusion_graph.rs:636-652"""
    g = OracleGraph([[244, 172, 168, 192], [83, 248, 38, 204]],
                    [[218, 138, 65, 18], [225, 7, 114, 127]], 4, 1.)
    assert g.edge_exists(0, 0, 2, 0, 0)
    assert g.edge_exists(0, 1, 0, 0, 0)
    assert g.edge_exists(0, 0, 2, 0, 1)
    assert g.edge_exists(0, 1, 0, 1, 0)
    assert not g.edge_exists(0, 0, 2, 1, 0)
    assert not g.edge_exists(0, 0, 2, 1, 1)
    assert not g.edge_exists(0, 0, 2, 2, 0)
    assert not g.edge_exists(0, 1, 0, 2, 1)
    assert not g.edge_exists(0, 0, 0, 0, 0)
    assert not g.edge_exists(0, 3, 0, 3, 0)
    assert not g.edge_exists(0, 2, 2, 0, 3)
    assert not g.edge_exists(0, 0, 2, 1, 0)


def test_min_edge_between_neighbors():
    """diffusion_graph.rs:655-668"""
    g = _graph_1x2()
    assert approx(g.min_edge_between_neighbors(0, 0, 2, 0), 0.8)
    g.potentials[0][1][0][0] = 2.
    assert approx(g.min_edge_between_neighbors(0, 0, 2, 0), 1.9)
    assert g.min_edge_between_neighbors(0, 0, 2, 1) == INF


def test_update_vertex_potential():
    """diffusion_graph.rs:671-695 (dummy_potentials start at 0: the write buffer is
    not the read buffer, so expected values are 0 - min_edge)."""
    g = OracleGraph([[25, 50], [5, 145], [248, 62]], [[39, 15], [77, 145], [31, 71]], 2, 1.)
    p = g.potentials
    p[0][0][3][0] = -0.3
    p[1][0][1][0] = 2.06
    p[1][0][2][0] = 2.06
    p[1][0][2][1] = 2.06
    p[1][0][3][0] = 2.06
    p[2][0][1][0] = 0.44
    p[1][1][0][0] = -1.42
    p[1][1][1][0] = -1.42
    p[1][1][0][1] = -0.62
    p[1][1][1][1] = -0.62
    p[1][1][3][1] = -0.62
    assert g.dummy_potentials[1][0][1][0] == 0.
    g.update_vertex_potential(1, 0, 1, 0)
    assert approx(g.dummy_potentials[1][0][1][0], -1.76)
    assert g.dummy_potentials[1][0][2][0] == 0.
    g.update_vertex_potential(1, 0, 2, 0)
    assert approx(g.dummy_potentials[1][0][2][0], -0.64)
    assert g.dummy_potentials[1][0][3][0] == 0.
    g.update_vertex_potential(1, 0, 3, 0)
    assert approx(g.dummy_potentials[1][0][3][0], -2.5)


def test_update_edge_potential():
    """diffusion_graph.rs:698-713"""
    img = [[1, 1, 1], [1, 0, 1], [1, 1, 1]]
    g = OracleGraph(img, img, 1, 1.)
    g.potentials[0][0][2][0] = -0.3
    g.potentials[0][0][3][0] = 0.1
    g.potentials[2][1][0][0] = 0.8
    g.potentials[2][1][2][0] = -0.62
    assert g.dummy_potentials[0][0][2][0] == 0.
    g.update_edge_potential(0, 0, 2, 0)
    assert approx(g.dummy_potentials[0][0][2][0], 0.1)
    g.update_edge_potential(1, 1, 0, 0)
    assert approx(g.dummy_potentials[1][1][0][0], 0.)
    g.update_edge_potential(2, 1, 2, 0)
    assert approx(g.dummy_potentials[2][1][2][0], -0.06)


def _graph_2x1():
    g = OracleGraph([[1], [0]], [[1], [0]], 2, 1.)
    g.potentials[0][0][3][0] = -0.3
    g.potentials[1][0][1][0] = 0.1
    assert g.dummy_potentials[0][0][3][0] == 0.
    return g


def test_diffusion_act_vertexes():
    """diffusion_graph.rs:716-729"""
    g = _graph_2x1()
    g.diffusion_act_vertexes(0, 0)
    assert approx(g.dummy_potentials[0][0][3][0], 0.2)
    assert approx(g.potentials[0][0][3][0], 0.2)
    g.diffusion_act_vertexes(1, 0)
    assert approx(g.dummy_potentials[1][0][1][0], -0.3)
    assert approx(g.potentials[1][0][1][0], -0.3)


def test_diffusion_act_edges():
    """diffusion_graph.rs:732-745"""
    g = _graph_2x1()
    g.diffusion_act_edges(0, 0)
    assert approx(g.dummy_potentials[0][0][3][0], 0.3)
    assert approx(g.potentials[0][0][3][0], 0.3)
    g.diffusion_act_edges(1, 0)
    assert approx(g.dummy_potentials[1][0][1][0], -0.1)
    assert approx(g.potentials[1][0][1][0], -0.1)


def test_diffusion_act():
    """diffusion_graph.rs:748-760"""
    g = _graph_2x1()
    g.diffusion_act()
    assert approx(g.dummy_potentials[0][0][3][0], 0.)
    assert approx(g.potentials[0][0][3][0], 0.)
    assert approx(g.dummy_potentials[1][0][1][0], 0.)
    assert approx(g.potentials[1][0][1][0], 0.)


def _graph_2x2_energy():
    g = OracleGraph([[1, 1], [0, 0]], [[1, 0], [0, 0]], 2, 1.)
    g.potentials[0][0][2][0] = 0.6
    g.potentials[0][0][3][0] = 358.
    g.potentials[0][1][0][0] = -13.7
    g.potentials[0][1][0][1] = 80.
    g.potentials[0][1][3][1] = -1E9
    return g


def test_min_penalty_vertex():
    """diffusion_graph.rs:763-776 (exact equality in the reference)"""
    g = _graph_2x2_energy()
    assert g.min_penalty_vertex(0, 0) == (0, -358.6)
    assert g.min_penalty_vertex(0, 1) == (0, 14.7)


def test_min_penalty_edge():
    """diffusion_graph.rs:779-790 (exact equality in the reference)"""
    g = _graph_2x2_energy()
    assert g.min_penalty_edge(0, 0, 2) == -13.1
    assert g.min_penalty_edge(0, 1, 0) == -13.1


def test_energy():
    """diffusion_graph.rs:793-806 (exact equality in the reference)"""
    g = _graph_2x2_energy()
    assert g.energy() == -999999999.0
    assert g.potentials[0][0][1][0] == 0.
    assert g.potentials[0][0][1][1] == 0.
    assert g.potentials[0][1][1][1] == 0.


# --------------------------------------------------- crossing_out_graph.rs

def _co(left, right, D, vfill, efill):
    g = OracleGraph(left, right, D, 1.)
    g.co_alloc(True)
    g.vertices[...] = 1 if vfill else 0
    g.edges[...] = 1 if efill else 0
    return g


def test_min_penalty_zero():
    """diffusion_stereo/src/crossing_out_graph.rs:368-380"""
    g = _co([[0, 0]], [[0, 0]], 2, True, True)
    assert g.min_penalty_edge(0, 0, 2) == 0.
    assert g.min_penalty_vertex(0, 0)[1] == 0.


def test_empty_crossing_out_graph():
    """crossing_out_graph.rs:383-399"""
    g = _co([[0, 0]], [[0, 0]], 1, True, False)
    g.vertices[0][0][0] = 0
    g.edges[0][0][0][2][0] = 1
    g.crossing_out()
    assert not g.vertices_exist()
    assert not g.edges_exist()
    assert not g.is_not_empty()


def test_crossing_out():
    """crossing_out_graph.rs:402-428"""
    g = _co([[1, 1], [1, 0]], [[1, 0], [0, 0]], 2, True, True)
    e, v = g.edges, g.vertices
    e[0][0][0][2][0] = 0
    e[0][1][0][0][0] = 0
    e[0][1][1][3][1] = 0
    e[1][1][1][1][1] = 0
    g.crossing_out()
    assert v[0][0][0]
    assert v[0][1][1]
    assert v[1][0][0]
    assert v[1][1][0]
    assert not v[0][1][0]
    assert not v[1][1][1]
    assert e[0][0][0][2][1]
    assert e[0][0][0][3][0]
    assert e[1][0][0][2][0]
    assert not e[1][0][0][2][1]
    assert e[0][1][1][3][0]
    assert not e[0][1][1][3][1]
    assert not e[0][0][0][2][0]


def test_initialize_vertices():
    """crossing_out_graph.rs:431-446"""
    g = _co([[1, 1]], [[1, 0]], 2, False, False)
    g.potentials[0][0][2][0] = 0.6
    g.potentials[0][1][0][0] = -13.7
    g.potentials[0][1][0][1] = 80.
    g.initialize_vertices(1.)
    assert g.vertices[0][0][0]
    assert not g.vertices[0][1][0]
    assert g.vertices[0][1][1]


def test_initialize_edges():
    """crossing_out_graph.rs:449-471"""
    g = _co([[1, 1, 0]], [[1, 0, 0]], 2, False, False)
    p = g.potentials
    p[0][0][2][0] = 0.6
    p[0][1][0][0] = -13.7
    p[0][1][0][1] = 80.
    p[0][1][2][0] = 358.
    p[0][1][2][1] = -1E9
    p[0][2][0][0] = -0.3
    p[0][2][0][1] = 0.1
    g.initialize_edges(0.1)
    e = g.edges
    assert e[0][0][0][2][0]
    assert not e[0][0][0][2][1]
    assert not e[0][1][0][2][0]
    assert not e[0][1][0][2][1]
    assert not e[0][1][1][2][0]
    assert e[0][1][1][2][1]


def test_initialize_with_epsilon():
    """crossing_out_graph.rs:474-491"""
    g = _co([[1, 1]], [[1, 0]], 2, False, False)
    g.potentials[0][0][2][0] = 0.6
    g.potentials[0][1][0][0] = -0.3
    g.potentials[0][1][0][1] = 0.1
    g.initialize_with_epsilon(1.3)
    assert g.vertices[0][0][0]
    assert not g.vertices[0][1][0]
    assert g.vertices[0][1][1]
    assert g.edges[0][0][0][2][0]
    assert not g.edges[0][0][0][2][1]


def test_vertices_exist():
    """crossing_out_graph.rs:494-504"""
    g = _co([[1, 1]], [[1, 0]], 2, True, False)
    g.vertices[0][1][1] = 0
    assert g.vertices_exist()


def test_vertices_exist_negative():
    """crossing_out_graph.rs:507-517"""
    g = _co([[1, 1]], [[1, 0]], 1, True, False)
    g.vertices[0][0][0] = 0
    assert not g.vertices_exist()


def test_edge_exist():
    """crossing_out_graph.rs:520-540"""
    g = _co([[1, 1], [1, 0]], [[1, 0], [0, 0]], 2, True, True)
    e = g.edges
    e[0][0][0][2][0] = 0
    e[0][1][0][0][0] = 0
    e[0][1][0][3][1] = 0
    e[1][1][1][1][0] = 0
    e[0][1][1][3][0] = 0
    e[1][1][0][1][1] = 0
    e[0][1][1][3][1] = 0
    e[1][1][1][1][1] = 0
    assert g.edges_exist()
    e[0][0][0][3][0] = 0
    e[1][0][0][1][0] = 0
    assert not g.edges_exist()


def test_is_not_empty():
    """crossing_out_graph.rs:543-562"""
    g = _co([[1, 1]], [[1, 0]], 2, True, True)
    g.vertices[0][1][0] = 0
    g.edges[0][0][0][2][1] = 0
    g.edges[0][1][1][0][0] = 0
    assert g.is_not_empty()
    g.vertices[0][1][0] = 1
    g.vertices[0][0][0] = 0
    assert not g.is_not_empty()
    g.vertices[0][0][0] = 1
    g.edges[0][0][0][2][0] = 0
    g.edges[0][1][0][0][0] = 0
    assert not g.is_not_empty()


def _graph_1x3(efill):
    g = _co([[1, 1, 0]], [[1, 0, 0]], 3, True, efill)
    p = g.potentials
    p[0][0][2][0] = 0.6
    p[0][1][0][0] = -13.7
    p[0][1][0][1] = 80.
    p[0][1][2][0] = 358.
    p[0][1][2][1] = -1E9
    p[0][2][0][0] = -0.3
    p[0][2][0][1] = 0.1
    p[0][2][0][2] = 0.8
    return g


def test_min_vertex_between_existing():
    """crossing_out_graph.rs:565-586"""
    g = _graph_1x3(False)
    assert g.min_vertex_between_existing(0, 0) == 0
    g.vertices[0][1][1] = 0
    assert g.min_vertex_between_existing(0, 1) == 0
    g.vertices[0][2][1] = 0
    assert g.min_vertex_between_existing(0, 2) == 2


def test_min_vertex_between_existing_empty():
    """crossing_out_graph.rs:589-599 (#[should_panic], max_disparity = 0)"""
    g = OracleGraph([[1]], [[1]], 0, 1.)
    g.co_alloc(True)
    with pytest.raises(AssertionError):
        g.min_vertex_between_existing(0, 0)


def test_cross_vertex_that_is_crossed():
    """crossing_out_graph.rs:602-613 (#[should_panic])"""
    g = _co([[1, 1]], [[1, 0]], 2, True, False)
    g.vertices[0][1][0] = 0
    with pytest.raises(AssertionError):
        g.cross_vertex(0, 1, 0)


def test_cross_vertex_one_disparity():
    """crossing_out_graph.rs:616-626"""
    g = _co([[1]], [[1]], 1, True, False)
    g.cross_vertex(0, 0, 0)
    assert g.vertices[0][0][0]


def test_cross_vertex_two_disparities():
    """crossing_out_graph.rs:629-640"""
    g = _co([[1, 1]], [[1, 0]], 2, True, False)
    g.cross_vertex(0, 1, 0)
    assert g.vertices[0][1][0]
    assert not g.vertices[0][1][1]


def test_find_best_labeling():
    """crossing_out_graph.rs:643-663"""
    g = _graph_1x3(True)
    m = g.find_best_labeling()
    assert m[0][0] == 0
    assert m[0][1] == 0
    assert m[0][2] == 1


# ---------------------------------------------------------- pgm_handler.rs

def test_pgm_reader(tmp_path):
    """diffusion_stereo/src/pgm_handler.rs:145-171"""
    p = tmp_path / "image.pgm"
    p.write_text("P2\n3 2\n255\n1 2 3\n4 5 6\n")
    m, w, h = po.pgm_reader(str(p))
    assert (w, h) == (3, 2)
    assert m.tolist() == [[1, 2, 3], [4, 5, 6]]


def test_pgm_writer(tmp_path):
    """pgm_handler.rs:174-200"""
    p = tmp_path / "image.pgm"
    po.pgm_writer([[1, 2, 3], [4, 5, 6]], str(p), 255)
    assert p.read_text() == "P2\n3 2\n255\n1 2 3\n4 5 6\n"
    m, w, h = po.pgm_reader(str(p))
    assert (w, h) == (3, 2)
    assert m.tolist() == [[1, 2, 3], [4, 5, 6]]


def test_pgm_reader_rejects_wrong_count(tmp_path):
    """pgm_handler.rs:65 (assert_eq!(width * height, array.len()))"""
    p = tmp_path / "bad.pgm"
    p.write_text("P2\n3 2\n255\n1 2 3\n4 5\n")
    with pytest.raises(IOError):
        po.pgm_reader(str(p))


# ------------------------------------------------------- epsilon_search.rs

def test_epsilon_search_zero():
    """diffusion_stereo/src/epsilon_search.rs:115-128"""
    g = po.OracleGraph(np.zeros((1, 2), np.uint32), np.zeros((1, 2), np.uint32), 2, 1.0)
    g.co_alloc(True)
    array = g.create_array_of_epsilons(0.01)
    assert g.epsilon_search(array) == 0.0


def test_epsilon_search():
    """epsilon_search.rs:131-145"""
    g = po.OracleGraph([[1, 1]], [[1, 0]], 2, 1.0)
    g.potentials[0][0][2][0] = 9.4
    g.potentials[0][1][0][0] = -1.8
    g.potentials[0][1][0][1] = 6.7
    g.co_alloc(True)
    assert g.epsilon_search([0., 1., 2., 3., 4., 5., 6., 7., 8., 9.]) == 9.0


def test_create_array_of_epsilons():
    """epsilon_search.rs:148-170"""
    g = po.OracleGraph([[1, 1]], [[1, 0]], 2, 1.0)
    g.potentials[0][0][2][0] = 0.6
    g.potentials[0][1][0][0] = -0.3
    g.potentials[0][1][0][1] = 0.1
    g.co_alloc(True)
    array = g.create_array_of_epsilons(0.0)
    assert len(array) == 3
    assert array[0] == 0.0
    assert po.approx_equal(1.4, array[1], 1e-6)
    assert po.approx_equal(1.4, array[2], 1e-6)
    array = g.create_array_of_epsilons(1.5)
    assert len(array) == 2
    assert array[0] == 0.0
    assert po.approx_equal(1.4, array[1], 1e-6)


# ------------------------------------------------------------ f32 variant

def test_f32_variant_agrees_on_a_kat():
    """The f32 instantiation follows the same code; diffusion_graph.rs:748-760."""
    g = OracleGraph([[1], [0]], [[1], [0]], 2, 1., dtype="f32")
    g.potentials[0][0][3][0] = -0.3
    g.potentials[1][0][1][0] = 0.1
    g.diffusion_act()
    assert abs(float(g.potentials[0][0][3][0])) < 1e-6
    assert abs(float(g.potentials[1][0][1][0])) < 1e-6
    assert not math.isnan(g.energy())
