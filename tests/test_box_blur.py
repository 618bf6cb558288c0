"""DiffusionGraph::box_blur (diffusion_stereo/src/diffusion_graph.rs:399-424).

The reference has no test for it (the function is never called), so the restatement is
"parity unpinned" by reference fixtures: it is pinned here by values worked out by hand from
the source text, under the arithmetic of a release build (usize subtraction wraps; a debug
build panics on the same inputs).  The GPU entry point ds_box_blur is then compared with the
restatement, exactly (integer work)."""
import numpy as np
import pytest

from oracle import pyoracle

MAP = np.array([[0, 1, 2, 3], [1, 1, 2, 5], [4, 4, 4, 4], [0, 0, 9, 9]], dtype=np.uint32)


def test_restatement_by_hand():
    out = pyoracle.box_blur(MAP, 1, 1, 1)
    # rows / columns below the radius: the window's start wraps, the window is empty
    assert out[0].tolist() == [0, 255, 510, 765] and out[:, 0].tolist() == [0, 255, 1020, 0]
    # (1,1): window {(0,0),(0,1),(1,0),(1,1)}; 0 < centre 1 never counts: (1+1+1+1)*255/4
    assert out[1, 1] == 255
    # (2,3): centre 4; 2 is smaller (wraps), 5 is within the threshold: (4+5+4+4)*255/4
    assert out[2, 3] == 1083
    # (1,3): centre 5; window values 2, 3, 2, 5: only itself counts: (5+5)*255/2
    assert out[1, 3] == 1275
    # the normalisation term divides; a radius of 0 gives an empty window everywhere
    assert pyoracle.box_blur(MAP, 9, 0, 5).tolist() == (MAP.astype(np.uint64) * 255 // 5).tolist()
    with pytest.raises(ZeroDivisionError):
        pyoracle.box_blur(MAP, 1, 1, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,D,thr,r,norm", [(4, 4, 10, 1, 1, 1), (37, 61, 40, 3, 2, 40), (64, 300, 256, 0, 5, 256),
                                             (9, 5, 7, 100, 20, 3), (50, 50, 128, 2, 0, 1)])
def test_gpu_box_blur_equals_restatement(H, W, D, thr, r, norm):
    from __graft_entry__ import load_package
    load_package()
    from stereo_vision_amd import diffstereo
    rng = np.random.default_rng(H * W + r)
    m = MAP if (H, W) == (4, 4) else rng.integers(0, D, size=(H, W), dtype=np.uint32)
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    g = diffstereo.DiffusionGraph(img, img, D, 1.2)
    assert np.array_equal(g.box_blur(m, thr, r, norm), pyoracle.box_blur(m, thr, r, norm))
    with pytest.raises(diffstereo.DiffStereoError):
        g.box_blur(m, thr, r, 0)
    g.close()


@pytest.mark.gpu
def test_f32_handles_refuse_an_epsilon_they_cannot_resolve():
    from __graft_entry__ import load_package
    load_package()
    from stereo_vision_amd import diffstereo
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(6, 9), dtype=np.uint8)
    for dtype, ok in (("f32", False), ("f64", True)):
        g = diffstereo.DiffusionGraph(img, img[::-1].copy(), 8, 1.2, dtype=dtype)
        g.iterate(2)
        if ok:
            assert g.crossing_out(1e-8)["passes"] >= 1
        else:
            with pytest.raises(diffstereo.DiffStereoError) as err:
                g.crossing_out(1e-8)
            assert err.value.code == -1 and "f64" in str(err.value)
            with pytest.raises(diffstereo.DiffStereoError):
                g.co_begin(1e-8)
        assert g.crossing_out(1e-3)["passes"] >= 1  # 2^-23 * (255 + 1.2 * 7) = 3.1e-5
        g.close()
