"""Host-side pieces of stereo-vision_amd/pipeline.py that need no GPU: the PGM reader and
writer against the reference's own known answers (diffusion_stereo/src/pgm_handler.rs:144-200)
and the consistency check (utils.rs:121-130, tests :249-269)."""
import numpy as np
import pytest

from __graft_entry__ import load_package

load_package()
from stereo_vision_amd import pipeline  # noqa: E402


def test_pgm_reader(tmp_path):
    """pgm_handler.rs:145-171"""
    p = tmp_path / "image.pgm"
    p.write_text("P2\n3 2\n255\n1 2 3\n4 5 6\n")
    m, maxval = pipeline.pgm_reader(str(p))
    assert maxval == 255 and m.dtype == np.uint32
    assert m.tolist() == [[1, 2, 3], [4, 5, 6]]


def test_pgm_writer(tmp_path):
    """pgm_handler.rs:174-200"""
    p = tmp_path / "image.pgm"
    pipeline.pgm_writer([[1, 2, 3], [4, 5, 6]], str(p), 255)
    assert p.read_text() == "P2\n3 2\n255\n1 2 3\n4 5 6\n"
    assert pipeline.pgm_reader(str(p))[0].tolist() == [[1, 2, 3], [4, 5, 6]]


def test_pgm_reader_rejects_wrong_count(tmp_path):
    """pgm_handler.rs:65 (assert_eq!(width * height, array.len()))"""
    p = tmp_path / "bad.pgm"
    p.write_text("P2\n3 2\n255\n1 2 3\n4 5\n")
    with pytest.raises(ValueError):
        pipeline.pgm_reader(str(p))


def test_bundled_images_load(golden_dir):
    left, maxval = pipeline.pgm_reader(golden_dir + "/left.pgm")
    assert left.shape == (5, 5) and int(left.max()) <= maxval


def test_check_disparity_map():
    """utils.rs:249-269"""
    assert pipeline.check_disparity_map([[0, 0], [0, 1]])
    assert pipeline.check_disparity_map([[0, 1, 2], [2, 1, 0]])
    assert not pipeline.check_disparity_map([[0, 2, 1]])
