"""Write the synthetic pair of stereo-vision_amd/synth.py as two ASCII PGM files.
usage: make_pgm_pair.py H W D LEFT.pgm RIGHT.pgm"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import pipeline, synth  # noqa: E402

H, W, D = (int(x) for x in sys.argv[1:4])
left, right, _ = synth.synthetic_pair(H, W, D)
pipeline.pgm_writer(left, sys.argv[4], 255)
pipeline.pgm_writer(right, sys.argv[5], 255)
