"""Deterministic inputs and synthetic quantizer states of the fixtures: the generators live in the package
(quantization_amd/synthetic.py, also used by bench.py) and are re-exported here for the fixture scripts."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from quantization_amd.synthetic import checksum, make_gaussian, make_kind, make_x, synthetic_state  # noqa: E402,F401
