"""oracle/_ref: the one file of the reference's hot path that compiles standalone (add_src/lineIterator.cpp)
is built from the sources under /root/reference by `make -C oracle _ref`; the oracle's restatement of that Bresenham walk
(behind Frame::AssignFeaturesToGridForLine, src/Frame.cc:286-309, row a22) is pinned against the reference's own code.
Skipped where oracle/_ref/ is absent (it cannot be built without /root/reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_lineiterator.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_grid_walk_equals_the_reference_line_iterator():
    ref = C.CDLL(REF_SO)
    orc = oracle_lib.load()
    for L in (ref.ref_line_iterator_walk, orc.pso_line_iterator_walk):
        L.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_int]
    rng = np.random.default_rng(9)
    cases = [(0, 0, 63.9, 47.9), (10.5, 3.2, 10.5, 40.0), (5, 5, 5, 5), (63.2, 1.0, 0.4, 46.5), (-3.5, 10.0, 20.0, -8.0), (2.0, 2.0, 2.9, 2.1),
             (0.0, 47.99, 63.99, 0.0), (30.0, 10.0, 30.0, 10.0)]
    cases += [tuple(rng.uniform(-8, 72, 4) * np.array([1, 0.75, 1, 0.75])) for _ in range(20000)]
    # the grid coordinates Frame.cc feeds it: key-line end points times (64 / width, 48 / height), computed in float
    for _ in range(5000):
        p = rng.uniform(0, 640, 4).astype(np.float32) * np.array([1, 0.75, 1, 0.75], np.float32)
        inv = np.array([np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)] * 2, np.float32)
        cases.append(tuple(float(v) for v in p * inv))
    a = np.zeros((4096, 2), np.int32)
    b = np.zeros((4096, 2), np.int32)
    total = 0
    for c in cases:
        na = ref.ref_line_iterator_walk(*[float(v) for v in c], a.ctypes.data, 4096)
        nb = orc.pso_line_iterator_walk(*[float(v) for v in c], b.ctypes.data, 4096)
        assert na == nb and np.array_equal(a[:na], b[:nb]), c
        total += na
    assert total > 100000
