"""oracle/_ref: the files of the reference tree that compile standalone (no OpenCV / Eigen), built from the sources under
/root/reference by `make -C oracle _ref`, and the oracle's restatements pinned against the reference's own code:
  * add_src/lineIterator.cpp - the Bresenham walk behind Frame::AssignFeaturesToGridForLine (src/Frame.cc:286-309, row a22);
  * Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240 - nfa() / log_gamma(), the in-tree TWIN of the arithmetic inside the
    LSD the reference links (OpenCV's lsd.cpp, not in the tree; row a10).  The twin multiplies by a tabulated 1/i where lsd.cpp
    divides by i: log_gamma and every value that needs no series are compared bit for bit; of the summed tails 98.7 % are
    bit-identical too, the rest differ in the last bits, and a handful stop the series one term apart (the truncation test sits on
    that last bit): then the values differ by up to ~4e-5 relative, far inside the series' own 10 % error budget; no accept / reject
    decision (value > 0) differs;
  * Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-84 + FeatureVector.cpp:31-45 - the accumulation Frame::ComputeBoW links (row f2).
Skipped where oracle/_ref/ is absent (it cannot be built without /root/reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_lineiterator.so")
REF_NFA_SO = os.path.join(ROOT, "oracle", "_ref", "libref_nfa.so")
REF_DBOW2_SO = os.path.join(ROOT, "oracle", "_ref", "libref_dbow2.so")


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


@pytest.mark.skipif(not os.path.exists(REF_NFA_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_log_gamma_and_nfa_equal_the_reference_twin():
    ref = C.CDLL(REF_NFA_SO)
    orc = oracle_lib.load()
    for f in (ref.ref_log_gamma, orc.pso_lsd_log_gamma):
        f.argtypes, f.restype = [C.c_double], C.c_double
    for f in (ref.ref_nfa, orc.pso_lsd_nfa_lognt):
        f.argtypes, f.restype = [C.c_int, C.c_int, C.c_double, C.c_double], C.c_double
    old = orc.pso_set_nfa_math(0)   # the host's libm on both sides: what the reference calls
    try:
        # log_gamma: both branches (Lanczos x <= 15, Windschitl above), every integer argument nfa() can pass for a 640x480 ... 1280x960 image
        xs = list(range(1, 3000)) + list(range(3000, 400000, 37)) + [15.0, 15.5, 16.0, 0.5, 2.25, 1e6 + 1]
        for x in xs:
            a, b = ref.ref_log_gamma(float(x)), orc.pso_lsd_log_gamma(float(x))
            assert a == b, (x, a, b)
        # nfa(n, k, p, logNT): the (n, k, p) ranges rect_improve produces - rectangle pixel counts up to a few thousand, aligned counts 0..n,
        # p = 1/8 halved up to five times (LSD_REFINE_ADV), logNT of the scaled 512x384 and 1024x768 images
        rng = np.random.default_rng(3)
        lognts = [5 * (np.log10(512.0) + np.log10(384.0)) / 2 + np.log10(11.0), 5 * (np.log10(1024.0) + np.log10(768.0)) / 2 + np.log10(11.0)]
        cases = [(0, 0), (1, 0), (1, 1), (7, 7), (10, 3), (100, 100), (100, 0), (5000, 1), (5000, 4999)]
        for _ in range(60000):
            n = int(rng.integers(1, 9000)) if rng.random() < 0.7 else int(rng.integers(1, 200))
            u = rng.random()
            k = int(rng.integers(0, n + 1)) if u < 0.4 else int(min(n, max(0, round(n * rng.uniform(0.0, 0.45)))))
            cases.append((n, k))
        exact = series = flips = far = 0
        worst = 0.0
        for i, (n, k) in enumerate(cases):
            p = 0.125 / (1 << (i % 6))
            lnt = lognts[i & 1]
            a, b = ref.ref_nfa(n, k, p, lnt), orc.pso_lsd_nfa_lognt(n, k, p, lnt)
            if a == b:
                exact += 1
            else:   # only the summed tail may differ: (n-i+1) * (1/i) in the twin, (n-i+1) / i in lsd.cpp
                series += 1
                err = abs(a - b) / max(abs(a), abs(b), 1e-300)
                worst = max(worst, err)
                far += int(err > 1e-11)   # the series stopped one term apart
                assert err < 1e-3, (n, k, p, a, b)
            flips += int((a > 0) != (b > 0))
        assert flips == 0 and exact > 0.98 * len(cases) and far < 1e-3 * len(cases), (flips, exact, series, far, worst)
        # the cases that need no series are bit-identical by construction; check that family explicitly
        for n, k, p in ((0, 0, 0.125), (50, 0, 0.125), (64, 64, 0.0625), (3000, 2900, 0.125), (4000, 5, 0.125)):
            assert ref.ref_nfa(n, k, p, lognts[0]) == orc.pso_lsd_nfa_lognt(n, k, p, lognts[0]), (n, k, p)
    finally:
        orc.pso_set_nfa_math(old)


@pytest.mark.skipif(not os.path.exists(REF_DBOW2_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_bow_accumulation_equals_the_reference_dbow2():
    """The (word, weight, node) stream the oracle's tree descent produces, accumulated by the reference's own BowVector::addWeight /
    normalize(L1) / FeatureVector::addFeature, gives the oracle's BowVector (f64 values bit for bit) and FeatureVector."""
    import bow_vocab
    ref = C.CDLL(REF_DBOW2_SO)
    ref.ref_bow_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]
    rng = np.random.default_rng(12)
    checked = 0
    for seed, (k, L, ragged, n, levelsup) in enumerate([(10, 3, False, 1000, 2), (6, 4, True, 700, 2), (10, 3, False, 0, 4), (3, 5, True, 1500, 4),
                                                           (10, 2, False, 2000, 1), (4, 3, False, 5, 3)]):
        vocab = bow_vocab.make_vocab(k, L, seed, ragged, stopped=0.1)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if n > 10:
            desc[n // 2:] = desc[: n - n // 2]   # repeated words: addWeight's accumulate branch
        o = oracle_lib.compute_bow(*vocab[:4], vocab[4], desc, levelsup)
        m = max(n, 1)
        bow_id, bow_val = np.zeros(m, np.int32), np.zeros(m, np.float64)
        fv_node, fv_start, fv_idx = np.zeros(m, np.int32), np.zeros(m + 1, np.int32), np.zeros(m, np.int32)
        nf = C.c_int()
        w, wt, nid = (np.ascontiguousarray(o[key]) for key in ("word", "weight", "nid"))
        nb = ref.ref_bow_accumulate(w.ctypes.data, wt.ctypes.data, nid.ctypes.data, n, bow_id.ctypes.data, bow_val.ctypes.data, fv_node.ctypes.data,
                                    fv_start.ctypes.data, fv_idx.ctypes.data, C.byref(nf))
        assert nb == len(o["bow_id"]) and nf.value == len(o["fv_node"])
        assert np.array_equal(bow_id[:nb], o["bow_id"]) and bow_val[:nb].tobytes() == o["bow_val"].tobytes()
        assert np.array_equal(fv_node[:nf.value], o["fv_node"]) and np.array_equal(fv_start[:nf.value + 1], o["fv_start"])
        assert np.array_equal(fv_idx[:fv_start[nf.value]], o["fv_idx"])
        checked += nb
    assert checked > 400
