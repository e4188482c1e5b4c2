"""GPU parity of the KeyFrame-rate matchers through the C ABI vs oracle/kf_oracle.cpp: the candidate loop
of both ORBmatcher::Fuse overloads, SearchBySim3, SearchForTriangulation, LSDmatcher::Fuse / SearchForTriangulation and
ComputeDistinctiveDescriptors.  All comparisons are exact (integer / index results)."""
import numpy as np
import pytest

import kf_scene as ks

pytestmark = pytest.mark.gpu


def _grid(P, k, d, uright=None, slots=1):
    g = P.FrameGrid(2048, slots)
    g.set(0, k, d, ks.BOUNDS, uright=uright)
    return g


@pytest.mark.parametrize("chi2", [False, True])
@pytest.mark.parametrize("th", [3.0, 8.0])
def test_fuse_window_best(chi2, th):
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(5)
    q = ks.proj_queries(k1, rng, th=th, jitter=1.5 if chi2 else 3.0)
    qd = ks.noisy_desc(d1, rng)
    ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(5, 60, len(k1)), -1).astype(np.float32)
    g = _grid(P, k1, d1, ur)
    kf = P.KeyFrameMatcher()
    bi, bd = kf.window_best(g, 0, q, qd, chi2, ks.INV_SIGMA2)
    rbi, rbd = oracle_lib.window_best(k1, d1, ur, ks.BOUNDS, q, qd, chi2, ks.INV_SIGMA2)
    np.testing.assert_array_equal(bi, rbi)
    np.testing.assert_array_equal(bd, rbd)
    assert (bi[q["radius"] < 0] == -1).all() and (bi >= 0).sum() > 200
    idx, fused = (kf.Fuse(g, 0, q, qd, ks.INV_SIGMA2) if chi2 else kf.FuseSim3(g, 0, q, qd))
    np.testing.assert_array_equal(fused, rbd <= 50)


def test_fuse_more_points_than_keypoints_and_empty():
    """vpMapPoints of Fuse is a neighbourhood's map, usually larger than one keyframe's feature set"""
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(6)
    rep = np.concatenate([np.arange(len(k1))] * 6)
    q = ks.proj_queries(k1[rep], rng, th=4.0, jitter=4.0)
    qd = ks.noisy_desc(d1[rep], rng, flips=30)
    g = _grid(P, k1, d1)
    kf = P.KeyFrameMatcher()
    bi, bd = kf.window_best(g, 0, q, qd)
    rbi, rbd = oracle_lib.window_best(k1, d1, None, ks.BOUNDS, q, qd)
    np.testing.assert_array_equal(bi, rbi)
    np.testing.assert_array_equal(bd, rbd)
    bi, bd = kf.window_best(g, 0, q[:0], qd[:0])
    assert len(bi) == 0
    with pytest.raises(P.PslfeError):
        kf.window_best(g, 0, q, qd, True, None)  # chi2 gates without mvInvLevelSigma2


def test_search_by_sim3():
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(7)
    # map point i1 of KF1 "is" keypoint perm[i1] of KF2 for the pairs that exist in both; the rest project to random places
    n1, n2 = len(k0), len(k1)
    n = min(n1, n2)
    p12 = rng.permutation(n2)[:n1] if n2 >= n1 else np.concatenate([rng.permutation(n2), rng.integers(0, n2, n1 - n2)])
    q12 = ks.proj_queries(k1[p12], rng, th=7.5, jitter=2.0, p_drop=0.2)
    qd1 = ks.noisy_desc(d1[p12], rng, flips=20)       # descriptors of KF1's map points look like their KF2 partner
    inv = np.full(n2, -1)
    inv[p12] = np.arange(n1)
    src = np.where(inv >= 0, inv, rng.integers(0, n1, n2))
    q21 = ks.proj_queries(k0[src], rng, th=7.5, jitter=2.0, p_drop=0.2)
    qd2 = ks.noisy_desc(d0[src], rng, flips=20)
    g = P.FrameGrid(2048, 2)
    g.set(0, k0, d0, ks.BOUNDS)
    g.set(1, k1, d1, ks.BOUNDS)
    nf, m = P.KeyFrameMatcher().SearchBySim3(g, 0, g, 1, q12, qd1, q21, qd2)
    rnf, rm = oracle_lib.search_by_sim3(k0, d0, ks.BOUNDS, k1, d1, ks.BOUNDS, q12, qd1, q21, qd2)
    assert nf == rnf and nf > 100
    np.testing.assert_array_equal(m, rm)


@pytest.mark.parametrize("nnodes", [8, 60, 500])
@pytest.mark.parametrize("only_stereo,check_ori", [(False, True), (True, True), (False, False)])
def test_search_for_triangulation(nnodes, only_stereo, check_ori):
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(9)
    has1 = rng.random(len(k0)) < 0.4
    st1 = rng.random(len(k0)) < 0.5
    ur2 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 20, -1).astype(np.float32)
    taken2 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    # KF2 features resemble KF1's so that distances fall under TH_LOW: swap in noisy copies
    m = min(len(k0), len(k1))
    d2 = d1.copy()
    d2[:m] = ks.noisy_desc(d0[:m], rng, flips=25)
    fidx, q, qd, idx1 = ks.tri_inputs(k0, d0, has1, st1, ks.feature_vector(d0, nnodes), ks.feature_vector(d2, nnodes), only_stereo,
                                      oracle_lib.TRIQUERY_DTYPE)
    F12, epi = ks.fundamental(rng)
    F12 = (F12 * np.float32(rng.uniform(0.02, 0.2))).astype(np.float32)
    # a loose epipolar band so that several candidates per query survive: scale sigma2
    sig2 = (ks.SIGMA2 * np.float32(4000.0)).astype(np.float32)
    g = _grid(P, k1, d2, ur2)
    nm, match = P.KeyFrameMatcher().SearchForTriangulation(g, 0, fidx, taken2, q, qd, F12, epi, ks.SCALE, sig2, only_stereo, check_ori)
    rnm, rmatch = oracle_lib.search_for_triangulation(k1, d2, ur2, taken2, fidx, q, qd, F12, epi, ks.SCALE, sig2, only_stereo, check_ori)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    assert not taken2[match[match >= 0]].any()
    if nnodes <= 60 and not only_stereo:
        assert nm > 20


def _lines(n, rng):
    import psl_slam_amd as P
    kl = np.zeros(n, P.KEYLINE_DTYPE)
    sx, sy = rng.uniform(20, 620, n), rng.uniform(20, 460, n)
    ang = rng.choice([0.0, 0.01, np.pi / 2, 0.7, -0.7], n) + rng.normal(0, 0.01, n)
    ln = rng.uniform(20, 120, n)
    kl["startPointX"], kl["startPointY"] = sx, sy
    kl["endPointX"], kl["endPointY"] = sx + ln * np.cos(ang), sy + ln * np.sin(ang)
    kl["pt_x"] = (kl["startPointX"] + kl["endPointX"]) / 2
    kl["pt_y"] = (kl["startPointY"] + kl["endPointY"]) / 2
    kl["octave"] = rng.integers(0, 2, n)
    kl["lineLength"] = ln
    return kl


@pytest.mark.parametrize("ndesc_short", [False, True])
def test_line_fuse_best(ndesc_short):
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(13)
    kl = _lines(300, rng)
    desc = rng.integers(0, 256, (250 if ndesc_short else 300, 32), dtype=np.uint8)
    src = rng.integers(0, len(kl), 500)
    q = np.zeros(len(src), P.LINEFUSEQUERY_DTYPE)
    for a, b in (("x1", "startPointX"), ("y1", "startPointY"), ("x2", "endPointX"), ("y2", "endPointY")):
        q[a] = kl[b][src] + rng.normal(0, 1.0, len(src)).astype(np.float32)
    q["level"] = kl["octave"][src] + rng.integers(0, 2, len(src))
    q["radius"] = np.float32(30.0) * np.float32(1.2) ** q["level"]
    q["radius"][rng.random(len(src)) < 0.1] = -1
    q["x2"][5], q["y2"][5] = q["x1"][5], q["y1"][5]          # degenerate projection: NaN direction, CosSita test never rejects
    qd = ks.noisy_desc(desc[np.minimum(src, len(desc) - 1)], rng, flips=40)
    bi, bd = P.KeyFrameMatcher().LineFuse(kl, desc, q, qd)
    rbi, rbd = oracle_lib.line_fuse_best(kl, desc, q, qd)
    np.testing.assert_array_equal(bi, rbi)
    np.testing.assert_array_equal(bd, rbd)
    assert (bi >= 0).sum() > 200 and (bi < len(desc)).all()


def test_lsd_search_for_triangulation_is_mutual_frame_bf_match():
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(14)
    d1 = rng.integers(0, 256, (180, 32), dtype=np.uint8)
    d2 = np.concatenate([ks.noisy_desc(d1[:150], rng, flips=16)[rng.permutation(150)], rng.integers(0, 256, (40, 32), dtype=np.uint8)])
    has1 = rng.random(len(d1)) < 0.2
    has2 = rng.random(len(d2)) < 0.2
    lm = P.LSDmatcher(0.95, True)
    for TH, dbl in ((lm.TH_LOW, True), (lm.TH_HIGH, False)):
        n, pairs = lm.SearchForTriangulation(d1, d2, has1, has2, TH, dbl)
        m12 = oracle_lib.frame_bf_match(d1, d2, 0.95, TH)
        m21 = oracle_lib.frame_bf_match(d2, d1, 0.95, TH)
        ref = np.full(len(d1), -1, np.int32)
        for i, j in enumerate(m12):
            if j >= 0 and (not dbl or m21[j] == i) and not has1[i] and not has2[j]:
                ref[i] = j
        np.testing.assert_array_equal(pairs, ref)
        assert n == (ref >= 0).sum() and n > 50


def test_distinctive_descriptors():
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(15)
    sizes = [0, 1, 2, 3, 4, 5, 8, 17, 63, 64, 65, 130, 700, 1024] + list(rng.integers(1, 40, 300))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    desc = np.zeros((off[-1], 32), np.uint8)
    for p, s in enumerate(sizes):
        desc[off[p]:off[p + 1]] = ks.noisy_desc(np.repeat(base[p:p + 1], s, 0), rng, flips=40)
    desc[off[7]:off[8]] = desc[off[7]]                      # all equal: every median is 0, the first row wins
    kf = P.KeyFrameMatcher()
    best = kf.ComputeDistinctiveDescriptors(desc, off)
    ref = oracle_lib.distinctive_descriptors(desc, off)
    np.testing.assert_array_equal(best, ref)
    assert best[0] == -1 and best[7] == 0
    with pytest.raises(P.PslfeError):
        kf.ComputeDistinctiveDescriptors(np.zeros((1025, 32), np.uint8), [0, 1025])
