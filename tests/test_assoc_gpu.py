"""GPU parity (exact) of the small association routines: LSDmatcher::SearchByGeomNApearance,
LSDmatcher::FrameBFMatch (+ lineDescriptorMAD), Map::AssociatePlanesByBoundary (live) and
InsectLineMatch::SearchMapInsectline (dead upstream, H14)."""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu


def _two_frames(style="struct", seed=3):
    import oracle_lib
    sc = sf.Scene(640, 480, style, seed)
    return [oracle_lib.line_extract(sc.gray(t), 200) for t in (0, 1)]


@pytest.mark.parametrize("style", ["struct", "desk"])
def test_search_by_geom_appearance(style):
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0, _), (k1, d1, _) = _two_frames(style)
    rng = np.random.default_rng(0)
    has = (rng.random(len(k0)) < 0.8).astype(np.uint8)
    k1 = k1.copy()
    k1["startPointX"][:2] = 0.0  # the reference skips current lines whose startPointX == 0 (:64-65)
    bounds = (0.0, 640.0, 0.0, 480.0)
    for th in (0.95, 0.8):
        n, m12, asg = P.LSDmatcher().SearchByGeomNApearance(k0, d0, k1, d1, has, th, bounds)
        rn, rm12, rasg = oracle_lib.search_by_geom_appearance(k0, d0, k1, d1, has, th, bounds)
        assert n == rn and n > 5
        np.testing.assert_array_equal(m12, rm12)
        np.testing.assert_array_equal(asg, rasg)
    n, m12, asg = P.LSDmatcher().SearchByGeomNApearance(k0, d0, k1[:0], d1[:0], has, 0.95, bounds)
    assert n == 0 and (m12 == -1).all()


def test_frame_bf_match_and_mad():
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0, _), (k1, d1, _) = _two_frames()
    for ratio, TH in ((0.95, 80.0), (0.85, 60.0)):
        got = P.LSDmatcher(ratio).FrameBFMatch(d0, d1, TH)
        ref = oracle_lib.frame_bf_match(d0, d1, ratio, TH)
        np.testing.assert_array_equal(got, ref)
    assert (ref >= 0).sum() > 5
    assert (P.LSDmatcher().FrameBFMatch(d0, d1[:1], 80.0) == -1).all()


@pytest.mark.parametrize("live", [True, False])
def test_associate_planes(live):
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(4)
    M, N = 40, 12
    nrm = rng.standard_normal((M, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    mp = np.concatenate([nrm, rng.uniform(-3, 3, (M, 1))], 1).astype(np.float32)
    sel = rng.integers(0, M, N)
    planes = mp[sel].copy()
    planes[:, :3] += rng.normal(0, 0.01, (N, 3)).astype(np.float32)
    pts = np.zeros((N, 5, 3))
    for i in range(N):  # points close to the chosen map plane
        n3, d = mp[sel[i], :3].astype(np.float64), float(mp[sel[i], 3])
        p = rng.uniform(-2, 2, (5, 3))
        p -= ((p @ n3 + d)[:, None]) * n3[None, :]
        pts[i] = p + rng.normal(0, 0.01, (5, 3))
    bad = (rng.random(M) < 0.2).astype(np.uint8)
    n, assoc = P.associate_planes(planes, pts, mp, 0.05, 0.999, live=live, map_bad=bad)
    rn, rassoc = oracle_lib.associate_planes(planes, pts, mp, 0.05, 0.999, live, bad)
    assert n == rn
    np.testing.assert_array_equal(assoc, rassoc)
    assert (assoc >= 0).sum() >= 1
    n, assoc = P.associate_planes(planes[:0], pts[:0], mp, 0.05, 0.999, live=live)
    assert n == 0


def _line_queries(k0, rng, radius, th_cos, jitter=2.0, p_block=0.8):
    import psl_slam_amd as P
    q = np.zeros(len(k0), P.LINEQUERY_DTYPE)
    for a, b in (("x1", "startPointX"), ("y1", "startPointY"), ("x2", "endPointX"), ("y2", "endPointY")):
        q[a] = k0[b] + rng.uniform(-jitter, jitter, len(k0)).astype(np.float32)
    q["radius"], q["th_cos"] = radius, th_cos
    q["vx"] = k0["ePointInOctaveX"] - k0["sPointInOctaveX"]
    q["vy"] = k0["ePointInOctaveY"] - k0["sPointInOctaveY"]
    q["length"] = k0["lineLength"]
    q["blocks"] = (rng.random(len(k0)) < p_block).astype(np.int32)
    return q


@pytest.mark.parametrize("style", ["struct", "desk"])
def test_line_search_by_projection_last(style):
    """LSDmatcher::SearchByProjection(cur,last,th) incl. mGridForLine (Bresenham) and GetFeaturesInAreaForLine."""
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0, _), (k1, d1, e1) = _two_frames(style)
    rng = np.random.default_rng(7)
    bounds = (0.0, 0.0, 640.0, 480.0)
    q = _line_queries(k0, rng, 6.0, 0.96)
    taken = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    nm, match, asg, gs, gi = P.LSDmatcher().SearchByProjection(k1, d1, e1, bounds, q, d0, mode=0, taken=taken, want_grid=True)
    rgs, rgi = oracle_lib.line_grid_build(k1, bounds)
    np.testing.assert_array_equal(gs, rgs)
    np.testing.assert_array_equal(gi, rgi)
    rnm, rmatch, rasg = oracle_lib.line_search_by_projection(k1, d1, e1, bounds, q, d0, 0, None, taken)
    assert nm == rnm and nm > 5
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(asg, rasg)


def test_line_search_by_projection_map():
    """LSDmatcher::SearchByProjection(F, vpMapLines, ..): 3-D direction gate, best / second best ratio."""
    import psl_slam_amd as P
    import oracle_lib
    (k0, d0, _), (k1, d1, e1) = _two_frames("struct")
    rng = np.random.default_rng(9)
    bounds = (0.0, 0.0, 640.0, 480.0)
    q = _line_queries(k0, rng, 8.0, 0.998, jitter=1.0, p_block=1.0)
    dir3d = rng.standard_normal((len(k1), 3))
    # make most map-line normals agree with the frame line they should match (same index when it exists)
    w = rng.standard_normal((len(k0), 3))
    m = min(len(k0), len(k1))
    w[:m] = dir3d[:m] + rng.normal(0, 0.05, (m, 3))
    q["wdir"] = w
    for ratio in (0.95, 0.7):
        nm, match, asg = P.LSDmatcher(ratio).SearchByProjection(k1, d1, e1, bounds, q, d0, mode=1, dir3d=dir3d)
        rnm, rmatch, rasg = oracle_lib.line_search_by_projection(k1, d1, e1, bounds, q, d0, 1, dir3d, None, ratio)
        assert nm == rnm
        np.testing.assert_array_equal(match, rmatch)
        np.testing.assert_array_equal(asg, rasg)
