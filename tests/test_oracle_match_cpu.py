"""Oracle of the matching functions vs independent numpy restatements and the golden vectors."""
import os

import numpy as np

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BOUNDS = (0.0, 0.0, 640.0, 480.0)


def popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_swar_hamming_equals_popcount():
    rng = np.random.default_rng(0)
    for _ in range(500):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert oracle_lib.hamming256(a, b) == popcount_dist(a, b)
    z = np.zeros(32, np.uint8)
    assert oracle_lib.hamming256(z, ~z) == 256 and oracle_lib.hamming256(z, z) == 0


def test_knn2_ordering_and_ties():
    rng = np.random.default_rng(1)
    t = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t[7] = t[3]
    q[0] = t[3]
    idx, dist = oracle_lib.hamming_knn2(q, t)
    D = np.array([[popcount_dist(a, b) for b in t] for a in q])
    order = np.argsort(D, axis=1, kind="stable")  # stable: lower train index first on ties
    np.testing.assert_array_equal(idx, order[:, :2])
    np.testing.assert_array_equal(dist, np.take_along_axis(D, order[:, :2], 1))
    assert list(idx[0]) == [3, 7]
    n, m12 = oracle_lib.line_match_nnr(q, t, 0.9)
    exp = np.where(dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * np.float32(0.9), idx[:, 0], -1)
    np.testing.assert_array_equal(m12, exp)
    assert n == (exp >= 0).sum()


def test_grid_against_bruteforce():
    g = np.load(os.path.join(GOLD, "match_640x480_desk.npz"))
    k1 = g["k1"]
    start, idx = oracle_lib.grid_build(k1, BOUNDS)
    px = np.floor((k1["x"] - np.float32(0)) * np.float32(64.0 / 640.0) + np.float32(0.5)).astype(int)  # round half away (x >= 0)
    py = np.floor((k1["y"] - np.float32(0)) * np.float32(48.0 / 480.0) + np.float32(0.5)).astype(int)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    cell = px * 48 + py
    exp = [i for c in range(64 * 48) for i in np.flatnonzero(ok & (cell == c))]
    assert list(idx) == exp and start[-1] == len(exp)


def test_search_by_projection_golden_and_invariants():
    g = np.load(os.path.join(GOLD, "match_640x480_desk.npz"))
    nm, match, assigned = oracle_lib.search_by_projection_last(g["k1"], g["d1"], None, BOUNDS, g["queries"], g["qdesc"], None, True)
    assert nm == int(g["nmatches"])
    np.testing.assert_array_equal(match, g["match"])
    np.testing.assert_array_equal(assigned, g["assigned"])
    m = match[match >= 0]
    assert len(np.unique(m)) == len(m)            # every query blocks: a keypoint is given away once
    for qi in np.flatnonzero(match >= 0)[:200]:   # TH_HIGH gate and window gate
        c = match[qi]
        assert popcount_dist(g["qdesc"][qi], g["d1"][c]) <= 100
        assert abs(g["k1"]["x"][c] - g["queries"]["u"][qi]) < g["queries"]["radius"][qi]
        assert assigned[c] == qi
    idx, dist = oracle_lib.hamming_knn2(g["qdesc"][:200], g["d1"][:200])
    np.testing.assert_array_equal(idx, g["knn_idx"])
    np.testing.assert_array_equal(dist, g["knn_dist"])


def test_search_by_projection_sequential_semantics_small():
    """Hand-built case: two queries prefer the same keypoint; the later one must fall back to its
    second choice iff the earlier one blocks (src/ORBmatcher.cc:1401-1403)."""
    kps = np.zeros(3, oracle_lib.KEYPOINT_DTYPE)
    kps["x"], kps["y"], kps["octave"] = [100, 104, 300], [100, 100, 300], 0
    desc = np.zeros((3, 32), np.uint8)
    desc[1, 0] = 0b00000111                      # keypoint 1 is 3 bits from keypoint 0
    q = np.zeros(2, oracle_lib.PROJQUERY_DTYPE)
    q["u"], q["v"], q["radius"], q["min_level"], q["max_level"] = 102, 100, 15, -1, -1
    qd = np.zeros((2, 32), np.uint8)
    qd[1, 0] = 0b00000001
    for blocks, exp in ((1, [0, 1]), (0, [0, 0])):
        q["blocks"] = blocks
        nm, match, assigned = oracle_lib.search_by_projection_last(kps, desc, None, BOUNDS, q, qd, None, False)
        assert list(match) == exp and nm == 2
        assert assigned[0] == (0 if blocks else 1)


def test_frame_bf_match_against_numpy():
    """FrameBFMatch + lineDescriptorMAD (add_src/LSDmatcher.cpp:492-516, 660-685) re-derived in numpy."""
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (120, 32), dtype=np.uint8)
    q = t[rng.integers(0, 120, 100)].copy()
    q[:, :4] ^= rng.integers(0, 256, (100, 4), dtype=np.uint8)
    idx, dist = oracle_lib.hamming_knn2(q, t)
    d0, d1 = dist[:, 0].astype(np.float32), dist[:, 1].astype(np.float32)
    gap = d1 - d0
    med = np.sort(gap)[len(gap) // 2]
    mad = 1.4826 * np.sort(np.abs(gap - med))[len(gap) // 2]
    for ratio, TH in ((0.95, 80.0), (0.8, 40.0)):
        exp = np.where((gap > mad * 0.5) & (d0 < TH) & (d0 < np.float32(ratio) * d1), idx[:, 0], -1)
        np.testing.assert_array_equal(oracle_lib.frame_bf_match(q, t, ratio, TH), exp)


def test_associate_planes_running_threshold():
    """The live variant (src/Map.cc:249-251) overwrites dTh with the signed distance: once a negative
    distance is accepted nothing can be associated any more; the dead variant keeps a per-plane copy."""
    planes = np.array([[0, 0, 1, -1.0], [0, 0, 1, -1.0]], np.float32)
    pts = np.zeros((2, 5, 3))
    pts[0, :, 2] = 0.99   # 0.99 - 1 = -0.01 -> dis = -0.01
    pts[1, :, 2] = 1.02   # dis = +0.02
    mp = np.array([[0, 0, 1, -1.0]], np.float32)
    n, assoc = oracle_lib.associate_planes(planes, pts, mp, 0.05, 0.999, True)
    assert n == 1 and list(assoc) == [0, -1]          # dTh became -0.01 after the first association
    n, assoc = oracle_lib.associate_planes(planes, pts, mp, 0.05, 0.999, False)
    assert n == 2 and list(assoc) == [0, 0]


def test_line_grid_and_area_small():
    """mGridForLine via the Bresenham iterator: a horizontal line crosses one row of cells, and a probe
    next to it finds it (GetFeaturesInAreaForLine) while a perpendicular query line does not (cos gate)."""
    kl = np.zeros(1, oracle_lib.KEYLINE_DTYPE)
    kl["startPointX"], kl["startPointY"], kl["endPointX"], kl["endPointY"] = 105.0, 205.0, 305.0, 205.0
    for a, b in (("sPointInOctaveX", "startPointX"), ("sPointInOctaveY", "startPointY"), ("ePointInOctaveX", "endPointX"), ("ePointInOctaveY", "endPointY")):
        kl[a] = kl[b]
    kl["lineLength"] = 200.0
    bounds = (0.0, 0.0, 640.0, 480.0)
    start, idx = oracle_lib.line_grid_build(kl, bounds)
    cells = [c for c in range(64 * 48) if start[c + 1] > start[c]]
    assert [c // 48 for c in cells] == list(range(10, 31)) and {c % 48 for c in cells} == {20}
    eq = np.array([[0.0, 1.0, -205.0]])
    desc = np.zeros((1, 32), np.uint8)
    q = np.zeros(2, oracle_lib.LINEQUERY_DTYPE)
    q["x1"], q["y1"], q["x2"], q["y2"] = [100, 200], [207, 100], [300, 200], [207, 300]
    q["radius"], q["th_cos"], q["vx"], q["vy"], q["length"], q["blocks"] = 6.0, 0.96, [200, 0], [0, 200], 200.0, 1
    n, match, asg = oracle_lib.line_search_by_projection(kl, desc, eq, bounds, q, np.zeros((2, 32), np.uint8), 0)
    assert n == 1 and list(match) == [0, -1] and list(asg) == [0]


# ---- frame post-processing (gray, depth, undistortion, stereo-from-RGBD) ---------------------------------
TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0)


def _cam(vals):
    c = np.zeros((), oracle_lib.CAMERA_DTYPE)
    for k, v in zip(oracle_lib.CAMERA_DTYPE.names, vals):
        c[k] = np.float32(v)
    return c


def test_rgb_to_gray_matches_float_formula_within_one():
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    g = oracle_lib.rgb_to_gray(rgb, True).astype(np.float64)
    f = rgb[..., 0] * 0.299 + rgb[..., 1] * 0.587 + rgb[..., 2] * 0.114
    assert np.abs(g - f).max() <= 0.51
    gb = oracle_lib.rgb_to_gray(rgb[..., ::-1].copy(), False)
    np.testing.assert_array_equal(gb, g.astype(np.uint8))
    assert oracle_lib.rgb_to_gray(np.full((1, 1, 3), 255, np.uint8))[0, 0] == 255  # 4899+9617+1868 == 1<<14


def test_depth_to_float_is_float_product():
    d = np.array([0, 1, 5000, 65535], np.uint16)
    f = np.float32(1.0 / 5000.0)
    np.testing.assert_array_equal(oracle_lib.depth_to_float(d, f), d.astype(np.float32) * f)


def test_undistort_inverts_the_distortion_model():
    """Independent check of the five-iteration undistortion: distort ideal points with the forward
    radial-tangential model (float64 numpy), undistort with the oracle, recover the ideal points."""
    cam = _cam(TUM1)
    fx, fy, cx, cy, k1, k2, p1, p2, k3, _ = [float(np.float32(v)) for v in TUM1]
    rng = np.random.default_rng(2)
    ideal = np.stack([rng.uniform(40, 600, 400), rng.uniform(40, 440, 400)], 1)
    x, y = (ideal[:, 0] - cx) / fx, (ideal[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    kps = np.zeros(400, oracle_lib.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = xd * fx + cx, yd * fy + cy
    depth = np.full((480, 640), 2.0, np.float32)
    inside = (kps["x"] >= 0) & (kps["x"] < 640) & (kps["y"] >= 0) & (kps["y"] < 480)
    kps = kps[inside]
    un, dep, ur = oracle_lib.frame_post_rgbd(kps, depth, cam)
    err = np.hypot(un["x"] - ideal[inside, 0], un["y"] - ideal[inside, 1])
    assert err.max() < 0.05  # five fixed-point iterations, not a converged solve
    np.testing.assert_array_equal(dep, np.float32(2.0))
    np.testing.assert_array_equal(ur, un["x"] - np.float32(40.0) / np.float32(2.0))


def test_image_bounds_and_zero_distortion_identity():
    cam0 = _cam((535.4, 539.2, 320.1, 247.6, 0, 0, 0, 0, 0, 40.0))
    np.testing.assert_array_equal(oracle_lib.image_bounds(cam0, 640, 480), [0, 0, 640, 480])
    b = oracle_lib.image_bounds(_cam(TUM1), 640, 480)
    # k1 > 0: distorted radii exceed ideal ones, so the undistorted corners lie inside the image
    assert 0 < b[0] < 30 and 0 < b[1] < 30 and 600 < b[2] < 640 and 450 < b[3] < 480
    kps = np.zeros(3, oracle_lib.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = [10.7, 300.2, 639.9], [5.5, 200.9, 479.9]
    depth = np.zeros((480, 640), np.float32)
    depth[200, 300] = 1.5
    un, dep, ur = oracle_lib.frame_post_rgbd(kps, depth, cam0)
    assert un.tobytes() == kps.tobytes()
    np.testing.assert_array_equal(dep, [-1, 1.5, -1])
    np.testing.assert_array_equal(ur, np.array([-1, np.float32(300.2) - np.float32(40.0) / np.float32(1.5), -1], np.float32))


# ---- a18: SearchByBoW / SearchByProjection(cur, KF) against plain-Python restatements -----------------------------
def _py_three_maxima(h):
    m1 = m2 = m3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(h):
        if s > m1:
            m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
        elif s > m2:
            m3, m2, i3, i2 = m2, s, i2, i
        elif s > m3:
            m3, i3 = s, i
    if m2 < 0.1 * m1:
        i2 = i3 = -1
    elif m3 < 0.1 * m1:
        i3 = -1
    return i1, i2, i3


def _py_bow(fdesc, fangle, fidx, runs, qdesc, qangle, ratio, ori):
    owner = {}
    match = [-1] * len(runs)
    hist = [[] for _ in range(30)]
    for i, (s, ln) in enumerate(runs):
        b1, b2, bi = 256, 256, -1
        for p in range(s, s + ln):
            f = int(fidx[p])
            if f in owner:
                continue
            d = popcount_dist(qdesc[i], fdesc[f])
            if d < b1:
                b2, b1, bi = b1, d, f
            elif d < b2:
                b2 = d
        if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
            owner[bi] = i
            match[i] = bi
            if ori:
                rot = np.float32(qangle[i]) - np.float32(fangle[bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                b = int(np.round(np.float32(rot * np.float32(1.0 / 30))))  # round half away == np.round only off exact .5: avoided below
                hist[0 if b == 30 else b].append((i, bi))
    if ori:
        keep = _py_three_maxima([len(h) for h in hist])
        for b, h in enumerate(hist):
            if b not in keep:
                for i, f in h:
                    match[i] = -1
                    del owner[f]
    return sum(m >= 0 for m in match), np.array(match, np.int32), owner


def test_search_by_bow_against_plain_python():
    rng = np.random.default_rng(4)
    nf, nkf, nnodes = 300, 280, 12
    base = rng.integers(0, 256, (nf, 32), dtype=np.uint8)
    fdesc = base.copy()
    qsrc = base[rng.permutation(nf)[:nkf]].copy()
    flip = rng.integers(0, 32, nkf)
    qsrc[np.arange(nkf), flip] ^= rng.integers(1, 64, nkf).astype(np.uint8)  # near-duplicates: matches with small distances
    fangle = rng.uniform(0, 360, nf).astype(np.float32) + np.float32(0.013)
    qangle_all = rng.uniform(0, 360, nkf).astype(np.float32)
    fnode = fdesc[:, 1] % nnodes
    qnode = qsrc[:, 1] % nnodes
    fidx, start = [], {}
    for nd in range(nnodes):
        ids = np.nonzero(fnode == nd)[0]
        start[nd] = (len(fidx), len(ids))
        fidx.extend(ids.tolist())
    runs, qd, qa = [], [], []
    for nd in range(nnodes):
        for i in np.nonzero(qnode == nd)[0]:
            runs.append(start[nd]); qd.append(qsrc[i]); qa.append(qangle_all[i])
    runs = np.array(runs, np.int32); qd = np.array(qd, np.uint8); qa = np.array(qa, np.float32)
    for ori in (False, True):
        nm, match, assigned = oracle_lib.search_by_bow(fdesc, fangle, np.array(fidx, np.int32), runs, qd, qa, 0.8, ori)
        pnm, pmatch, powner = _py_bow(fdesc, fangle, fidx, runs, qd, qa, 0.8, ori)
        assert nm == pnm and nm > 30
        np.testing.assert_array_equal(match, pmatch)
        for f, i in powner.items():
            assert assigned[f] == i
        assert (assigned >= 0).sum() == len(powner)


def test_search_by_projection_kf_semantics_small():
    """two queries want the same keypoint: the first takes it, the second falls back to its next candidate or to nothing;
    an occupied keypoint is never used; the distance gate is ORBdist."""
    kps = np.zeros(3, oracle_lib.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = [100, 104, 300], [100, 100, 300]
    desc = np.zeros((3, 32), np.uint8)
    desc[1, 0] = 0x0F  # 4 bits from keypoint 0
    desc[2, :] = 0xFF
    q = np.zeros(3, oracle_lib.PROJQUERY_DTYPE)
    q["u"], q["v"], q["radius"] = [101, 102, 300], [100, 100, 300], 20
    q["min_level"], q["max_level"] = -1, -1
    qd = np.zeros((3, 32), np.uint8)
    qd[2, :4] = 0x00
    qd[2, 4:] = 0xFF  # 32 bits from keypoint 2
    nm, match, assigned = oracle_lib.search_by_projection_kf(kps, desc, BOUNDS, q, qd, None, 100, False)
    assert nm == 3 and list(match) == [0, 1, 2] and list(assigned) == [0, 1, 2]
    nm, match, _ = oracle_lib.search_by_projection_kf(kps, desc, BOUNDS, q, qd, None, 3, False)  # ORBdist 3: 4 and 32 bits fail
    assert nm == 1 and list(match) == [0, -1, -1]
    nm, match, _ = oracle_lib.search_by_projection_kf(kps, desc, BOUNDS, q, qd, np.array([1, 0, 0], np.uint8), 100, False)
    assert list(match) == [1, -1, 2]  # keypoint 0 is occupied: query 0 takes 1, query 1 finds nothing free
