"""GPU parity of the matching part of the path vs the sequential CPU oracle (oracle/match_oracle.cpp):
frame grid (AssignFeaturesToGrid), the two SearchByProjection variants (including the reference's
first-come-first-served skipping of taken keypoints and the rotation-histogram filter), brute-force
Hamming kNN-2 and LSDmatcher::matchNNR.  All comparisons are exact."""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu

BOUNDS = (0.0, 0.0, 640.0, 480.0)


def _frames():
    import oracle_lib
    orc = oracle_lib.OracleORB()
    sc = sf.Scene(640, 480, "desk", seed=11)
    return [orc(sc.gray(t)) for t in range(2)], sc


def make_queries(kps, desc, rng, th=15.0, jitter=2.0, p_block=0.7, with_ur=False):
    import psl_slam_amd as P
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    scale = sf.orb_scale_factors()
    q = np.zeros(len(kps), P.PROJQUERY_DTYPE)
    q["u"] = kps["x"] + rng.uniform(-jitter, jitter, len(kps)).astype(np.float32)
    q["v"] = kps["y"] + rng.uniform(-jitter, jitter, len(kps)).astype(np.float32)
    q["radius"] = np.float32(th) * scale[kps["octave"]]
    q["min_level"] = kps["octave"] - 1
    q["max_level"] = kps["octave"] + 1
    q["angle"] = kps["angle"]
    q["blocks"] = (rng.random(len(kps)) < p_block).astype(np.int32)
    q["ur"] = q["u"] - np.float32(40.0) / np.float32(2.0) if with_ur else 0
    return q, desc.copy()


def test_grid_matches_reference_order():
    import psl_slam_amd as P
    import oracle_lib
    (f0, _), _ = _frames()
    kps, desc = f0
    g = P.FrameGrid(2048, 2)
    g.set(1, kps, desc, BOUNDS)
    start, idx = g.debug_grid(1)
    rstart, ridx = oracle_lib.grid_build(kps, BOUNDS)
    np.testing.assert_array_equal(start, rstart)
    np.testing.assert_array_equal(idx, ridx)
    # keypoints that fall off the grid (PosInGrid false) are dropped, as in the reference
    k2 = kps.copy()
    k2["x"][:5] = 700.0
    g.set(0, k2, desc, BOUNDS)
    start, idx = g.debug_grid(0)
    rstart, ridx = oracle_lib.grid_build(k2, BOUNDS)
    np.testing.assert_array_equal(start, rstart)
    np.testing.assert_array_equal(idx, ridx)
    assert len(idx) == len(kps) - 5


@pytest.mark.parametrize("check_ori", [True, False])
@pytest.mark.parametrize("p_block,with_ur", [(0.7, False), (1.0, True), (0.0, False)])
def test_search_by_projection_last(check_ori, p_block, with_ur):
    import psl_slam_amd as P
    import oracle_lib
    (f0, f1), _ = _frames()
    rng = np.random.default_rng(3)
    kps1, desc1 = f1
    q, qd = make_queries(f0[0], f0[1], rng, p_block=p_block, with_ur=with_ur)
    uright = None
    if with_ur:
        uright = np.where(rng.random(len(kps1)) < 0.6, kps1["x"] - 20.0 + rng.uniform(-30, 30, len(kps1)), -1.0).astype(np.float32)
    g = P.FrameGrid(2048, 1)
    g.set(0, kps1, desc1, BOUNDS, uright)
    nm, match, assigned = P.ORBmatcher(0.9, check_ori).SearchByProjectionLast(g, 0, q, qd)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_last(kps1, desc1, uright, BOUNDS, q, qd, None, check_ori)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)
    assert nm > 300  # the drifted scene really matches


def test_search_by_projection_last_contention():
    """Many queries fight for few keypoints: exercises the first-come-first-served fixpoint."""
    import psl_slam_amd as P
    import oracle_lib
    (f0, f1), _ = _frames()
    rng = np.random.default_rng(5)
    kps1, desc1 = f1[0][:60], f1[1][:60]
    sel = rng.integers(0, len(f0[0]), 900)
    q, qd = make_queries(f0[0][sel], f0[1][sel], rng, th=60.0, jitter=30.0, p_block=0.9)
    q["min_level"], q["max_level"] = -1, -1
    qd = desc1[rng.integers(0, 60, 900)].copy()  # descriptors close to the few targets
    flip = rng.integers(0, 256, (900, 4))
    for i in range(900):
        for b in flip[i]:
            qd[i, b // 8] ^= 1 << (b % 8)
    taken = (rng.random(60) < 0.2).astype(np.uint8)
    g = P.FrameGrid(64, 1)
    g.set(0, kps1, desc1, BOUNDS)
    nm, match, assigned = P.ORBmatcher(0.9, True).SearchByProjectionLast(g, 0, q, qd, taken)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_last(kps1, desc1, None, BOUNDS, q, qd, taken, True)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)


@pytest.mark.parametrize("nnratio", [0.8, 0.6])
def test_search_by_projection_map(nnratio):
    import psl_slam_amd as P
    import oracle_lib
    (f0, f1), _ = _frames()
    rng = np.random.default_rng(9)
    kps1, desc1 = f1
    q, qd = make_queries(f0[0], f0[1], rng, th=4.0, jitter=1.5, p_block=1.0)
    q["min_level"] = f0[0]["octave"] - 1  # GetFeaturesInArea(..., nPredictedLevel-1, nPredictedLevel) :66
    q["max_level"] = f0[0]["octave"]
    taken = (rng.random(len(kps1)) < 0.1).astype(np.uint8)
    g = P.FrameGrid(2048, 1)
    g.set(0, kps1, desc1, BOUNDS)
    nm, match, assigned = P.ORBmatcher(nnratio, True).SearchByProjectionMap(g, 0, q, qd, taken)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_map(kps1, desc1, None, BOUNDS, q, qd, taken, nnratio)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)
    assert nm > 100


def test_matcher_from_device_resident_extraction():
    """extract on the GPU -> grid built from HBM-resident results -> same matches as the host path."""
    import psl_slam_amd as P
    import oracle_lib
    frames = sf.stream(2, 640, 480, "desk", seed=11)
    orb = P.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    res = orb.extract_batch(frames)
    g = P.FrameGrid(orb.max_keypoints(640, 480), 2)
    g.set_from_orb(orb, BOUNDS)
    g.n = [len(res[0][0]), len(res[1][0])]
    rng = np.random.default_rng(1)
    q, qd = make_queries(res[0][0], res[0][1], rng)
    nm, match, assigned = P.ORBmatcher(0.9, True).SearchByProjectionLast(g, 1, q, qd)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_last(res[1][0], res[1][1], None, BOUNDS, q, qd, None, True)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)


def test_hamming_knn2_and_match_nnr():
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(2)
    t = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    q = t[rng.integers(0, 200, 180)].copy()
    q[:, :3] ^= rng.integers(0, 256, (180, 3), dtype=np.uint8)
    t[50] = t[10]
    t[51] = t[10]  # exact duplicates: lower train index must rank first
    q[0] = t[10]
    idx, dist = P.hamming_knn2(q, t)
    ridx, rdist = oracle_lib.hamming_knn2(q, t)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)
    assert list(idx[0]) == [10, 50] and list(dist[0]) == [0, 0]
    for nnr in (0.95, 0.85, 0.5):
        nm, m12 = P.LSDmatcher().matchNNR(q, t, nnr)
        rnm, rm12 = oracle_lib.line_match_nnr(q, t, nnr)
        assert nm == rnm
        np.testing.assert_array_equal(m12, rm12)
    # edge cases the reference leaves undefined (add_src/LSDmatcher.cpp:369): defined as "no match"
    idx, dist = P.hamming_knn2(q[:3], t[:1])
    assert (idx[:, 1] == -1).all() and (idx[:, 0] == 0).all()
    nm, m12 = P.LSDmatcher().matchNNR(q[:3], t[:1], 0.9)
    assert nm == 0 and (m12 == -1).all()
    idx, dist = P.hamming_knn2(q[:0], t)
    assert idx.shape == (0, 2)
    # large brute force (ORB-sized, 1000 x 1000)
    T = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    Q = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    idx, dist = P.hamming_knn2(Q, T)
    ridx, rdist = oracle_lib.hamming_knn2(Q, T)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)


def test_hip_reproduces_committed_match_golden():
    import os
    import psl_slam_amd as P
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_640x480_desk.npz"))
    grid = P.FrameGrid(2048, 1)
    grid.set(0, g["k1"], g["d1"], BOUNDS)
    nm, match, assigned = P.ORBmatcher(0.9, True).SearchByProjectionLast(grid, 0, g["queries"], g["qdesc"])
    assert nm == int(g["nmatches"])
    np.testing.assert_array_equal(match, g["match"])
    np.testing.assert_array_equal(assigned, g["assigned"])
    idx, dist = P.hamming_knn2(g["qdesc"][:200], g["d1"][:200])
    np.testing.assert_array_equal(idx, g["knn_idx"])
    np.testing.assert_array_equal(dist, g["knn_dist"])


# ---- SearchByProjection(cur, KF) and SearchByBoW(KF, F) ----------------------------------------
@pytest.mark.parametrize("orb_dist", [100, 64])
@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_projection_kf(orb_dist, check_ori):
    import psl_slam_amd as P
    import oracle_lib
    ((k0, d0), (k1, d1)), _ = _frames()
    rng = np.random.default_rng(17)
    q, qd = make_queries(k0, d0, rng, th=10.0, jitter=3.0)
    taken = (rng.random(len(k1)) < 0.2).astype(np.uint8)  # keypoints that already hold a map point
    g = P.FrameGrid(2048, 1)
    g.set(0, k1, d1, BOUNDS, uright=rng.uniform(1, 600, len(k1)).astype(np.float32))  # mvuRight must not gate this variant
    nm, match, assigned = P.ORBmatcher(0.9, check_ori).SearchByProjectionKF(g, 0, q, qd, taken, ORBdist=orb_dist)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_kf(k1, d1, BOUNDS, q, qd, taken, orb_dist, check_ori)
    assert nm == rnm and nm > 100
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)
    assert not taken[match[match >= 0]].any()


def _feature_vector(desc, nnodes):
    """stand-in for DBoW2's FeatureVector: node = a hash of the descriptor; indices ascending inside a node"""
    node = (desc[:, 0].astype(np.int32) * 7 + desc[:, 5]) % nnodes
    return {int(nd): np.nonzero(node == nd)[0].astype(np.int32) for nd in np.unique(node)}


def _bow_inputs(dkf, akf, valid, fv_kf, fv_f):
    fidx, start = [], {}
    for nd in sorted(fv_f):
        start[nd] = (len(fidx), len(fv_f[nd]))
        fidx.extend(fv_f[nd].tolist())
    runs, qa, qd = [], [], []
    for nd in sorted(fv_kf):          # common nodes ascending, vIndicesKF order, features without a good map point dropped
        if nd not in start:
            continue
        for i in fv_kf[nd]:
            if valid[i]:
                runs.append(start[nd]); qa.append(akf[i]); qd.append(dkf[i])
    return (np.array(fidx, np.int32), np.array(runs, np.int32).reshape(-1, 2), np.array(qa, np.float32),
            np.array(qd, np.uint8).reshape(-1, 32))


@pytest.mark.parametrize("nnodes,ratio", [(40, 0.7), (6, 0.9), (400, 0.75)])
@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow(nnodes, ratio, check_ori):
    """few nodes -> long runs (more than the 8 cached candidates: the rescan path), many nodes -> short and empty runs"""
    import psl_slam_amd as P
    import oracle_lib
    ((k0, d0), (k1, d1)), _ = _frames()
    rng = np.random.default_rng(23)
    valid = rng.random(len(k0)) < 0.8
    fidx, runs, qa, qd = _bow_inputs(d0, k0["angle"], valid, _feature_vector(d0, nnodes), _feature_vector(d1, nnodes))
    g = P.FrameGrid(2048, 1)
    g.set(0, k1, d1, BOUNDS)
    nm, match, assigned = P.ORBmatcher(ratio, check_ori).SearchByBoW(g, 0, fidx, runs, qa, qd)
    rnm, rmatch, rassigned = oracle_lib.search_by_bow(d1, k1["angle"], fidx, runs, qd, qa, ratio, check_ori)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)
    if nnodes <= 40:
        assert nm > 20


def test_search_by_bow_empty_and_bad_input():
    import psl_slam_amd as P
    ((k0, d0), (k1, d1)), _ = _frames()
    g = P.FrameGrid(2048, 1)
    g.set(0, k1, d1, BOUNDS)
    nm, match, assigned = P.ORBmatcher(0.7, True).SearchByBoW(g, 0, np.zeros(0, np.int32), np.zeros((0, 2), np.int32), [], np.zeros((0, 32), np.uint8))
    assert nm == 0 and len(match) == 0
    with pytest.raises(P.PslfeError):  # a run that leaves the feature vector
        P.ORBmatcher(0.7, True).SearchByBoW(g, 0, np.arange(10, dtype=np.int32), [(5, 10)], [0.0], d0[:1])


def test_many_frames_search_by_projection_staged_window_equals_oracle():
    """The many-frames launch of SearchByProjection(cur,last) (>= 64 frames of <= 1280 keypoints: k_window_eval_staged, the frame's
    grid / keypoints / descriptors in LDS, one workgroup per frame) through the batched pipeline: ORB extraction of 96 frames,
    frame f matched against frame f - 1; sampled frames equal the oracle (keypoints, descriptors, matches), and the whole match
    table equals the one the wave-per-query kernel gives for the same frames in launches of 48 (< 64: not staged)."""
    import torch
    import psl_slam_amd as P
    import batch_pipeline as BP
    B, w, h = 96, 640, 480
    frames = []
    for style, seed in (("desk", 31), ("struct", 32), ("sticks", 33)):
        sc = sf.Scene(w, h, style, seed)
        frames += [sc.gray(t) for t in range(8)]
    gray = np.ascontiguousarray(np.stack(frames * (B // len(frames)), 0))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    pipe = BP.BatchPipeline(P, torch, dev, stream, 0, B, w, h, lines=False)
    assert pipe.cap <= 1280
    d_gray = torch.from_numpy(gray).to(dev)
    pipe.step(d_gray.data_ptr())
    torch.cuda.synchronize(dev)
    match_staged, nm_staged = pipe.match.cpu().numpy().copy(), pipe.nmatches.cpu().numpy().copy()
    cache = {}
    for f in (0, 1, 7, 8, 16, 23, 24, 95):   # incl. the cuts between scenes (frame 8: struct after desk) and the cyclic predecessor of frame 0
        pf = (f - 1) % B
        ref = BP.oracle_frame((pf % 24, gray[pf]), (f % 24, gray[f]), None, f, w, h, False, pipe.cam, cache=cache)
        BP.compare_frame(pipe.fetch_frame(f), ref, f"frame {f}: ")
    assert int(nm_staged.sum()) > 100 * B // 4
    # the same frames in two launches of 48: the wave-per-query kernel; frame 0 / 48 see another predecessor there, every other row is equal
    for half in (0, 1):
        pipe.step(d_gray[48 * half:].data_ptr(), None, 48)
        torch.cuda.synchronize(dev)
        m = pipe.match.cpu().numpy()[:48]
        assert np.array_equal(m[1:], match_staged[48 * half + 1:48 * half + 48]), f"half {half}: staged and wave-per-query match tables differ"
