"""Shared inputs for the KeyFrame-rate matcher tests: two oracle-extracted keyframes of the synthetic
desk scene, projected map points with the reference's gates already applied by the "host", a stand-in FeatureVector."""
import numpy as np

import synth_frames as sf

BOUNDS = (0.0, 0.0, 640.0, 480.0)
SCALE = __import__("synth_frames").orb_scale_factors()
SIGMA2 = (SCALE * SCALE).astype(np.float32)
INV_SIGMA2 = (np.float32(1.0) / SIGMA2).astype(np.float32)

_cache = {}


def keyframes():
    if "kf" not in _cache:
        import oracle_lib
        orc = oracle_lib.OracleORB()
        sc = sf.Scene(640, 480, "desk", seed=11)
        _cache["kf"] = [orc(sc.gray(t)) for t in (0, 3)]
    return _cache["kf"]


def proj_queries(kps, rng, th=3.0, jitter=2.0, p_drop=0.15, dtype=None):
    """map points that project near the keypoints `kps` (of the OTHER keyframe / of fuse candidates)"""
    import oracle_lib
    q = np.zeros(len(kps), dtype or oracle_lib.PROJQUERY_DTYPE)
    q["u"] = kps["x"] + rng.normal(0, jitter, len(kps)).astype(np.float32)
    q["v"] = kps["y"] + rng.normal(0, jitter, len(kps)).astype(np.float32)
    lvl = np.clip(kps["octave"] + rng.integers(-1, 2, len(kps)), 0, 7).astype(np.int32)   # PredictScale is not exact
    q["max_level"] = lvl
    q["min_level"] = lvl - 1
    q["radius"] = np.float32(th) * SCALE[lvl]
    q["ur"] = q["u"] - np.float32(40.0) / rng.uniform(0.5, 4.0, len(kps)).astype(np.float32)
    q["radius"][rng.random(len(kps)) < p_drop] = -1.0    # no map point / bad / behind the camera / outside the image ...
    return q


def noisy_desc(desc, rng, flips=12):
    """descriptor of the map point: the keypoint's with a few bits flipped (pMP->GetDescriptor() is another observation's)"""
    d = desc.copy()
    for i in range(len(d)):
        for b in rng.integers(0, 256, rng.integers(0, flips + 1)):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def feature_vector(desc, nnodes):
    """stand-in for DBoW2's FeatureVector: node = a hash of the descriptor; indices ascending inside a node"""
    node = (desc[:, 0].astype(np.int32) * 7 + desc[:, 5]) % nnodes
    return {int(nd): np.nonzero(node == nd)[0].astype(np.int32) for nd in np.unique(node)}


def tri_inputs(k1, d1, has_mp1, stereo1, fv1, fv2, only_stereo, dtype):
    """flatten KF2's FeatureVector; one query per KF1 feature in the reference's order (src/ORBmatcher.cc:689-711)"""
    fidx, start = [], {}
    for nd in sorted(fv2):
        start[nd] = (len(fidx), len(fv2[nd]))
        fidx.extend(fv2[nd].tolist())
    rows, qd, idx1s = [], [], []
    for nd in sorted(fv1):
        if nd not in start:
            continue
        for i in fv1[nd]:
            if has_mp1[i] or (only_stereo and not stereo1[i]):
                continue
            rows.append((start[nd][0], start[nd][1], k1["x"][i], k1["y"][i], k1["angle"][i], int(stereo1[i])))
            qd.append(d1[i]); idx1s.append(i)
    q = np.array(rows, dtype) if rows else np.zeros(0, dtype)
    return np.array(fidx, np.int32), q, np.array(qd, np.uint8).reshape(-1, 32), np.array(idx1s, np.int32)


def fundamental(rng):
    """a plausible F12 (skew(t) R, pixel units) and the epipole in image 2; values only need to make the gates bite"""
    fx, fy, cx, cy = 525.0, 525.0, 319.5, 239.5
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    a = 0.02
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.12, 0.01, 0.02])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    return F.astype(np.float32), (np.float32(330.0), np.float32(250.0))
