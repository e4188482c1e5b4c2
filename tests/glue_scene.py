"""Synthetic RGB-D line scene for the Frame-glue tests: a two-plane "room corner" seen by the TUM1 pinhole camera,
random image segments as keylines, pairs of segments as fan rows.  Pure numpy; used by CPU and GPU tests."""
import numpy as np

import oracle_lib

W, H = 640, 480
CAM = (517.306408, 516.469215, 318.643040, 255.313989, 0.0, 0.0, 0.0, 0.0, 0.0, 40.0)


def camera():
    c = np.zeros((), oracle_lib.CAMERA_DTYPE)
    for k, v in zip(oracle_lib.CAMERA_DTYPE.names, CAM):
        c[k] = np.float32(v)
    return c


def depth_image(rng, noise=0.004, holes=True):
    """Left wall n1.P = d1, right wall n2.P = d2 meeting near the image centre; metres, f32."""
    fx, fy, cx, cy = CAM[:4]
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    x, y = (u - cx) / fx, (v - cy) / fy
    n1, d1 = np.array([0.6, 0.05, 0.8]), 2.0
    n2, d2 = np.array([-0.6, 0.02, 0.8]), 2.1
    z1 = d1 / (n1[0] * x + n1[1] * y + n1[2])
    z2 = d2 / (n2[0] * x + n2[1] * y + n2[2])
    z = np.minimum(z1, z2)  # the nearer wall is visible
    z = z + rng.normal(0, noise, z.shape) * z * z
    if holes:
        z[200:230, 100:180] = 0.0
        z[rng.random(z.shape) < 0.02] = 0.0
    return z.astype(np.float32), (n1, d1), (n2, d2)


def keylines(rng, n=60):
    kls = np.zeros(n, oracle_lib.KEYLINE_DTYPE)
    for i in range(n):
        while True:
            p = rng.uniform([5, 5], [W - 5, H - 5])
            ang = rng.uniform(0, np.pi)
            ln = rng.uniform(3, 220) if i % 7 else rng.uniform(0.2, 6)  # a few very short ones
            q = p + ln * np.array([np.cos(ang), np.sin(ang)])
            if 0 <= q[0] < W and 0 <= q[1] < H:
                break
        if i % 11 == 0:  # integer end points: the "boundary issue" branch of the sampler
            p, q = np.floor(p), np.floor(q)
        kls["startPointX"][i], kls["startPointY"][i], kls["endPointX"][i], kls["endPointY"][i] = p[0], p[1], q[0], q[1]
        kls["sPointInOctaveX"][i], kls["sPointInOctaveY"][i], kls["ePointInOctaveX"][i], kls["ePointInOctaveY"][i] = p[0], p[1], q[0], q[1]
        kls["lineLength"][i] = ln
        kls["class_id"][i] = i
    return kls


def fans(rng, kls, n=150):
    """rows (x, y, index1, index2): pairs of keylines whose end points are close, plus random pairs"""
    m = len(kls)
    ends = np.stack([np.stack([kls["startPointX"], kls["startPointY"]], 1), np.stack([kls["endPointX"], kls["endPointY"]], 1)], 1)
    rows = []
    for i in range(m):
        for j in range(i + 1, m):
            d = np.linalg.norm(ends[i][:, None, :] - ends[j][None, :, :], axis=2)
            if d.min() < 60:
                a, b = np.unravel_index(d.argmin(), d.shape)
                c = 0.5 * (ends[i][a] + ends[j][b])
                rows.append((c[0], c[1], i, j))
    while len(rows) < n:
        i, j = rng.integers(0, m, 2)
        if i != j:
            rows.append((rng.uniform(0, W), rng.uniform(0, H), i, j))
    rng.shuffle(rows)
    return np.array(rows[:n], np.float32)


def scene(seed=3, nlines=60, nfans=150):
    rng = np.random.default_rng(seed)
    depth, p1, p2 = depth_image(rng)
    kls = keylines(rng, nlines)
    return kls, fans(rng, kls, nfans), depth, camera(), (p1, p2)
