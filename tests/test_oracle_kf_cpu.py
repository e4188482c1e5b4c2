"""CPU checks of oracle/kf_oracle.cpp (KeyFrame-rate matchers) against independent plain-numpy
restatements of the same reference loops.  The reference holds no fixtures for these functions: parity unpinned."""
import numpy as np

import kf_scene as ks
import oracle_lib


def _ham(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_window_best_against_bruteforce():
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(1)
    sel = rng.permutation(len(k1))[:150]
    q = ks.proj_queries(k1[sel], rng, th=6.0, jitter=2.0)
    qd = ks.noisy_desc(d1[sel], rng)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 30, -1).astype(np.float32)
    for chi2 in (False, True):
        bi, bd = oracle_lib.window_best(k1, d1, ur, ks.BOUNDS, q, qd, chi2, ks.INV_SIGMA2)
        nhit = 0
        for i in range(len(q)):
            if q["radius"][i] < 0:
                assert bi[i] == -1
                continue
            u, v, r, lvl = q["u"][i], q["v"][i], q["radius"][i], q["max_level"][i]
            cand = []
            for j in range(len(k1)):
                if not (abs(np.float32(k1["x"][j] - u)) < r and abs(np.float32(k1["y"][j] - v)) < r):
                    continue
                if k1["octave"][j] < lvl - 1 or k1["octave"][j] > lvl:
                    continue
                if chi2:
                    ex, ey = np.float32(u - k1["x"][j]), np.float32(v - k1["y"][j])
                    e2 = np.float32(np.float32(ex * ex) + np.float32(ey * ey))
                    lim = 5.99
                    if ur[j] >= 0:
                        er = np.float32(q["ur"][i] - ur[j])
                        e2 = np.float32(e2 + np.float32(er * er))
                        lim = 7.8
                    if float(np.float32(e2 * ks.INV_SIGMA2[k1["octave"][j]])) > lim:
                        continue
                cand.append((_ham(qd[i], d1[j]), j))
            if not cand:
                assert bi[i] == -1 and bd[i] == 0x7fffffff
                continue
            dmin = min(c[0] for c in cand)
            assert bd[i] == dmin
            winners = [j for d, j in cand if d == dmin]
            assert bi[i] in winners
            nhit += 1
        assert nhit > (30 if chi2 else 60)


def test_sim3_agreement_small():
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(2)
    n1, n2 = len(k0), len(k1)
    src12 = rng.integers(0, n2, n1)
    q12 = ks.proj_queries(k1[src12], rng, th=7.5)
    qd1 = ks.noisy_desc(d1[src12], rng, flips=10)
    src21 = rng.integers(0, n1, n2)
    q21 = ks.proj_queries(k0[src21], rng, th=7.5)
    qd2 = ks.noisy_desc(d0[src21], rng, flips=10)
    nf, m = oracle_lib.search_by_sim3(k0, d0, ks.BOUNDS, k1, d1, ks.BOUNDS, q12, qd1, q21, qd2)
    b1, e1 = oracle_lib.window_best(k1, d1, None, ks.BOUNDS, q12, qd1)
    b2, e2 = oracle_lib.window_best(k0, d0, None, ks.BOUNDS, q21, qd2)
    v1 = np.where(e1 <= 100, b1, -1)
    v2 = np.where(e2 <= 100, b2, -1)
    ref = np.array([v1[i] if v1[i] >= 0 and v2[v1[i]] == i else -1 for i in range(n1)])
    np.testing.assert_array_equal(m, ref)
    assert nf == (ref >= 0).sum()


def _py_tri(k2, d2, ur2, taken2, fidx, q, qd, F, epi, sf, s2, only_stereo, ori):
    F = F.reshape(3, 3)
    match = np.full(len(q), -1, np.int32)
    hist = [[] for _ in range(30)]
    f32 = np.float32
    for i in range(len(q)):
        x, y = q["x"][i], q["y"][i]
        a = f32(f32(f32(x * F[0, 0]) + f32(y * F[1, 0])) + F[2, 0])
        b = f32(f32(f32(x * F[0, 1]) + f32(y * F[1, 1])) + F[2, 1])
        c = f32(f32(f32(x * F[0, 2]) + f32(y * F[1, 2])) + F[2, 2])
        best, bidx = 50, -1
        for p in range(q["start"][i], q["start"][i] + q["len"][i]):
            j = fidx[p]
            if taken2[j] or (only_stereo and not ur2[j] >= 0):
                continue
            d = _ham(qd[i], d2[j])
            if d > 50 or d > best:
                continue
            if not q["stereo"][i] and not ur2[j] >= 0:
                dx, dy = f32(epi[0] - k2["x"][j]), f32(epi[1] - k2["y"][j])
                if f32(f32(dx * dx) + f32(dy * dy)) < f32(f32(100) * sf[k2["octave"][j]]):
                    continue
            num = f32(f32(f32(a * k2["x"][j]) + f32(b * k2["y"][j])) + c)
            den = f32(f32(a * a) + f32(b * b))
            if den == 0:
                continue
            if float(f32(f32(num * num) / den)) < 3.84 * float(s2[k2["octave"][j]]):
                best, bidx = d, j
        if bidx >= 0:
            match[i] = bidx
            if ori:
                rot = f32(q["angle"][i] - k2["angle"][bidx])
                if rot < 0:
                    rot = f32(rot + f32(360))
                b_ = int(np.floor(float(f32(rot * f32(1.0 / 30))) + 0.5))
                hist[0 if b_ == 30 else b_].append(i)
    if ori:
        sz = [len(h) for h in hist]
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i, s in enumerate(sz):
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for b_ in range(30):
            if b_ not in (i1, i2, i3):
                for i in hist[b_]:
                    match[i] = -1
    return int((match >= 0).sum()), match


def test_search_for_triangulation_against_plain_python():
    (k0, d0), (k1, d1) = ks.keyframes()
    rng = np.random.default_rng(3)
    k0, d0 = k0[:400], d0[:400]
    has1 = rng.random(len(k0)) < 0.3
    st1 = rng.random(len(k0)) < 0.5
    ur2 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 20, -1).astype(np.float32)
    taken2 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    d2 = d1.copy()
    d2[:len(d0)] = ks.noisy_desc(d0, rng, flips=25)
    F12, epi = ks.fundamental(rng)
    F12 = (F12 * np.float32(0.1)).astype(np.float32)
    sig2 = (ks.SIGMA2 * np.float32(4000.0)).astype(np.float32)
    total = 0
    for nnodes, only_stereo, ori in ((8, False, True), (40, True, True), (40, False, False)):
        fidx, q, qd, _ = ks.tri_inputs(k0, d0, has1, st1, ks.feature_vector(d0, nnodes), ks.feature_vector(d2, nnodes), only_stereo,
                                       oracle_lib.TRIQUERY_DTYPE)
        nm, match = oracle_lib.search_for_triangulation(k1, d2, ur2, taken2, fidx, q, qd, F12, epi, ks.SCALE, sig2, only_stereo, ori)
        rnm, rmatch = _py_tri(k1, d2, ur2, taken2, fidx, q, qd, F12, epi, ks.SCALE, sig2, only_stereo, ori)
        np.testing.assert_array_equal(match, rmatch)
        assert nm == rnm
        total += nm
    assert total > 30


def test_line_fuse_best_against_numpy():
    rng = np.random.default_rng(4)
    n = 120
    kl = np.zeros(n, oracle_lib.KEYLINE_DTYPE)
    sx, sy = rng.uniform(20, 620, n), rng.uniform(20, 460, n)
    ang = rng.choice([0.0, np.pi / 2, 0.7], n) + rng.normal(0, 0.01, n)
    ln = rng.uniform(20, 120, n)
    kl["startPointX"], kl["startPointY"] = sx, sy
    kl["endPointX"], kl["endPointY"] = sx + ln * np.cos(ang), sy + ln * np.sin(ang)
    kl["pt_x"] = (kl["startPointX"] + kl["endPointX"]) / 2
    kl["pt_y"] = (kl["startPointY"] + kl["endPointY"]) / 2
    kl["octave"] = rng.integers(0, 2, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = rng.integers(0, n, 200)
    q = np.zeros(len(src), oracle_lib.LINEFUSEQUERY_DTYPE)
    for a, b in (("x1", "startPointX"), ("y1", "startPointY"), ("x2", "endPointX"), ("y2", "endPointY")):
        q[a] = kl[b][src] + rng.normal(0, 1.0, len(src)).astype(np.float32)
    q["level"] = kl["octave"][src] + rng.integers(0, 2, len(src))
    q["radius"] = 40.0
    qd = ks.noisy_desc(desc[src], rng, flips=30)
    bi, bd = oracle_lib.line_fuse_best(kl, desc, q, qd)
    f32 = np.float32
    hits = 0
    for i in range(len(q)):
        d1x, d1y = f32(q["x1"][i] - q["x2"][i]), f32(q["y1"][i] - q["y2"][i])
        n1 = np.sqrt(f32(f32(d1x * d1x) + f32(d1y * d1y)))
        d1x, d1y = f32(d1x / n1), f32(d1y / n1)
        mx, my = 0.5 * float(f32(q["x1"][i] + q["x2"][i])), 0.5 * float(f32(q["y1"][i] + q["y2"][i]))
        best, bidx = 256, -1
        for k in range(n):
            dist = f32((mx - float(kl["pt_x"][k])) ** 2 + (my - float(kl["pt_y"][k])) ** 2)
            if dist > f32(q["radius"][i] * q["radius"][i]):
                continue
            d2x, d2y = f32(kl["startPointX"][k] - kl["endPointX"][k]), f32(kl["startPointY"][k] - kl["endPointY"][k])
            n2 = np.sqrt(f32(f32(d2x * d2x) + f32(d2y * d2y)))
            d2x, d2y = f32(d2x / n2), f32(d2y / n2)
            if abs(f32(f32(d1x * d2x) + f32(d1y * d2y))) < f32(0.998):
                continue
            if kl["octave"][k] < q["level"][i] - 1 or kl["octave"][k] > q["level"][i]:
                continue
            d = _ham(qd[i], desc[k])
            if d < best:
                best, bidx = d, k
        assert (bi[i], bd[i]) == (bidx, best)
        hits += bidx >= 0
    assert hits > 100


def test_distinctive_descriptors_against_numpy():
    rng = np.random.default_rng(5)
    sizes = [0, 1, 2, 3, 4, 7, 20, 33]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = rng.integers(0, 256, (off[-1], 32), dtype=np.uint8)
    desc[off[6]:off[7]] = ks.noisy_desc(np.repeat(desc[off[6]:off[6] + 1], 20, 0), rng, flips=40)
    best = oracle_lib.distinctive_descriptors(desc, off)
    for p, s in enumerate(sizes):
        if s == 0:
            assert best[p] == -1
            continue
        D = desc[off[p]:off[p + 1]]
        M = np.array([[_ham(a, b) for b in D] for a in D])
        med = np.sort(M, axis=1)[:, int(0.5 * (s - 1))]
        assert best[p] == int(np.argmin(med))   # argmin returns the first minimum, like `median < BestMedian`
