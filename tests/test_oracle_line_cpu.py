"""Line-path oracle (oracle/line_oracle.cpp): definition-level checks, no GPU."""
import ctypes as C
import os

import numpy as np

import oracle_lib
import synth_frames as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _edge_image():
    img = np.full((240, 320), 60, np.uint8)
    yy, xx = np.mgrid[0:240, 0:320]
    img[yy > 0.5 * xx + 40] = 190          # one long straight edge: y = 0.5 x + 40
    return img


def test_lsd_finds_the_synthetic_edge():
    seg = oracle_lib.lsd_detect(_edge_image())
    assert len(seg) >= 1
    L = np.hypot(seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1])
    s = seg[L.argmax()]
    assert L.max() > 250
    for x, y in ((s[0], s[1]), (s[2], s[3])):      # both end points lie on the edge (within a pixel)
        assert abs(y - (0.5 * x + 40)) < 1.5
    assert len(oracle_lib.lsd_detect(np.full((240, 320), 99, np.uint8))) == 0


def test_lsd_gradient_definition():
    img = sf.Scene(160, 120, "struct", seed=2, n_poly=30).gray(0)
    scaled, ang, mod = oracle_lib.lsd_gradient(img)
    assert scaled.shape == (96, 128)
    A, B, Cc, D = scaled[:-1, :-1], scaled[:-1, 1:], scaled[1:, :-1], scaled[1:, 1:]
    gx, gy = (B + D) - (A + Cc), (Cc + D) - (A + B)
    np.testing.assert_allclose(mod[:-1, :-1], np.sqrt((gx * gx + gy * gy) / 4.0), rtol=1e-12)
    rho = 2.0 / np.sin(np.pi * 22.5 / 180)
    defined = ang[:-1, :-1] != -1024.0
    np.testing.assert_array_equal(defined, mod[:-1, :-1] > rho)
    ref = np.mod(np.arctan2(gx, -gy), 2 * np.pi)     # level-line angle = atan2(gx, -gy)
    d = np.abs(ang[:-1, :-1] - ref)[defined]
    assert np.minimum(d, 2 * np.pi - d).max() < 0.006  # fastAtan2 is a 0.3-degree polynomial


def test_line_iterator_count_and_clipping():
    f = lambda *a: oracle_lib.load().pso_line_iterator_count(640, 480, *[C.c_float(v) for v in a])
    assert f(10, 10, 110, 60) == 101
    assert f(10.4, 10.5, 10.4, 10.5) == 1
    assert f(-50, 100, 50, 100) == 51            # clipped at x = 0
    assert f(600, 470, 700, 500) > 0 and f(700, 500, 800, 600) == 0


def test_merge_two_collinear_segments():
    seg = np.array([[10, 100, 110, 100.5], [118, 100.6, 260, 101.2], [300, 300, 340, 305]], np.float32)
    kl = oracle_lib.optimize_and_merge(seg, 640, 480)
    assert len(kl) == 1                          # the two collinear pieces merge, the 40-px one is filtered (< 50)
    assert kl["startPointX"][0] < 12 and kl["endPointX"][0] > 258 and abs(kl["lineLength"][0] - 250) < 3
    assert kl["class_id"][0] == 0 and kl["octave"][0] == 0
    assert abs(kl["response"][0] - kl["lineLength"][0] / 640) < 1e-6


def test_lbd_sobel_and_descriptor_basics():
    img = _edge_image()
    dx, dy = oracle_lib.lbd_sobel(img)
    flat = np.full((50, 60), 90, np.uint8)
    fx, fy = oracle_lib.lbd_sobel(flat)
    assert (fx == 0).all() and (fy == 0).all()
    assert dy[90, 100] > 0 and dx[90, 100] < 0     # at (x=100, y=90) on the edge: brighter below, darker to the right
    seg = oracle_lib.lsd_detect(img)
    kl = oracle_lib.optimize_and_merge(seg, 320, 240)
    desc, fdesc = oracle_lib.lbd_compute(img, kl, want_float=True)
    assert desc.shape == (len(kl), 32) and np.isfinite(fdesc).all()
    np.testing.assert_allclose(np.linalg.norm(fdesc, axis=1), 1.0, atol=1e-5)   # re-normalised 72-vector
    assert fdesc.max() <= 0.4 / 0.4 + 1e-6


def test_pairing_right_angle_corner():
    lines = np.array([[100, 100, 200, 100], [205, 105, 205, 200], [100, 300, 200, 300]], np.float32)
    fans = oracle_lib.lil_pair(lines, 20.0, np.float32(np.pi / 4), 640, 480)
    assert len(fans) == 1 and {int(fans[0, 2]), int(fans[0, 3])} == {0, 1}
    assert abs(fans[0, 0] - 205) < 1e-3 and abs(fans[0, 1] - 100) < 1e-3     # intersection of the two supports
    par = np.array([[100, 100, 200, 100], [205, 104, 300, 104]], np.float32)     # near-parallel: rejected by fanThr
    assert len(oracle_lib.lil_pair(par, 20.0, np.float32(np.pi / 4), 640, 480)) == 0


def test_line_golden_vectors():
    g = np.load(os.path.join(GOLD, "line_640x480_struct.npz"))
    kls, desc, eq = oracle_lib.line_extract(g["image"], 200)
    assert kls.tobytes() == g["kls"].tobytes()
    np.testing.assert_array_equal(desc, g["desc"])
    np.testing.assert_array_equal(eq, g["eq"])
    np.testing.assert_array_equal(oracle_lib.lsd_detect(g["image"]), g["segments"])          # LSD_REFINE_ADV (default)
    oracle_lib.set_lsd_refine(1)
    try:
        np.testing.assert_array_equal(oracle_lib.lsd_detect(g["image"]), g["segments_std"])  # LSD_REFINE_STD
    finally:
        oracle_lib.set_lsd_refine(2)
    assert 0 < len(g["segments"]) < len(g["segments_std"])
    L = np.stack([kls[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
    np.testing.assert_array_equal(oracle_lib.lil_pair(L, 20.0, np.float32(np.pi / 4), 640, 480), g["fans"])


def test_seed_cos_sin_of_k_lsd_grad_equals_libm_for_every_float_angle(tmp_path):
    """k_lsd_grad tabulates (float)cos(a), (float)sin(a), a = (double)deg * DEG_TO_RADS, with a restricted-range evaluation
    (psl-slam_amd/csrc/psl_sincos64.h) instead of the device library's general f64 cos / sin.  oracle/sincos64_check.c runs
    the same header on the host against libm for all 1 135 869 953 floats in [0, 360] (a few seconds on 8 threads)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sincos64_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(root, "oracle", "sincos64_check.c"), "-lm", "-lpthread"],
                   check=True)
    out = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "sin mismatches 0, cos mismatches 0" in out.stdout


def test_fast_atan2_error_bound_used_by_the_lazy_region_angle():
    """k_lsd_grow4 decides most neighbours of a growing region from the running sum itself instead of the reference's
    fastAtan2(sum); its margin of 2e-3 rad (0.115 degrees, line_kernels.h) has to cover the error of the fastAtan2 polynomial.
    The polynomial's error against atan2 is measured here: < 0.02 degrees on vectors of every direction and of the magnitudes a
    region's sum can have (>= 1)."""
    L = oracle_lib.load()
    L.pso_fast_atan2_f.restype = C.c_float
    L.pso_fast_atan2_f.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(3)
    worst = 0.0
    for scale in (1.0, 3.7, 41.0, 977.0, 2.0e5):
        th = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), np.linspace(0, 2 * np.pi, 2881), np.arange(8) * np.pi / 4 + 1e-7,
                             np.arange(8) * np.pi / 4 - 1e-7])
        x = (np.cos(th) * scale).astype(np.float32)
        y = (np.sin(th) * scale).astype(np.float32)
        f = np.array([L.pso_fast_atan2_f(float(b), float(a)) for a, b in zip(x, y)], np.float64)
        t = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
        d = np.abs(f - t)
        worst = max(worst, float(np.minimum(d, 360.0 - d).max()))
    assert worst < 0.02, worst


def test_restated_atanf_atan2f_equal_libm(tmp_path):
    """The line merging calls atanf / atan2f; the device uses glibc's float algorithms restated (psl-slam_amd/csrc/psl_atanf.h).
    oracle/atanf_check.c runs that header on the host against libm: atanf for every float, atan2f on 2e8 pairs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "atanf_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(root, "oracle", "atanf_check.c"), "-lm", "-lpthread"], check=True)
    for mode in ("0", "1"):
        out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_cross_dot_test_of_the_window_rounds_never_contradicts_the_reference_decision():
    """k_lsd_grow4 decides a neighbour from the running sum S and the pixel's unit vector u: |S x u| < tan(prec - m) max(S.u, 0) => joins,
    |S x u| >= tan(prec + m) max(S.u, 0) => does not (m = 2e-3 rad, line_kernels.h lsdg_fast_setup / lsdg_decide); everything in between
    takes the reference's arithmetic.  Here the rule is evaluated in float32 on vectors of every direction, of the magnitudes a region's
    sum can have, with the pixel angle placed around the threshold, for the detector's tolerance and for refinement tolerances, and
    compared with the reference's decision fold(|fastAtan2(S) - a|) <= prec: a sure answer must never differ from it."""
    L = oracle_lib.load()
    L.pso_fast_atan2_f.restype = C.c_float
    L.pso_fast_atan2_f.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(11)
    f32 = np.float32
    K = np.pi / 180.0
    m = f32(2.0e-3)
    sure = 0
    for prec in (np.pi * 22.5 / 180.0, 1.0e-3, 0.05, 0.2, 0.8, 1.2, 1.49, 1.6):
        p = f32(prec)
        ok = bool(p + m < f32(1.5))
        t_hi = f32(np.tan(f32(p - m), dtype=f32) * f32(1 - 1e-5)) if ok and p - m > 0 else f32(-1)
        t_lo = f32(np.tan(f32(p + m), dtype=f32) * f32(1 + 1e-5)) if ok else f32(np.inf)
        n = 6000
        th = rng.uniform(0, 360, n)
        mag = np.exp(rng.uniform(np.log(0.93), np.log(2.0e5), n))
        sx = (np.cos(th * K) * mag).astype(f32)
        sy = (np.sin(th * K) * mag).astype(f32)
        # pixel angles: around +-prec of the sum's direction (three widths), and anywhere
        off = np.degrees(prec) * rng.choice([-1.0, 1.0], n) + np.where(rng.random(n) < 0.5, rng.normal(0, 0.05, n), rng.normal(0, 0.5, n))
        a = np.where(rng.random(n) < 0.8, th + off, rng.uniform(0, 360, n)) % 360.0
        a = a.astype(f32)
        ar = (a.astype(np.float64) * K).astype(f32)
        cs, sn = np.cos(ar, dtype=f32), np.sin(ar, dtype=f32)
        dot = (sx * cs + sy * sn).astype(f32)
        cr = np.abs((sx * sn - sy * cs).astype(f32))
        dp = np.maximum(dot, f32(0))
        with np.errstate(invalid="ignore"):
            RA = cr < t_hi * dp
            RN = cr >= t_lo * dp
        reg = np.array([L.pso_fast_atan2_f(float(y), float(x)) for x, y in zip(sx, sy)], np.float64) * K
        d = np.abs(reg - a.astype(np.float64) * K)
        d = np.where(d > 1.5 * np.pi, np.abs(d - 2 * np.pi), d)
        exact = d <= prec
        assert not np.any(RA & RN)
        assert not np.any(RA & ~exact), (prec, np.flatnonzero(RA & ~exact)[:5])
        assert not np.any(RN & exact), (prec, np.flatnonzero(RN & exact)[:5])
        if not ok:
            assert not RA.any() and not RN.any()
        sure += int(RA.sum() + RN.sum())
    assert sure > 20000  # the rule decides most cases; the rest goes to the exact path


def test_reduce_region_radius_by_rank_equals_the_walk():
    """lsdw_refine (psl-slam_amd/csrc/line_kernels.h) does not walk the region's list as reduce_region_radius does (OpenCV lsd.cpp; the oracle's
    Lsd::reduce_region_radius): with m stayers, a stayer in front of position m keeps its place and the k-th hole in front of m receives the k-th stayer
    behind m counted down from the end.  Both formulations on random lists and stay / leave flags, several radius steps in a row."""
    import random

    def walk(a, stays):
        a = list(a); n = len(a); i = 0
        while i < n:
            if not stays[a[i]]:
                a[i], a[n - 1] = a[n - 1], a[i]
                n -= 1
                i -= 1
            i += 1
        return a[:n]

    def by_rank(a, stays):
        n = len(a); m = sum(1 for v in a if stays[v])
        behind = [a[q] for q in range(n - 1, m - 1, -1) if stays[a[q]]]
        out = list(a[:m]); r = 0
        for j in range(m):
            if not stays[a[j]]:
                out[j] = behind[r]; r += 1
        assert r == len(behind)
        return out

    rnd = random.Random(7)
    for _ in range(30000):
        n = rnd.randint(1, 90)
        a = list(range(n)); rnd.shuffle(a)
        keep = rnd.random()
        for _step in range(3):   # the list a step leaves is the next step's input
            stays = [rnd.random() < keep for _ in range(n)]
            stays[a[0]] = True   # the region's first pixel is the centre of the radius
            w, r = walk(a, stays), by_rank(a, stays)
            assert w == r
            a = w
            if len(a) < 2:
                break
