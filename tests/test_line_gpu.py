"""GPU parity of the line front-end vs the CPU oracle (oracle/line_oracle.cpp).

Stages that are pure IEEE arithmetic in a fixed order (LSD working image, gradient norm, level-line
angle, LBD given keylines, merge, pairing) are compared exactly.  LSD segments depend on double
cos/sin from libm (host) vs the device math library inside a serial chain; the stated tolerance
 is: >= 95 % of oracle segments recovered with endpoints within 0.5 px — in practice the
lists are identical and the test reports it.
"""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu

DEG2RAD = np.pi / 180


def _scene(style, seed, t=0, w=640, h=480):
    return sf.Scene(w, h, style, seed).gray(t)


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4)])
def test_lsd_gradient_stage_exact(style, seed):
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    le = P.LINEextractor()
    le.lsd_detect(img)
    scaled, angdeg, mod = le.debug_gradient(0)
    rs, ra, rm = oracle_lib.lsd_gradient(img)
    np.testing.assert_array_equal(scaled, rs)
    np.testing.assert_array_equal(mod[:-1, :-1], rm[:-1, :-1])
    gang = np.where(angdeg == -1024.0, -1024.0, angdeg.astype(np.float64) * DEG2RAD)
    np.testing.assert_array_equal(gang, ra)


def _match_segments(got, ref, tol=0.5):
    if len(ref) == 0:
        return 1.0
    hit = 0
    for r in ref:
        d = np.abs(got - r).max(1) if len(got) else np.array([9e9])
        hit += d.min() <= tol
    return hit / len(ref)


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4), ("struct", 8)])
def test_lsd_segments(style, seed):
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    got = P.LINEextractor().lsd_detect(img)
    ref = oracle_lib.lsd_detect(img)
    frac = _match_segments(got, ref)
    exact = got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()
    print(f"LSD {style}/{seed}: {len(got)} vs oracle {len(ref)} segments, recovered {frac:.4f}, bit-identical {exact}")
    assert len(ref) > 100
    assert frac >= 0.95 and abs(len(got) - len(ref)) <= 0.05 * len(ref)


def test_lsd_large_regions_exercise_queue_overflow():
    """Wide smooth ramps give regions of several thousand pixels (> the 1024-entry LDS ring of the queue),
    thick bars give regions that fail the density test and go through refine / reduce_region_radius."""
    import psl_slam_amd as P
    import oracle_lib
    yy, xx = np.mgrid[0:480, 0:640]
    img = np.clip(40 + 0.9 * (xx - 100) * (xx > 100) * (xx < 300) + 180 * (xx >= 300), 0, 255)
    img = np.where(yy > 300, np.clip(30 + 1.2 * (yy - 300), 0, 255), img)
    img[100:140, 350:600] = 20
    img[180:190, 340:620] = 240
    rng = np.random.default_rng(5)
    img = np.clip(img + rng.normal(0, 1.0, img.shape), 0, 255).astype(np.uint8)
    got = P.LINEextractor().lsd_detect(img)
    ref = oracle_lib.lsd_detect(img)
    exact = got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()
    print(f"LSD ramps: {len(got)} vs oracle {len(ref)} segments, bit-identical {exact}")
    assert len(ref) >= 4 and _match_segments(got, ref) >= 0.95 and abs(len(got) - len(ref)) <= max(1, 0.05 * len(ref))
    assert exact


def test_lsd_flat_image_gives_no_segments():
    import psl_slam_amd as P
    assert len(P.LINEextractor().lsd_detect(np.full((480, 640), 128, np.uint8))) == 0


def _kl_equal(a, b, what, skip=()):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)} keylines"
    for name in a.dtype.names:
        if name in skip:
            continue
        x, y = a[name], b[name]
        bad = np.flatnonzero(x.view(np.uint32) != y.view(np.uint32)) if x.dtype.kind == "f" else np.flatnonzero(x != y)
        assert bad.size == 0, f"{what}: field {name} differs at {bad[:5]}: {x[bad[:5]]} vs {y[bad[:5]]}"


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4), ("struct", 8)])
def test_merge_stage_on_oracle_segments(style, seed):
    """optimizeAndMergeLines_lsd on identical input segments.  atanf / atan2f are glibc's algorithms restated (pinned against
    libm in the CPU suite); the double sin, cos of MergeTwoLines come from the device math library vs glibc in the oracle
    (last-ulp differences possible), so the contract is a tolerance: same number of lines, endpoints within 0.01 px; bit-identity is
    reported."""
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    seg = oracle_lib.lsd_detect(img)
    ref = oracle_lib.optimize_and_merge(seg, 640, 480)
    got = P.LINEextractor().optimize_and_merge(seg, 640, 480)
    assert len(got) == len(ref) and len(ref) > 10
    ge = np.stack([got[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    re_ = np.stack([ref[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    assert np.abs(ge - re_).max() <= 0.01
    np.testing.assert_array_equal(got["numOfPixels"], ref["numOfPixels"])
    np.testing.assert_array_equal(got["class_id"], ref["class_id"])
    exact = got.tobytes() == ref.tobytes()
    print(f"merge {style}/{seed}: {len(seg)} segments -> {len(got)} keylines, bit-identical {exact}")


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4)])
def test_lbd_exact_on_oracle_keylines(style, seed):
    """LBD given keylines: blur, Sobel, 72-float vector and the 256-bit code, all bit-exact."""
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    kls, rdesc, _ = oracle_lib.line_extract(img, 200)
    le = P.LINEextractor()
    desc, fdesc = le.lbd_compute(img, kls, want_float=True)
    dx, dy = le.debug_sobel(640, 480)
    rdx, rdy = oracle_lib.lbd_sobel(img)
    np.testing.assert_array_equal(dx, rdx)
    np.testing.assert_array_equal(dy, rdy)
    rdesc2, rfdesc = oracle_lib.lbd_compute(img, kls, want_float=True)
    np.testing.assert_array_equal(rdesc2, rdesc)
    np.testing.assert_array_equal(fdesc.view(np.uint32), rfdesc.view(np.uint32))
    np.testing.assert_array_equal(desc, rdesc)
    assert len(kls) > 20


def test_lbd_lines_touching_the_border():
    import psl_slam_amd as P
    import oracle_lib
    img = _scene("desk", 12)
    kls = np.zeros(4, P.KEYLINE_DTYPE)
    ends = [(0, 0, 639, 479), (-20.5, 100.2, 300.7, -10.0), (620, 5, 700, 470), (5.5, 470.25, 630.25, 478.5)]
    for i, (x1, y1, x2, y2) in enumerate(ends):
        for a, b in (("startPointX", x1), ("startPointY", y1), ("endPointX", x2), ("endPointY", y2), ("sPointInOctaveX", x1),
                     ("sPointInOctaveY", y1), ("ePointInOctaveX", x2), ("ePointInOctaveY", y2)):
            kls[a][i] = b
        kls["angle"][i] = np.float32(np.arctan2(np.float32(y2) - np.float32(y1), np.float32(x2) - np.float32(x1)))
        kls["numOfPixels"][i] = oracle_lib.load().pso_line_iterator_count(640, 480, *[__import__("ctypes").c_float(v) for v in (x1, y1, x2, y2)])
        kls["class_id"][i] = i
    assert (kls["numOfPixels"] > 0).all()
    desc, fdesc = P.LINEextractor().lbd_compute(img, kls, want_float=True)
    rdesc, rfdesc = oracle_lib.lbd_compute(img, kls, want_float=True)
    np.testing.assert_array_equal(fdesc.view(np.uint32), rfdesc.view(np.uint32))
    np.testing.assert_array_equal(desc, rdesc)


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4), ("struct", 8)])
def test_pairing_on_oracle_lines(style, seed):
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    kls, _, _ = oracle_lib.line_extract(img, 200)
    lines = np.stack([kls[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
    ref = oracle_lib.lil_pair(lines, 20.0, np.float32(np.pi / 4), 640, 480)
    got = P.LINEextractor().pair(lines, 20.0, np.float32(np.pi / 4), 640, 480)
    assert got.shape == ref.shape and len(ref) > 5
    np.testing.assert_array_equal(got[:, 2:], ref[:, 2:])
    assert np.abs(got[:, :2] - ref[:, :2]).max() <= 1e-3
    print(f"pairing {style}/{seed}: {len(ref)} fans, bit-identical {got.tobytes() == ref.tobytes()}")
    # degenerate inputs: no lines, one line
    assert len(P.LINEextractor().pair(np.zeros((0, 4), np.float32), 20.0, 0.785, 640, 480)) == 0
    assert len(P.LINEextractor().pair(lines[:1], 20.0, 0.785, 640, 480)) == 0


@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4), ("struct", 8)])
def test_full_line_extractor(style, seed):
    """LINEextractor::operator(): tolerance contract: >= 95 % of the oracle's keylines
    recovered with endpoints within 0.5 px and, on the recovered ones, LBD Hamming distance <= 8."""
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    rk, rd, re_ = oracle_lib.line_extract(img, 200)
    gk, gd, ge = P.LINEextractor(1, 1.2, 200, 0.0)(img)
    assert len(rk) > 20 and abs(len(gk) - len(rk)) <= 0.05 * len(rk)
    R = np.stack([rk[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    G = np.stack([gk[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    hit = 0
    for i, r in enumerate(R):
        d = np.abs(G - r).max(1)
        j = int(d.argmin())
        if d[j] <= 0.5:
            hit += 1
            assert int(np.unpackbits(gd[j] ^ rd[i]).sum()) <= 8
            assert np.abs(ge[j] - re_[i]).max() <= 1e-2 * max(1.0, np.abs(re_[i]).max())
    exact = gk.tobytes() == rk.tobytes() and (gd == rd).all() and ge.tobytes() == re_.tobytes()
    print(f"line extract {style}/{seed}: {len(gk)} vs {len(rk)} keylines, recovered {hit / len(rk):.4f}, bit-identical {exact}")
    assert hit >= 0.95 * len(rk)


def test_line_extractor_batch_and_pairing_device():
    import psl_slam_amd as P
    frames = np.stack([_scene("struct", 3, t) for t in range(3)], 0)
    le = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=3)
    d_ptr, _ = le.ctx.device_array(frames)
    le.extract_batch_device(d_ptr, 3, 640, 480, 640, 640 * 480)
    le.pair_batch_device(20.0, np.float32(np.pi / 4))
    single = P.LINEextractor(1, 1.2, 200, 0.0)
    for f in range(3):
        k, dsc, eq, st = le.fetch(f)
        k1, d1, e1 = single(frames[f])
        assert st == 0 and k.tobytes() == k1.tobytes() and (dsc == d1).all() and eq.tobytes() == e1.tobytes()
        lines = np.stack([k[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
        fans = le.fans_fetch(f)
        ref = single.pair(lines, 20.0, np.float32(np.pi / 4), 640, 480)
        assert fans.tobytes() == ref.tobytes()
    # batched LSDmatcher::match(last, cur, 0.9) on the HBM-resident descriptors vs the oracle
    import oracle_lib
    import ctypes as C
    cap = le.results_device()[4]
    d_m, _ = le.ctx.device_array(np.full((3, cap), -7, np.int32))
    d_n, _ = le.ctx.device_array(np.zeros(3, np.int32))
    le.match_batch_device(1, 0.9, d_m, d_n)
    le.ctx.synchronize()
    m = np.zeros((3, cap), np.int32)
    nm = np.zeros(3, np.int32)
    P._check(P.lib().pslfe_device_download(le.ctx._h, P._ptr(m), C.c_void_p(d_m), C.c_size_t(m.nbytes)), "download")
    P._check(P.lib().pslfe_device_download(le.ctx._h, P._ptr(nm), C.c_void_p(d_n), C.c_size_t(nm.nbytes)), "download")
    descs = [le.fetch(f)[1] for f in range(3)]
    for f in range(3):
        rn, rm = oracle_lib.line_match_nnr(descs[(f - 1) % 3], descs[f], 0.9)
        assert nm[f] == rn
        np.testing.assert_array_equal(m[f, :len(rm)], rm)
    for d in (d_ptr, d_m, d_n):
        le.ctx.device_free(d)


def test_line_extractor_empty_inputs():
    import psl_slam_amd as P
    le = P.LINEextractor()
    k, d, e = le(np.zeros((0, 0), np.uint8))
    assert len(k) == 0
    k, d, e = le(np.full((480, 640), 77, np.uint8))   # no gradients: LSD finds nothing; upstream would hit UB (H12)
    assert len(k) == 0 and d.shape == (0, 32)
    with pytest.raises(P.PslfeError):
        P.LINEextractor(2, 1.2, 200, 0.0)              # only numOctaves == 1 (every reference YAML)


def test_hip_vs_committed_line_golden():
    """HIP vs tests/golden/line_640x480_struct.npz: LSD segments, LBD on the golden keylines and pairing on the
    golden lines are exact; the end-to-end extractor stays within the stated tolerance."""
    import os
    import psl_slam_amd as P
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "line_640x480_struct.npz"))
    le = P.LINEextractor(1, 1.2, 200, 0.0)
    seg = le.lsd_detect(g["image"])
    np.testing.assert_array_equal(seg.view(np.uint32), g["segments"].view(np.uint32))
    np.testing.assert_array_equal(le.lbd_compute(g["image"], g["kls"]), g["desc"])
    L = np.stack([g["kls"][n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
    np.testing.assert_array_equal(le.pair(L, 20.0, np.float32(np.pi / 4), 640, 480), g["fans"])
    k, d, e = le(g["image"])
    assert len(k) == len(g["kls"])
    E = np.stack([k[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    R = np.stack([g["kls"][n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    assert np.abs(E - R).max() <= 0.5


def test_line_extractor_xcd_grid_ragged_batch():
    """9 frames: the XCD-aware grids of the scale / LBD pre-pass kernels with a ragged last group; every frame equals the
    single-frame path (which uses the plain grids and is checked against the oracle above)."""
    import psl_slam_amd as P
    frames = np.stack([_scene("struct", 5, t) for t in range(9)], 0)
    le = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=9)
    d_ptr, _ = le.ctx.device_array(frames)
    le.extract_batch_device(d_ptr, 9, 640, 480, 640, 640 * 480)
    single = P.LINEextractor(1, 1.2, 200, 0.0)
    for f in range(9):
        k, dsc, eq, st = le.fetch(f)
        k1, d1, e1 = single(frames[f])
        assert st == 0 and k.tobytes() == k1.tobytes() and (dsc == d1).all() and eq.tobytes() == e1.tobytes(), f"frame {f}"
    le.ctx.device_free(d_ptr)


@pytest.mark.parametrize("nscenes", [2, 4])
def test_merge_stage_more_segments_than_the_small_lds_instance(nscenes):
    """k_line_merge exists for 512 and 1024 lines in LDS, beyond that its working set is in HBM: segment lists of ~700 and
    ~1400 lines (several scenes' segments in one list) exercise the other two paths; same contract as above."""
    import psl_slam_amd as P
    import oracle_lib
    seg = np.concatenate([oracle_lib.lsd_detect(_scene(st, sd)) for st, sd in (("desk", 4), ("struct", 3), ("desk", 9), ("desk", 12))[:nscenes]])
    assert len(seg) > (512 if nscenes == 2 else 1024)
    ref = oracle_lib.optimize_and_merge(seg, 640, 480, cap=4096)
    got = P.LINEextractor().optimize_and_merge(seg, 640, 480, cap=4096)
    assert len(got) == len(ref) and len(ref) > 100
    ge = np.stack([got[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    re_ = np.stack([ref[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1)
    assert np.abs(ge - re_).max() <= 0.01
    np.testing.assert_array_equal(got["numOfPixels"], ref["numOfPixels"])
    print(f"merge of {len(seg)} segments -> {len(got)} keylines, bit-identical {got.tobytes() == ref.tobytes()}")


def test_lsd_segments_bit_identical_over_many_frames():
    """The region growing decides most neighbours against an angle that is NOT up to date (a rigorous bound on its drift
    replaces the per-pixel fastAtan2, line_kernels.h): every shortcut must give the reference's decision, so the segment
    lists of many different frames are compared bit for bit (textured and structure-like scenes, several time steps)."""
    import psl_slam_amd as P
    import oracle_lib
    le = P.LINEextractor()
    nseg = 0
    for style, seed in (("desk", 31), ("struct", 32), ("desk", 33), ("struct", 34)):
        sc = sf.Scene(640, 480, style, seed)
        for t in range(0, 12, 2):
            img = sc.gray(t)
            got = le.lsd_detect(img)
            ref = oracle_lib.lsd_detect(img)
            assert got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all(), (style, seed, t)
            nseg += len(ref)
    assert nseg > 5000
