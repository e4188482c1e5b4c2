"""GPU parity of the line front-end vs the CPU oracle (oracle/line_oracle.cpp): every stage and the whole extractor are
compared BIT FOR BIT (segments, keylines, LBD descriptors, line equations, fans), in both LSD refinement modes
(LSD_REFINE_ADV = default, LSD_REFINE_STD).  The device calls no math library on this path: sinf / cosf / atanf / atan2f / tanf
are glibc's float algorithms restated and pinned exhaustively against libm in the CPU suite, and so are the double sin / cos of
MergeTwoLines and region2rect (glibc's table-driven algorithm restated, psl_sincos_glibc.h: 0 differences on 6e7 arguments);
only the log / exp / log10 of the NFA, whose glibc form cannot be reproduced offline, are fdlibm's, within 1-2 ulp of glibc
(tests/test_f64math_cpu.py) - a difference that can reach an output only on an exact tie of two NFA values; none occurs on any
frame compared here, and the CPU suite counts decision flips over 240 frames (0).
"""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu

DEG2RAD = np.pi / 180


ADV, STD = 2, 1


def _scene(style, seed, t=0, w=640, h=480):
    return sf.Scene(w, h, style, seed).gray(t)


@pytest.fixture
def refine_mode(request):
    """Sets the oracle's LSD refinement mode for one test and restores the default (ADV) afterwards."""
    import oracle_lib
    mode = getattr(request, "param", ADV)
    oracle_lib.set_lsd_refine(mode)
    yield mode
    oracle_lib.set_lsd_refine(ADV)


def _extractor(mode, *a, **kw):
    import psl_slam_amd as P
    le = P.LINEextractor(*a, **kw)
    le.set_refine(mode)
    return le


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


@pytest.mark.parametrize("refine_mode", [ADV, STD], indirect=True)
@pytest.mark.parametrize("style,seed", [("struct", 3), ("desk", 4), ("struct", 8)])
def test_lsd_segments(style, seed, refine_mode):
    import oracle_lib
    img = _scene(style, seed)
    got = _extractor(refine_mode).lsd_detect(img)
    ref = oracle_lib.lsd_detect(img)
    exact = got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()
    print(f"LSD {style}/{seed} refine {refine_mode}: {len(got)} vs oracle {len(ref)} segments, recovered {_match_segments(got, ref):.4f}, bit-identical {exact}")
    assert len(ref) > (100 if refine_mode == STD else 40)
    assert exact


@pytest.mark.parametrize("refine_mode", [ADV, STD], indirect=True)
def test_line_stages_at_1280x960(refine_mode):
    """BASELINE configs[4]'s geometry (LSD on 1024x768): gradient stage, segment list, merge, LBD, pairing and the whole extractor,
    bit for bit."""
    import oracle_lib
    img = _scene("struct", 41, 0, 1280, 960)
    le = _extractor(refine_mode, 1, 1.2, 200, 0.0)
    seg = le.lsd_detect(img)
    ref_seg = oracle_lib.lsd_detect(img)
    assert len(ref_seg) > (200 if refine_mode == STD else 100) and seg.shape == ref_seg.shape and (seg.view(np.uint32) == ref_seg.view(np.uint32)).all()
    scaled, angdeg, mod = le.debug_gradient(0)
    rs, ra, rm = oracle_lib.lsd_gradient(img)
    np.testing.assert_array_equal(scaled, rs)
    gang = np.where(angdeg == -1024.0, -1024.0, angdeg.astype(np.float64) * DEG2RAD)
    np.testing.assert_array_equal(gang, ra)
    _kl_equal(le.optimize_and_merge(ref_seg, 1280, 960, cap=4096), oracle_lib.optimize_and_merge(ref_seg, 1280, 960, cap=4096), "merge 1280x960")
    ref = oracle_lib.line_extract(img, 200)
    _assert_extract_equal(le(img), ref, "line extract 1280x960")
    L = np.stack([ref[0][n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
    np.testing.assert_array_equal(le.pair(L, 20.0, np.float32(np.pi / 4), 1280, 960).view(np.uint32),
                                  oracle_lib.lil_pair(L, 20.0, np.float32(np.pi / 4), 1280, 960).view(np.uint32))
    assert len(ref[0]) > 20


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
    for mode in (ADV, STD):
        oracle_lib.set_lsd_refine(mode)
        try:
            ref = oracle_lib.lsd_detect(img)
        finally:
            oracle_lib.set_lsd_refine(ADV)
        got = _extractor(mode).lsd_detect(img)
        exact = got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()
        print(f"LSD ramps refine {mode}: {len(got)} vs oracle {len(ref)} segments, bit-identical {exact}")
        assert len(ref) >= (4 if mode == STD else 2) and exact


def test_lsd_reduce_region_radius_on_a_queue_of_most_of_the_image():
    """A noisy diagonal ramp on a small image: one region is most of the image and its rectangle, turned by 45 degrees, is twice its size - the
    density test fails, the refinement regrows it, and reduce_region_radius starts on a list of ~80 % of the pixels.  lsdw_refine's compaction by rank
    keeps a scratch list behind the queue; here list + removed pixels exceed the image (checked on the oracle's counts), so these steps take the walk that
    remains for that case, and the later, shorter ones the rank passes - segments bit for bit, both refine modes."""
    import oracle_lib
    cases = [(40, 40, 0.3, 0), (40, 40, 0.3, 9), (40, 40, 0.3, 14), (40, 40, 0.45, 5), (40, 40, 0.6, 2), (40, 40, 0.6, 3), (40, 40, 0.8, 6), (36, 36, 0.6, 21), (36, 36, 0.8, 17)]
    over = steps = 0
    for w, h, noise, seed in cases:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(254.0 / (w + h - 2) * (xx + yy) + np.random.default_rng(seed).normal(0, noise, (h, w)), 0, 255).astype(np.uint8)
        st = oracle_lib.lsd_refine_stats(img)
        over += int(st[8] > st[9])
        steps += int(st[5])
        for mode in (ADV, STD):
            oracle_lib.set_lsd_refine(mode)
            try:
                ref = oracle_lib.lsd_detect(img)
            finally:
                oracle_lib.set_lsd_refine(ADV)
            got = _extractor(mode).lsd_detect(img)
            assert got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all(), (w, h, noise, seed, mode)
    assert over == len(cases) and steps > 100, (over, steps)


@pytest.mark.parametrize("w,h", [(1536, 1152), (1600, 1200)])
def test_lsd_one_large_frame_on_both_sides_of_the_lds_map_limit(w, h):
    """A few-frames launch keeps the `used` bits of the scaled image in LDS when they fit beside the kernel's own arrays (144 KB: ~1.18 M pixels):
    1536x1152 scales to 1228x921 = 138 KB of bits (k_lsd_grow4<3, 1> with nearly all of the CU's LDS as dynamic shared memory), 1600x1200 to
    1280x960 = 150 KB (the map stays in memory: k_lsd_grow4<3, 0>).  Segments bit for bit on either side."""
    import oracle_lib
    img = _scene("struct", 23, 1, w, h)
    ref = oracle_lib.lsd_detect(img)
    got = _extractor(ADV).lsd_detect(img)
    assert len(ref) > 100 and got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()


def _adversarial_images():
    """Inputs that stress the queue order of the region growing rather than look like a room: rings (regions that turn and close on
    themselves), stripes of every thickness in both diagonals (frontiers several entries wide, growth up and to the left of the seed),
    smoothed noise (blobs, many tiny regions), a checker board (corners everywhere), raw noise, and a frame that touches all four
    borders."""
    import synth_frames as sf
    h, w = 300, 400
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = {}
    r = np.hypot(xx - 190.3, yy - 140.7)
    out["rings"] = np.clip(128 + 100 * np.sin(r / 3.1), 0, 255).astype(np.uint8)
    st = np.zeros((h, w))
    for k, (t, s) in enumerate([(1, 1), (2, -1), (3, 1), (5, -1), (8, 1), (13, -1)]):
        d = (xx + s * yy * (0.35 + 0.2 * k)) - (40 + 55 * k) - (0 if s > 0 else -120)
        st += 170.0 * (np.abs(d) < t)
    out["stripes"] = np.clip(30 + st, 0, 255).astype(np.uint8)
    out["blobs"] = sf.random_gray(w, h, 12, "blobs")
    out["checker"] = sf.random_gray(w, h, 13, "checker")
    out["noise"] = sf.random_gray(w, h, 14, "noise")
    fr = np.full((h, w), 60, np.uint8)
    fr[:3] = 250; fr[-3:] = 250; fr[:, :3] = 250; fr[:, -3:] = 250
    fr[40:44, :] = 200; fr[:, 100:103] = 10
    out["frame"] = fr
    # one region of 22 000 pixels whose breadth-first frontier is 85 entries behind the queue's end (tools/grow_stats.py): the queue's
    # LDS ring wraps 20 times; in a build with -DPSL_LSD_RING=128 the frontier is mapped from the HBM copy of the queue
    by, bx = np.mgrid[0:200, 0:640].astype(np.float64)
    out["band"] = np.rint(np.clip(4.4 * (by - 40 + 0.05 * np.abs(bx - 320)), 0, 255)).astype(np.uint8)
    return out


@pytest.mark.parametrize("refine_mode", [ADV, STD], indirect=True)
def test_lsd_adversarial_queue_orders(refine_mode):
    import oracle_lib
    for name, img in _adversarial_images().items():
        img = np.ascontiguousarray(img)
        ref = oracle_lib.lsd_detect(img)
        got = _extractor(refine_mode).lsd_detect(img)
        exact = got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all()
        print(f"LSD {name} refine {refine_mode}: {len(got)} vs oracle {len(ref)} segments, bit-identical {exact}")
        assert exact, name


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
    """optimizeAndMergeLines_lsd on identical input segments (the STD segment list: more lines, more merges): every KeyLine field
    bit-identical.  atanf / atan2f and the double sin / cos of MergeTwoLines are glibc's algorithms restated (header)."""
    import psl_slam_amd as P
    import oracle_lib
    img = _scene(style, seed)
    oracle_lib.set_lsd_refine(STD)
    try:
        seg = oracle_lib.lsd_detect(img)
    finally:
        oracle_lib.set_lsd_refine(ADV)
    ref = oracle_lib.optimize_and_merge(seg, 640, 480)
    got = P.LINEextractor().optimize_and_merge(seg, 640, 480)
    assert len(ref) > 10
    _kl_equal(got, ref, f"merge {style}/{seed}")
    print(f"merge {style}/{seed}: {len(seg)} segments -> {len(got)} keylines, bit-identical")


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
    assert got.shape == ref.shape and len(ref) > 3
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    # degenerate inputs: no lines, one line
    assert len(P.LINEextractor().pair(np.zeros((0, 4), np.float32), 20.0, 0.785, 640, 480)) == 0
    assert len(P.LINEextractor().pair(lines[:1], 20.0, 0.785, 640, 480)) == 0


def _assert_extract_equal(got, ref, what):
    gk, gd, ge = got
    rk, rd, re_ = ref
    _kl_equal(gk, rk, what)
    np.testing.assert_array_equal(gd, rd, err_msg=what)
    np.testing.assert_array_equal(ge.view(np.uint64), re_.view(np.uint64), err_msg=what)


@pytest.mark.parametrize("refine_mode", [ADV, STD], indirect=True)
@pytest.mark.parametrize("style,seed,t", [("struct", 3, 0), ("desk", 4, 0), ("struct", 8, 0), ("struct", 5, 0), ("desk", 7, 7), ("struct", 9, 15),
                                          ("desk", 11, 23), ("sticks", 13, 0), ("sticks", 14, 5)])
def test_full_line_extractor(style, seed, t, refine_mode):
    """LINEextractor::operator() bit for bit: keylines (all 17 fields), LBD descriptors, line equations.  struct/5 t=0 is the
    frame of round 1's stress-run mismatch (one LBD row, |dx| 1.2e-4, before atanf was restated); the next three are frames of
    that stress set; 'sticks' is the dense scene of the headline bench: ~220 refinements and ~150 reduce_region_radius calls with
    ~580 radius steps per frame (the compaction by rank of lsdw_refine; 'desk' has ~120 calls, 'struct' ~10)."""
    import oracle_lib
    img = _scene(style, seed, t)
    ref = oracle_lib.line_extract(img, 200)
    got = _extractor(refine_mode, 1, 1.2, 200, 0.0)(img)
    assert len(ref[0]) > (20 if refine_mode == STD else 8)
    _assert_extract_equal(got, ref, f"line extract {style}/{seed}/{t} refine {refine_mode}")


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
    """HIP vs tests/golden/line_640x480_struct.npz, everything exact: LSD segments (both refinement modes), LBD on the golden
    keylines, pairing on the golden lines and the end-to-end extractor."""
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
    assert k.tobytes() == g["kls"].tobytes() and (d == g["desc"]).all() and e.tobytes() == g["eq"].tobytes()
    le.set_refine(STD)
    np.testing.assert_array_equal(le.lsd_detect(g["image"]).view(np.uint32), g["segments_std"].view(np.uint32))


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


@pytest.mark.parametrize("w,h", [(633, 479), (321, 243), (1001, 437)])
def test_line_extractor_odd_image_sizes(w, h):
    """Sizes that are multiples of nothing: the scaled image's width is neither a multiple of the 64-pixel scan chunks nor of the
    256-pixel scan trips (a chunk straddles two rows), tiles are ragged on both edges, rows are unaligned.  One frame (helper waves,
    static singletons) and the same frame inside a 70-frame launch (neither) against the oracle, bit for bit."""
    import psl_slam_amd as P
    import oracle_lib
    img = np.ascontiguousarray(_scene("struct", 6, 3, 1280, 960)[100:100 + h, 200:200 + w])
    ref_seg = oracle_lib.lsd_detect(img)
    ref = oracle_lib.line_extract(img, 200)
    le = P.LINEextractor(1, 1.2, 200, 0.0)
    seg = le.lsd_detect(img)
    assert seg.shape == ref_seg.shape and (seg.view(np.uint32) == ref_seg.view(np.uint32)).all()
    k1, d1, e1 = le(img)
    assert k1.tobytes() == ref[0].tobytes() and (d1 == ref[1]).all() and e1.tobytes() == ref[2].tobytes()
    assert len(ref[0]) > 5
    nb = 70
    frames = np.ascontiguousarray(np.stack([img] * nb, 0))
    lb = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=nb)
    d_ptr, _ = lb.ctx.device_array(frames)
    lb.extract_batch_device(d_ptr, nb, w, h, w, w * h)
    for f in (0, 33, 69):
        k, dsc, eq, st = lb.fetch(f)
        assert st == 0 and k.tobytes() == k1.tobytes() and (dsc == d1).all() and eq.tobytes() == e1.tobytes(), f
    lb.ctx.device_free(d_ptr)


@pytest.mark.parametrize("nscenes", [2, 4])
def test_merge_stage_more_segments_than_the_small_lds_instance(nscenes):
    """k_line_merge exists for 512 and 1024 lines in LDS, beyond that its working set is in HBM: segment lists of ~700 and
    ~1400 lines (several scenes' STD segments in one list) exercise the other two paths; bit-identical as above."""
    import psl_slam_amd as P
    import oracle_lib
    oracle_lib.set_lsd_refine(STD)
    try:
        seg = np.concatenate([oracle_lib.lsd_detect(_scene(st, sd)) for st, sd in (("desk", 4), ("struct", 3), ("desk", 9), ("desk", 12))[:nscenes]])
    finally:
        oracle_lib.set_lsd_refine(ADV)
    assert len(seg) > (512 if nscenes == 2 else 1024)
    ref = oracle_lib.optimize_and_merge(seg, 640, 480, cap=4096)
    got = P.LINEextractor().optimize_and_merge(seg, 640, 480, cap=4096)
    assert len(ref) > 100
    _kl_equal(got, ref, f"merge of {len(seg)} segments")


@pytest.mark.parametrize("refine_mode", [ADV, STD], indirect=True)
def test_lsd_segments_bit_identical_over_many_frames(refine_mode):
    """The region growing decides most neighbours against an angle that is NOT up to date (a rigorous bound on its drift
    replaces the per-pixel fastAtan2, line_kernels.h), and the NFA validation runs as its own kernel with restated log / exp
    (line_kernels3.h): every shortcut must give the reference's decision, so the segment lists of many different frames are
    compared bit for bit (textured and structure-like scenes, several time steps)."""
    import oracle_lib
    le = _extractor(refine_mode)
    nseg = 0
    for style, seed in (("desk", 31), ("struct", 32), ("desk", 33), ("struct", 34)):
        sc = sf.Scene(640, 480, style, seed)
        for t in range(0, 12, 2):
            img = sc.gray(t)
            got = le.lsd_detect(img)
            ref = oracle_lib.lsd_detect(img)
            assert got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all(), (style, seed, t)
            nseg += len(ref)
    assert nseg > (5000 if refine_mode == STD else 2000)


def test_full_size_line_batch_properties():
    """768 frames in one launch (6 LSD waves per SIMD on 128 CUs' worth of workgroups, XCD-aware grids, both merge instances):
    96 distinct frames x 8; twins of a frame are byte-identical, a second run of the batch is byte-identical, and 16 sampled
    frames equal the CPU oracle bit for bit (keylines, descriptors, line equations, fans)."""
    import zlib
    import oracle_lib
    import psl_slam_amd as P
    frames = []
    for style, seed in (("struct", 5), ("desk", 7), ("struct", 9), ("desk", 11)):
        sc = sf.Scene(640, 480, style, seed)
        frames += [sc.gray(t) for t in range(24)]
    frames = np.stack(frames, 0)
    n = 768
    batch = np.ascontiguousarray(np.concatenate([frames] * (n // len(frames)), 0))
    le = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=n)
    d_ptr, _ = le.ctx.device_array(batch)
    crcs = []
    for rep in range(2):
        le.extract_batch_device(d_ptr, n, 640, 480, 640, 640 * 480)
        le.pair_batch_device(20.0, np.float32(np.pi / 4))
        per = []
        for f in range(n):
            k, dsc, eq, st = le.fetch(f)
            assert st == 0
            per.append(zlib.crc32(k.tobytes()) ^ zlib.crc32(dsc.tobytes()) ^ zlib.crc32(eq.tobytes()) ^ zlib.crc32(le.fans_fetch(f).tobytes()))
        crcs.append(per)
    assert crcs[0] == crcs[1], "a second run of the same batch differs"
    for f in range(len(frames), n):
        assert crcs[0][f] == crcs[0][f % len(frames)], f"frame {f} differs from its twin {f % len(frames)}"
    for f in np.linspace(0, n - 1, 16).astype(int):
        f = int(f)
        k, dsc, eq, _ = le.fetch(f)
        _assert_extract_equal((k, dsc, eq), oracle_lib.line_extract(batch[f], 200), f"batch frame {f}")
        L = np.stack([k[m] for m in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
        np.testing.assert_array_equal(le.fans_fetch(f).view(np.uint32), oracle_lib.lil_pair(L, 20.0, np.float32(np.pi / 4), 640, 480).view(np.uint32))
    le.ctx.device_free(d_ptr)


def test_line_extractor_many_random_scenes_in_one_batch():
    """40 frames of 40 different scenes (textured and structure-like, random time steps) in one launch, every one compared with the
    oracle bit for bit: keylines, LBD rows, line equations and fans."""
    import oracle_lib
    import psl_slam_amd as P
    rng = np.random.default_rng(2024)
    frames = np.stack([_scene("desk" if i % 3 == 0 else "struct", 100 + i, int(rng.integers(0, 30))) for i in range(40)], 0)
    le = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=len(frames))
    d_ptr, _ = le.ctx.device_array(frames)
    le.extract_batch_device(d_ptr, len(frames), 640, 480, 640, 640 * 480)
    le.pair_batch_device(20.0, np.float32(np.pi / 4))
    nkl = 0
    for f in range(len(frames)):
        k, dsc, eq, st = le.fetch(f)
        assert st == 0
        _assert_extract_equal((k, dsc, eq), oracle_lib.line_extract(frames[f], 200), f"scene {f}")
        L = np.stack([k[m] for m in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32) if len(k) else np.zeros((0, 4), np.float32)
        np.testing.assert_array_equal(le.fans_fetch(f).view(np.uint32), oracle_lib.lil_pair(L, 20.0, np.float32(np.pi / 4), 640, 480).view(np.uint32))
        nkl += len(k)
    assert nkl > 400
    le.ctx.device_free(d_ptr)
