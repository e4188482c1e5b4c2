"""GPU parity of the ORB extraction path: HIP (through the C ABI) vs the CPU oracle, bit-exact.

Stage order follows src/ORBextractor.cc: ComputePyramid -> FAST per cell -> DistributeOctTree ->
IC_Angle -> GaussianBlur -> rBRIEF.  Every comparison is exact (integer / index work; the angle is
an f32 computed from exact integer moments with a non-contracted polynomial).
"""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu

CFG = dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)


def _pair(max_batch=1, **kw):
    import psl_slam_amd as P
    import oracle_lib
    cfg = dict(CFG)
    cfg.update(kw)
    return P.ORBextractor(max_batch=max_batch, **cfg), oracle_lib.OracleORB(**cfg)


def _assert_same_frame(got, ref, what=""):
    gk, gd = got
    rk, rd = ref
    assert len(gk) == len(rk), f"{what}: {len(gk)} keypoints vs oracle {len(rk)}"
    for name in ("octave", "class_id", "x", "y", "size", "response", "angle"):
        a, b = gk[name], rk[name]
        bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32)) if a.dtype.kind == "f" else np.flatnonzero(a != b)
        assert bad.size == 0, f"{what}: field {name} differs at {bad[:5]}: {a[bad[:5]]} vs {b[bad[:5]]}"
    bad = np.flatnonzero((gd != rd).any(1))
    assert bad.size == 0, f"{what}: {bad.size} descriptors differ, first {bad[:5]}"


@pytest.fixture(scope="module")
def desk():
    return sf.Scene(640, 480, "desk").gray(0)


def test_stages_640x480(desk):
    gpu, orc = _pair()
    got = gpu(desk)
    ref = orc(desk)
    for l in range(8):
        np.testing.assert_array_equal(gpu.debug_level_image(0, l), orc.level_image(l), err_msg=f"pyramid level {l}")
    for l in range(8):
        gc, rc = gpu.debug_candidates(0, l), orc.candidates(l)
        assert gc.shape == rc.shape, f"FAST level {l}: {len(gc)} vs {len(rc)} candidates"
        np.testing.assert_array_equal(gc, rc, err_msg=f"FAST candidates level {l}")
    for l in range(8):
        gk = gpu.debug_level_keypoints(0, l)
        rk = orc.level_keypoints(l)
        assert len(gk) == len(rk), f"octree level {l}: {len(gk)} vs {len(rk)}"
        np.testing.assert_array_equal(gk[:, 0] + 16, rk["x"].astype(np.int32), err_msg=f"octree x level {l}")
        np.testing.assert_array_equal(gk[:, 1] + 16, rk["y"].astype(np.int32), err_msg=f"octree y level {l}")
        np.testing.assert_array_equal(gk[:, 2], rk["response"].astype(np.int32), err_msg=f"octree response level {l}")
    for l in range(8):
        rb = orc.level_image(l, blurred=True)
        if rb is not None:
            np.testing.assert_array_equal(gpu.debug_level_image(0, l, blurred=True), rb, err_msg=f"blur level {l}")
    _assert_same_frame(got, ref, "desk frame")
    assert 900 <= len(got[0]) <= gpu.max_keypoints(640, 480)


@pytest.mark.parametrize("kind", ["noise", "flat", "checker", "blobs"])
def test_adversarial_images(kind):
    img = sf.random_gray(640, 480, 7, kind)
    gpu, orc = _pair()
    _assert_same_frame(gpu(img), orc(img), kind)


def test_struct_scene_and_strided_input():
    img = sf.Scene(640, 480, "struct", seed=99).gray(3)
    gpu, orc = _pair()
    big = np.zeros((480, 700), np.uint8)
    big[:, :640] = img
    _assert_same_frame(gpu(big[:, :640]), orc(img), "strided")


@pytest.mark.parametrize("w,h,nf,nl", [(320, 240, 500, 4), (752, 480, 1200, 8), (1280, 960, 2000, 8), (333, 517, 300, 3)])
def test_other_geometries(w, h, nf, nl):
    img = sf.Scene(w, h, "desk", seed=w * 7 + h).gray(1)
    gpu, orc = _pair(nfeatures=nf, nlevels=nl)
    _assert_same_frame(gpu(img), orc(img), f"{w}x{h}")


def test_batch_equals_single_and_oracle():
    frames = sf.stream(6, 640, 480, "desk", seed=5)
    gpu, orc = _pair(max_batch=6)
    res = gpu.extract_batch(frames)
    for f in range(6):
        _assert_same_frame(res[f], orc(frames[f]), f"batch frame {f}")


def test_empty_and_error_behaviour():
    import psl_slam_amd as P
    gpu, _ = _pair()
    k, d = gpu(np.zeros((0, 0), np.uint8))          # src/ORBextractor.cc:1046: empty image -> nothing
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = gpu(np.full((480, 640), 90, np.uint8))   # no corners -> 0 keypoints (descriptors released, :1064)
    assert len(k) == 0
    with pytest.raises(P.PslfeError):
        gpu(np.zeros((100, 100), np.uint8))         # level 7 smaller than a FAST cell: reference UB -> error code
    with pytest.raises(P.PslfeError):
        P.ORBextractor(0, 1.2, 8, 20, 7)


def test_getters_match_reference_tables():
    gpu, orc = _pair()
    assert gpu.GetLevels() == 8
    assert list(gpu.features_per_level()) == [217, 181, 151, 126, 105, 87, 73, 60] == orc.quota()
    s = gpu.GetScaleFactors()
    assert s[0] == 1.0 and abs(s[7] - 1.2 ** 7) < 1e-5
    np.testing.assert_array_equal(gpu.GetInverseScaleFactors(), (np.float32(1.0) / s).astype(np.float32))


@pytest.mark.parametrize("name", ["orb_640x480_desk", "orb_640x480_struct", "orb_320x240_desk"])
def test_hip_reproduces_committed_golden_vectors(name):
    """HIP path vs the fixtures under tests/golden/ (made by the oracle; make_golden.py)."""
    import os
    import psl_slam_amd as P
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    nf, nl, ini, mn = [int(v) for v in g["cfg"]]
    gpu = P.ORBextractor(nf, 1.2, nl, ini, mn)
    kps, desc = gpu(g["image"])
    assert kps.tobytes() == g["kps"].tobytes()
    np.testing.assert_array_equal(desc, g["desc"])
    assert [len(gpu.debug_candidates(0, l)) for l in range(nl)] == list(g["ncand"])


def test_full_size_batch_properties():
    """BASELINE full size (256 frames of 640x480 per launch): size-independent properties — every
    copy of a frame inside the batch gives the identical result (no cross-frame leakage), counts stay
    within quota + overshoot, octaves are sorted, and frame 0 equals the single-frame path."""
    import psl_slam_amd as P
    base = sf.stream(8, 640, 480, "desk", seed=21)
    frames = np.ascontiguousarray(np.concatenate([base] * 32, 0))
    gpu = P.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=256)
    res = gpu.extract_batch(frames)
    single = P.ORBextractor(1000, 1.2, 8, 20, 7)
    k0, d0 = single(base[0])
    assert res[0][0].tobytes() == k0.tobytes() and (res[0][1] == d0).all()
    cap = gpu.max_keypoints(640, 480)
    for f in range(256):
        k, d = res[f]
        kb, db = res[f % 8]
        assert k.tobytes() == kb.tobytes() and (d == db).all(), f"frame {f} differs from its twin {f % 8}"
        assert 0 < len(k) <= cap and (np.diff(k["octave"]) >= 0).all()


def test_xcd_grid_with_a_ragged_last_group():
    """11 frames: the many-frames launches use the XCD-aware grid (8, items, ceil(F / 8)), whose last group holds 3
    frames and 5 idle XCD slots; every frame must still equal the oracle."""
    frames = sf.stream(11, 640, 480, "desk", seed=9)
    gpu, orc = _pair(max_batch=11)
    res = gpu.extract_batch(frames)
    for f in range(11):
        _assert_same_frame(res[f], orc(frames[f]), f"xcd batch frame {f}")
