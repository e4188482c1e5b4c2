"""Oracle of the RGB-D line glue (oracle/glue_oracle.cpp) vs independent checks: the restated glibc rand() against the
host's libc, 3-D lines against the planes the depth image was rendered from, planes against those planes."""
import ctypes as C

import numpy as np

import glue_scene
import oracle_lib


def test_glibc_rand_matches_host_libc():
    libc = C.CDLL("libc.so.6")
    for seed in (1, 0, 7, 20250418, 0xFFFFFFFF):
        libc.srand(C.c_uint(seed))
        ref = np.array([libc.rand() for _ in range(2000)], np.int32)
        np.testing.assert_array_equal(oracle_lib.glibc_rand(seed, 2000), ref)


def _plane_residual(P, planes):
    return min(abs(np.dot(n, P) - d) / np.linalg.norm(n) for n, d in planes)


def test_lines3d_lie_on_the_rendered_planes_and_project_onto_their_keylines():
    kls, fans, depth, cam, planes = glue_scene.scene()
    r = oracle_lib.frame_glue(kls, fans, depth, cam, seed=1)
    good = np.abs(r["lines3d"]).sum(1) > 0
    assert 20 < good.sum() < len(kls)  # most lines are good, the very short / hole-covered ones are not
    np.testing.assert_array_equal(r["lineEq"][~good], -1.0)
    fx, fy, cx, cy = [float(cam[k]) for k in ("fx", "fy", "cx", "cy")]
    for i in np.nonzero(good)[0]:
        A, B = r["lines3d"][i, :3], r["lines3d"][i, 3:]
        assert np.linalg.norm(A - B) > 0.02
        for P in (A, B):
            assert P[2] > 0.3
            # end points are depth samples: on one of the walls up to the depth noise
            assert _plane_residual(P, planes) < 0.06
            # and they back-project onto the 2-D segment (nearest-pixel sampling: within ~1.5 px of the line)
            u, v = fx * P[0] / P[2] + cx, fy * P[1] / P[2] + cy
            s = np.array([kls["startPointX"][i], kls["startPointY"][i]])
            e = np.array([kls["endPointX"][i], kls["endPointY"][i]])
            t = np.clip(np.dot([u, v] - s, e - s) / np.dot(e - s, e - s), 0, 1)
            assert np.linalg.norm(s + t * (e - s) - [u, v]) < 1.6
        np.testing.assert_allclose(np.linalg.norm(r["lineEq"][i]), 1.0, atol=1e-6)
        np.testing.assert_allclose(r["lineEq"][i], ((B - A) / np.linalg.norm(B - A)).astype(np.float32), atol=1e-6)


def test_crossings_and_planes():
    kls, fans, depth, cam, planes = glue_scene.scene()
    r = oracle_lib.frame_glue(kls, fans, depth, cam, seed=1)
    assert len(r["pair"]) > 0 and len(r["planes"]) > 0
    # crossings keep the fan order and only pair good lines
    good = np.abs(r["lines3d"]).sum(1) > 0
    k = 0
    for f in fans:
        if k < len(r["pair"]) and (int(f[2]), int(f[3])) == tuple(r["pair"][k]) and np.allclose(r["xy"][k], f[:2]):
            k += 1
    assert k == len(r["pair"])
    # every accepted plane is one of the two walls (up to sign) and its five points are coplanar within 5 cm
    for pl, nn, ln, c3 in zip(r["planes"], r["normals"], r["lineNo"], r["cross3d"]):
        assert pl[3] >= 0
        np.testing.assert_allclose(np.linalg.norm(pl[:3]), 1.0, atol=1e-5)
        np.testing.assert_allclose(pl[:3], nn.astype(np.float32), atol=1e-7)
        assert good[ln[0]] and good[ln[1]]
        pts = np.concatenate([r["lines3d"][ln[0]].reshape(2, 3), r["lines3d"][ln[1]].reshape(2, 3), c3[None]], 0)
        d = pts @ nn
        assert d.max() - d.min() <= 0.0501
    # no two accepted planes are "the same" by Frame::OldPlane
    P = r["planes"]
    for i in range(len(P)):
        for j in range(i):
            same = abs(P[i][3] - P[j][3]) <= 0.2 and abs(np.dot(P[i][:3], P[j][:3])) >= 0.9397
            assert not same
    # mvle_l: normalised 2-D line equations through the keylines' end points
    for (l1, l2), le in zip(r["pair"], r["le_l"]):
        for s, l in ((0, l1), (1, l2)):
            a = le[3 * s:3 * s + 3]
            np.testing.assert_allclose(np.hypot(a[0], a[1]), 1.0, atol=1e-12)
            for x, y in ((kls["startPointX"][l], kls["startPointY"][l]), (kls["endPointX"][l], kls["endPointY"][l])):
                assert abs(a[0] * x + a[1] * y + a[2]) < 1e-6 * max(1.0, abs(a[2]))


def test_seed_changes_the_ransac_stream_but_not_the_contract():
    kls, fans, depth, cam, _ = glue_scene.scene()
    a = oracle_lib.frame_glue(kls, fans, depth, cam, seed=1)
    b = oracle_lib.frame_glue(kls, fans, depth, cam, seed=1)
    c = oracle_lib.frame_glue(kls, fans, depth, cam, seed=2)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["lines3d"], c["lines3d"])  # different samples, different end points somewhere
    ga, gc = np.abs(a["lines3d"]).sum(1) > 0, np.abs(c["lines3d"]).sum(1) > 0
    assert (ga == gc).mean() > 0.9


def test_glue_golden_vectors():
    """The committed fixture (tests/golden/glue_640x480_corner.npz, made by make_golden.py from the oracle) still holds."""
    import os
    import zlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_640x480_corner.npz"))
    kls, fans, depth, cam, _ = glue_scene.scene(seed=3)
    assert zlib.crc32(depth.tobytes()) == int(g["depth_crc"])
    assert kls.tobytes() == g["kls"].tobytes() and fans.tobytes() == g["fans"].tobytes()
    r = oracle_lib.frame_glue(kls, fans, depth, cam, seed=int(g["seed"]))
    for k, v in r.items():
        assert v.tobytes() == g["out_" + k].tobytes(), k
