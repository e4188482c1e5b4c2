"""GPU parity of the RGB-D line glue (isLineGood with the 3-D RANSAC, convertFansToKeyLines, the
plane-from-pair loop) vs the sequential CPU oracle (oracle/glue_oracle.cpp).  Device and oracle execute the same IEEE
double statements in the same order, so every output - including the discrete ones that depend on the rand() stream - is
compared exactly."""
import numpy as np
import pytest

import glue_scene

pytestmark = pytest.mark.gpu

KEYS = ("lines3d", "lineEq", "pair", "xy", "cross", "le_l", "planes", "normals", "lineNo", "cross3d", "cross2d")


def _same(a, b):
    for k in KEYS:
        assert a[k].shape == b[k].shape, k
        assert a[k].tobytes() == b[k].tobytes(), k


@pytest.mark.parametrize("seed", [1, 2, 12345])
def test_frame_glue_exact(seed):
    import psl_slam_amd as P
    import oracle_lib
    kls, fans, depth, cam, _ = glue_scene.scene(seed=3)
    g = P.FrameGlue(max_lines=256, max_fans=512)
    got = g.run(kls, fans, depth, cam, seed=seed)
    ref = oracle_lib.frame_glue(kls, fans, depth, cam, seed=seed)
    assert (np.abs(ref["lines3d"]).sum(1) > 0).sum() > 20 and len(ref["planes"]) > 0
    _same(got, ref)


def test_frame_glue_edge_cases():
    import psl_slam_amd as P
    import oracle_lib
    kls, fans, depth, cam, _ = glue_scene.scene(seed=5, nlines=40, nfans=60)
    g = P.FrameGlue(max_lines=64, max_fans=64)
    # no depth at all: every line fails, no crossings, no planes
    z = np.zeros_like(depth)
    got = g.run(kls, fans, z, cam)
    _same(got, oracle_lib.frame_glue(kls, fans, z, cam))
    assert np.all(got["lines3d"] == 0) and np.all(got["lineEq"] == -1) and len(got["pair"]) == 0 and len(got["planes"]) == 0
    # no fans / no lines
    got = g.run(kls, np.zeros((0, 4), np.float32), depth, cam)
    _same(got, oracle_lib.frame_glue(kls, np.zeros((0, 4), np.float32), depth, cam))
    got = g.run(kls[:0], np.zeros((0, 4), np.float32), depth, cam)
    assert len(got["lines3d"]) == 0 and len(got["planes"]) == 0
    # a fan that refers to a line that does not exist is refused, as the reference would index out of range
    bad = fans.copy()
    bad[0, 2] = 1000
    with pytest.raises(P.PslfeError):
        g.run(kls, bad, depth, cam)


def test_frame_glue_batch_device():
    import psl_slam_amd as P
    import oracle_lib
    ctx = P.default_context()
    F, ML, MF = 3, 64, 128
    scenes = [glue_scene.scene(seed=10 + f, nlines=50 + f, nfans=100 + 5 * f) for f in range(F)]
    kls = np.zeros((F, ML), P.KEYLINE_DTYPE)
    fans = np.zeros((F, MF, 4), np.float32)
    nkl, nfans = np.zeros(F, np.int32), np.zeros(F, np.int32)
    depth = np.zeros((F, glue_scene.H, glue_scene.W), np.float32)
    for f, (k, fa, d, cam, _) in enumerate(scenes):
        kls[f, :len(k)] = k
        fans[f, :len(fa)] = fa
        nkl[f], nfans[f] = len(k), len(fa)
        depth[f] = d
    d_kls, _ = ctx.device_array(kls)
    d_fans, _ = ctx.device_array(fans)
    d_nkl, _ = ctx.device_array(nkl)
    d_nfans, _ = ctx.device_array(nfans)
    d_depth, _ = ctx.device_array(depth)
    g = P.FrameGlue(max_lines=ML, max_fans=MF, max_batch=F)
    g.run_batch_device(F, d_kls, ML, d_nkl, d_fans, MF, d_nfans, d_depth, glue_scene.W, glue_scene.H, scenes[0][3], seed0=7)
    for f, (k, fa, d, cam, _) in enumerate(scenes):
        _same(g.fetch(f, len(k)), oracle_lib.frame_glue(k, fa, d, cam, seed=7 + f))
    ctx.synchronize()
    for p in (d_kls, d_fans, d_nkl, d_nfans, d_depth):
        ctx.device_free(p)


def test_hip_vs_committed_glue_golden():
    import os
    import psl_slam_amd as P
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_640x480_corner.npz"))
    _, _, depth, cam, _ = glue_scene.scene(seed=3)
    got = P.FrameGlue(max_lines=256, max_fans=512).run(g["kls"], g["fans"], depth, cam, seed=int(g["seed"]))
    for k in KEYS:
        assert got[k].tobytes() == g["out_" + k].tobytes(), k
