"""GPU parity of the conversions either side of the extractors vs the CPU oracle:
RGB/BGR->gray, depth u16->f32, UndistortKeyPoints + ComputeStereoFromRGBD + ComputeImageBounds + grid.
All comparisons are exact (integer arithmetic, or IEEE double/float in the reference's operation order)."""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu

# Examples/RGB-D/TUM1.yaml (non-zero distortion) and TUM3.yaml (zero distortion)
TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0)
TUM3 = (535.4, 539.2, 320.1, 247.6, 0.0, 0.0, 0.0, 0.0, 0.0, 40.0)


def _cam(P, vals):
    c = np.zeros((), P.CAMERA_DTYPE)
    for k, v in zip(P.CAMERA_DTYPE.names, vals):
        c[k] = np.float32(v)
    return c


@pytest.mark.parametrize("is_rgb", [True, False])
@pytest.mark.parametrize("shape", [(480, 640), (37, 53), (1, 1)])
def test_rgb_to_gray_exact(is_rgb, shape):
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    np.testing.assert_array_equal(P.rgb_to_gray(rgb, is_rgb), oracle_lib.rgb_to_gray(rgb, is_rgb))


def test_rgb_to_gray_all_channel_extremes():
    import psl_slam_amd as P
    import oracle_lib
    v = np.array([0, 1, 127, 128, 254, 255], np.uint8)
    rgb = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, -1, 3)
    np.testing.assert_array_equal(P.rgb_to_gray(rgb, True), oracle_lib.rgb_to_gray(rgb, True))
    assert P.rgb_to_gray(np.full((2, 2, 3), 255, np.uint8))[0, 0] == 255


@pytest.mark.parametrize("n", [640 * 480, 1, 7])
def test_depth_to_float_exact(n):
    import psl_slam_amd as P
    import oracle_lib
    rng = np.random.default_rng(6)
    d = rng.integers(0, 65536, n, dtype=np.uint16)
    d[:1] = 0
    f = np.float32(1.0 / 5000.0)
    got, ref = P.depth_to_float(d, f), oracle_lib.depth_to_float(d, f)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("camvals", [TUM1, TUM3], ids=["tum1", "tum3"])
def test_frame_rgbd_post_exact(camvals):
    import psl_slam_amd as P
    import oracle_lib
    cam = _cam(P, camvals)
    sc = sf.Scene(640, 480, "desk", seed=11)
    kps, desc = oracle_lib.OracleORB()(sc.gray(0))
    depth = oracle_lib.depth_to_float(sc.depth_u16(0), np.float32(1.0 / 5000.0))
    depth[100:200, 100:300] = 0.0  # holes: mvDepth = mvuRight = -1
    g = P.FrameGrid(2048, 2)
    np.testing.assert_array_equal(g.image_bounds(cam, 640, 480).view(np.uint32), oracle_lib.image_bounds(cam, 640, 480).view(np.uint32))
    g.set_rgbd(1, kps, desc, depth, cam)
    un, dep, ur = g.fetch(1)
    run, rdep, rur = oracle_lib.frame_post_rgbd(kps, depth, cam)
    assert un.tobytes() == run.tobytes()
    np.testing.assert_array_equal(dep.view(np.uint32), rdep.view(np.uint32))
    np.testing.assert_array_equal(ur.view(np.uint32), rur.view(np.uint32))
    assert (dep < 0).sum() > 0 and (dep > 0).sum() > 0
    b = oracle_lib.image_bounds(cam, 640, 480)
    start, idx = g.debug_grid(1)
    rstart, ridx = oracle_lib.grid_build(run, tuple(float(x) for x in b))
    np.testing.assert_array_equal(start, rstart)
    np.testing.assert_array_equal(idx, ridx)
    if camvals is TUM1:
        assert np.abs(un["x"] - kps["x"]).max() > 1.0  # the distortion really moves points


def test_frame_rgbd_feeds_matcher():
    """mvuRight produced on the device takes part in SearchByProjection exactly as a host-provided one."""
    import psl_slam_amd as P
    import oracle_lib
    from test_match_gpu import make_queries
    cam = _cam(P, TUM1)
    sc = sf.Scene(640, 480, "desk", seed=11)
    kps, desc = oracle_lib.OracleORB()(sc.gray(0))
    depth = oracle_lib.depth_to_float(sc.depth_u16(0), np.float32(1.0 / 5000.0))
    g = P.FrameGrid(2048, 1)
    g.set_rgbd(0, kps, desc, depth, cam)
    un, dep, ur = g.fetch(0)
    q, qd = make_queries(un, desc, np.random.default_rng(3), with_ur=True)
    q["ur"] = ur + np.random.default_rng(4).uniform(-30, 30, len(ur)).astype(np.float32)
    b = tuple(float(x) for x in oracle_lib.image_bounds(cam, 640, 480))
    nm, match, assigned = P.ORBmatcher(0.9, True).SearchByProjectionLast(g, 0, q, qd)
    rnm, rmatch, rassigned = oracle_lib.search_by_projection_last(un, desc, ur, b, q, qd, None, True)
    assert nm == rnm
    np.testing.assert_array_equal(match, rmatch)
    np.testing.assert_array_equal(assigned, rassigned)


def test_frame_rgbd_batch_from_orb():
    import psl_slam_amd as P
    import oracle_lib
    cam = _cam(P, TUM1)
    sc = sf.Scene(640, 480, "desk", seed=12)
    frames = np.stack([sc.gray(t) for t in range(3)], 0)
    depth = np.stack([oracle_lib.depth_to_float(sc.depth_u16(t), np.float32(1.0 / 5000.0)) * np.float32(1 + 0.1 * t) for t in range(3)], 0)
    ctx = P.default_context()
    orb = P.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=3)
    d_gray, _ = ctx.device_array(frames)
    d_depth, _ = ctx.device_array(depth)
    orb.extract_batch_device(d_gray, 3, 640, 480, 640, 640 * 480)
    g = P.FrameGrid(orb.max_keypoints(640, 480), 3)
    g.set_from_orb_rgbd(orb, d_depth, 640, 480, cam)
    orc = oracle_lib.OracleORB()
    for t in range(3):
        kps, desc = orc(frames[t])
        un, dep, ur = g.fetch(t)
        run, rdep, rur = oracle_lib.frame_post_rgbd(kps, depth[t], cam)
        assert un.tobytes() == run.tobytes()
        np.testing.assert_array_equal(dep.view(np.uint32), rdep.view(np.uint32))
        np.testing.assert_array_equal(ur.view(np.uint32), rur.view(np.uint32))
    ctx.synchronize()
    ctx.device_free(d_gray)
    ctx.device_free(d_depth)
