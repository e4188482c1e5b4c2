"""Failure paths of the C ABI on the GPU: a prepare() whose allocations fail must leave the object empty - the next call with
the same geometry has to fail again (or allocate cleanly), never launch kernels on freed or undersized buffers."""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu


def test_line_prepare_allocation_failure_is_clean_and_repeatable():
    import psl_slam_amd as P
    import oracle_lib
    img = sf.Scene(640, 480, "struct", 3).gray(0)
    le = P.LINEextractor(1, 1.2, 200, 0.0, max_batch=65535)   # ~17 MB of working buffers per frame: > 1 TB, cannot be allocated
    for _ in range(2):
        with pytest.raises(P.PslfeError):
            le(img)
    with pytest.raises(P.PslfeError):
        le.lsd_detect(img)
    le.close()
    ok = P.LINEextractor(1, 1.2, 200, 0.0)                    # the context is still healthy
    k, d, e = ok(img)
    rk, rd, re_ = oracle_lib.line_extract(img, 200)
    assert k.tobytes() == rk.tobytes() and (d == rd).all()


def test_orb_prepare_allocation_failure_is_clean_and_repeatable():
    import psl_slam_amd as P
    import oracle_lib
    small = sf.Scene(640, 480, "desk", 3).gray(0)
    big = sf.Scene(1280, 960, "desk", 3).gray(0)
    orb = P.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=60000)  # 1280x960: ~18 MB per frame -> > 1 TB
    for _ in range(2):
        with pytest.raises(P.PslfeError):
            orb(big)
    orb.close()
    ok = P.ORBextractor(1000, 1.2, 8, 20, 7)
    k, d = ok(small)
    rk, rd = oracle_lib.OracleORB(1000, 1.2, 8, 20, 7)(small)
    assert k.tobytes() == rk.tobytes() and (d == rd).all()
    # a geometry change after a successful prepare, then back: results unchanged
    k2, d2 = ok(big)
    k3, d3 = ok(small)
    assert len(k2) > 0 and k3.tobytes() == k.tobytes() and (d3 == d).all()
