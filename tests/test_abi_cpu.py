"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol
include/pslfe.h declares, has the POD layouts the reference types have, and fails loudly (no CPU
fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pslfe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pslfe_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import psl_slam_amd as P
    P.build()
    lib = P.lib()
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/pslfe.h but not exported: {missing}"
    assert lib.pslfe_version().decode().startswith("pslfe")


def test_pod_layouts_match_reference_types():
    import psl_slam_amd as P
    assert P.KEYPOINT_DTYPE.itemsize == 28      # cv::KeyPoint
    assert P.KEYLINE_DTYPE.itemsize == 68       # line_descriptor::KeyLine (descriptor_custom.hpp:107-146)
    assert P.KEYLINE_DTYPE.names[:3] == ("angle", "class_id", "octave") and P.KEYLINE_DTYPE.names[-1] == "numOfPixels"
    assert P.PROJQUERY_DTYPE.itemsize == 32


def test_no_cpu_fallback():
    import psl_slam_amd as P
    lib = P.lib()
    h = C.c_void_p()
    rc = lib.pslfe_ctx_create(C.c_int(0), C.byref(h))
    if rc == 0:
        lib.pslfe_ctx_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == -2 and b"no CPU fallback" in lib.pslfe_last_error()
    with pytest.raises(P.PslfeError):
        P.Context(0)


def test_null_arguments_return_error_codes():
    import psl_slam_amd as P
    lib = P.lib()
    assert lib.pslfe_ctx_create(C.c_int(0), None) == -1
    assert lib.pslfe_orb_create(None, 1000, C.c_float(1.2), 8, 20, 7, 1, None) == -1
    assert lib.pslfe_orb_levels(None) == -1
    assert lib.pslfe_frame_create(None, 10, 1, None) == -1
    n = C.c_int(5)
    assert lib.pslfe_hamming_knn2(None, None, 0, None, 0, None, None) == -1
    lib.pslfe_orb_destroy(None)
    lib.pslfe_frame_destroy(None)
    lib.pslfe_ctx_destroy(None)
