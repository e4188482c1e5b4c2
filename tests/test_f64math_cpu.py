"""Pins the product's restated libm functions (psl-slam_amd/csrc/psl_f64math.h, psl_sincos64.h) against this host's glibc,
and the NFA arithmetic of LSD_REFINE_ADV in the oracle against a plain-Python restatement.  No GPU."""
import math
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import synth_frames as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def check_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("f64") / "f64math_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "oracle", "f64math_check.c"), "-lm", "-lpthread"],
                   check=True)
    return exe


def test_restated_tanf_equals_libm_for_every_float_in_range(check_exe):
    """CPartiallyRecoverConnectivity's `abs(tan(arcAng)) > 1` (add_src/PartiallyRecoverConnectivity.cpp:39): arcAng is a float
    number of degrees in [0, 360] times pi/180, i.e. in [0, 6.29].  psl_tanf must equal glibc's tanf for EVERY float in [0, 8]."""
    out = subprocess.run([check_exe, "tanf"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_restated_log_exp_log10_within_one_ulp_of_libm(check_exe):
    """nfa() / log_gamma() of LSD_REFINE_ADV: the device evaluates fdlibm's log / exp / log10; glibc's are table-driven and not
    reproducible offline.  Contract: <= 1 ulp (log10: <= 2) on 4e6 samples of the ranges nfa() uses."""
    out = subprocess.run([check_exe, "f64", "4000000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    print(out.stdout.strip())


def test_ratio_from_reciprocal_table_equals_the_division_for_every_pair(check_exe):
    """k_lsd_nfa_series forms (n - i + 1) / i of nfa()'s binomial tail (OpenCV lsd.cpp nfa(); twin
    Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:199-216, which multiplies by a tabulated 1 / i WITHOUT the correction and is therefore
    not the division) as psl_ratio_inv(a, b, 1 / b): it must be the correctly rounded quotient for all 65536 x 16383 pairs it is used for."""
    out = subprocess.run([check_exe, "ratio"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_restated_double_sin_cos_equal_libm(check_exe):
    """MergeTwoLines' double sin / cos of thr in [-pi/2, pi/2] (add_src/uselongline.cpp:320-329) and region2rect's cos / sin of
    theta in [0, 3 pi): glibc's table-driven algorithm restated with a table regenerated from the series
    (psl-slam_amd/csrc/psl_sincos_glibc.h, tools/gen_sincostab.py).  Bit-identical to this host's libm on 3 x 2e7 arguments."""
    out = subprocess.run([check_exe, "sincos", "20000000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_sincos_table_is_what_the_generator_writes(tmp_path):
    inc = os.path.join(ROOT, "psl-slam_amd", "csrc", "psl_sincostab.inc")
    before = open(inc).read()
    subprocess.run(["python", os.path.join(ROOT, "tools", "gen_sincostab.py")], check=True, capture_output=True)
    assert open(inc).read() == before


# ---- nfa(): plain-Python restatement of OpenCV 3.x lsd.cpp (twin: Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240)
def _log_gamma(x):
    if x > 15.0:
        return 0.918938533204673 + (x - 0.5) * math.log(x) - x + 0.5 * x * math.log(x * math.sinh(1 / x) + 1 / (810.0 * x ** 6.0))
    q = [75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511]
    a = (x + 0.5) * math.log(x + 5.5) - (x + 5.5)
    b = 0.0
    for n in range(7):
        a -= math.log(x + n)
        b += q[n] * x ** n
    return a + math.log(b)


def _nfa(n, k, p, log_nt):
    if n == 0 or k == 0:
        return -log_nt
    if n == k:
        return -log_nt - n * math.log10(p)
    p_term = p / (1 - p)
    log1term = _log_gamma(n + 1.0) - _log_gamma(k + 1.0) - _log_gamma(n - k + 1.0) + k * math.log(p) + (n - k) * math.log(1.0 - p)
    term = math.exp(log1term)
    if term == 0.0 or abs(term) / max(abs(term), 2.2250738585072014e-308) <= 100 * 2.2204460492503131e-16:
        return -log1term / math.log(10.0) - log_nt if k > n * p else -log_nt
    bin_tail = term
    for i in range(k + 1, n + 1):
        bin_term = (n - i + 1) / i
        mult_term = bin_term * p_term
        term *= mult_term
        bin_tail += term
        if bin_term < 1:
            err = term * ((1 - mult_term ** (n - i + 1)) / (1 - mult_term) - 1)
            if err < 0.1 * abs(-math.log10(bin_tail) - log_nt) * bin_tail:
                break
    return -math.log10(bin_tail) - log_nt


def test_oracle_nfa_against_plain_python_and_restated_math():
    W, H = 512, 384
    log_nt = 5 * (math.log10(W) + math.log10(H)) / 2 + math.log10(11.0)
    rng = np.random.default_rng(11)
    cases = [(0, 0, 0.125), (10, 0, 0.125), (17, 17, 0.125), (40, 39, 0.0625), (3, 1, 0.125), (15, 14, 0.125), (16, 2, 0.125)]
    for _ in range(400):
        n = int(rng.integers(1, 4000))
        k = int(rng.integers(0, n + 1))
        cases.append((n, k, 0.125 / 2 ** int(rng.integers(0, 6))))
    for n, k, p in cases:
        ref = _nfa(n, k, p, log_nt)
        oracle_lib.set_nfa_math(0)
        a = oracle_lib.lsd_nfa(n, k, p, W, H)
        oracle_lib.set_nfa_math(1)
        b = oracle_lib.lsd_nfa(n, k, p, W, H)
        oracle_lib.set_nfa_math(0)
        assert a == pytest.approx(ref, rel=1e-9, abs=1e-9), (n, k, p)
        assert b == pytest.approx(a, rel=1e-11, abs=1e-10), (n, k, p)   # restated log / exp / log10: last-ulp differences only
    assert _nfa(100, 90, 0.125, log_nt) > 0 > _nfa(100, 12, 0.125, log_nt)   # many aligned pixels: meaningful; chance level: not


@pytest.mark.parametrize("style,seed", [("struct", 5), ("desk", 4), ("struct", 8)])
def test_lsd_refine_adv_rejects_a_subset_and_is_insensitive_to_the_math_library(style, seed):
    """LSD_REFINE_ADV (the default, oracle/line_oracle.cpp header) = LSD_REFINE_STD's rectangles, each improved and validated:
    fewer segments, none new; and the decisions are identical whether nfa() calls this host's libm (as the reference does) or
    the restated functions the device evaluates."""
    img = sf.Scene(640, 480, style, seed).gray(0)
    oracle_lib.set_lsd_refine(1)
    std = oracle_lib.lsd_detect(img)
    oracle_lib.set_lsd_refine(2)
    adv = oracle_lib.lsd_detect(img)
    oracle_lib.set_nfa_math(1)
    adv_restated = oracle_lib.lsd_detect(img)
    oracle_lib.set_nfa_math(0)
    rects = oracle_lib.lsd_rects(img)
    assert len(rects) == len(std) and 0 < len(adv) < len(std)
    assert adv.tobytes() == adv_restated.tobytes()
    # the rectangles handed to rect_improve are exactly the STD segments (+0.5, / 0.8, clamped by the contrib wrapper)
    e = ((rects[:, :4] + 0.5) / 0.8).astype(np.float32)
    for cols, lim in ((slice(0, 4, 2), 640), (slice(1, 4, 2), 480)):   # checkLineExtremes: < 0 -> 0, >= size -> size - 1
        v = e[:, cols]
        v[v < 0] = 0
        v[v >= lim] = np.float32(lim - 1)
    np.testing.assert_array_equal(e, std)
    # an accepted segment is its rectangle, possibly shifted sideways by the one-sided width reductions (<= 5 * 0.25 px at scale 0.8)
    j = 0
    for s in adv:
        while j < len(std) and np.abs(std[j] - s).max() > 5 * 0.25 / 0.8 + 1e-3:
            j += 1
        assert j < len(std), "an ADV segment that is not one of the STD rectangles, in order"
        j += 1


def test_nfa_decisions_do_not_flip_between_libm_and_the_restated_math_over_many_frames():
    """The device evaluates fdlibm-style log / exp / log10 (<= 1 - 2 ulp from glibc, not bit-identical): a near-tie in `log_nfa > 0` or
    `v > log_nfa` could flip an accept / reject against the reference's libm.  Counted instead of assumed: 240 synthetic frames,
    every segment list compared between pso_set_nfa_math(0) (host libm, what the reference calls) and (1) (what the device runs)."""
    flips = frames = segments = 0
    try:
        for style, seeds in (("struct", range(100, 130)), ("desk", range(200, 230))):
            for seed in seeds:
                sc = sf.Scene(320, 240, style, seed)
                for t in range(4):
                    img = sc.gray(7 * t)
                    oracle_lib.set_nfa_math(0)
                    a = oracle_lib.lsd_detect(img)
                    oracle_lib.set_nfa_math(1)
                    b = oracle_lib.lsd_detect(img)
                    frames += 1
                    segments += len(a)
                    if a.tobytes() != b.tobytes():
                        flips += 1
    finally:
        oracle_lib.set_nfa_math(0)
    assert frames == 240 and segments > 5000, (frames, segments)
    assert flips == 0, f"{flips} of {frames} frames changed their segment list with the restated math"
