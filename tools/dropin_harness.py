"""Builds and runs tools/dropin/dropin_main (the compiled C++ consumer of psl-slam_amd/host/pslfe.hpp) on a synthetic
RGB-D stream.  Harness code shared by bench.py (--workload dropin / tracking) and tests/test_dropin_gpu.py."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "dropin", "dropin_main.cpp")
EXE = os.path.join(ROOT, "tools", "dropin", "dropin_main")
MAGIC = 0x50534C46


def build(force=False):
    """g++ on the consumer, linked against the in-tree libpslfe.so (no GPU needed to build)."""
    hdrs = [os.path.join(ROOT, "psl-slam_amd", "host", "pslfe.hpp"), os.path.join(ROOT, "include", "pslfe.h"), SRC]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(p) for p in hdrs):
        return EXE
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", EXE, SRC, "-L" + os.path.join(ROOT, "psl-slam_amd"), "-lpslfe",
                    "-Wl,-rpath,$ORIGIN/../../psl-slam_amd"], check=True, capture_output=True)
    return EXE


def synth_stream(w, h, n, style, seed):
    """n gray frames (u8) + depth frames (f32 metres) of one drifting synthetic scene (tools/synth_frames.py)."""
    import synth_frames as sf
    sc = sf.Scene(w, h, style, seed)
    gray = np.stack([sc.gray(t) for t in range(n)], 0)
    depth = np.stack([sc.depth_u16(t).astype(np.float32) / np.float32(5000.0) for t in range(n)], 0)
    return gray, depth


def write_frames(path, gray, depth):
    n, h, w = gray.shape
    with open(path, "wb") as f:
        np.array([MAGIC, w, h, n], np.int32).tofile(f)
        np.ascontiguousarray(gray, np.uint8).tofile(f)
        np.ascontiguousarray(depth, np.float32).tofile(f)


def run(frames_path, nfeatures, nlines, warmup, results_path=None, stages=False, timeout=600, lookahead=1):
    """lookahead K > 1: the Frame members come from pslfe::FramePrefetcher (K frames extracted per batched launch)."""
    env = dict(os.environ)
    if stages:
        env["PSLFE_DROPIN_STAGES"] = "1"
    cmd = [build(), frames_path, str(nfeatures), str(nlines), str(warmup), results_path or "-", str(int(lookahead))]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"dropin_main failed ({p.returncode}): {p.stderr[-2000:]}")
    return json.loads(p.stdout.strip().splitlines()[-1])


SECTIONS = [("mvKeys", "kp"), ("mDescriptors", np.uint8), ("mvKeysUn", "kp"), ("mvDepth", np.float32), ("mvuRight", np.float32),
            ("mvKeylinesUn", "kl"), ("mLdesc", np.uint8), ("mvKeyLineFunctions", np.float64), ("fans", np.float32),
            ("lines3d", np.float64), ("planes", np.float32), ("lineNo", np.int32), ("match", np.int32), ("lm12", np.int32),
            ("lassigned", np.int32), ("plane_assoc", np.int32)]


def read_results(path, nframes):
    """Per frame: dict of the arrays dropin_main dumped (see its dump section)."""
    import psl_slam_amd as P
    out = []
    with open(path, "rb") as f:
        for _ in range(nframes):
            d = {}
            for name, ty in SECTIONS:
                n = int(np.fromfile(f, np.int64, 1)[0])
                dt = P.KEYPOINT_DTYPE if ty == "kp" else P.KEYLINE_DTYPE if ty == "kl" else np.dtype(ty)
                d[name] = np.fromfile(f, dt, n)
            out.append(d)
    return out


def camera(w):
    """The camera dropin_main builds: Examples/RGB-D/TUM3.yaml scaled with the image width, zero distortion (f32 arithmetic)."""
    import psl_slam_amd as P
    s = np.float32(w) / np.float32(640.0)
    cam = np.zeros((), P.CAMERA_DTYPE)
    for k, v in zip(("fx", "fy", "cx", "cy"), (535.4, 539.2, 320.1, 247.6)):
        cam[k] = np.float32(v) * s
    cam["bf"] = np.float32(40.0) * s
    return cam


def oracle_sequence(gray, depth, nfeatures, nlines, stage_ms=None, first=0, lines=True, frame_ms=None):
    """The same Frame::Frame + TrackWithMotionModel call sequence as tools/dropin/dropin_main.cpp, on the CPU oracle
    (tests/oracle_lib.py).  Returns one dict per frame with the arrays read_results() yields; stage_ms (a dict) accumulates the
    wall time of every call, so this is also the per-stage CPU baseline of the drop-in workloads."""
    import time
    import oracle_lib as O
    n, h, w = gray.shape
    cam = camera(w)
    orb = O.OracleORB(nfeatures, 1.2, 8, 20, 7)
    import synth_frames as sf
    scale = sf.orb_scale_factors()
    b = O.image_bounds(cam, w, h)  # mnMinX, mnMinY, mnMaxX, mnMaxY
    bounds = tuple(float(x) for x in b)
    out, last = [], None

    def lap(name, t0):
        if stage_ms is not None:
            stage_ms[name] = stage_ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    empty_kl = (np.zeros(0, O.KEYLINE_DTYPE), np.zeros((0, 32), np.uint8), np.zeros((0, 3)))
    for t in range(n):
        t0 = tf = time.perf_counter()
        cur = {}
        cur["mvKeys"], cur["mDescriptors"] = orb(gray[t])
        t0 = lap("ORBextractor()", t0)
        kls, ldesc, eq = O.line_extract(gray[t], nlines) if lines else empty_kl
        cur["mvKeylinesUn"], cur["mLdesc"], cur["mvKeyLineFunctions"] = kls, ldesc, eq
        t0 = lap("LINEextractor()", t0)
        L4 = np.stack([kls[k] for k in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32) if len(kls) else np.zeros((0, 4), np.float32)
        cur["fans"] = O.lil_pair(L4, 20.0, np.float32(np.pi / 4), w, h)
        t0 = lap("CPartiallyRecoverConnectivity", t0)
        g = O.frame_glue(kls, cur["fans"], depth[t], cam, seed=1 + first + t)
        cur["lines3d"], cur["planes"], cur["lineNo"], cur["cross3d"] = g["lines3d"], g["planes"], g["lineNo"], g["cross3d"]
        t0 = lap("isLineGood+fans+planes", t0)
        cur["mvKeysUn"], cur["mvDepth"], cur["mvuRight"] = O.frame_post_rgbd(cur["mvKeys"], depth[t], cam)
        O.grid_build(cur["mvKeysUn"], bounds)
        t0 = lap("Undistort+StereoFromRGBD+Grid", t0)
        for k in ("match", "lm12", "lassigned", "plane_assoc"):
            cur[k] = np.zeros(0, np.int32)
        if last is not None and len(cur["mvKeys"]):
            lm = 0
            if lines:
                lm, cur["lm12"], cur["lassigned"] = O.search_by_geom_appearance(last["mvKeylinesUn"], last["mLdesc"], kls, ldesc,
                                                                                np.ones(len(last["mvKeylinesUn"]), np.uint8), 0.95,
                                                                                (bounds[0], bounds[2], bounds[1], bounds[3]))
            t0 = lap("SearchByGeomNApearance", t0)

            def project(th):
                ku = last["mvKeysUn"]
                q = np.zeros(len(ku), O.PROJQUERY_DTYPE)
                q["u"], q["v"] = ku["x"], ku["y"]
                q["radius"] = np.float32(th) * scale[ku["octave"]]
                q["ur"] = last["mvuRight"]
                q["min_level"], q["max_level"] = ku["octave"] - 1, ku["octave"] + 1
                q["angle"], q["blocks"] = ku["angle"], 1
                return O.search_by_projection_last(cur["mvKeysUn"], cur["mDescriptors"], cur["mvuRight"], bounds, q, last["mDescriptors"], None, True)
            nm, cur["match"], _ = project(15)
            t0 = lap("SearchByProjection(cur,last)", t0)
            if nm + lm < 20:
                nm, cur["match"], _ = project(30)
                t0 = lap("SearchByProjection retry", t0)
            npl, nmap = len(cur["planes"]), len(last["planes"])
            if npl and nmap:
                pts = np.zeros((npl, 15), np.float64)
                pts[:, 0:6] = cur["lines3d"][cur["lineNo"][:, 0]]
                pts[:, 6:12] = cur["lines3d"][cur["lineNo"][:, 1]]
                pts[:, 12:15] = cur["cross3d"]
                _, cur["plane_assoc"] = O.associate_planes(cur["planes"], pts, last["planes"], 0.05, 0.999, 1)
            t0 = lap("AssociatePlanesByBoundary", t0)
        if frame_ms is not None:
            frame_ms.append((time.perf_counter() - tf) * 1e3)
        out.append(cur)
        last = cur
    return out


def compare(got, ref, what=""):
    """Bit-for-bit comparison of one frame's dropin_main results with the oracle's; raises AssertionError naming the array."""
    for name, _ in SECTIONS:
        a, b = got[name], ref[name]
        if b.ndim > 1:
            b = b.reshape(-1) if b.dtype.names is None else b
        a = a.reshape(-1) if a.dtype.names is None else a
        b = np.ascontiguousarray(b).reshape(-1) if b.dtype.names is None else b
        assert a.shape == b.shape, f"{what}{name}: {a.shape} vs oracle {b.shape}"
        b = np.ascontiguousarray(b).astype(a.dtype, copy=False)
        if a.tobytes() != b.tobytes():
            detail = ""
            if a.dtype.names is None:
                bad = np.flatnonzero(a.view(np.uint8).reshape(len(a), -1).any(1) | True) if False else np.flatnonzero(a != b)
                detail = f": {len(bad)} of {len(a)} entries, first at {bad[:8].tolist()}: got {a[bad[:8]].tolist()} oracle {b[bad[:8]].tolist()}"
            raise AssertionError(f"{what}{name} differs from the oracle{detail}")
