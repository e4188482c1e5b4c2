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


def run(frames_path, nfeatures, nlines, warmup, results_path=None, stages=False, timeout=600):
    env = dict(os.environ)
    if stages:
        env["PSLFE_DROPIN_STAGES"] = "1"
    cmd = [build(), frames_path, str(nfeatures), str(nlines), str(warmup)] + ([results_path] if results_path else [])
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
