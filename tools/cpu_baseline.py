"""The CPU baseline bench.py reports beside the GPU number: the CPU oracle (oracle/, kind "port":
the reference itself cannot be built here) on the same synthetic frames and the same call sequence,
  (a) single-threaded - the reference runs ExtractORB; ExtractLSD serially on the tracking thread (src/Frame.cc:179-180) - with
      10 warm-up frames, median and mean ms/frame and per-call ms;
  (b) one stream per core over the host cores this process may use (one worker process per core, each its own stream).
Runs BEFORE the GPU is initialised (worker processes are forked)."""
import multiprocessing as mp
import os
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_JOB = {}


def native_oracle():
    """-O3 -march=native build of the oracle for timing, made on this host (the portable -O2 build is what the tests load)."""
    import oracle_lib
    odir = os.path.join(ROOT, "oracle")
    native = os.path.join(odir, "libpsl_oracle_native.so")
    try:
        srcs = sorted(os.path.join(odir, f) for f in os.listdir(odir) if f.endswith(".cpp") and not f.startswith("ref_"))
        if not os.path.exists(native) or any(os.path.getmtime(s) > os.path.getmtime(native) for s in srcs):
            subprocess.run(["g++", "-O3", "-march=native", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", native] + srcs + ["-lm", "-lpthread"],
                           check=True, capture_output=True)
        oracle_lib.SO = native
        oracle_lib._lib = None
        return "-O3 -march=native"
    except Exception:
        return "-O2 (portable build)"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _worker(i):
    import dropin_harness as D
    J = _JOB
    n = J["warm"] + J["per_worker"]
    lo = (i * 7) % max(1, len(J["gray"]) - n + 1)
    fm = []
    D.oracle_sequence(J["gray"][lo:lo + n], J["depth"][lo:lo + n], J["nfeatures"], J["nlines"], lines=J["lines"], frame_ms=fm)
    return fm[J["warm"]:]


def run(gray, depth, nfeatures, nlines, lines, budget_s=20.0, min_frames=200, warm=10):
    """gray [n][h][w] u8, depth [n][h][w] f32: consecutive frames of one synthetic stream."""
    import dropin_harness as D
    build = native_oracle()
    n_avail = len(gray)
    # (a) one thread
    D.oracle_sequence(gray[:warm], depth[:warm], nfeatures, nlines, lines=lines)   # warm-up, discarded
    stage, fm = {}, []
    t0 = time.perf_counter()
    done = 0
    while done < min_frames and (time.perf_counter() - t0 < budget_s or done < 32):
        m = min(16, min_frames - done)
        lo = (warm + done) % max(1, n_avail - m + 1)
        D.oracle_sequence(gray[lo:lo + m], depth[lo:lo + m], nfeatures, nlines, stage_ms=stage, first=lo, lines=lines, frame_ms=fm)
        done += m
    timed = np.array(fm)
    ntimed = stage_n = len(timed)
    out = {"value": round(1e3 / float(timed.mean()), 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "ms_per_frame": {"median": round(float(np.median(timed)), 3), "mean": round(float(timed.mean()), 3)},
           "calls_ms_mean": {k: round(v / stage_n, 3) for k, v in stage.items()},
           "sample": f"{ntimed} frames {gray.shape[2]}x{gray.shape[1]} after {warm} warm-up frames, synthetic stream, the Frame::Frame + TrackWithMotionModel "
                     f"call sequence of tools/dropin_harness.py:oracle_sequence ({'ORB + lines' if lines else 'ORB only'}; a SUPERSET of the batched GPU step's call list: "
                     f"it adds UndistortKeyPoints / ComputeStereoFromRGBD / grid, SearchByGeomNApearance and AssociatePlanesByBoundary, about 0.3 ms per frame), "
                     f"oracle/ built {build}, 1 thread",
           "cpu_model": cpu_model()}
    # (b) one stream per core
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    per_worker = max(12, -(-min_frames // cores))
    _JOB.update(gray=gray, depth=depth, nfeatures=nfeatures, nlines=nlines, lines=lines, warm=4, per_worker=per_worker)
    try:
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res = pool.map(_worker, range(cores))
        wall = time.perf_counter() - t0
        allms = np.concatenate([np.array(r) for r in res])
        # throughput from the workers' own timed frames: cores x (frames / time of a worker), start-up and warm-up excluded
        fps = float(sum(len(r) / (sum(r) * 1e-3) for r in res))
        out["all_cores"] = {"value": round(fps, 1), "unit": "frames/s", "cores": cores, "frames": int(len(allms)),
                            "ms_per_frame": {"median": round(float(np.median(allms)), 3), "mean": round(float(allms.mean()), 3)},
                            "wall_s": round(wall, 2),
                            "how": "one worker process per core, each running its own stream one frame after the other (4 warm-up frames per worker excluded)"}
    except Exception as e:  # a box without fork / with too little memory still reports (a)
        out["all_cores"] = {"error": repr(e)}
    return out
