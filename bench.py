#!/usr/bin/env python3
"""bench.py — frames/sec of the PSL-SLAM per-frame feature front-end on MI355X.

Workloads (BASELINE.json configs):
  lines     (default) configs[2], the configuration the headline metric "frames/sec ORB+line extract+match, 640x480 RGB-D" is
            quoted on: a step = one pass over a batch of B = 12288 frames already resident in HBM - ORB extraction (pyramid,
            per-cell FAST, octree, orientation, blur, rBRIEF) + frame grid + ORBmatcher::SearchByProjection(cur,last) against
            the predecessor frame, and the line path (LSD with LSD_REFINE_ADV, merge, top-200, LBD, LIL pairing, the RGB-D
            line glue of the Frame constructor, LSDmatcher::match).  The batch holds 256 DISTINCT frames (8 scenes x 32 time
            steps).  With N > 1 every rank runs its own stream (weak scaling) and the per-frame result records
            are all-gathered with RCCL through the C ABI (pslfe_record_pack_device + pslfe_gather_all).
  orb       configs[1]: ORB-only extract + match, B = 256.
  --host-io (either of the above, any --batch): the batch comes from pinned host memory (gray u8 + depth u16) and the packed
            result records go back to pinned host memory INSIDE the timed region - the drop-in boundary with copies.
  dropin    the single-frame drop-in (B = 1), 640x480 / 1000 ORB / 200 lines: the compiled C++ consumer of host/pslfe.hpp
            (tools/dropin/dropin_main.cpp) runs Frame::Frame + TrackWithMotionModel one frame at a time, host buffers in, host
            results out, wall-clock per frame (what the reference itself reports: Examples/RGB-D/rgbd_tum.cc:103-119).
  tracking  configs[4]: the same consumer at 1280x960, 2000 ORB + 200 lines.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement), including
  roofline      for the dominant kernel: algorithmic bytes per launch / mean launch time (HIP events on the launch stream over the
                timed region) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference itself cannot be built here) on the same frames and the same
                call sequence: 1 thread (median / mean ms per frame, per-call ms) and one stream per host core
  parity_checked_frames  frames of the last timed step (or of the consumer's run) compared bit for bit with the oracle; the run
                FAILS (exit 1) on a mismatch.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

W, H = 640, 480
NFEATURES, NLEVELS = 1000, 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# ALGORITHMIC bytes per frame, SURVEY.md §8(d) / BASELINE.md §3 (640x480, N = 1000, P = 950 532 px):
#   gray read WH + pyramid write P-WH + pyramid read P + blurred write P + sampling 512N + outputs 60N
P_PIX, WH = 950532, W * H
ORB_BYTES_PER_FRAME = WH + (P_PIX - WH) + P_PIX + P_PIX + 512 * NFEATURES + 60 * NFEATURES  # 3 423 596
MATCH_BYTES_PER_FRAME = 2 * 32 * NFEATURES + 8 * NFEATURES                                   # 72 000
LINE_BYTES_PER_FRAME = 17040800 + 12800
# per-kernel shares of that accounting: what each kernel must move at least
STAGE_BYTES_PER_FRAME = {
    "orb.pyramid": WH + (P_PIX - WH),        # read level 0, write levels 1..7
    "orb.fast": P_PIX,                       # read every level once
    "orb.octree": 8 * 4000,                  # candidate list in + out (~4 k candidates x 8 B); not in §8(d)
    "orb.blur": P_PIX + P_PIX,               # read every level, write its blurred copy
    "orb.describe": 512 * NFEATURES + 60 * NFEATURES,
    "match.grid": 2 * 60 * NFEATURES,
    "match.window": MATCH_BYTES_PER_FRAME,
    # line path, SURVEY.md §8(d) "Lines @640x480" = 17 040 800 B/frame split over the kernels that move them
    "line.lsd_scale": WH + 8 * 196608,                 # gray read + f64 working image write
    "line.lsd_grad": 8 * 196608 + 2 * 8 * 196608,      # working image read + angle & modgrad write
    "line.lsd_grow": 2 * 8 * 196608 + 2 * 8 * 196608 + 393216,  # angle & modgrad read + coordinate list w+r + used map
    # LSD_REFINE_ADV (not in §8(d)'s accounting, which predates the refine mode): per launch of a phase the angle map is read once
    # and the rectangle state (96 B) read and written; the stage is launched 6 times per step (first test + 5 phases)
    "line.nfa_count": 4 * 196608 + 400 * (96 + 40),
    "line.nfa_eval": 400 * (96 + 40 + 96),
    "line.merge": 2 * 16 * 500,
    "line.lbd_pre": WH + 4 * WH,                       # gray read, Sobel (dx, dy) s16 write (the blurred image stays in LDS)
    "line.lbd": 5040000 + 20000,                       # LSR gathers + outputs
    "line.pair": 2 * 16 * 200,
    "line.match": 12800,
    "line.good": 200 * 21 * 4 + 200 * (68 + 48 + 12),  # depth samples + keylines in, 3-D lines out
    "line.planes": 4096 * 16 + 200 * 60,
    "gather.pack": 2 * 85000,
}
# stage -> kernels it launches (kernels per step); HBM traffic of a stage = sum over its launches
STAGE_KERNELS = {
    "orb.pyramid": [("k_pyr_resize_tiled", NLEVELS - 1)], "orb.fast": [("k_fast_cells4", 1)], "orb.octree": [("k_octree<256>", 1)],
    "orb.blur": [("k_blur7", 1)], "orb.describe": [("k_orient_describe", 1)],
    "match.grid": [("k_frame_import", 1), ("k_build_grid", 1)], "match.window": [("k_window_eval", 1), ("k_window_resolve<0, 4096, 1024>", 1)],
    "line.lsd_scale": [("k_lsd_scale_tiled", -2048)], "line.lsd_grad": [("k_lsd_grad", -2048)],   # -n: one launch per sub-batch of n frames
    "line.lsd_grow": [("k_lsd_grow4<0>", 1)],
    "line.nfa_count": [("k_lsd_nfa_count<%d>" % ph, 1) for ph in (-2, -1, 0, 1, 2, 3)],
    "line.nfa_eval": [("k_lsd_nfa_%s<%d>" % (k, ph), 1) for k in ("setup", "series", "select") for ph in (-2, -1, 0, 1, 2, 3)],
    "line.merge": [("k_line_merge<512>", 1), ("k_line_merge<1024>", 1)], "line.lbd_pre": [("k_lbd_pre", 1)], "line.lbd": [("k_lbd", 1)], "line.pair": [("k_lil_pair", 1)],
    "line.match": [("k_line_match_batch", 1)], "line.good": [("k_line_good", 1)], "line.planes": [("k_fans_planes", 1)],
}
PER_LAUNCH_STAGES = ("line.nfa_count", "line.nfa_eval")   # STAGE_BYTES_PER_FRAME is per launch for these (6 launches per step), per step otherwise
STAGE_NAMES = ["orb.pyramid", "orb.fast", "orb.octree", "orb.blur", "orb.describe", "match.grid", "match.window", "line.lsd_scale",
               "line.lsd_grad", "line.lsd_grow", "line.nfa_count", "line.nfa_eval", "line.merge", "line.lbd_pre", "line.lbd", "line.pair", "line.match", "line.good",
               "line.planes", "gather.pack"]


def pmc_traffic(workload, stage, batch):
    """HBM bytes per launch of a stage from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/summarize_round.py: FETCH_SIZE / WRITE_SIZE in their own passes, corrected with the calibration measured in
    the same session).  None when no pass exists for this workload at this batch size."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]
        if j["frames_per_launch"] != batch:
            return None, None
        total = 0.0
        for kern, n in STAGE_KERNELS[stage]:
            k = j["kernels"].get(kern)
            if k is None:   # a template argument list grew after the pass (k_lsd_grow4<0> is k_lsd_grow4<0, 0> since the few-frames variant has a second one)
                k = next(v for name, v in j["kernels"].items() if kern.endswith(">") and name.startswith(kern[:-1] + ","))
            if n < 0:
                n = -(-batch // -n)
            total += n * (k["read_bytes"] + k["write_bytes"])
        return int(total), f"profiles/{j['tag']}_pmc_summary.txt"
    except Exception:
        return None, None


# ---- synthetic input: 8 scenes x 32 time steps = 256 distinct frames per rank, generated by a process pool, cached in /tmp -----
def _gen_scene(args):
    import synth_frames as sf
    w, h, style, seed, nt, zscale = args
    sc = sf.Scene(w, h, style, seed)
    gray = np.stack([sc.gray(t) for t in range(nt)], 0)
    depth = np.rint(sc.depth_u16(0).astype(np.float64) * zscale).astype(np.uint16)
    return gray, depth


def distinct_frames(w, h, style, seed0, nscenes=8, nt=32, serial=False):
    """(gray [nscenes*nt][h][w] u8, depth_u16 [nscenes][h][w]): the frames of a scene are consecutive (drift <= 2 px per frame), scenes
    follow each other (a cut every nt frames).  Depth is a tilted plane per scene (x5000, TUM convention).  serial: no worker
    processes (under a profiler whose preload has initialised the GPU this process must not fork)."""
    import multiprocessing as mp
    path = os.path.join(tempfile.gettempdir(), f"pslfe_bench_{w}x{h}_{style}_{seed0}_{nscenes}x{nt}.npz")
    if os.path.exists(path):
        try:
            z = np.load(path)
            return z["gray"], z["depth"]
        except Exception:
            pass
    jobs = [(w, h, style, seed0 + 17 * s, nt, 1.0 + 0.04 * s) for s in range(nscenes)]
    if serial:
        print("[bench] generating the input frames in this process (profiler preload present: no fork); a few minutes", file=sys.stderr, flush=True)
        res = [_gen_scene(j) for j in jobs]
    else:
        try:
            with mp.get_context("fork").Pool(min(nscenes, max(1, len(os.sched_getaffinity(0))))) as pool:
                res = pool.map(_gen_scene, jobs)
        except Exception:
            res = [_gen_scene(j) for j in jobs]
    gray = np.ascontiguousarray(np.concatenate([r[0] for r in res], 0))
    depth = np.stack([r[1] for r in res], 0)
    try:
        np.savez(path + ".tmp.npz", gray=gray, depth=depth)
        os.replace(path + ".tmp.npz", path)
    except Exception:
        pass
    return gray, depth


def depth_f32(depth_u16):
    """imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) (src/Tracking.cc:230-235): u16 * (1 / 5000) in float - the same values
    pslfe_depth_to_float_device produces, so HBM-resident and --host-io runs see identical depth."""
    return np.ascontiguousarray(depth_u16).astype(np.float32) * (np.float32(1.0) / np.float32(5000.0))   # u16 -> f32 is exact, one f32 product


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; the contract is one JSON line there."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def seed_for(rank):
    import synth_frames as sf
    return sf.SEED + 1000 * rank


# ---- the single-frame drop-in workloads (C++ consumer) ---------------------------------------------------------------------
def run_consumer(args):
    import dropin_harness as D
    import cpu_baseline as CB
    tracking = args.workload == "tracking"
    w, h, nf, nl = (1280, 960, 2000, 200) if tracking else (640, 480, 1000, 200)
    K = max(1, args.lookahead)
    if K > 1:   # whole batches inside and outside the timed region: the launch of a batch is paid by its first frame
        args.warmup = -(-args.warmup // K) * K
        args.steps = max(K, -(-args.steps // K) * K)
    nframes = args.steps + args.warmup
    scene = args.scene or "struct"
    gray, depth = D.synth_stream(w, h, nframes, scene, seed_for(0))
    cpu = None
    if not args.no_cpu_baseline:
        cpu = CB.run(gray, depth, nf, nl, True, budget_s=20.0, min_frames=min(200, max(16, nframes - 10)), warm=min(10, nframes // 4))
    tmp = tempfile.mkdtemp()
    fpath = os.path.join(tmp, "frames.bin")
    D.write_frames(fpath, gray, depth)
    ncheck = min(8, nframes)
    rpath = os.path.join(tmp, "results.bin")
    cpath = os.path.join(tmp, "check.bin")
    D.write_frames(cpath, gray[:ncheck], depth[:ncheck])
    D.run(cpath, nf, nl, 0, results_path=rpath, lookahead=K)        # untimed: the results the parity check reads
    got = D.read_results(rpath, ncheck)
    ref = D.oracle_sequence(gray[:ncheck], depth[:ncheck], nf, nl)
    for t in range(ncheck):
        D.compare(got[t], ref[t], f"frame {t}: ")
    r = D.run(fpath, nf, nl, args.warmup, lookahead=K)               # the timed run: host clock around every frame, copies included
    rs = D.run(fpath, nf, nl, args.warmup, stages=True, lookahead=K)  # same again with the library's per-kernel HIP-event timers on
    stages = rs.get("gpu_stage_ms_per_frame", {})
    dom = max(stages, key=lambda s: stages[s]) if stages else "line.lsd_grow"
    scale_bytes = (w * h) / float(W * H)
    dom_bytes = int(STAGE_BYTES_PER_FRAME.get(dom, 0) * scale_bytes)
    dom_ms = stages.get(dom, 0.0)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    out = {
        "metric": ("frames/sec ORB+line extract+match, single-frame drop-in (B = 1), host buffers in / host results out" if K == 1 else
                   f"frames/sec ORB+line extract+match, Tracking loop with a look-ahead of {K} frames (FramePrefetcher), host buffers in / host results out"),
        "value": round(1e3 / r["ms_per_frame"]["mean"], 2), "unit": "frames/s", "n_gpus": 1, "steps": r["frames_timed"], "warmup": args.warmup,
        "ms_per_step": round(r["ms_per_frame"]["mean"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": (f"configs[4]: {w}x{h} synthetic structure-like RGB-D stream, 2000 ORB + 200 lines, the C-ABI sequence a Tracking loop issues "
                                "(Frame::Frame src/Frame.cc:133-208 + TrackWithMotionModel src/Tracking.cc:1164-1214) through the compiled C++ consumer "
                                "tools/dropin/dropin_main.cpp, " + ("one frame at a time" if K == 1 else f"frames extracted {K} at a time by pslfe::FramePrefetcher, tracked one at a time") +
                                ", H2D / D2H inside the timed region") if tracking else
                               (f"configs[2] shape, B = 1: {w}x{h} synthetic structure-like RGB-D stream, 1000 ORB + 200 lines, Frame::Frame + "
                                "TrackWithMotionModel through the compiled C++ consumer tools/dropin/dropin_main.cpp, " +
                                ("one frame at a time" if K == 1 else f"frames extracted {K} at a time by pslfe::FramePrefetcher, tracked one at a time") +
                                ", H2D / D2H inside the timed region"),
                   "frames_per_step_per_gpu": 1, "lookahead": K, "scene": scene, "mean_keypoints": r["mean_keypoints"], "mean_keylines": r["mean_keylines"],
                   "mean_matches": r["mean_matches"], "mean_line_matches": r["mean_line_matches"], "lsd_refine": "LSD_REFINE_ADV"},
        "latency_ms_per_frame": r["ms_per_frame"], "frame_ctor_ms": r["frame_ctor_ms"], "track_ms": r["track_ms"],
        "calls_ms_mean": r["calls_ms_mean"], "gpu_stage_ms_per_frame": stages,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": None, "algorithmic_bytes_per_launch": dom_bytes,
                     "ms_per_launch": round(dom_ms, 4),
                     "note": ("one frame per launch: a serial chain on one wave, latency-bound (DESIGN.md §5)" if K == 1 else
                              f"{K} frames per launch; time and bytes are per frame (launch / {K})")},
        "parity_checked_frames": ncheck,
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    return 0


def under_profiler():
    """rocprofv3's preload initialises the GPU before this program starts (with --pmc it does): such a process must not fork."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ)


def build_once():
    """libpslfe.so is built by ONE process (the others wait on a file lock and find it up to date)."""
    import fcntl
    import psl_slam_amd as P
    with open(os.path.join(tempfile.gettempdir(), "pslfe_build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            P.build()
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return P


def probe_device(local_rank):
    """pslfe_ctx_create on this rank's device in a short-lived child process (this process forks worker pools before it touches
    the GPU itself).  Returns None when the device is usable, else the library's message (PSLFE_E_NODEVICE: no CPU fallback)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import psl_slam_amd as P\n"
            "try:\n    P.Context(%d).close()\nexcept P.PslfeError as e:\n    print(str(e)); sys.exit(3)\n" % (ROOT, local_rank))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode == 0:
        return None
    return (r.stdout.strip() or r.stderr.strip() or f"exit code {r.returncode}").splitlines()[-1]


# ---- python bench.py --gpus N without a launcher: this process starts the N ranks itself and never touches the GPU ------------
def spawn_ranks(n, argv):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its environment, as
    torch.distributed.run would set them), started BEFORE anything in this process has touched the GPU.  Rank 0's stdout - the one
    JSON line - is relayed; every other rank's stdout goes to stderr.  Exit code = the worst of the children; when a rank dies the
    others get 60 s to notice (a failed rendezvous or collective) and are then terminated by PID."""
    import socket
    import subprocess
    import threading
    build_once()   # hipcc only; no GPU call
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PSLFE_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []

    def relay():
        for line in procs[0].stdout:
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    first_fail = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.25)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_fail is None:
            first_fail = time.time()
        if first_fail is not None and time.time() - first_fail > 60.0:
            for p in procs:
                if p.poll() is None:
                    print(f"[bench] rank process {p.pid} still running 60 s after another rank failed: terminating it", file=sys.stderr)
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
    th.join(timeout=10.0)
    rcs = [p.returncode for p in procs]
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
    return min(worst, 255)


def launcher_selftest(rank, world):
    """CPU rehearsal of the N > 1 choreography with the gloo backend (tests/test_multigpu_cpu.py drives it through spawn_ranks):
    rendezvous, the collective agreement that selects the gather, gather-to-rank-0 of per-rank record buffers, barrier-bracketed
    timing with MAX over ranks, one JSON line from rank 0.  No product code path runs here - there is no CPU path to run."""
    import zlib
    import torch
    import torch.distributed as dist
    from importlib import import_module
    mg = import_module("psl_slam_amd.multigpu")
    dist.init_process_group("gloo")
    fail_rank = int(os.environ.get("PSLFE_SELFTEST_FAIL_RANK", "-1"))      # this rank "cannot create its communicator"
    agreed = mg.agree_all_ranks(rank != fail_rank, world)
    rng = np.random.default_rng(100 + rank)
    rec = torch.from_numpy(rng.integers(0, 256, (16, 4096), dtype=np.uint8))
    recv = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.barrier()
    t0 = time.perf_counter()
    dist.gather(rec, recv, dst=0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    seen = torch.full((world,), -1, dtype=torch.int32)
    dist.all_gather_into_tensor(seen, torch.tensor([rank], dtype=torch.int32))
    if rank == 0:
        want = [zlib.crc32(np.random.default_rng(100 + r).integers(0, 256, (16, 4096), dtype=np.uint8).tobytes()) for r in range(world)]
        got = [zlib.crc32(r_.numpy().tobytes()) for r_ in recv]
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": [int(v) for v in seen], "gather_ok": got == want,
                          "rccl_gather_agreed": agreed, "launcher": os.environ.get("PSLFE_BENCH_LAUNCHER", "external"),
                          "max_s": float(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 20; 100 frames for dropin / 40 for tracking)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default 3; 10 frames for dropin / tracking)")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default 12288 = two rounds of the 6144 wave slots of the LSD growing when the "
                                                         "GPU's free memory holds it, for every N; else 6144; 256 for --workload orb)")
    ap.add_argument("--workload", choices=["orb", "lines", "dropin", "tracking"], default="lines",
                    help="lines = BASELINE configs[2], the configuration of the headline metric; orb = configs[1]; dropin = B = 1 through the "
                         "C++ consumer; tracking = configs[4] through the C++ consumer")
    ap.add_argument("--scene", choices=["sticks", "struct", "desk"], default=None,
                    help="synthetic scene family (tools/synth_frames.py).  lines: 'sticks' (default: the structure scene at the configured line load, "
                         "160 - 185 keylines per frame) or 'struct' (rounds 1 - 2: ~29 keylines per frame); orb: 'desk'")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: the line pipeline on its own context / stream beside the ORB pipeline (the two extractor objects of a Frame are "
                         "independent); stages then overlap and their event timings stop meaning what they say, so 1 is the default")
    ap.add_argument("--host-io", action="store_true", help="host gray + depth in, host result records out, inside the timed region")
    ap.add_argument("--gather", choices=["root", "all"], default="root",
                    help="N > 1: per-frame result records to rank 0 (pslfe_gather_to_root: ncclSend / ncclRecv) or to every rank (pslfe_gather_all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-like-for-like", action="store_true", help="skip the extra steps at 6144 frames per launch after the timed region")
    ap.add_argument("--lookahead", type=int, default=1, help="dropin / tracking: frames extracted per launch by the FramePrefetcher (1 = one frame at a time)")
    ap.add_argument("--parity-frames", type=int, default=8)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--prepare-inputs", action="store_true", help="generate (and cache in /tmp) the input frames of the workload, then exit: run before profiler passes")
    args = ap.parse_args()
    consumer = args.workload in ("dropin", "tracking")
    if args.steps <= 0:
        args.steps = (100 if args.workload == "dropin" else 40) if consumer else 20
    if args.warmup < 0:
        args.warmup = 10 if consumer else 3

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])          # no launcher: start the ranks here; this process never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: running {world} ranks", file=sys.stderr)
    if args.launcher_selftest:
        return launcher_selftest(rank, world)
    profiled = under_profiler()   # the profiler's preload owns the GPU already: this process starts no child process at all
    if args.prepare_inputs:
        distinct_frames(W, H, args.scene or ("sticks" if args.workload == "lines" else "desk"), seed_for(rank))
        return 0
    if profiled:
        import psl_slam_amd as P
    else:
        P = build_once()
    msg = None if profiled else probe_device(local_rank)
    if msg is not None:
        print(f"[bench] rank {rank}: {msg}", file=sys.stderr, flush=True)
        return 3
    if consumer:
        if world != 1:
            print("[bench] the single-frame workloads run on one GPU", file=sys.stderr)
            return 2
        return run_consumer(args)

    LINES = args.workload == "lines"
    scene = args.scene or ("sticks" if LINES else "desk")
    # ---- everything that forks worker processes happens BEFORE the GPU is touched: input frames and the CPU baseline
    gray256, depth8 = distinct_frames(W, H, scene, seed_for(rank), serial=profiled)
    # continuity with rounds 1 - 2, whose headline ran on the 'struct' scene (29 keylines per frame): a few steps on it after the timed region
    gray_prev = None
    if LINES and scene == "sticks" and world == 1 and not args.host_io and not args.no_like_for_like and not profiled:
        gray_prev, _ = distinct_frames(W, H, "struct", seed_for(rank))
    ND = len(gray256)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not profiled:
        import cpu_baseline as CB
        d32 = depth_f32(depth8[0])
        cpu = CB.run(gray256[:64], np.broadcast_to(d32, (64,) + d32.shape), NFEATURES, 200, LINES, budget_s=20.0, min_frames=200, warm=10)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev)

    from importlib import import_module
    import batch_pipeline as BP
    mg = import_module("psl_slam_amd.multigpu")

    # Frames per launch.  12288 = two rounds of the 6144 wave slots of the LSD growing (+3.8 % frames/s over 6144: the launch lasts as long
    # as its longest frames) when the device's FREE memory holds it: 21.7 MB per frame (pipeline buffers 15.8 + 4.4, gray, depth f32: 265 GB in use at 12288) +
    # the gather's buffers (N > 1: two send buffers per rank, ONE receive buffer of world x batch records on rank 0; --host-io: pinned
    # staging is host memory, the u16 depth copy 0.6 MB per frame) + 6 GB of slack.  Every rank takes the same decision (MIN over ranks).
    REC_BYTES = 104 * 1024   # upper estimate of one result record (101 120 B at 1100 keypoint rows / 200 lines / 512 fans / 64 planes)
    def need_bytes(b):
        n = b * 21.7e6 + 6e9
        if world > 1 or args.host_io:
            n += 2 * b * REC_BYTES + (world * b * REC_BYTES * (1 if args.gather == "root" else 2) if world > 1 else 0)
        if args.host_io:
            n += b * W * H * (2 * 2 + 1)   # two u16 depth slots and the second gray slot
        return n
    if args.batch:
        B = args.batch
    elif not LINES:
        B = 256
    else:
        free_b = torch.cuda.mem_get_info(dev)[0]
        ok = mg.agree_all_ranks(free_b >= need_bytes(12288), world, dev)
        B = 12288 if ok else 6144
        if not ok and rank == 0:
            print(f"[bench] {free_b / 1e9:.1f} GB free on the device: 6144 frames per launch instead of 12288", file=sys.stderr)

    # a real (non-null) torch stream: the library launches on it, so torch ops and the HIP kernels are ordered on one stream and
    # torch.cuda events / synchronize see everything
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    stream_l = torch.cuda.Stream(dev) if (args.streams == 2 and LINES) else None
    pipe = BP.BatchPipeline(P, torch, dev, stream, local_rank, B, W, H, lines=LINES, second_stream=stream_l)
    ctx = pipe.ctx

    # the batch: B frames = the 256 distinct frames repeated; depth (lines): the scene's plane, f32 metres
    idx = np.arange(B) % ND
    g256 = torch.from_numpy(gray256).to(dev)
    frames_d = g256[torch.from_numpy(idx).to(dev)].contiguous() if B != ND else g256
    depth_d = None
    if LINES:
        d8 = torch.from_numpy(np.stack([depth_f32(d) for d in depth8], 0)).to(dev)
        depth_d = d8[torch.from_numpy((idx // 32) % len(depth8)).to(dev)].contiguous()
    host_io = None
    if args.host_io:
        # Host buffers in, host records out, inside the timed region, as a steady-state pipeline: while the kernels of batch n run, the copy stream
        # brings in batch n + 1 (double-buffered device inputs) and takes out the records of batch n - 1 - the way a host that reads a recording
        # ahead (pslfe::FramePrefetcher's caller) overlaps its copies.  One H2D and one D2H per step inside the timed region.
        host_io = {"gray": torch.from_numpy(gray256[idx]).pin_memory(), "cs": torch.cuda.Stream(dev), "n": 0,
                   "d_gray": [frames_d, torch.empty_like(frames_d)], "ev_in": [torch.cuda.Event(), torch.cuda.Event()], "ev_done": [None, None],
                   "ev_pack": torch.cuda.Event()}
        if LINES:
            host_io["depth16"] = torch.from_numpy(np.ascontiguousarray(depth8[(idx // 32) % len(depth8)]).view(np.int16)).pin_memory()  # u16 bits
            host_io["d_depth16"] = [torch.empty((B, H, W), dtype=torch.int16, device=dev) for _ in range(2)]

        def h2d(slot):   # on the copy stream, after the step that last read this slot
            cs = host_io["cs"]
            if host_io["ev_done"][slot] is not None:
                cs.wait_event(host_io["ev_done"][slot])
            with torch.cuda.stream(cs):
                host_io["d_gray"][slot].copy_(host_io["gray"], non_blocking=True)
                if LINES:
                    host_io["d_depth16"][slot].copy_(host_io["depth16"], non_blocking=True)
                host_io["ev_in"][slot].record(cs)
        host_io["h2d"] = h2d
        h2d(0)

    gather = None
    gather_info = None
    layout = None
    rec_host = None
    nB = [B]   # frames per launch of step(): B, or 6144 for the like-for-like steps after the timed region

    def step():
        n = nB[0]
        d_gray = frames_d
        if host_io is not None:   # depth arrives as the sensor's u16 and is converted on the device (src/Tracking.cc:230-235)
            slot = host_io["n"] & 1
            host_io["n"] += 1
            host_io["h2d"](slot ^ 1)                       # the next batch comes in while this one is computed
            stream.wait_event(host_io["ev_in"][slot])      # this batch has arrived
            d_gray = host_io["d_gray"][slot]
            if LINES:
                P._check(P.lib().pslfe_depth_to_float_device(ctx._h, __import__("ctypes").c_void_p(host_io["d_depth16"][slot].data_ptr()),
                                                             __import__("ctypes").c_size_t(B * H * W), __import__("ctypes").c_float(1.0 / 5000.0),
                                                             __import__("ctypes").c_void_p(depth_d.data_ptr())), "pslfe_depth_to_float_device")
        if stream_l is not None:
            stream_l.wait_stream(stream)     # the inputs (and the previous step's record pack) are ordered on the main stream
        pipe.step(d_gray.data_ptr(), depth_d.data_ptr() if LINES else None, n)
        if stream_l is not None:
            stream.wait_stream(stream_l)     # join: the step ends when both pipelines have
        if host_io is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            host_io["ev_done"][(host_io["n"] - 1) & 1] = ev   # the slot may be overwritten once this step's kernels are through
        if gather is not None and n == B:
            if rec_host is not None and host_io.get("ev_d2h", [None, None])[gather.k] is not None:
                stream.wait_event(host_io["ev_d2h"][gather.k])   # the record buffer this pack overwrites has left for the host
            k = gather.submit(pipe.record_sources(mg))
            if rec_host is not None:   # D2H of this rank's packed records on the copy stream, behind the pack
                host_io["ev_pack"].record(stream)
                host_io["cs"].wait_event(host_io["ev_pack"])
                with torch.cuda.stream(host_io["cs"]):
                    rec_host.copy_(gather.send[k], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(host_io["cs"])
                host_io.setdefault("ev_d2h", [None, None])[k] = ev

    for c in pipe.contexts():
        c.profile(True)
    step()  # first call allocates buffers / builds tables (not one of the W warm-up steps)
    torch.cuda.synchronize(dev)
    if world > 1 or args.host_io:
        layout = pipe.record_layout(mg)
        assert layout.bytes <= REC_BYTES, layout.bytes
        root = None if args.gather == "all" else 0

        def bcast(uid, ok):   # rank 0's 128-byte ncclUniqueId and whether it got one, to every rank (always the SAME collective on all ranks)
            if world == 1:
                return uid, ok
            t = torch.from_numpy(np.concatenate([uid, np.array([1 if ok else 0], np.uint8)])).to(dev)
            dist.broadcast(t, 0)
            a = t.cpu().numpy()
            return a[:128].copy(), bool(a[128])
        with stdout_to_stderr():
            try:
                gather = mg.RecordGather(ctx, layout, B, rank, world, dev, bcast, root=root, recv_slots=1 if root is not None else 2)
                kind = "pslfe_gather_to_root (one group of ncclSend / ncclRecv, RCCL through the C ABI)" if root is not None else "pslfe_gather_all (ncclAllGather, RCCL through the C ABI)"
            except Exception as e:   # taken on EVERY rank together (RecordGather agrees before it raises): the same records through torch.distributed
                print(f"[bench] rank {rank}: pslfe_gather unavailable ({e}); gathering the records with torch.distributed", file=sys.stderr)
                gather = mg.TorchRecordGather(ctx, layout, B, world, dev, rank=rank, root=root, recv_slots=1 if root is not None else 2)
                kind = "torch.distributed " + ("gather" if root is not None else "all_gather_into_tensor") + " (backend nccl = RCCL)"
            step()   # the first exchange initialises the communicator's channels
            gather.wait()
            seen = gather.ranks_seen()
        gather_info = {"kind": kind, "root": root, "record_bytes": int(layout.bytes), "ranks_seen": seen,
                       "hardware_status": ("N > 1 exchange first exercised by this run" if world > 1 else "world-size-1 communicator")}
        if args.host_io:
            rec_host = torch.empty((B, layout.bytes), dtype=torch.uint8).pin_memory()

    def read_stages(nsteps):
        out = {}
        for s in STAGE_NAMES:
            for c in pipe.contexts():
                ms, n = c.stage_time(s)
                if n:
                    out[s] = {"ms_per_launch": ms / n, "launches": n, "steps": nsteps}
        return out

    # Warm-up, with every stage timed: finds the dominant stage.  Each timed stage puts two HIP event records between
    # kernels (~10 us of idle GPU per stage boundary), so inside the timed region only the dominant stage is timed.
    if args.warmup > 0:
        for c in pipe.contexts():
            c.profile_reset()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    warm = read_stages(max(1, args.warmup))
    dom = max(warm, key=lambda s: warm[s]["ms_per_launch"])
    for c in pipe.contexts():
        c.profile_reset()
        c.profile_only(dom)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dom_stage = read_stages(args.steps)[dom]  # the dominant kernel over exactly the timed steps

    # ---- parity: frames of the LAST TIMED STEP against the oracle (bit for bit); a mismatch fails the run
    nchk = 0
    if rank == 0 and args.parity_frames > 0:
        cand = sorted(set(int(v) for v in np.linspace(0, B - 1, args.parity_frames)) | {0, min(B - 1, 32), B - 1})[:max(args.parity_frames, 3)]
        cache = {}
        cam = pipe.cam
        for f in cand:
            got = pipe.fetch_frame(f)
            pf = (f - 1) % B
            dep = depth_f32(depth8[(idx[f] // 32) % len(depth8)]) if LINES else None
            ref = BP.oracle_frame((int(idx[pf]), gray256[idx[pf]]), (int(idx[f]), gray256[idx[f]]), dep, f, W, H, LINES, cam, cache=cache)
            BP.compare_frame(got, ref, f"bench frame {f} (distinct frame {int(idx[f])}): ")
            nchk += 1
        if gather is not None and rec_host is not None:   # and the packed record that went to the host is that frame
            u = layout.unpack(rec_host[cand[-1]].numpy())
            g = pipe.fetch_frame(cand[-1])
            assert u["kps"].tobytes() == g["kps"].tobytes() and u["n_match"] == g["nmatches"], "host record differs from the fetched frame"
        if gather is not None and world > 1 and gather.recv[0] is not None:   # what rank 0 received from itself is its own last frame
            got_rec = layout.unpack(gather.result(gather.k ^ 1)[0, B - 1].cpu().numpy())
            g = pipe.fetch_frame(B - 1)
            assert got_rec["kps"].tobytes() == g["kps"].tobytes() and got_rec["n_kl"] == len(g.get("kls", [])), "gathered record differs from the fetched frame"

    # what the line stages worked on (the step's results are still in place): keylines, the LBD's sample count, fans, matches
    line_load = None
    if LINES:
        d_kls, _, _, d_nkl, klcap = pipe.le.results_device()
        nkl = torch.as_tensor(P._DevArray(d_nkl, (B,), "<i4"), device=dev)[:ND].clone()
        kl = torch.as_tensor(P._DevArray(d_kls, (B, klcap, 17), "<i4"), device=dev)[:ND, :, 16]   # PslKeyLine.numOfPixels
        live = torch.arange(klcap, device=dev)[None, :] < nkl[:, None]
        npx = (kl * live).sum(1).double()
        d_fans, d_nfans = pipe.le.fans_device()
        nfans = torch.as_tensor(P._DevArray(d_nfans, (B,), "<i4"), device=dev)[:ND].double()
        line_load = {"mean_keylines": round(float(nkl.double().mean().item()), 1), "mean_fans": round(float(nfans.mean().item()), 1),
                     "mean_lbd_pixels_per_line": round(float((npx.sum() / max(1, int(nkl.sum().item()))).item()), 1),
                     "lbd_sample_bytes_per_frame": int(round(float(npx.mean().item()) * 63 * 4))}

    # per-stage table: a few extra steps after the timed region, every stage timed
    for c in pipe.contexts():
        c.profile_reset()
        c.profile_only(None)
    for _ in range(min(args.steps, 5)):
        step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize(dev)
    stages = read_stages(min(args.steps, 5))
    stages[dom] = dom_stage
    counts = torch.as_tensor(P.orb_results_as_arrays(pipe.orb, B)[2], device=dev)
    mean_kp = float(counts.float().mean().item())
    mean_matches = float(pipe.nmatches.float().mean().item())
    mean_lm = float(pipe.lnm.float().mean().item()) if LINES else 0.0

    # like for like with rounds 1 - 2 and with runs that do not get 12288 frames into memory: 6144 frames per launch, 5 steps, untimed stages
    lfl = None
    if LINES and B > 6144 and not args.no_like_for_like and stream_l is None:
        for c in pipe.contexts():
            c.profile(False)
        nB[0] = 6144
        step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        d1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([d1], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d1 = float(t.item())
        lfl = {"frames_per_step_per_gpu": 6144, "steps": 5, "value": round(world * 6144 * 5 / d1, 1), "unit": "frames/s", "ms_per_step": round(d1 / 5 * 1e3, 3),
               "note": "same pipeline object, 6144 frames per launch (no gather in these steps)"}
        nB[0] = B
    prev_scene = None
    if gray_prev is not None and stream_l is None:
        tmp = torch.from_numpy(gray_prev).to(dev)
        for lo in range(0, B, ND):   # the 256 distinct frames repeated, without a second batch-sized temporary
            n_ = min(ND, B - lo)
            frames_d[lo:lo + n_].copy_(tmp[:n_])
        del tmp
        torch.cuda.empty_cache()
        step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        prev_scene = {"scene": "struct", "frames_per_step_per_gpu": B, "steps": 5, "value": round(B * 5 / d1, 1), "unit": "frames/s", "ms_per_step": round(d1 / 5 * 1e3, 3),
                      "note": "the scene the headline of rounds 1 - 2 was measured on (overlapping polygons: ~29 keylines per frame), same pipeline object and "
                              "batch; BENCH_r02: 52 495.5 frames/s"}
    segs = None
    if LINES and rank == 0 and not profiled:   # LSD segments in front of the merging, on 16 of the distinct frames (single-frame entry point; not under
        # a profiler: its single-frame launches would dilute the per-kernel averages of the trace)
        le1 = P.LINEextractor(1, 1.2, 200, 0.0, ctx=ctx)
        segs = round(float(np.mean([len(le1.lsd_detect(gray256[i])) for i in range(0, ND, max(1, ND // 16))])), 1)
        le1.close()
    for c in pipe.contexts():
        c.profile(False)

    if rank == 0:
        fps = world * B * args.steps / dt
        stage_bytes = dict(STAGE_BYTES_PER_FRAME)
        line_bytes = LINE_BYTES_PER_FRAME
        if line_load is not None:   # the LBD's gathers from the MEASURED sum of numOfPixels x 63 rows x 4 B instead of NL = 200, L = 100
            stage_bytes["line.lbd"] = line_load["lbd_sample_bytes_per_frame"] + 100 * int(round(line_load["mean_keylines"]))
            line_bytes = LINE_BYTES_PER_FRAME - (5040000 + 20000) + stage_bytes["line.lbd"]
        dom_bytes = stage_bytes[dom] * B
        dom_s = stages[dom]["ms_per_launch"] * 1e-3
        achieved = dom_bytes / dom_s / 1e9
        traffic, traffic_src = pmc_traffic(args.workload + ("" if scene in ("sticks", "desk") else "_" + scene), dom, B)
        per_frame = ORB_BYTES_PER_FRAME + MATCH_BYTES_PER_FRAME + (line_bytes if LINES else 0)
        io = ("host gray + depth(u16) in, host result records out inside the timed region (one H2D and one D2H per step on a copy stream: the next batch comes in "
              "and the previous records go out while a batch is computed)" if args.host_io else "frames resident in HBM")
        scene_txt = {"sticks": "synthetic structure scene at the configured line load ('sticks': non-overlapping high-contrast bars)",
                     "struct": "synthetic structure-notexture-like stream ('struct': overlapping polygons, the scene of rounds 1 - 2)",
                     "desk": "synthetic RGB-D stream (desk-like)"}[scene]
        out = {
            "metric": ("frames/sec ORB+line extract+match, 640x480 RGB-D, 1/2/4/8 MI355X" if LINES
                       else "frames/sec ORB-only extract+match, 640x480 RGB-D (BASELINE configs[1])"),
            "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"configs[2]: 640x480 {scene_txt}, ORB 1000/1.2/8 FAST 20/7 + LSD (LSD_REFINE_ADV) / merge / top-200 / LBD "
                                    "+ LIL pairing + RGB-D line glue (isLineGood, crossings, planes) extract, SearchByProjection(cur,last) + LSDmatcher::match, "
                                    + io) if LINES else
                                   (f"configs[1]: 640x480 {scene_txt}, ORB 1000/1.2/8 FAST 20/7 "
                                    "extract + SearchByProjection(cur,last) match, " + io),
                       "scene": scene, "frames_per_step_per_gpu": B, "distinct_frames_per_batch": int(min(B, ND)), "mean_keypoints": round(mean_kp, 1),
                       "mean_matches": round(mean_matches, 1), "host_io": bool(args.host_io), "streams": args.streams,
                       "hbm_in_use_GB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                       "launcher": os.environ.get("PSLFE_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none"),
                       "multi_gpu": ("independent stream per rank, no data-path collective; per-frame result records (counts, keypoints, descriptors, point "
                                     "matches, keylines, LBD descriptors, line equations, line matches, fans, planes) packed on the device and gathered "
                                     f"per batch: {gather_info['kind']}") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": round(stages[dom]["ms_per_launch"], 4)},
            "pipeline_roofline": {"bytes_per_frame": per_frame,
                                  "achieved_GBs": round(fps / world * per_frame / 1e9, 2),
                                  "frac_of_8TBs": round(fps / world * per_frame / 1e9 / HBM_PEAK_GBS, 5),
                                  "note": "SURVEY.md §8(d)'s accounting with the LBD term taken from the measured keylines of this workload"},
            "stages_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in stages.items()},
            "stages_ms_per_step": {k: round(v["ms_per_launch"] * v["launches"] / max(1, v["steps"]), 4) for k, v in stages.items()},
            # algorithmic bytes of a stage (per launch for the NFA stages, whose table entry is per launch; per step otherwise) / its time
            "stages_frac_of_hbm_peak": {k: round(stage_bytes[k] * B / ((v["ms_per_launch"] if k in PER_LAUNCH_STAGES else
                                                                        v["ms_per_launch"] * v["launches"] / max(1, v["steps"])) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                        for k, v in stages.items() if k in stage_bytes and v["ms_per_launch"] > 0},
            "parity_checked_frames": nchk,
        }
        if gather_info is not None:
            out["gather"] = gather_info
        if prev_scene is not None:
            out["previous_rounds_scene"] = prev_scene
            out["note"] = ("round 3 measures configs[2] on a scene that carries the configured line load (config.mean_keylines per frame; 29 in rounds 1 - 2, whose "
                           "scene is measured in the same run: previous_rounds_scene); a frame of this scene is 4.3 times the LSD region-growing work of that one")
        if lfl is not None:
            out["like_for_like"] = lfl
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if LINES:
            out["config"]["mean_line_matches"] = round(mean_lm, 1)
            out["config"]["lsd_refine"] = "LSD_REFINE_ADV"
            out["config"].update(line_load)
            out["config"]["mean_segments"] = segs
        print(json.dumps(out), flush=True)
    if gather is not None:
        gather.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
