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
    "line.lsd_scale": [("k_lsd_scale_tiled", 1)], "line.lsd_grad": [("k_lsd_grad", 1)], "line.lsd_grow": [("k_lsd_grow4<0>", 1)],
    "line.nfa_count": [("k_lsd_nfa_count<%d>" % ph, 1) for ph in (-2, -1, 0, 1, 2, 3)],
    "line.nfa_eval": [("k_lsd_nfa_%s<%d>" % (k, ph), 1) for k in ("setup", "series", "select") for ph in (-2, -1, 0, 1, 2, 3)],
    "line.merge": [("k_line_merge<512>", 1), ("k_line_merge<1024>", 1)], "line.lbd_pre": [("k_lbd_pre", 1)], "line.lbd": [("k_lbd", 1)], "line.pair": [("k_lil_pair", 1)],
    "line.match": [("k_line_match_batch", 1)], "line.good": [("k_line_good", 1)], "line.planes": [("k_fans_planes", 1)],
}
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
            k = j["kernels"][kern]
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


def distinct_frames(w, h, style, seed0, nscenes=8, nt=32):
    """(gray [nscenes*nt][h][w] u8, depth_u16 [nscenes][h][w]): the frames of a scene are consecutive (drift <= 2 px per frame), scenes
    follow each other (a cut every nt frames).  Depth is a tilted plane per scene (x5000, TUM convention)."""
    import multiprocessing as mp
    path = os.path.join(tempfile.gettempdir(), f"pslfe_bench_{w}x{h}_{style}_{seed0}_{nscenes}x{nt}.npz")
    if os.path.exists(path):
        try:
            z = np.load(path)
            return z["gray"], z["depth"]
        except Exception:
            pass
    jobs = [(w, h, style, seed0 + 17 * s, nt, 1.0 + 0.04 * s) for s in range(nscenes)]
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
    import oracle_lib
    return oracle_lib.depth_to_float(np.ascontiguousarray(depth_u16), np.float32(1.0) / np.float32(5000.0))


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
    nframes = args.steps + args.warmup
    gray, depth = D.synth_stream(w, h, nframes, "struct", seed_for(0))
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
    D.run(cpath, nf, nl, 0, results_path=rpath)                     # untimed: the results the parity check reads
    got = D.read_results(rpath, ncheck)
    ref = D.oracle_sequence(gray[:ncheck], depth[:ncheck], nf, nl)
    for t in range(ncheck):
        D.compare(got[t], ref[t], f"frame {t}: ")
    r = D.run(fpath, nf, nl, args.warmup)                            # the timed run: host clock around every frame, copies included
    rs = D.run(fpath, nf, nl, args.warmup, stages=True)              # same again with the library's per-kernel HIP-event timers on
    stages = rs.get("gpu_stage_ms_per_frame", {})
    dom = max(stages, key=lambda s: stages[s]) if stages else "line.lsd_grow"
    scale_bytes = (w * h) / float(W * H)
    dom_bytes = int(STAGE_BYTES_PER_FRAME.get(dom, 0) * scale_bytes)
    dom_ms = stages.get(dom, 0.0)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    out = {
        "metric": "frames/sec ORB+line extract+match, single-frame drop-in (B = 1), host buffers in / host results out",
        "value": round(1e3 / r["ms_per_frame"]["mean"], 2), "unit": "frames/s", "n_gpus": 1, "steps": r["frames_timed"], "warmup": args.warmup,
        "ms_per_step": round(r["ms_per_frame"]["mean"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": (f"configs[4]: {w}x{h} synthetic structure-like RGB-D stream, 2000 ORB + 200 lines, the C-ABI sequence a Tracking loop issues "
                                "(Frame::Frame src/Frame.cc:133-208 + TrackWithMotionModel src/Tracking.cc:1164-1214) through the compiled C++ consumer "
                                "tools/dropin/dropin_main.cpp, one frame at a time, H2D / D2H inside the timed region") if tracking else
                               (f"configs[2] shape, B = 1: {w}x{h} synthetic structure-like RGB-D stream, 1000 ORB + 200 lines, Frame::Frame + "
                                "TrackWithMotionModel through the compiled C++ consumer tools/dropin/dropin_main.cpp, one frame at a time, H2D / D2H inside "
                                "the timed region"),
                   "frames_per_step_per_gpu": 1, "mean_keypoints": r["mean_keypoints"], "mean_keylines": r["mean_keylines"],
                   "mean_matches": r["mean_matches"], "mean_line_matches": r["mean_line_matches"], "lsd_refine": "LSD_REFINE_ADV"},
        "latency_ms_per_frame": r["ms_per_frame"], "frame_ctor_ms": r["frame_ctor_ms"], "track_ms": r["track_ms"],
        "calls_ms_mean": r["calls_ms_mean"], "gpu_stage_ms_per_frame": stages,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": None, "algorithmic_bytes_per_launch": dom_bytes,
                     "ms_per_launch": round(dom_ms, 4),
                     "note": "one frame per launch: a serial chain on one wave, latency-bound (DESIGN.md §5)"},
        "parity_checked_frames": ncheck,
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 20; 100 frames for dropin / 40 for tracking)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default 3; 10 frames for dropin / tracking)")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default 12288 = two rounds of the 6144 wave slots of the LSD growing, 281 GB of HBM; 6144 per rank with N > 1; 256 for --workload orb)")
    ap.add_argument("--workload", choices=["orb", "lines", "dropin", "tracking"], default="lines",
                    help="lines = BASELINE configs[2], the configuration of the headline metric; orb = configs[1]; dropin = B = 1 through the "
                         "C++ consumer; tracking = configs[4] through the C++ consumer")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: the line pipeline on its own context / stream beside the ORB pipeline (the two extractor objects of a Frame are "
                         "independent); stages then overlap and their event timings stop meaning what they say, so 1 is the default")
    ap.add_argument("--host-io", action="store_true", help="host gray + depth in, host result records out, inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-frames", type=int, default=8)
    args = ap.parse_args()
    consumer = args.workload in ("dropin", "tracking")
    if args.steps <= 0:
        args.steps = (100 if args.workload == "dropin" else 40) if consumer else 20
    if args.warmup < 0:
        args.warmup = 10 if consumer else 3

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if consumer:
        assert world == 1, "the single-frame workloads run on one GPU"
        return run_consumer(args)

    LINES = args.workload == "lines"
    # 12288 frames per launch = two rounds of the 6144 wave slots of the LSD growing (+3.8 % frames/s over 6144: the launch lasts as long as
    # its longest frames) and 281 GB of the GPU's 309 GB; with N > 1 the all-gather's receive buffers (world x batch x 85 KB, twice) do not
    # fit beside that, so every rank runs 6144 frames per launch (~165 GB); the same with --host-io (pinned staging + record buffers: 295 GB at 12288)
    B = args.batch or ((12288 if world == 1 and not args.host_io else 6144) if LINES else 256)
    # ---- everything that forks worker processes happens BEFORE the GPU is touched: input frames and the CPU baseline
    gray256, depth8 = distinct_frames(W, H, "struct" if LINES else "desk", seed_for(rank))
    ND = len(gray256)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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

    import psl_slam_amd as P
    from importlib import import_module
    import batch_pipeline as BP
    mg = import_module("psl_slam_amd.multigpu")
    P.build()

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
        host_io = {"gray": torch.from_numpy(gray256[idx]).pin_memory()}
        if LINES:
            host_io["depth16"] = torch.from_numpy(np.ascontiguousarray(depth8[(idx // 32) % len(depth8)]).view(np.int16)).pin_memory()  # u16 bits
            host_io["d_depth16"] = torch.empty((B, H, W), dtype=torch.int16, device=dev)

    gather = None
    gather_kind = None
    layout = None
    rec_host = None

    def step():
        if host_io is not None:   # H2D inside the step; depth arrives as the sensor's u16 and is converted on the device (src/Tracking.cc:230-235)
            frames_d.copy_(host_io["gray"], non_blocking=True)
            if LINES:
                host_io["d_depth16"].copy_(host_io["depth16"], non_blocking=True)
                P._check(P.lib().pslfe_depth_to_float_device(ctx._h, __import__("ctypes").c_void_p(host_io["d_depth16"].data_ptr()),
                                                             __import__("ctypes").c_size_t(B * H * W), __import__("ctypes").c_float(1.0 / 5000.0),
                                                             __import__("ctypes").c_void_p(depth_d.data_ptr())), "pslfe_depth_to_float_device")
        if stream_l is not None:
            stream_l.wait_stream(stream)     # the inputs (and the previous step's record pack) are ordered on the main stream
        pipe.step(frames_d.data_ptr(), depth_d.data_ptr() if LINES else None)
        if stream_l is not None:
            stream.wait_stream(stream_l)     # join: the step ends when both pipelines have
        if gather is not None:
            k = gather.submit(pipe.record_sources(mg))
            if rec_host is not None:   # D2H of this rank's packed records inside the step
                rec_host.copy_(gather.send[k], non_blocking=True)

    for c in pipe.contexts():
        c.profile(True)
    step()  # first call allocates buffers / builds tables (not one of the W warm-up steps)
    torch.cuda.synchronize(dev)
    if world > 1 or args.host_io:
        layout = pipe.record_layout(mg)

        def bcast(uid):
            if world == 1:
                return uid
            t = torch.from_numpy(uid.copy()).to(dev)
            dist.broadcast(t, 0)
            return t.cpu().numpy()
        with stdout_to_stderr():
            try:
                gather = mg.RecordGather(ctx, layout, B, rank, world, dev, bcast)
            except Exception as e:   # RCCL not loadable through the C ABI on this node: the same records through torch.distributed
                print(f"[bench] pslfe_gather unavailable ({e}); gathering the records with torch.distributed", file=sys.stderr)
                gather = mg.TorchRecordGather(ctx, layout, B, world, dev)
            gather_kind = type(gather).__name__
            step()   # the first exchange initialises the communicator's channels
            gather.wait()
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
        if gather is not None and world == 1:   # and the packed record that went to the host is that frame
            u = layout.unpack(rec_host[cand[-1]].numpy())
            g = pipe.fetch_frame(cand[-1])
            assert u["kps"].tobytes() == g["kps"].tobytes() and u["n_match"] == g["nmatches"], "host record differs from the fetched frame"

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
    for c in pipe.contexts():
        c.profile(False)
    stages[dom] = dom_stage
    counts = torch.as_tensor(P.orb_results_as_arrays(pipe.orb, B)[2], device=dev)
    mean_kp = float(counts.float().mean().item())
    mean_matches = float(pipe.nmatches.float().mean().item())

    if rank == 0:
        fps = world * B * args.steps / dt
        dom_bytes = STAGE_BYTES_PER_FRAME[dom] * B
        dom_s = stages[dom]["ms_per_launch"] * 1e-3
        achieved = dom_bytes / dom_s / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, dom, B)
        per_frame = ORB_BYTES_PER_FRAME + MATCH_BYTES_PER_FRAME + (LINE_BYTES_PER_FRAME if LINES else 0)
        io = ("host gray + depth(u16) in, host result records out inside the timed region" if args.host_io else "frames resident in HBM")
        out = {
            "metric": ("frames/sec ORB+line extract+match, 640x480 RGB-D, 1/2/4/8 MI355X" if LINES
                       else "frames/sec ORB-only extract+match, 640x480 RGB-D (BASELINE configs[1])"),
            "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[2]: 640x480 synthetic structure-notexture-like stream, ORB 1000/1.2/8 FAST 20/7 + LSD (LSD_REFINE_ADV) / merge / LBD 200 lines "
                                    "+ LIL pairing + RGB-D line glue (isLineGood, crossings, planes) extract, SearchByProjection(cur,last) + LSDmatcher::match, "
                                    + io) if LINES else
                                   ("configs[1]: 640x480 synthetic RGB-D stream (desk-like), ORB 1000/1.2/8 FAST 20/7 "
                                    "extract + SearchByProjection(cur,last) match, " + io),
                       "frames_per_step_per_gpu": B, "distinct_frames_per_batch": int(min(B, ND)), "mean_keypoints": round(mean_kp, 1),
                       "mean_matches": round(mean_matches, 1), "host_io": bool(args.host_io), "streams": args.streams,
                       "hbm_in_use_GB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                       "multi_gpu": ("independent stream per rank; per-frame result records (counts, keypoints, descriptors, point matches, keylines, LBD "
                                     "descriptors, line equations, line matches, fans, planes) packed and all-gathered with RCCL through the C ABI "
                                     f"({'pslfe_gather_all' if gather_kind == 'RecordGather' else 'torch.distributed all_gather_into_tensor'}), "
                                     f"{layout.bytes} B per frame") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": round(stages[dom]["ms_per_launch"], 4)},
            "pipeline_roofline": {"bytes_per_frame": per_frame,
                                  "achieved_GBs": round(fps / world * per_frame / 1e9, 2),
                                  "frac_of_8TBs": round(fps / world * per_frame / 1e9 / HBM_PEAK_GBS, 5)},
            "stages_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in stages.items()},
            "stages_ms_per_step": {k: round(v["ms_per_launch"] * v["launches"] / max(1, v["steps"]), 4) for k, v in stages.items()},
            "parity_checked_frames": nchk,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if LINES:
            out["config"]["mean_line_matches"] = round(float(pipe.lnm.float().mean().item()), 1)
            out["config"]["lsd_refine"] = "LSD_REFINE_ADV"
        print(json.dumps(out), flush=True)
    if gather is not None:
        gather.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
