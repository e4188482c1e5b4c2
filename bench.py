#!/usr/bin/env python3
"""bench.py — frames/sec of the PSL-SLAM per-frame feature front-end on MI355X.

A "step" = one pass of the hot path over one batch of B synthetic 640x480 RGB-D frames that are already
resident in HBM.  Default workload = the one BASELINE.json's metric ("frames/sec ORB+line extract+match, 640x480
RGB-D") is quoted on, configs[2]: ORB extraction (pyramid, per-cell FAST, octree distribution, orientation, blur,
rBRIEF) + frame grid + ORBmatcher::SearchByProjection(cur,last) against the predecessor frame, and the line path
(LSD, merge, LBD, top-200, LIL pairing, the RGB-D line glue of the Frame constructor, LSDmatcher::match), 6144
frames per launch.  `--workload orb` = configs[1] (ORB-only extract+match, 256 frames per launch).  Results stay in
HBM; with N > 1 every rank runs its own independent stream (weak scaling) and the per-frame result
records are all-gathered over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement), including
  roofline     for the dominant kernel: algorithmic bytes per launch / mean launch time (HIP events
               on the launch stream over the timed region) vs the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (oracle/, kind "port": the reference itself cannot be built here)
               timed single-threaded on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 640, 480
NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH = 1000, 1.2, 8, 20, 7
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# ALGORITHMIC bytes per frame, SURVEY.md §8(d) / BASELINE.md §3 (640x480, N = 1000, P = 950 532 px):
#   gray read WH + pyramid write P-WH + pyramid read P + blurred write P + sampling 512N + outputs 60N
P_PIX, WH = 950532, W * H
ORB_BYTES_PER_FRAME = WH + (P_PIX - WH) + P_PIX + P_PIX + 512 * NFEATURES + 60 * NFEATURES  # 3 423 596
MATCH_BYTES_PER_FRAME = 2 * 32 * NFEATURES + 8 * NFEATURES                                   # 72 000
# per-kernel shares of that accounting: what each kernel must move at least
STAGE_BYTES_PER_FRAME = {
    "orb.pyramid": WH + (P_PIX - WH),        # read level 0, write levels 1..7
    "orb.fast": P_PIX,                       # read every level once
    "orb.fast0": WH,                         # PSLFE_OVERLAP=1 only: level 0 in its own launch beside the pyramid
    "orb.octree": 8 * 4000,                  # candidate list in + out (~4 k candidates x 8 B); not in §8(d)
    "orb.blur": P_PIX + P_PIX,               # read every level, write its blurred copy
    "orb.describe": 512 * NFEATURES + 60 * NFEATURES,
    "match.grid": 2 * 60 * NFEATURES,
    "match.window": MATCH_BYTES_PER_FRAME,
    # line path, SURVEY.md §8(d) "Lines @640x480" = 17 040 800 B/frame split over the kernels that move them
    "line.lsd_scale": WH + 8 * 196608,                 # gray read + f64 working image write
    "line.lsd_grad": 8 * 196608 + 2 * 8 * 196608,      # working image read + angle & modgrad write
    "line.lsd_grow": 2 * 8 * 196608 + 2 * 8 * 196608 + 393216,  # angle & modgrad read + coordinate list w+r + used map
    "line.merge": 2 * 16 * 500,
    "line.lbd_pre": WH + 4 * WH,                       # gray read, Sobel (dx, dy) s16 write (the blurred image stays in LDS)
    "line.lbd": 5040000 + 20000,                       # LSR gathers + outputs
    "line.pair": 2 * 16 * 200,
    "line.match": 12800,
    "line.good": 200 * 21 * 4 + 200 * (68 + 48 + 12),  # depth samples + keylines in, 3-D lines out
    "line.planes": 4096 * 16 + 200 * 60,
}
LINE_BYTES_PER_FRAME = 17040800 + 12800
if os.environ.get("PSLFE_OVERLAP"):
    STAGE_BYTES_PER_FRAME["orb.fast"] = P_PIX - WH  # levels 1..7 in the launch after the pyramid

# stage -> kernels it launches (kernels per step); HBM traffic of a stage = sum over its launches
STAGE_KERNELS = {
    "orb.pyramid": [("k_pyr_resize_tiled", NLEVELS - 1)], "orb.fast": [("k_fast_cells4<1>", 1)], "orb.fast0": [("k_fast_cells4<0>", 1)], "orb.octree": [("k_octree<256>", 1)],
    "orb.blur": [("k_blur7", 1)], "orb.describe": [("k_orient_describe", 1)],
    "match.grid": [("k_frame_import", 1), ("k_build_grid", 1)], "match.window": [("k_window_eval", 1), ("k_window_resolve<0, 4096, 1024>", 1)],
    "line.lsd_scale": [("k_lsd_scale_tiled", 1)], "line.lsd_grad": [("k_lsd_grad", 1)], "line.lsd_grow": [("k_lsd_grow3", 1)],
    "line.merge": [("k_line_merge<512>", 1), ("k_line_merge<1024>", 1)], "line.lbd_pre": [("k_lbd_pre", 1)], "line.lbd": [("k_lbd", 1)], "line.pair": [("k_lil_pair", 1)],
    "line.match": [("k_line_match_batch", 1)], "line.good": [("k_line_good", 1)], "line.planes": [("k_fans_planes", 1)],
}


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


def synth_batch(batch, seed, n_distinct=16, style="desk"):
    import synth_frames as sf
    sc = sf.Scene(W, H, style, seed)
    base = np.stack([sc.gray(t) for t in range(min(n_distinct, batch))], 0)
    reps = (batch + len(base) - 1) // len(base)
    # forward then backward in time so consecutive frames always differ by one drift step
    seq = np.concatenate([base, base[::-1]], 0)
    return np.ascontiguousarray(np.concatenate([seq] * reps, 0)[:batch])


def bench_kernels():
    """Harness-only HIP kernel (tools/bench_kernels/bench_kernels.hip): builds the projection queries of a step in one
    launch.  Not part of libpslfe."""
    import ctypes as C
    d = os.path.join(ROOT, "tools", "bench_kernels")
    so, src = os.path.join(d, "libbench_kernels.so"), os.path.join(d, "bench_kernels.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so, src], check=True, capture_output=True)
    lib = C.CDLL(so)
    lib.bench_queries_from_prev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def cpu_baseline(sample_frames, lines=False, depth_frames=None, cam=None):
    """Oracle (CPU restatement) timed single-threaded on the same workload: extract + match."""
    import ctypes as C
    import oracle_lib
    odir = os.path.join(ROOT, "oracle")
    native = os.path.join(odir, "libpsl_oracle_native.so")
    try:  # -O3 -march=native build for timing, made on this host
        srcs = sorted(os.path.join(odir, f) for f in os.listdir(odir) if f.endswith(".cpp"))
        subprocess.run(["g++", "-O3", "-march=native", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", native] + srcs + ["-lm"],
                       check=True, capture_output=True)
        oracle_lib.SO = native
        oracle_lib._lib = None
    except Exception:
        pass
    orc = oracle_lib.OracleORB(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH)
    scale = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(NLEVELS - 1, SCALE, np.float64)])).astype(np.float32)
    prev = None
    t0 = time.perf_counter()
    n = 0
    for img in sample_frames:
        kps, desc = orc(img)
        if prev is not None:
            q = np.zeros(len(prev[0]), oracle_lib.PROJQUERY_DTYPE)
            q["u"], q["v"] = prev[0]["x"], prev[0]["y"]
            q["radius"] = np.float32(15.0) * scale[prev[0]["octave"]]
            q["min_level"], q["max_level"] = prev[0]["octave"] - 1, prev[0]["octave"] + 1
            q["angle"], q["blocks"] = prev[0]["angle"], 1
            oracle_lib.search_by_projection_last(kps, desc, None, (0.0, 0.0, float(W), float(H)), q, prev[1], None, True)
        if lines:
            kls, ldesc, _ = oracle_lib.line_extract(img, 200)
            L4 = np.stack([kls[k] for k in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32) if len(kls) else np.zeros((0, 4), np.float32)
            fans = oracle_lib.lil_pair(L4, 20.0, np.float32(np.pi / 4), W, H)
            if depth_frames is not None:  # isLineGood + convertFansToKeyLines + planes (src/Frame.cc:500-660)
                oracle_lib.frame_glue(kls, fans, depth_frames[n % len(depth_frames)], cam, seed=1 + n)
            if prev is not None and len(prev) > 2:
                oracle_lib.line_match_nnr(prev[2], ldesc, 0.9)
            prev = (kps, desc, ldesc)
        else:
            prev = (kps, desc)
        n += 1
        if time.perf_counter() - t0 > 20.0:
            break
    dt = time.perf_counter() - t0
    what = ("ORB 1000 + LSD/merge/LBD 200 + LIL pairing + RGB-D line glue extract, SearchByProjection(cur,last) + matchNNR" if lines
            else "ORB 1000 extract + SearchByProjection(cur,last)")
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames 640x480 synthetic stream, {what}, "
                      f"oracle/ built -O3 -march=native, 1 thread, {dt:.1f} s on {os.cpu_count()} host cores available"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default 6144 = 6 LSD waves per SIMD; 256 for --workload orb)")
    ap.add_argument("--workload", choices=["orb", "lines"], default="lines",
                    help="lines = BASELINE configs[2], the configuration of the headline metric (ORB + LSD/LBD + pairing + glue, "
                         "extract+match); orb = configs[1] (ORB-only extract+match)")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2],
                    help="2: the line pipeline runs on its own context/stream beside the ORB pipeline (they are independent, as the "
                         "two extractor objects of a Frame are; +6.5 %% frames/s measured, but stages then overlap and their event timings "
                         "stop meaning what they say); 1 = default: everything on one stream, every stage timed alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import psl_slam_amd as P
    from importlib import import_module
    multigpu = import_module("psl_slam_amd.multigpu")
    P.build()
    LINES = args.workload == "lines"
    B = args.batch or (6144 if LINES else 256)
    frames_h = synth_batch(B, P_seed(rank), style="struct" if LINES else "desk")
    frames_d = torch.from_numpy(frames_h).to(dev)

    # a real (non-null) torch stream: the library launches on it, so torch ops, RCCL and the HIP
    # kernels are ordered on one stream and torch.cuda events/synchronize see everything
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = P.Context(local_rank, stream.cuda_stream)
    orb = P.ORBextractor(NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, ctx=ctx, max_batch=B)
    cap = orb.max_keypoints(W, H)
    grid = P.FrameGrid(cap, B, ctx=ctx)
    bounds = (0.0, 0.0, float(W), float(H))

    # device-side stand-in for Tracking's constant-velocity projection: the last frame's keypoints
    # are predicted at the same pixel (drift <= 2 px/frame), window th = 15 * scale[octave]
    scale_t = torch.tensor(np.cumprod(np.concatenate([[np.float32(1.0)], np.full(NLEVELS - 1, SCALE, np.float64)])).astype(np.float32), device=dev)
    queries = torch.zeros((B, cap, 8), dtype=torch.float32, device=dev)
    qdesc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    nq = torch.zeros((B,), dtype=torch.int32, device=dev)
    match = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
    nmatches = torch.zeros((B,), dtype=torch.int32, device=dev)
    q_i32 = queries.view(torch.int32)

    BK = bench_kernels()
    le = None
    nstreams = args.streams or 1
    ctx_l = ctx
    if LINES and nstreams == 2:
        stream_l = torch.cuda.Stream(dev)
        ctx_l = P.Context(local_rank, stream_l.cuda_stream)
    if LINES:
        le = P.LINEextractor(1, 1.2, 200, 0.0, ctx=ctx_l, max_batch=B)
        le.extract_batch_device(frames_d.data_ptr(), B, W, H, W, W * H)
        _, _, _, _, klcap = le.results_device()
        lmatch = torch.full((B, klcap), -1, dtype=torch.int32, device=dev)
        lnm = torch.zeros((B,), dtype=torch.int32, device=dev)
        # RGB-D glue of the Frame constructor (isLineGood, convertFansToKeyLines, planes): needs the depth frames
        import synth_frames as sf
        dsc = sf.Scene(W, H, "struct", P_seed(rank))
        depth_h = np.stack([dsc.depth_u16(t).astype(np.float32) / np.float32(5000.0) for t in range(16)], 0)
        # tiled on the device: the host never holds more than the 16 distinct depth frames (8 ranks share one node's memory)
        depth_d = torch.from_numpy(depth_h).to(dev).repeat((B + 15) // 16, 1, 1)[:B].contiguous()
        glue = P.FrameGlue(max_lines=klcap, max_fans=4096, max_batch=B, ctx=ctx_l)
        cam = np.zeros((), P.CAMERA_DTYPE)
        for k_, v_ in zip(P.CAMERA_DTYPE.names, (517.306408, 516.469215, 318.643040, 255.313989, 0, 0, 0, 0, 0, 40.0)):
            cam[k_] = np.float32(v_)

    gather = None
    ctxs = [ctx] if ctx_l is ctx else [ctx, ctx_l]

    def step():
        orb.extract_batch_device(frames_d.data_ptr(), B, W, H, W, W * H)
        k_arr, d_arr, c_arr, _ = P.orb_results_as_arrays(orb, B)
        kps = torch.as_tensor(k_arr, device=dev)
        desc = torch.as_tensor(d_arr, device=dev)
        counts = torch.as_tensor(c_arr, device=dev)
        grid.set_from_orb(orb, bounds)
        # queries of pair f = keypoints of frame f-1 (cyclic inside the batch), one harness kernel on the same stream
        rc = BK.bench_queries_from_prev(stream.cuda_stream, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), B, cap, NLEVELS,
                                        scale_t.data_ptr(), 15.0, queries.data_ptr(), qdesc.data_ptr(), nq.data_ptr())
        assert rc == 0
        P.search_by_projection_last_device(grid, 0, B, queries.data_ptr(), qdesc.data_ptr(), nq.data_ptr(), cap, True,
                                           match.data_ptr(), nmatches.data_ptr())
        if LINES:
            le.extract_batch_device(frames_d.data_ptr(), B, W, H, W, W * H)   # LSD -> merge -> top-200 -> LBD -> line equations
            le.pair_batch_device(20.0, float(np.float32(np.pi / 4)))          # CPartiallyRecoverConnectivity (src/Frame.cc:505)
            le.match_batch_device(1, 0.9, lmatch.data_ptr(), lnm.data_ptr())  # lmatcher.match(last, cur, 0.9) (src/Tracking.cc:901)
            d_kls_, _, _, d_nkl_, _ = le.results_device()
            d_fans_, d_nfans_ = le.fans_device()
            glue.run_batch_device(B, d_kls_, klcap, d_nkl_, d_fans_, 4096, d_nfans_, depth_d.data_ptr(), W, H, cam, 1)  # src/Frame.cc:500-660
        if gather is not None:
            gather.submit([counts, kps, desc, match, nmatches])
        return counts

    for c in ctxs:
        c.profile(True)
    counts = step()  # first call allocates buffers / builds tables (not one of the W warm-up steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        k_arr, d_arr, c_arr, _ = P.orb_results_as_arrays(orb, B)
        tmpl = [torch.as_tensor(c_arr, device=dev), torch.as_tensor(k_arr, device=dev), torch.as_tensor(d_arr, device=dev), match, nmatches]
        gather = multigpu.ResultGather(tmpl, world, dev)

    stage_names = ["orb.pyramid", "orb.fast0", "orb.fast", "orb.octree", "orb.blur", "orb.describe", "match.grid", "match.window",
                   "line.lsd_scale", "line.lsd_grad", "line.lsd_grow", "line.merge", "line.lbd_pre", "line.lbd", "line.pair", "line.match", "line.good", "line.planes"]

    def read_stages():
        out = {}
        for s in stage_names:
            for c in ctxs:
                ms, n = c.stage_time(s)
                if n:
                    out[s] = {"ms_per_launch": ms / n, "launches": n}
        return out

    # Warm-up, with every stage timed: finds the dominant stage.  Each timed stage puts two HIP event records between
    # kernels (~10 us of idle GPU per stage boundary), so inside the timed region only the dominant stage is timed.
    if args.warmup > 0:
        for c in ctxs:
            c.profile_reset()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    warm = read_stages()
    dom = max(warm, key=lambda s: warm[s]["ms_per_launch"])
    for c in ctxs:
        c.profile_reset()
        c.profile_only(dom)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
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
    dom_stage = read_stages()[dom]  # the dominant kernel over exactly the timed steps
    # per-stage table: a few extra steps after the timed region, every stage timed
    for c in ctxs:
        c.profile_reset()
        c.profile_only(None)
    for _ in range(min(args.steps, 5)):
        step()
    torch.cuda.synchronize(dev)
    stages = read_stages()
    for c in ctxs:
        c.profile(False)
    stages[dom] = dom_stage
    mean_kp = float(counts.float().mean().item())
    mean_matches = float(nmatches.float().mean().item())

    if rank == 0:
        fps = world * B * args.steps / dt
        dom_bytes = STAGE_BYTES_PER_FRAME[dom] * B
        dom_s = stages[dom]["ms_per_launch"] * 1e-3
        achieved = dom_bytes / dom_s / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, dom, B)
        per_frame = ORB_BYTES_PER_FRAME + MATCH_BYTES_PER_FRAME + (LINE_BYTES_PER_FRAME if LINES else 0)
        out = {
            "metric": ("frames/sec ORB+line extract+match, 640x480 RGB-D, 1/2/4/8 MI355X" if LINES
                       else "frames/sec ORB-only extract+match, 640x480 RGB-D (BASELINE configs[1])"),
            "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[2]: 640x480 synthetic structure-notexture-like stream, ORB 1000/1.2/8 FAST 20/7 + LSD/merge/LBD 200 lines "
                                    "+ LIL pairing + RGB-D line glue (isLineGood, crossings, planes) extract, SearchByProjection(cur,last) + LSDmatcher::match, "
                                    "frames resident in HBM") if LINES else
                                   ("configs[1]: 640x480 synthetic RGB-D stream (desk-like), ORB 1000/1.2/8 FAST 20/7 "
                                    "extract + SearchByProjection(cur,last) match, frames resident in HBM"),
                       "frames_per_step_per_gpu": B, "mean_keypoints": round(mean_kp, 1), "mean_matches": round(mean_matches, 1),
                       "streams": nstreams,
                       "multi_gpu": "independent stream per rank, RCCL all-gather of result records" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": round(stages[dom]["ms_per_launch"], 4)},
            "pipeline_roofline": {"bytes_per_frame": per_frame,
                                  "achieved_GBs": round(fps / world * per_frame / 1e9, 2),
                                  "frac_of_8TBs": round(fps / world * per_frame / 1e9 / HBM_PEAK_GBS, 5)},
            "stages_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in stages.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_h, lines=LINES, depth_frames=depth_h if LINES else None,
                                               cam=cam if LINES else None)  # stops after ~25 s of CPU work
        if LINES:
            out["config"]["mean_line_matches"] = round(float(lnm.float().mean().item()), 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def P_seed(rank):
    import synth_frames as sf
    return sf.SEED + 1000 * rank


if __name__ == "__main__":
    main()
