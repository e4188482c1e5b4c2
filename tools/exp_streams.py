"""Experiment: the ORB workload split over N concurrent streams (N pipelines of B/N frames each)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import psl_slam_amd as P
import bench

def run(B, N, steps=20, warmup=3):
    dev = torch.device("cuda", 0)
    frames_h = bench.synth_batch(B, 20250418, style="desk")
    frames_d = torch.from_numpy(frames_h).to(dev)
    b = B // N
    pipes = []
    scale_t = torch.tensor(__import__("synth_frames").orb_scale_factors(), device=dev)
    for i in range(N):
        st = torch.cuda.Stream(dev)
        ctx = P.Context(0, st.cuda_stream)
        orb = P.ORBextractor(1000, 1.2, 8, 20, 7, ctx=ctx, max_batch=b)
        cap = orb.max_keypoints(640, 480)
        grid = P.FrameGrid(cap, b, ctx=ctx)
        with torch.cuda.stream(st):
            q = torch.zeros((b, cap, 8), dtype=torch.float32, device=dev)
            qd = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
            nq = torch.zeros((b,), dtype=torch.int32, device=dev)
            match = torch.full((b, cap), -1, dtype=torch.int32, device=dev)
            nm = torch.zeros((b,), dtype=torch.int32, device=dev)
        pipes.append(dict(st=st, ctx=ctx, orb=orb, grid=grid, cap=cap, q=q, qd=qd, nq=nq, match=match, nm=nm, fr=frames_d[i * b:(i + 1) * b]))
    torch.cuda.synchronize()

    def step():
        for p in pipes:
            with torch.cuda.stream(p["st"]):
                orb, grid, cap, q = p["orb"], p["grid"], p["cap"], p["q"]
                orb.extract_batch_device(p["fr"].data_ptr(), b, 640, 480, 640, 640 * 480)
                k_arr, d_arr, c_arr, _ = P.orb_results_as_arrays(orb, b)
                kps = torch.as_tensor(k_arr, device=dev); desc = torch.as_tensor(d_arr, device=dev); counts = torch.as_tensor(c_arr, device=dev)
                grid.set_from_orb(orb, (0.0, 0.0, 640.0, 480.0))
                prev = torch.roll(kps, 1, 0)
                octv = prev[..., 5].view(torch.int32)
                q[..., 0] = prev[..., 0]; q[..., 1] = prev[..., 1]
                q[..., 2] = 15.0 * scale_t[octv.clamp(0, 7).long()]
                q[..., 3] = 0.0
                qi = q.view(torch.int32)
                qi[..., 4] = octv - 1; qi[..., 5] = octv + 1
                q[..., 6] = prev[..., 3]; qi[..., 7] = 1
                p["qd"].copy_(torch.roll(desc, 1, 0)); p["nq"].copy_(torch.roll(counts, 1, 0))
                P.search_by_projection_last_device(grid, 0, b, q.data_ptr(), p["qd"].data_ptr(), p["nq"].data_ptr(), cap, True,
                                                   p["match"].data_ptr(), p["nm"].data_ptr())
    for _ in range(warmup + 1):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} streams={N}: {B * steps / dt:.0f} fps, {dt / steps * 1e3:.3f} ms/step", flush=True)

if __name__ == "__main__":
    P.build()
    for B, N in [(256, 1), (256, 2), (256, 4), (512, 1), (512, 2), (512, 4), (1024, 1), (1024, 4)]:
        run(B, N)
