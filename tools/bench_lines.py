"""Stage timing of the line front-end on a batch (no torch). Usage: python tools/bench_lines.py [B] [style]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import psl_slam_amd as P  # noqa: E402
import synth_frames as sf  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
style = sys.argv[2] if len(sys.argv) > 2 else "struct"
sc = sf.Scene(640, 480, style, seed=3)
base = np.stack([sc.gray(t) for t in range(8)], 0)
frames = np.ascontiguousarray(np.concatenate([base] * ((B + 7) // 8), 0)[:B])
ctx = P.default_context()
le = P.LINEextractor(1, 1.2, 200, 0.0, ctx=ctx, max_batch=B)
d_ptr, _ = ctx.device_array(frames)
le.extract_batch_device(d_ptr, B, 640, 480, 640, 640 * 480)
le.pair_batch_device()
ctx.synchronize()
ctx.profile_reset()
ctx.profile(True)
t0 = time.perf_counter()
R = 3
for _ in range(R):
    le.extract_batch_device(d_ptr, B, 640, 480, 640, 640 * 480)
    le.pair_batch_device()
ctx.synchronize()
dt = (time.perf_counter() - t0) / R
ctx.profile(False)
print(f"B={B} style={style}: {dt * 1e3:.2f} ms per batch -> {B / dt:.1f} frames/s")
for s in ("line.lsd_scale", "line.lsd_grad", "line.lsd_grow", "line.merge", "line.lbd_pre", "line.lbd", "line.pair"):
    ms, n = ctx.stage_time(s)
    if n:
        print(f"  {s:16s} {ms / n:10.3f} ms/launch")
k, d, e, st = le.fetch(0)
print("frame 0:", len(k), "keylines, status", st)
