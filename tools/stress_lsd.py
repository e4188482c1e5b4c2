"""Determinism / parity stress of the line extractor on the GPU box: a batch of distinct synthetic frames is extracted
several times; every repetition must give byte-identical keylines and descriptors, and a sample of frames must equal the
CPU oracle.  usage: python tools/stress_lsd.py [frames] [repeats] [oracle_samples]"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import psl_slam_amd as P
    import synth_frames as sf
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    W, H = 640, 480
    frames = []
    for style, seed in (("struct", 5), ("desk", 7), ("sticks", 13), ("struct", 9), ("desk", 11), ("sticks", 17)):   # 'sticks': the dense scene of the headline bench
        sc = sf.Scene(W, H, style, seed)
        frames += [sc.gray(t) for t in range(24)]
    frames = np.stack(frames, 0)
    batch = np.ascontiguousarray(np.concatenate([frames] * ((n + len(frames) - 1) // len(frames)), 0)[:n])
    dev = torch.device("cuda:0")
    d = torch.from_numpy(batch).to(dev)
    ctx = P.default_context()
    le = P.LINEextractor(1, 1.2, 200, 0.0, ctx=ctx, max_batch=n)
    sums = []
    for r in range(reps):
        le.extract_batch_device(d.data_ptr(), n, W, H, W, W * H)
        ctx.synchronize()
        crc = 0
        per = []
        for f in range(n):
            kl, desc, eq = le.fetch(f)[:3]
            c = zlib.crc32(kl.tobytes()) ^ zlib.crc32(desc.tobytes())
            per.append(c)
        sums.append(per)
        print("rep", r, "crc", hex(zlib.crc32(np.array(per, np.uint32).tobytes())), flush=True)
    bad = [f for f in range(n) if any(sums[r][f] != sums[0][f] for r in range(1, reps))]
    # identical input frames must give identical results too
    for f in range(len(frames), n):
        if sums[0][f] != sums[0][f % len(frames)]:
            bad.append(f)
    print("nondeterministic frames:", bad[:20], len(bad))
    import oracle_lib
    mism = 0
    for f in np.linspace(0, min(n, len(frames)) - 1, ns).astype(int):
        kl, desc = le.fetch(int(f))[:2]
        rk, rd = oracle_lib.line_extract(batch[f], 200)[:2]
        same = len(kl) == len(rk) and np.array_equal(desc, rd) and np.allclose(kl["startPointX"], rk["startPointX"], atol=0.01)
        print("frame", f, "lines", len(kl), "oracle", len(rk), "ok" if same else "MISMATCH", flush=True)
        if not same and len(kl) == len(rk):
            dd = np.nonzero((desc != rd).any(1))[0]
            print("   desc rows differing:", dd[:10], "max |dx|", float(np.abs(kl["startPointX"] - rk["startPointX"]).max()))
        mism += not same
    sys.exit(1 if bad or mism else 0)


if __name__ == "__main__":
    main()
