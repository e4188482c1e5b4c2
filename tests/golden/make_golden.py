"""Generates the committed golden vectors from the CPU oracle (oracle/libpsl_oracle.so).

The reference ships no fixtures and cannot be built here, so these vectors
certify HIP == oracle and guard the oracle against regressions; they do NOT certify
oracle == OpenCV (parity unpinned).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib  # noqa: E402
import synth_frames as sf  # noqa: E402

CASES = [
    # name, w, h, style, seed, t, nfeatures, nlevels
    ("orb_640x480_desk", 640, 480, "desk", sf.SEED, 0, 1000, 8),
    ("orb_640x480_struct", 640, 480, "struct", sf.SEED + 1, 2, 1000, 8),
    ("orb_320x240_desk", 320, 240, "desk", sf.SEED + 2, 1, 500, 4),
]


def main():
    for name, w, h, style, seed, t, nf, nl in CASES:
        img = sf.Scene(w, h, style, seed).gray(t)
        orc = oracle_lib.OracleORB(nf, 1.2, nl, 20, 7)
        kps, desc = orc(img)
        cands = [len(orc.candidates(l)) for l in range(nl)]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, kps=kps, desc=desc,
                            cfg=np.array([nf, nl, 20, 7], np.int32), ncand=np.array(cands, np.int32))
        print(name, img.shape, len(kps), cands)
    # matching golden: frame pair of the desk scene, queries = frame 0 keypoints at their own pixels
    sc = sf.Scene(640, 480, "desk", sf.SEED)
    orc = oracle_lib.OracleORB(1000, 1.2, 8, 20, 7)
    k0, d0 = orc(sc.gray(0))
    k1, d1 = orc(sc.gray(1))
    scale = sf.orb_scale_factors()
    q = np.zeros(len(k0), oracle_lib.PROJQUERY_DTYPE)
    q["u"], q["v"] = k0["x"], k0["y"]
    q["radius"] = np.float32(15.0) * scale[k0["octave"]]
    q["min_level"], q["max_level"] = k0["octave"] - 1, k0["octave"] + 1
    q["angle"], q["blocks"] = k0["angle"], 1
    bounds = (0.0, 0.0, 640.0, 480.0)
    nm, match, assigned = oracle_lib.search_by_projection_last(k1, d1, None, bounds, q, d0, None, True)
    idx, dist = oracle_lib.hamming_knn2(d0[:200], d1[:200])
    np.savez_compressed(os.path.join(HERE, "match_640x480_desk.npz"), k1=k1, d1=d1, queries=q, qdesc=d0,
                        nmatches=np.int32(nm), match=match, assigned=assigned, knn_idx=idx, knn_dist=dist)
    print("match", nm)


def lines():
    img = sf.Scene(640, 480, "struct", sf.SEED + 1).gray(2)
    kls, desc, eq = oracle_lib.line_extract(img, 200)
    seg = oracle_lib.lsd_detect(img)            # LSD_REFINE_ADV, the default
    oracle_lib.set_lsd_refine(1)
    seg_std = oracle_lib.lsd_detect(img)        # LSD_REFINE_STD
    oracle_lib.set_lsd_refine(2)
    L = np.stack([kls[n] for n in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32)
    fans = oracle_lib.lil_pair(L, 20.0, np.float32(np.pi / 4), 640, 480)
    np.savez_compressed(os.path.join(HERE, "line_640x480_struct.npz"), image=img, segments=seg, segments_std=seg_std, kls=kls, desc=desc, eq=eq, fans=fans)
    print("lines", len(seg), len(kls), len(fans))


def glue():
    import glue_scene
    kls, fans, depth, cam, _ = glue_scene.scene(seed=3)
    r = oracle_lib.frame_glue(kls, fans, depth, cam, seed=1)
    # the depth image is re-rendered by glue_scene (1.2 MB as f32); its CRC pins it
    import zlib
    np.savez_compressed(os.path.join(HERE, "glue_640x480_corner.npz"), kls=kls, fans=fans, depth_crc=np.uint32(zlib.crc32(depth.tobytes())),
                        seed=np.uint32(1), **{"out_" + k: v for k, v in r.items()})
    print("glue", int((np.abs(r["lines3d"]).sum(1) > 0).sum()), len(r["pair"]), len(r["planes"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["lines", "main", "glue"]
    if "lines" in which:
        lines()
    if "main" in which:
        main()
    if "glue" in which:
        glue()
