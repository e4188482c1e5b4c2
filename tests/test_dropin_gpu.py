"""The compiled C++ consumer of psl-slam_amd/host/pslfe.hpp (tools/dropin/dropin_main.cpp): built with g++, run as a child
process on the GPU, one frame at a time through the call sequence of Frame::Frame (src/Frame.cc:133-208) and
Tracking::TrackWithMotionModel (src/Tracking.cc:1164-1214); every array it produces is compared bit for bit with the same
sequence on the CPU oracle.  640x480 / 1000 ORB / 200 lines = BASELINE configs[2]'s shape, 1280x960 / 2000 / 200 = configs[4]."""
import os

import numpy as np
import pytest

import dropin_harness as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,nfeatures,nlines,style,nframes", [(640, 480, 1000, 200, "struct", 5), (640, 480, 1000, 200, "desk", 3),
                                                                (1280, 960, 2000, 200, "struct", 3)],
                         ids=["configs2-struct", "configs2-desk", "configs4-1280x960"])
def test_cpp_consumer_frame_and_tracking_sequence_equals_oracle(tmp_path, w, h, nfeatures, nlines, style, nframes):
    gray, depth = D.synth_stream(w, h, nframes, style, seed=77)
    frames, results = str(tmp_path / "frames.bin"), str(tmp_path / "results.bin")
    D.write_frames(frames, gray, depth)
    D.build(force=True)   # compile + link against libpslfe.so here: the header and the ABI are consumed by a real C++ program
    summary = D.run(frames, nfeatures, nlines, warmup=0, results_path=results)
    assert summary["frames_timed"] == nframes and summary["w"] == w and summary["mean_keypoints"] > 0.9 * nfeatures
    got = D.read_results(results, nframes)
    ref = D.oracle_sequence(gray, depth, nfeatures, nlines)
    nkl = 0
    for t in range(nframes):
        D.compare(got[t], ref[t], f"frame {t}: ")
        nkl += len(ref[t]["mvKeylinesUn"])
    assert nkl > 3 * nframes and any(len(r["match"]) and (r["match"] >= 0).sum() > 100 for r in ref[1:])
    print(f"{w}x{h}: {summary['ms_per_frame']['median']:.2f} ms/frame (median), {summary['mean_keylines']:.1f} keylines, "
          f"{summary['mean_matches']:.0f} point matches, {summary['mean_line_matches']:.1f} line matches")


@pytest.mark.parametrize("w,h,nfeatures,nlines,style,nframes,K", [(640, 480, 1000, 200, "struct", 7, 3), (640, 480, 1000, 200, "desk", 5, 8),
                                                                  (1280, 960, 2000, 200, "struct", 4, 4)],
                         ids=["configs2-struct-K3", "configs2-desk-K8", "configs4-1280x960-K4"])
def test_cpp_consumer_with_lookahead_prefetcher_equals_oracle(tmp_path, w, h, nfeatures, nlines, style, nframes, K):
    """pslfe::FramePrefetcher (host/pslfe.hpp): K frames pushed ahead and extracted by ONE batched launch (a last batch that is
    shorter included), every frame then tracked as before - the same arrays as the one-frame-at-a-time path and as the oracle."""
    gray, depth = D.synth_stream(w, h, nframes, style, seed=78)
    frames, results = str(tmp_path / "frames.bin"), str(tmp_path / "results.bin")
    D.write_frames(frames, gray, depth)
    D.build(force=True)
    summary = D.run(frames, nfeatures, nlines, warmup=0, results_path=results, lookahead=K)
    assert summary["frames_timed"] == nframes and summary["lookahead"] == K
    got = D.read_results(results, nframes)
    ref = D.oracle_sequence(gray, depth, nfeatures, nlines)
    for t in range(nframes):
        D.compare(got[t], ref[t], f"lookahead {K}, frame {t}: ")
    assert any(len(r["match"]) and (r["match"] >= 0).sum() > 100 for r in ref[1:])
