"""Row (e) on one GPU: the batch step's results packed into per-frame records by pslfe_record_pack_device and moved by
pslfe_gather_all - a real ncclAllGather from RCCL with a communicator of world size 1 (a one-GPU box has no peers; the N > 1
exchange is the same call).  Records are compared with the per-frame fetch entry points, with the numpy packer the CPU (gloo)
test uses, and, through those, with the CPU oracle."""
import numpy as np
import pytest

import synth_frames as sf

pytestmark = pytest.mark.gpu


def test_record_pack_and_rccl_gather_world1():
    import torch
    from importlib import import_module
    import psl_slam_amd as P
    import batch_pipeline as BP
    import oracle_lib
    mg = import_module("psl_slam_amd.multigpu")
    B, w, h = 6, 640, 480
    sc = sf.Scene(w, h, "struct", 21)
    gray = np.stack([sc.gray(t) for t in range(B)], 0)
    depth = np.stack([oracle_lib.depth_to_float(sc.depth_u16(t), np.float32(1.0 / 5000.0)) * np.float32(1 + 0.02 * t) for t in range(B)], 0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    pipe = BP.BatchPipeline(P, torch, dev, stream, 0, B, w, h, lines=True)
    d_gray, d_depth = torch.from_numpy(gray).to(dev), torch.from_numpy(depth).to(dev)
    layout = None
    gat = None
    for rep in range(3):   # both record buffers, and an exchange overlapping the next step
        pipe.step(d_gray.data_ptr(), d_depth.data_ptr())
        if gat is None:
            layout = pipe.record_layout(mg)
            gat = mg.RecordGather(pipe.ctx, layout, B, 0, 1, dev, lambda uid: uid)
        k = gat.submit(pipe.record_sources(mg))
    rec = gat.result(k).cpu().numpy()
    torch.cuda.synchronize(dev)
    assert rec.shape == (1, B, layout.bytes)
    cache = {}
    for f in range(B):
        u = layout.unpack(rec[0, f])
        r = pipe.fetch_frame(f)
        assert u["frame"] == f and u["flags"] == 0
        assert u["n_kp"] == len(r["kps"]) and u["n_match"] == r["nmatches"] and u["n_kl"] == len(r["kls"]) and u["n_lmatch"] == r["lnm"]
        assert u["n_fan"] == len(r["fans"]) and u["n_planes"] == len(r["planes"])
        assert u["kps"].tobytes() == r["kps"].tobytes() and np.array_equal(u["desc"], r["desc"]) and np.array_equal(u["match"], r["match"][:layout.caps.kp_cap])
        assert u["kls"].tobytes() == r["kls"].tobytes() and np.array_equal(u["ldesc"], r["ldesc"]) and u["lineEq"].tobytes() == r["lineEq"].tobytes()
        assert np.array_equal(u["lmatch"], r["lmatch"][:layout.caps.kl_cap]) and u["fans"].tobytes() == r["fans"].tobytes()
        assert u["planes"].tobytes() == r["planes"].tobytes() and np.array_equal(u["plane_lines"], r["plane_lines"])
        # the numpy packer (the CPU test's stand-in for k_record_pack) gives the same bytes
        twin = layout.pack(f, r["kps"], r["desc"], r["match"], r["nmatches"], r["kls"], r["ldesc"], r["lineEq"], r["lmatch"], r["lnm"], r["fans"],
                           r["planes"], r["plane_lines"])
        assert np.array_equal(twin, rec[0, f]), f"record {f}: device pack differs from the numpy pack"
        if f in (0, 3):  # and the content is the oracle's
            ref = BP.oracle_frame(((f - 1) % B, gray[(f - 1) % B]), (f, gray[f]), depth[f], f, w, h, True, pipe.cam, cache=cache)
            BP.compare_frame(r, ref, f"frame {f}: ")
    gat.close()
