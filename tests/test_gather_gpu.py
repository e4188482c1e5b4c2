"""Row (e) on one GPU: the batch step's results packed into per-frame records by pslfe_record_pack_device and moved by
pslfe_gather_to_root (one group of ncclSend / ncclRecv - with a world of one the root's send to itself, the same calls every rank
issues) and pslfe_gather_all (ncclAllGather) - real RCCL with a communicator of world size 1 (a one-GPU box has no peers; the
N > 1 exchange is the same call).  Records are compared with the per-frame fetch entry points, with the numpy packer the CPU (gloo)
test uses, and, through those, with the CPU oracle."""
import numpy as np
import pytest

import synth_frames as sf

try:   # PyTorch - and its copy of the HIP runtime - must be loaded BEFORE libpslfe in a process that uses both on the GPU (tests/conftest.py
    import torch   # does this when the run is selected with -m gpu; at collection time it also holds for `pytest tests/test_gather_gpu.py`)
    torch.cuda.is_available()
except Exception:
    pass

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("root", [0, None])
def test_record_pack_and_rccl_gather_world1(root):
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
            gat = mg.RecordGather(pipe.ctx, layout, B, 0, 1, dev, lambda uid, ok: (uid, ok), root=root, recv_slots=1 if root is not None else 2)
            assert gat.ranks_seen() == [0]
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


def test_record_pack_short_match_rows_read_no_match():
    """A caller whose match buffers hold fewer rows than the record's capacity (match_stride < kp_cap): the rows beyond read -1
    ("no match"), as the header documents and the numpy packer does - not 0 = "matched to keypoint 0"."""
    import torch
    import ctypes as C
    from importlib import import_module
    import psl_slam_amd as P
    mg = import_module("psl_slam_amd.multigpu")
    dev = torch.device("cuda", 0)
    ctx = P.default_context()
    layout = mg.RecordLayout(64, 16, 0, 0)
    F = 3
    match = torch.arange(F * 40, dtype=torch.int32, device=dev).reshape(F, 40)
    lmatch = torch.arange(F * 10, dtype=torch.int32, device=dev).reshape(F, 10) + 1000
    S = mg.RecordSources()
    S.d_match, S.match_stride = match.data_ptr(), 40
    S.d_lmatch, S.lmatch_stride = lmatch.data_ptr(), 10
    rec = torch.empty((F, layout.bytes), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    P._check(P.lib().pslfe_record_pack_device(ctx._h, C.byref(layout.caps), C.byref(S), C.c_int(F), C.c_void_p(rec.data_ptr())), "pslfe_record_pack_device")
    ctx.synchronize()
    r = rec.cpu().numpy()
    for f in range(F):
        u = layout.unpack(r[f])
        assert np.array_equal(u["match"][:40], np.arange(f * 40, f * 40 + 40)) and (u["match"][40:] == -1).all() and len(u["match"]) == 64
        assert np.array_equal(u["lmatch"][:10], np.arange(f * 10, f * 10 + 10) + 1000) and (u["lmatch"][10:] == -1).all()
