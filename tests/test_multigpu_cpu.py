"""N > 1 path on CPU: world_size-2 gloo processes shard 8 streams (stream s -> rank s mod world), compute REAL results for them
with the CPU oracle, pack them into the per-frame result records of the C ABI (include/pslfe.h: pslfe_record_layout; the numpy
packer of psl-slam_amd/multigpu.py mirrors k_record_pack) and all-gather the records through the helper bench.py falls back to
without RCCL.  The gathered records must equal, byte for byte, the records a single process makes for all 8 streams."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KP_CAP, KL_CAP, FAN_CAP, PLANE_CAP = 1100, 200, 256, 64
N_STREAMS = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream_records(layout, stream):
    """Two frames of stream `stream` through the oracle's Frame + TrackWithMotionModel sequence -> the record of frame 1
    (the one that has matches against its predecessor)."""
    for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dropin_harness as D
    gray, depth = D.synth_stream(320, 240, 2, "struct" if stream % 2 else "desk", seed=500 + stream)
    r = D.oracle_sequence(gray, depth, 500, 200)[1]
    return layout.pack(1, r["mvKeys"], r["mDescriptors"], r["match"], int((r["match"] >= 0).sum()), r["mvKeylinesUn"], r["mLdesc"],
                       r["mvKeyLineFunctions"], r["lm12"], int((r["lm12"] >= 0).sum()), r["fans"], r["planes"], r["lineNo"])


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import psl_slam_amd  # noqa: F401
    mg = import_module("psl_slam_amd.multigpu")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        layout = mg.RecordLayout(KP_CAP, KL_CAP, FAN_CAP, PLANE_CAP)
        streams = mg.streams_for_rank(N_STREAMS, rank, world)
        mine = torch.from_numpy(np.stack([_stream_records(layout, s) for s in streams], 0))
        gat = mg.ResultGather([mine], world, torch.device("cpu"))
        outs = []
        for step in range(3):  # exercises both staging slots
            k = gat.submit([mine])
            outs.append(gat.result(k)[0].clone())
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        q.put((rank, "ok", streams, outs[0].numpy()))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e), None, None))
    finally:
        dist.destroy_process_group()


def test_streams_partition():
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import psl_slam_amd  # noqa: F401
    mg = import_module("psl_slam_amd.multigpu")
    for world in (1, 2, 4, 8):
        parts = [mg.streams_for_rank(8, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(8))
        assert all(len(p) == 8 // world for p in parts)


def test_record_layout_and_numpy_pack_roundtrip():
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import psl_slam_amd as P
    mg = import_module("psl_slam_amd.multigpu")
    L = mg.RecordLayout(KP_CAP, KL_CAP, FAN_CAP, PLANE_CAP)
    offs = [32, L.off_kps, L.off_desc, L.off_match, L.off_kls, L.off_ldesc, L.off_lineEq, L.off_lmatch, L.off_fans, L.off_planes,
            L.off_plane_lines, L.bytes]
    assert offs[1] == 32 and all(b > a for a, b in zip(offs[1:], offs[2:])) and all(o % 16 == 0 for o in offs[1:]) and L.bytes % 256 == 0
    assert L.off_desc - L.off_kps >= KP_CAP * 28 and L.off_ldesc - L.off_kls >= KL_CAP * 68 and L.bytes - L.off_plane_lines >= PLANE_CAP * 8
    rec = _stream_records(L, 3)
    u = L.unpack(rec)
    assert u["frame"] == 1 and u["flags"] == 0 and 300 < u["n_kp"] <= 520 and u["n_match"] > 100 and u["n_kl"] >= 1
    assert u["kps"].dtype == P.KEYPOINT_DTYPE and len(u["kps"]) == u["n_kp"] and u["desc"].shape == (u["n_kp"], 32)
    assert (u["match"][:u["n_kp"] + 8] >= -1).all() and int((u["match"] >= 0).sum()) == u["n_match"]
    # truncation is flagged, not silent
    small = mg.RecordLayout(100, 4, 1, 1)
    r2 = small.unpack(small.pack(0, u["kps"], u["desc"], u["match"], u["n_match"], u["kls"], u["ldesc"], u["lineEq"], u["lmatch"], u["n_lmatch"],
                                 u["fans"], u["planes"], u["plane_lines"]))
    assert r2["n_kp"] == u["n_kp"] and len(r2["kps"]) == 100 and (r2["flags"] & 1) and r2["kps"].tobytes() == u["kps"][:100].tobytes()
    with pytest.raises(P.PslfeError):
        mg.RecordLayout(-1, 1, 1, 1)


def test_result_gather_gloo_world2_real_records():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[1] for r in res) == ["ok", "ok"], res
    assert sorted(r[2] for r in res) == [[0, 2, 4, 6], [1, 3, 5, 7]]
    # every rank holds the same gathered array: rank 0's streams first, then rank 1's (rank-major = what ncclAllGather gives)
    assert np.array_equal(res[0][3], res[1][3])
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import psl_slam_amd  # noqa: F401
    mg = import_module("psl_slam_amd.multigpu")
    layout = mg.RecordLayout(KP_CAP, KL_CAP, FAN_CAP, PLANE_CAP)
    order = [0, 2, 4, 6, 1, 3, 5, 7]
    single = np.stack([_stream_records(layout, s) for s in order], 0)   # the 1-rank result
    assert res[0][3].shape == single.shape and np.array_equal(res[0][3], single)
    u = layout.unpack(res[0][3][5])
    assert u["n_kp"] > 300 and u["n_match"] > 100


# ---- bench.py's own launcher (python bench.py --gpus N without torchrun) and the torchrun launch, rehearsed with gloo -------------
def _last_json(text):
    import json
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text)


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSLFE_BENCH_LAUNCHER")}


def test_bench_spawns_its_own_ranks_gloo_rehearsal():
    """python bench.py --gpus 2 with no launcher: two rank processes, rendezvous on 127.0.0.1, the collective agreement, the gather to
    rank 0, ONE JSON line relayed from rank 0, exit code 0."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True,
                       env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["launcher_selftest"] and j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1] and j["gather_ok"] and j["rccl_gather_agreed"]
    assert j["launcher"] == "self"
    assert sum(1 for line in r.stdout.splitlines() if line.startswith("{")) == 1


def test_bench_gather_choice_is_collective():
    """One rank that cannot create its communicator makes EVERY rank take the fallback (all_reduce MIN), and the run still completes."""
    import subprocess
    env = dict(_clean_env(), PSLFE_SELFTEST_FAIL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["rccl_gather_agreed"] is False and j["gather_ok"] and j["ranks_seen"] == [0, 1]


def test_bench_under_torchrun_gloo_rehearsal():
    """The launch the driver documents: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2."""
    import subprocess
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["ranks_seen"] == [0, 1] and j["gather_ok"] and j["launcher"] == "external"


def test_bench_multi_gpu_without_devices_fails_with_the_librarys_message():
    """No GPU: both rank processes reach pslfe_ctx_create's PSLFE_E_NODEVICE and the launcher exits non-zero with that message
    (not an AssertionError, not a hang)."""
    import subprocess
    import ctypes as C
    import psl_slam_amd as P
    h = C.c_void_p()
    if P.lib().pslfe_ctx_create(C.c_int(0), C.byref(h)) == 0:
        P.lib().pslfe_ctx_destroy(h)
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("this library has no CPU fallback") == 2 and "rank 0" in r.stderr and "rank 1" in r.stderr
    assert "AssertionError" not in r.stderr and not r.stdout.strip()
