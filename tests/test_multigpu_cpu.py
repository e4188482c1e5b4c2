"""N > 1 path on CPU: world_size-2 gloo processes shard streams and all-gather result records
through the same helper bench.py uses on RCCL (psl-slam_amd/multigpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
        streams = mg.streams_for_rank(8, rank, world)
        F, cap = len(streams), 16
        g = torch.Generator().manual_seed(1234)
        allc = torch.randint(0, cap, (8,), generator=g, dtype=torch.int32)
        allk = torch.rand((8, cap, 7), generator=g)
        alld = torch.randint(0, 256, (8, cap, 32), generator=g, dtype=torch.uint8)
        mine = [allc[streams], allk[streams], alld[streams]]
        gat = mg.ResultGather(mine, world, torch.device("cpu"))
        for step in range(3):  # exercises both staging slots
            k = gat.submit([t + step if t.dtype != torch.uint8 else t for t in mine])
            out = gat.result(k)
            order = [s for r in range(world) for s in mg.streams_for_rank(8, r, world)]
            assert torch.equal(out[0], allc[order] + step)
            assert torch.equal(out[1], allk[order] + step)
            assert torch.equal(out[2], alld[order])
        q.put((rank, "ok", streams))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e), None))
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


def test_result_gather_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[1] for r in res) == ["ok", "ok"], res
    assert sorted(r[2] for r in res) == [[0, 2, 4, 6], [1, 3, 5, 7]]
