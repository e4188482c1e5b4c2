"""Batched many-frames mode across the GPUs of one node (BASELINE configs[3]) - host-side mirror of the
record / gather part of the C ABI (include/pslfe.h: pslfe_record_*, pslfe_gather_*).

Frames and whole streams are independent, so they are sharded with no data-path collective: stream s runs on rank
s mod world.  The one exchange step is the RESULT GATHER: every rank packs the results of its batch into fixed-size per-frame
records (counts, mvKeys, mDescriptors, point matches, mvKeylinesUn, mLdesc, mvKeyLineFunctions, line matches, fans, mvPlanes,
mvPlaneLineNo; ~85 KB at 1000 points / 200 lines) and ONE gather per batch moves them:
  * on the GPUs: pslfe_record_pack_device + pslfe_gather_to_root = one group of ncclSend / ncclRecv from RCCL over xGMI towards the
    consuming rank (only that rank holds world x batch records; root=None: pslfe_gather_all = ncclAllGather to every rank), issued
    from the C ABI on the gather's own stream (RecordGather below; torch.distributed only carries the 128-byte ncclUniqueId, the
    agreement on which gather is used, and the barriers);
  * in the CPU tests: the same records, packed with numpy by the same layout (pslfe_record_layout is host arithmetic), moved by
    torch.distributed's gloo backend (ResultGather).
"""
import ctypes as C

import numpy as np

HEADER_FIELDS = ("n_kp", "n_match", "n_kl", "n_lmatch", "n_fan", "n_planes", "flags", "frame")


def streams_for_rank(n_streams, rank, world):
    """Stream s -> rank s mod world."""
    return [s for s in range(n_streams) if s % world == rank]


class RecordCaps(C.Structure):
    _fields_ = [("kp_cap", C.c_int32), ("kl_cap", C.c_int32), ("fan_cap", C.c_int32), ("plane_cap", C.c_int32)]


class _Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("bytes", "off_kps", "off_desc", "off_match", "off_kls", "off_ldesc", "off_lineEq", "off_lmatch",
                                         "off_fans", "off_planes", "off_plane_lines")]


class RecordSources(C.Structure):
    _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("d_kp_counts", C.c_void_p), ("kp_stride", C.c_int32),
                ("d_match", C.c_void_p), ("d_nmatches", C.c_void_p), ("match_stride", C.c_int32),
                ("d_kls", C.c_void_p), ("d_ldesc", C.c_void_p), ("d_lineEq", C.c_void_p), ("d_kl_counts", C.c_void_p), ("kl_stride", C.c_int32),
                ("d_lmatch", C.c_void_p), ("d_nlmatches", C.c_void_p), ("lmatch_stride", C.c_int32),
                ("d_fans", C.c_void_p), ("d_fan_counts", C.c_void_p), ("fan_stride", C.c_int32),
                ("d_planes", C.c_void_p), ("d_plane_lines", C.c_void_p), ("d_plane_counts", C.c_void_p), ("plane_stride", C.c_int32)]


class RecordLayout:
    """pslfe_record_layout: offsets of the sections of one per-frame record (host arithmetic, works without a GPU)."""

    def __init__(self, kp_cap, kl_cap, fan_cap, plane_cap):
        import psl_slam_amd as P
        self.caps = RecordCaps(kp_cap, kl_cap, fan_cap, plane_cap)
        L = _Layout()
        P._check(P.lib().pslfe_record_layout(C.byref(self.caps), C.byref(L)), "pslfe_record_layout")
        for n, _ in _Layout._fields_:
            setattr(self, n, getattr(L, n))

    def sections(self):
        """name -> (offset, rows, numpy dtype per row)"""
        import psl_slam_amd as P
        c = self.caps
        return {"kps": (self.off_kps, c.kp_cap, P.KEYPOINT_DTYPE), "desc": (self.off_desc, c.kp_cap, np.dtype((np.uint8, 32))),
                "match": (self.off_match, c.kp_cap, np.dtype(np.int32)), "kls": (self.off_kls, c.kl_cap, P.KEYLINE_DTYPE),
                "ldesc": (self.off_ldesc, c.kl_cap, np.dtype((np.uint8, 32))), "lineEq": (self.off_lineEq, c.kl_cap, np.dtype((np.float64, 3))),
                "lmatch": (self.off_lmatch, c.kl_cap, np.dtype(np.int32)), "fans": (self.off_fans, c.fan_cap, np.dtype((np.float32, 4))),
                "planes": (self.off_planes, c.plane_cap, np.dtype((np.float32, 4))),
                "plane_lines": (self.off_plane_lines, c.plane_cap, np.dtype((np.int32, 2)))}

    def pack(self, frame, kps, desc, match, n_match, kls, ldesc, lineEq, lmatch, n_lmatch, fans, planes, plane_lines):
        """numpy twin of k_record_pack (csrc/pslfe_gather.hip): one record as a uint8 array."""
        rec = np.zeros(self.bytes, np.uint8)
        c = self.caps
        n = dict(kps=len(kps), kls=len(kls), fans=len(fans), planes=len(planes))
        flags = (n["kps"] > c.kp_cap) | ((n["kls"] > c.kl_cap) << 1) | ((n["fans"] > c.fan_cap) << 2) | ((n["planes"] > c.plane_cap) << 3)
        rec[:32] = np.array([n["kps"], n_match, n["kls"], n_lmatch, n["fans"], n["planes"], flags, frame], np.int32).view(np.uint8)
        sec = self.sections()
        for name in ("match", "lmatch"):   # "no match" = -1 in every row the query list does not reach (the device buffers are pre-filled)
            off, cap, dt = sec[name]
            rec[off:off + cap * 4] = 0xFF

        def put(name, arr, rows):
            off, cap, dt = sec[name]
            rows = min(rows, cap, len(arr))
            if rows > 0:
                b = np.ascontiguousarray(arr[:rows]).view(np.uint8).reshape(-1)
                rec[off:off + len(b)] = b
        put("kps", kps, n["kps"]); put("desc", desc, n["kps"]); put("match", match, c.kp_cap)   # match / lmatch: indexed by the query, fixed size
        put("kls", kls, n["kls"]); put("ldesc", ldesc, n["kls"]); put("lineEq", lineEq, n["kls"]); put("lmatch", lmatch, c.kl_cap)
        put("fans", fans, n["fans"]); put("planes", planes, n["planes"]); put("plane_lines", plane_lines, n["planes"])
        return rec

    def unpack(self, rec):
        """One record (uint8 array of `bytes`) -> dict: the header fields and every section cut to its count."""
        rec = np.ascontiguousarray(rec, np.uint8).reshape(-1)
        h = rec[:32].view(np.int32)
        out = dict(zip(HEADER_FIELDS, (int(v) for v in h)))
        c = self.caps
        counts = dict(kps=min(out["n_kp"], c.kp_cap), desc=min(out["n_kp"], c.kp_cap), match=c.kp_cap,
                      kls=min(out["n_kl"], c.kl_cap), ldesc=min(out["n_kl"], c.kl_cap), lineEq=min(out["n_kl"], c.kl_cap), lmatch=c.kl_cap,
                      fans=min(out["n_fan"], c.fan_cap), planes=min(out["n_planes"], c.plane_cap), plane_lines=min(out["n_planes"], c.plane_cap))
        for name, (off, cap, dt) in self.sections().items():
            out[name] = rec[off:off + counts[name] * dt.itemsize].view(dt.base if dt.subdtype else dt).reshape((counts[name],) + (dt.shape if dt.subdtype else ()))
        return out


def agree_all_ranks(ok, world, device=None):
    """True iff `ok` holds on EVERY rank (all_reduce MIN over torch.distributed): decisions that select which collective the ranks
    issue next must be taken together, or the ranks end up in different collectives and hang."""
    if world == 1:
        return bool(ok)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class RecordGather:
    """GPU path: pslfe_record_pack_device + pslfe_gather_to_root / pslfe_gather_all (RCCL).  Double-buffered: submit() packs this
    batch's records and starts the exchange on the gather's own stream; the previous exchange is waited for only when its buffers
    are reused.  root = r: records go to rank r only (the others hold no receive buffer); root = None: all-gather to every rank."""

    def __init__(self, ctx, layout, nframes, rank, world, device, broadcast_id, root=0, agree=None, recv_slots=2):
        """broadcast_id(uid, ok) -> (128-byte uint8 numpy array, ok) as rank 0 holds them, on every rank; agree(ok) -> True iff ok on
        every rank (default: agree_all_ranks); recv_slots = 1: one receive buffer (exchanges run in order on the gather's stream, so
        they never overlap each other; the consumer must be done with result(k) before the next submit).  COLLECTIVE and failure-safe: a rank that cannot load RCCL or create its communicator
        does not raise before every rank has taken part in the same broadcast / agreement, so all ranks raise together (and the
        caller can fall back to TorchRecordGather on all of them) instead of one rank leaving the others in a collective."""
        import torch
        import psl_slam_amd as P
        self.P, self.ctx, self.layout, self.nframes, self.world, self.rank, self.root = P, ctx, layout, nframes, world, rank, root
        agree = agree or (lambda ok: agree_all_ranks(ok, world, device))
        uid = np.zeros(128, np.uint8)
        ok, err = True, ""
        if rank == 0:
            rc = P.lib().pslfe_gather_unique_id(P._ptr(uid))
            if rc != 0:
                ok, err = False, f"pslfe_gather_unique_id: {P.lib().pslfe_last_error().decode()}"
        uid, ok0 = broadcast_id(uid, ok)
        self._h = C.c_void_p()
        if ok0:
            uid = np.ascontiguousarray(uid, np.uint8)
            rc = P.lib().pslfe_gather_create(ctx._h, C.c_int(rank), C.c_int(world), P._ptr(uid), C.byref(self._h))
            if rc != 0:
                ok, err = False, f"pslfe_gather_create: {P.lib().pslfe_last_error().decode()}"
        else:
            ok, err = False, err or "rank 0 could not obtain an ncclUniqueId"
        if not agree(ok):
            self.close()
            raise RuntimeError(err or "pslfe_gather could not be created on another rank")
        self.send = [torch.empty((nframes, layout.bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [None, None]
        if root is None or rank == root:
            r0 = torch.empty((world, nframes, layout.bytes), dtype=torch.uint8, device=device)
            self.recv = [r0, r0 if recv_slots == 1 else torch.empty_like(r0)]
        self.k = 0
        self.bytes_per_step = nframes * layout.bytes

    def ranks_seen(self):
        """The ranks of the communicator as the exchange itself reports them: every rank contributes its rank number through
        pslfe_gather_all (4 bytes); returns the list every rank received."""
        import torch
        dev = self.send[0].device
        s = torch.tensor([self.rank], dtype=torch.int32, device=dev)
        r = torch.full((self.world,), -1, dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        self.P._check(self.P.lib().pslfe_gather_all(self._h, C.c_void_p(s.data_ptr()), C.c_size_t(4), C.c_void_p(r.data_ptr())), "pslfe_gather_all")
        self.wait()
        return [int(v) for v in r.cpu().numpy()]

    def submit(self, sources):
        P, k = self.P, self.k
        # slot k was used two submits ago: its exchange is older than the one still allowed in flight, which this waits for
        P._check(P.lib().pslfe_gather_wait(self._h, C.c_int(0)), "pslfe_gather_wait")
        P._check(P.lib().pslfe_record_pack_device(self.ctx._h, C.byref(self.layout.caps), C.byref(sources), C.c_int(self.nframes),
                                                  C.c_void_p(self.send[k].data_ptr())), "pslfe_record_pack_device")
        recv = C.c_void_p(self.recv[k].data_ptr()) if self.recv[k] is not None else C.c_void_p()
        if self.root is None:
            P._check(P.lib().pslfe_gather_all(self._h, C.c_void_p(self.send[k].data_ptr()), C.c_size_t(self.bytes_per_step), recv), "pslfe_gather_all")
        else:
            P._check(P.lib().pslfe_gather_to_root(self._h, C.c_void_p(self.send[k].data_ptr()), C.c_size_t(self.bytes_per_step),
                                                  C.c_int(self.root), recv), "pslfe_gather_to_root")
        self.k ^= 1
        return k

    def wait(self):
        self.P._check(self.P.lib().pslfe_gather_wait(self._h, C.c_int(1)), "pslfe_gather_wait")

    def result(self, k):
        self.wait()
        return self.recv[k]

    def close(self):
        if self._h:
            self.P.lib().pslfe_gather_destroy(self._h)
            self._h = C.c_void_p()


class TorchRecordGather:
    """Same interface as RecordGather, the exchange through torch.distributed (backend nccl = RCCL as well): the fallback when the
    C ABI cannot load librccl on a node.  The pack kernel is the C ABI's either way.  root = r: dist.gather to rank r (receive
    buffers only there); root = None: all_gather_into_tensor."""

    def __init__(self, ctx, layout, nframes, world, device, rank=0, root=0, recv_slots=2):
        import torch
        import psl_slam_amd as P
        self.P, self.ctx, self.layout, self.nframes, self.world, self.rank, self.root = P, ctx, layout, nframes, world, rank, root
        self.send = [torch.empty((nframes, layout.bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [None, None]
        if root is None or rank == root:
            r0 = torch.empty((world, nframes, layout.bytes), dtype=torch.uint8, device=device)
            self.recv = [r0, r0 if recv_slots == 1 else torch.empty_like(r0)]
        self.work = [None, None]
        self.k = 0

    def ranks_seen(self):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return [self.rank]
        dev = self.send[0].device
        s = torch.tensor([self.rank], dtype=torch.int32, device=dev)
        r = torch.full((self.world,), -1, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(r, s)
        return [int(v) for v in r.cpu().numpy()]

    def submit(self, sources):
        import torch.distributed as dist
        P, k = self.P, self.k
        if self.work[k] is not None:
            self.work[k].wait()
        P._check(P.lib().pslfe_record_pack_device(self.ctx._h, C.byref(self.layout.caps), C.byref(sources), C.c_int(self.nframes),
                                                  C.c_void_p(self.send[k].data_ptr())), "pslfe_record_pack_device")
        if self.world == 1:
            self.work[k] = None
            self.recv[k][0].copy_(self.send[k], non_blocking=True)
        elif self.root is None:
            self.work[k] = dist.all_gather_into_tensor(self.recv[k].view(-1), self.send[k].view(-1), async_op=True)
        else:
            self.work[k] = dist.gather(self.send[k], list(self.recv[k].unbind(0)) if self.rank == self.root else None, dst=self.root, async_op=True)
        self.k ^= 1
        return k

    def wait(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def result(self, k):
        self.wait()
        return self.recv[k]

    def close(self):
        self.wait()


class ResultGather:
    """torch.distributed path (gloo in the CPU tests): double-buffered all-gather of one step's tensors, e.g. the [F][bytes] record
    tensor; overlap with the next step's compute."""

    def __init__(self, templates, world, device, group=None):
        import torch
        self.world, self.group = world, group
        self.stage = [[torch.empty_like(t, device=device) for t in templates] for _ in range(2)]
        self.out = [[torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=device) for t in templates]
                    for _ in range(2)]
        self.works = [[], []]
        self.k = 0

    def submit(self, tensors):
        """Copy this step's results to a staging slot (so the producers can be overwritten) and start
        the all-gather.  Returns the slot index."""
        import torch.distributed as dist
        k = self.k
        self.wait(k)
        for s, t in zip(self.stage[k], tensors):
            s.copy_(t, non_blocking=True)
        self.works[k] = [dist.all_gather_into_tensor(o, s, group=self.group, async_op=True)
                         for o, s in zip(self.out[k], self.stage[k])]
        self.k ^= 1
        return k

    def wait(self, k=None):
        for kk in ([0, 1] if k is None else [k]):
            for w in self.works[kk]:
                w.wait()
            self.works[kk] = []

    def result(self, k):
        self.wait(k)
        return self.out[k]
