"""The HBM-resident many-frames step of the front-end, as bench.py times it and tests/test_gather_gpu.py checks it: harness code
over the C ABI (psl_slam_amd), not part of libpslfe.

One step over a batch of B consecutive frames of a stream (frame f's predecessor is frame f-1, cyclic inside the batch):
  ORB:    pslfe_orb_extract_batch_device -> pslfe_frame_set_from_orb -> queries from the predecessor's keypoints (harness kernel,
          stand-in for Tracking's constant-velocity projection) -> pslfe_orb_search_by_projection_last_device
  lines:  pslfe_line_extract_batch_device (LSD, NFA, merge, top-N, LBD, line equations) -> pslfe_line_pair_batch_device ->
          pslfe_line_match_batch_device -> pslfe_glue_run_batch_device (isLineGood, crossings, planes)
and, with N > 1 ranks, the result gather (psl_slam_amd.multigpu.RecordGather)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NFEATURES, SCALE, NLEVELS, INI_TH, MIN_TH, NLINES = 1000, 1.2, 8, 20, 7, 200
TUM1 = (517.306408, 516.469215, 318.643040, 255.313989, 0, 0, 0, 0, 0, 40.0)


def bench_kernels():
    """Harness-only HIP kernel (tools/bench_kernels/bench_kernels.hip): builds the projection queries of a step in one launch."""
    d = os.path.join(ROOT, "tools", "bench_kernels")
    so, src = os.path.join(d, "libbench_kernels.so"), os.path.join(d, "bench_kernels.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so, src], check=True, capture_output=True)
    lib = C.CDLL(so)
    lib.bench_queries_from_prev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def scale_factors():
    import synth_frames as sf
    return sf.orb_scale_factors(NLEVELS, SCALE)


class BatchPipeline:
    def __init__(self, P, torch, dev, stream, local_rank, B, w, h, lines=True, nfeatures=NFEATURES, nlines=NLINES, second_stream=None):
        self.P, self.torch, self.dev, self.B, self.w, self.h, self.lines = P, torch, dev, B, w, h, lines
        self.stream = stream
        self.ctx = P.Context(local_rank, stream.cuda_stream)
        self.ctx_l = self.ctx if second_stream is None else P.Context(local_rank, second_stream.cuda_stream)
        self.stream_l = second_stream   # the line pipeline's stream when it runs beside the ORB pipeline (None: one stream)
        self.orb = P.ORBextractor(nfeatures, SCALE, NLEVELS, INI_TH, MIN_TH, ctx=self.ctx, max_batch=B)
        self.cap = self.orb.max_keypoints(w, h)
        self.grid = P.FrameGrid(self.cap, B, ctx=self.ctx)
        self.bounds = (0.0, 0.0, float(w), float(h))
        self.scale_t = torch.tensor(scale_factors(), device=dev)
        self.queries = torch.zeros((B, self.cap, 8), dtype=torch.float32, device=dev)
        self.qdesc = torch.zeros((B, self.cap, 32), dtype=torch.uint8, device=dev)
        self.nq = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.match = torch.full((B, self.cap), -1, dtype=torch.int32, device=dev)
        self.nmatches = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.BK = bench_kernels()
        self.cam = np.zeros((), P.CAMERA_DTYPE)
        for k_, v_ in zip(P.CAMERA_DTYPE.names, TUM1):
            self.cam[k_] = np.float32(v_)
        if lines:
            self.le = P.LINEextractor(1, 1.2, nlines, 0.0, ctx=self.ctx_l, max_batch=B)
            self.nlines = nlines
            self.glue = None
            self.klcap = None

    def _lazy_line_buffers(self):
        if self.klcap is None:
            _, _, _, _, self.klcap = self.le.results_device()
            self.lmatch = self.torch.full((self.B, self.klcap), -1, dtype=self.torch.int32, device=self.dev)
            self.lnm = self.torch.zeros((self.B,), dtype=self.torch.int32, device=self.dev)
            self.glue = self.P.FrameGlue(max_lines=self.klcap, max_fans=4096, max_batch=self.B, ctx=self.ctx_l)

    def contexts(self):
        return [self.ctx] if self.ctx_l is self.ctx else [self.ctx, self.ctx_l]

    def step(self, d_gray, d_depth=None, nframes=None):
        """d_gray: device address of [B][h][w] u8, d_depth: [B][h][w] f32 (lines).  Asynchronous.  nframes <= B: only the first
        nframes frames of the batch (frame 0's predecessor is then frame nframes-1)."""
        P, w, h = self.P, self.w, self.h
        B = self.B if nframes is None else int(nframes)
        assert 1 <= B <= self.B
        self.orb.extract_batch_device(d_gray, B, w, h, w, w * h)
        k, d, c, _ = self.orb.results_device()
        self.grid.set_from_orb(self.orb, self.bounds)
        rc = self.BK.bench_queries_from_prev(self.stream.cuda_stream, k, d, c, B, self.cap, NLEVELS, self.scale_t.data_ptr(), 15.0,
                                             self.queries.data_ptr(), self.qdesc.data_ptr(), self.nq.data_ptr())
        assert rc == 0
        P.search_by_projection_last_device(self.grid, 0, B, self.queries.data_ptr(), self.qdesc.data_ptr(), self.nq.data_ptr(), self.cap, True,
                                           self.match.data_ptr(), self.nmatches.data_ptr())
        if self.lines:
            self.le.extract_batch_device(d_gray, B, w, h, w, w * h)             # LSD -> NFA -> merge -> top-N -> LBD -> line equations
            self._lazy_line_buffers()
            self.le.pair_batch_device(20.0, float(np.float32(np.pi / 4)))       # CPartiallyRecoverConnectivity (src/Frame.cc:505)
            self.le.match_batch_device(1, 0.9, self.lmatch.data_ptr(), self.lnm.data_ptr())  # lmatcher.match(last, cur, 0.9) (src/Tracking.cc:901)
            d_kls, _, _, d_nkl, _ = self.le.results_device()
            d_fans, d_nfans = self.le.fans_device()
            self.glue.run_batch_device(B, d_kls, self.klcap, d_nkl, d_fans, 4096, d_nfans, d_depth, w, h, self.cam, 1)  # src/Frame.cc:500-660

    def record_layout(self, mg):
        return mg.RecordLayout(self.cap, self.nlines if self.lines else 0, 512 if self.lines else 0, 64 if self.lines else 0)

    def record_sources(self, mg):
        """Where this step's results live (pslfe_record_pack_device)."""
        k, d, c, cap = self.orb.results_device()
        S = mg.RecordSources()
        S.d_kps, S.d_desc, S.d_kp_counts, S.kp_stride = k, d, c, cap
        S.d_match, S.d_nmatches, S.match_stride = self.match.data_ptr(), self.nmatches.data_ptr(), self.cap
        if self.lines:
            d_kls, d_ldesc, d_eq, d_nkl, klcap = self.le.results_device()
            d_fans, d_nfans = self.le.fans_device()
            S.d_kls, S.d_ldesc, S.d_lineEq, S.d_kl_counts, S.kl_stride = d_kls, d_ldesc, d_eq, d_nkl, klcap
            S.d_lmatch, S.d_nlmatches, S.lmatch_stride = self.lmatch.data_ptr(), self.lnm.data_ptr(), klcap
            S.d_fans, S.d_fan_counts, S.fan_stride = d_fans, d_nfans, 4096
            pl, pln, pc, ps = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
            P = self.P
            P._check(P.lib().pslfe_glue_planes_device(self.glue._h, C.byref(pl), C.byref(pln), C.byref(pc), C.byref(ps)), "pslfe_glue_planes_device")
            S.d_planes, S.d_plane_lines, S.d_plane_counts, S.plane_stride = pl.value, pln.value, pc.value, ps.value
        return S

    def fetch_frame(self, f):
        """Results of frame f of the last step, on the host (through the per-frame fetch entry points)."""
        out = {}
        out["kps"], out["desc"] = self.orb.fetch(f, self.w, self.h)
        out["match"] = self.match[f].cpu().numpy()
        out["nmatches"] = int(self.nmatches[f].item())
        if self.lines:
            out["kls"], out["ldesc"], out["lineEq"], st = self.le.fetch(f)
            assert st == 0
            out["fans"] = self.le.fans_fetch(f)
            out["lmatch"] = self.lmatch[f].cpu().numpy()
            out["lnm"] = int(self.lnm[f].item())
            g = self.glue.fetch(f, len(out["kls"]))
            out["lines3d"], out["planes"], out["plane_lines"] = g["lines3d"], g["planes"], g["lineNo"]
        return out


def oracle_frame(prev_gray, gray, depth, frame_index, w, h, lines=True, cam=None, nfeatures=NFEATURES, nlines=NLINES, cache=None):
    """What BatchPipeline.step must produce for a frame, from the CPU oracle (tests/oracle_lib.py): same dict as fetch_frame.
    cache: dict keyed by id for the per-image extraction results (a frame is both somebody's predecessor and itself)."""
    import oracle_lib as O
    scale = scale_factors()

    def extract(img, key):
        if cache is not None and key in cache:
            return cache[key]
        orb = O.OracleORB(nfeatures, SCALE, NLEVELS, INI_TH, MIN_TH)
        r = dict(zip(("kps", "desc"), orb(img)))
        if lines:
            r["kls"], r["ldesc"], r["lineEq"] = O.line_extract(img, nlines)
        if cache is not None:
            cache[key] = r
        return r
    p, c = extract(prev_gray[1], prev_gray[0]), extract(gray[1], gray[0])
    out = dict(c)
    q = np.zeros(len(p["kps"]), O.PROJQUERY_DTYPE)
    q["u"], q["v"] = p["kps"]["x"], p["kps"]["y"]
    q["radius"] = np.float32(15.0) * scale[p["kps"]["octave"]]
    q["min_level"], q["max_level"] = p["kps"]["octave"] - 1, p["kps"]["octave"] + 1
    q["angle"], q["blocks"] = p["kps"]["angle"], 1
    nm, match, _ = O.search_by_projection_last(c["kps"], c["desc"], None, (0.0, 0.0, float(w), float(h)), q, p["desc"], None, True)
    out["match"], out["nmatches"] = match, nm
    if lines:
        kls = c["kls"]
        L4 = np.stack([kls[k] for k in ("startPointX", "startPointY", "endPointX", "endPointY")], 1).astype(np.float32) if len(kls) else np.zeros((0, 4), np.float32)
        out["fans"] = O.lil_pair(L4, 20.0, np.float32(np.pi / 4), w, h)
        out["lnm"], out["lmatch"] = O.line_match_nnr(p["ldesc"], c["ldesc"], 0.9)
        g = O.frame_glue(kls, out["fans"], depth, cam, seed=1 + frame_index)
        out["lines3d"], out["planes"], out["plane_lines"] = g["lines3d"], g["planes"], g["lineNo"]
    return out


def compare_frame(got, ref, what=""):
    """Bit-for-bit; raises AssertionError naming the first array that differs."""
    assert got["kps"].tobytes() == ref["kps"].tobytes(), f"{what}keypoints differ from the oracle"
    assert np.array_equal(got["desc"], ref["desc"]), f"{what}ORB descriptors differ from the oracle"
    assert got["nmatches"] == ref["nmatches"] and np.array_equal(got["match"][:len(ref["match"])], ref["match"]), f"{what}point matches differ from the oracle"
    if "kls" in ref:
        assert got["kls"].tobytes() == ref["kls"].tobytes(), f"{what}keylines differ from the oracle ({len(got['kls'])} vs {len(ref['kls'])})"
        assert np.array_equal(got["ldesc"], ref["ldesc"]), f"{what}LBD descriptors differ from the oracle"
        assert got["lineEq"].tobytes() == np.ascontiguousarray(ref["lineEq"]).tobytes(), f"{what}line equations differ from the oracle"
        assert got["fans"].tobytes() == ref["fans"].tobytes(), f"{what}fans differ from the oracle"
        assert got["lnm"] == ref["lnm"] and np.array_equal(got["lmatch"][:len(ref["lmatch"])], ref["lmatch"]), f"{what}line matches differ from the oracle"
        assert got["lines3d"].tobytes() == ref["lines3d"].tobytes(), f"{what}mvLines3D differs from the oracle"
        assert got["planes"].tobytes() == ref["planes"].tobytes() and np.array_equal(got["plane_lines"], ref["plane_lines"]), f"{what}planes differ from the oracle"
