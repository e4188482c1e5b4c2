"""psl-slam_amd — MI355X-native feature front-end for PSL-SLAM (host-side Python mirror).

Thin ctypes layer over the C ABI in include/pslfe.h (libpslfe.so, HIP/gfx950).  The class and
method names follow the reference's C++ interfaces (include/ORBextractor.h:59,
add_inc/LineExtractor.h:167, include/ORBmatcher.h, add_inc/LSDmatcher.h) so that the parity tests
read like calls into the reference.  There is no CPU fallback: without the built library or
without a gfx950 GPU every entry point raises.

Import name: ``psl_slam_amd`` (root shim psl_slam_amd.py; the directory keeps the project's
hyphenated name).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpslfe.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
KEYLINE_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                          ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                          ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"),
                          ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                          ("lineLength", "<f4"), ("numOfPixels", "<i4")])
PROJQUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("min_level", "<i4"),
                            ("max_level", "<i4"), ("angle", "<f4"), ("blocks", "<i4")])
BOWQUERY_DTYPE = np.dtype([("start", "<i4"), ("len", "<i4"), ("angle", "<f4")])
CAMERA_DTYPE = np.dtype([(k, "<f4") for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")])
assert KEYPOINT_DTYPE.itemsize == 28 and KEYLINE_DTYPE.itemsize == 68 and PROJQUERY_DTYPE.itemsize == 32


class PslfeError(RuntimeError):
    pass


_lib = None


def build(force=False):
    """Compile libpslfe.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_pslfe_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force)


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PslfeError(f"{LIB_PATH} is missing: run `python psl-slam_amd/build.py` (needs hipcc); "
                             "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.pslfe_version.restype = C.c_char_p
        L.pslfe_last_error.restype = C.c_char_p
        L.pslfe_orb_scale_factor.restype = C.c_float
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise PslfeError(f"{what} failed with code {rc}: {lib().pslfe_last_error().decode()}")


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Context:
    """One per GPU (pslfe_ctx)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().pslfe_ctx_create(C.c_int(device), C.byref(self._h)), "pslfe_ctx_create")
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, hip_stream):
        _check(lib().pslfe_ctx_set_stream(self._h, C.c_void_p(hip_stream)), "pslfe_ctx_set_stream")

    def synchronize(self):
        _check(lib().pslfe_ctx_synchronize(self._h), "pslfe_ctx_synchronize")

    def profile(self, enable=True):
        _check(lib().pslfe_ctx_profile(self._h, C.c_int(1 if enable else 0)), "pslfe_ctx_profile")

    def profile_only(self, stage=None):
        _check(lib().pslfe_ctx_profile_only(self._h, None if not stage else stage.encode()), "pslfe_ctx_profile_only")

    def profile_reset(self):
        _check(lib().pslfe_ctx_profile_reset(self._h), "pslfe_ctx_profile_reset")

    def stage_time(self, stage):
        ms, n = C.c_double(), C.c_int()
        _check(lib().pslfe_ctx_stage_time(self._h, stage.encode(), C.byref(ms), C.byref(n)), "pslfe_ctx_stage_time")
        return ms.value, n.value

    def device_array(self, host_array):
        """Upload a numpy array into freshly allocated HBM; returns (device address, nbytes)."""
        a = np.ascontiguousarray(host_array)
        d = C.c_void_p()
        _check(lib().pslfe_device_alloc(self._h, C.c_size_t(a.nbytes), C.byref(d)), "pslfe_device_alloc")
        _check(lib().pslfe_device_upload(self._h, d, _ptr(a), C.c_size_t(a.nbytes)), "pslfe_device_upload")
        return d.value, a.nbytes

    def device_free(self, d_ptr):
        _check(lib().pslfe_device_free(self._h, C.c_void_p(d_ptr)), "pslfe_device_free")

    def close(self):
        if self._h:
            lib().pslfe_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class ORBextractor:
    """== ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-114)."""

    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, ctx=None, max_batch=1):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        self.nlevels = nlevels
        self.max_batch = max_batch
        _check(lib().pslfe_orb_create(self.ctx._h, C.c_int(nfeatures), C.c_float(scaleFactor), C.c_int(nlevels),
                                      C.c_int(iniThFAST), C.c_int(minThFAST), C.c_int(max_batch), C.byref(self._h)),
               "pslfe_orb_create")

    # getters of include/ORBextractor.h:63-84
    def GetLevels(self):
        return lib().pslfe_orb_levels(self._h)

    def GetScaleFactor(self):
        return lib().pslfe_orb_scale_factor(self._h)

    def _factors(self):
        a = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        _check(lib().pslfe_orb_scale_factors(self._h, *[_ptr(x) for x in a]), "pslfe_orb_scale_factors")
        return a

    def GetScaleFactors(self):
        return self._factors()[0]

    def GetInverseScaleFactors(self):
        return self._factors()[1]

    def GetScaleSigmaSquares(self):
        return self._factors()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._factors()[3]

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        _check(lib().pslfe_orb_features_per_level(self._h, _ptr(q)), "pslfe_orb_features_per_level")
        return q

    def max_keypoints(self, w, h):
        n = lib().pslfe_orb_max_keypoints(self._h, C.c_int(w), C.c_int(h))
        if n < 0:
            _check(n, "pslfe_orb_max_keypoints")
        return n

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors): mask is ignored as in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 image expected"
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _check(lib().pslfe_orb_extract(self._h, _ptr(image), C.c_int(w), C.c_int(h), C.c_int(image.strides[0]),
                                       _ptr(kps), _ptr(desc), C.c_int(cap), C.byref(n)), "pslfe_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: (F, h, w) uint8 host array -> list of (keypoints, descriptors)."""
        assert images.dtype == np.uint8 and images.ndim == 3 and images.flags.c_contiguous
        F, h, w = images.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros((F, cap), KEYPOINT_DTYPE)
        desc = np.zeros((F, cap, 32), np.uint8)
        counts = np.zeros(F, np.int32)
        _check(lib().pslfe_orb_extract_batch(self._h, _ptr(images), C.c_int(F), C.c_int(w), C.c_int(h), C.c_int(w),
                                             C.c_size_t(w * h), _ptr(kps), _ptr(desc), C.c_int(cap), _ptr(counts)),
               "pslfe_orb_extract_batch")
        return [(kps[f, :counts[f]].copy(), desc[f, :counts[f]].copy()) for f in range(F)]

    def extract_batch_device(self, d_ptr, nframes, w, h, stride, frame_stride):
        """Asynchronous extraction of frames resident in HBM (d_ptr = device address as int)."""
        _check(lib().pslfe_orb_extract_batch_device(self._h, C.c_void_p(d_ptr), C.c_int(nframes), C.c_int(w), C.c_int(h),
                                                    C.c_int(stride), C.c_size_t(frame_stride)),
               "pslfe_orb_extract_batch_device")

    def results_device(self):
        k, d, c, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        _check(lib().pslfe_orb_results_device(self._h, C.byref(k), C.byref(d), C.byref(c), C.byref(cap)),
               "pslfe_orb_results_device")
        return k.value, d.value, c.value, cap.value

    def fetch(self, frame, w, h):
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _check(lib().pslfe_orb_fetch(self._h, C.c_int(frame), _ptr(kps), _ptr(desc), C.c_int(cap), C.byref(n)), "pslfe_orb_fetch")
        return kps[:n.value].copy(), desc[:n.value].copy()

    # stage taps (parity tests)
    def debug_level_image(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        _check(lib().pslfe_orb_debug_level_size(self._h, C.c_int(level), C.byref(w), C.byref(h)), "pslfe_orb_debug_level_size")
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().pslfe_orb_debug_level_image(self._h, C.c_int(frame), C.c_int(level), C.c_int(1 if blurred else 0),
                                                 _ptr(out), C.c_int(w.value)), "pslfe_orb_debug_level_image")
        return out

    def _debug_xys(self, fn, frame, level):
        n = C.c_int()
        _check(fn(self._h, C.c_int(frame), C.c_int(level), None, C.c_int(0), C.byref(n)), "pslfe_orb_debug")
        out = np.zeros((max(n.value, 1), 3), np.int32)
        _check(fn(self._h, C.c_int(frame), C.c_int(level), _ptr(out), C.c_int(out.shape[0]), C.byref(n)), "pslfe_orb_debug")
        return out[:n.value]

    def debug_candidates(self, frame, level):
        return self._debug_xys(lib().pslfe_orb_debug_candidates, frame, level)

    def debug_level_keypoints(self, frame, level):
        return self._debug_xys(lib().pslfe_orb_debug_level_keypoints, frame, level)

    def close(self):
        if self._h:
            lib().pslfe_orb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rgb_to_gray(rgb, is_rgb=True, ctx=None):
    """cvtColor(im, gray, CV_RGB2GRAY if mbRGB else CV_BGR2GRAY), src/Tracking.cc:219-232. rgb: (h, w, 3) u8."""
    ctx = ctx or default_context()
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    gray = np.zeros((h, w), np.uint8)
    _check(lib().pslfe_rgb_to_gray(ctx._h, _ptr(rgb), C.c_int(w), C.c_int(h), C.c_int(3 * w), C.c_int(1 if is_rgb else 0),
                                   _ptr(gray)), "pslfe_rgb_to_gray")
    return gray


def depth_to_float(depth, factor, ctx=None):
    """imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor), src/Tracking.cc:234-235. depth: u16 array."""
    ctx = ctx or default_context()
    depth = np.ascontiguousarray(depth, np.uint16)
    out = np.zeros(depth.shape, np.float32)
    _check(lib().pslfe_depth_to_float(ctx._h, _ptr(depth), C.c_size_t(depth.size), C.c_float(factor), _ptr(out)),
           "pslfe_depth_to_float")
    return out


class FrameGrid:
    """Keypoints of up to `max_frames` frames on the 64x48 grid of include/Frame.h:45-46
    (== the part of ORB_SLAM2::Frame the matchers read: mvKeysUn, mDescriptors, mvuRight, mGrid)."""

    def __init__(self, max_keypoints, max_frames=1, ctx=None):
        self.ctx = ctx or default_context()
        self.cap, self.max_frames = max_keypoints, max_frames
        self._h = C.c_void_p()
        _check(lib().pslfe_frame_create(self.ctx._h, C.c_int(max_keypoints), C.c_int(max_frames), C.byref(self._h)),
               "pslfe_frame_create")
        self.n = [0] * max_frames

    def set(self, slot, kps, desc, bounds, uright=None):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        _check(lib().pslfe_frame_set(self._h, C.c_int(slot), _ptr(kps), _ptr(desc), _ptr(ur), C.c_int(len(kps)),
                                     *[C.c_float(b) for b in bounds]), "pslfe_frame_set")
        self.n[slot] = len(kps)

    def set_from_orb(self, orb, bounds):
        _check(lib().pslfe_frame_set_from_orb(self._h, orb._h, *[C.c_float(b) for b in bounds]), "pslfe_frame_set_from_orb")

    def image_bounds(self, cam, cols, rows):
        """Frame::ComputeImageBounds src/Frame.cc:1135-1168 -> (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
        b = np.zeros(4, np.float32)
        cam = np.ascontiguousarray(cam, CAMERA_DTYPE).reshape(1)
        _check(lib().pslfe_image_bounds(self._h, _ptr(cam), C.c_int(cols), C.c_int(rows), _ptr(b)), "pslfe_image_bounds")
        return b

    def set_rgbd(self, slot, kps, desc, depth, cam):
        """UndistortKeyPoints + ComputeStereoFromRGBD + AssignFeaturesToGrid (src/Frame.cc:105-171) for one frame.
        depth: CV_32F image in metres."""
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        cam = np.ascontiguousarray(cam, CAMERA_DTYPE).reshape(1)
        _check(lib().pslfe_frame_set_rgbd(self._h, C.c_int(slot), _ptr(kps), _ptr(desc), C.c_int(len(kps)), _ptr(depth),
                                          C.c_int(depth.shape[1]), C.c_int(depth.shape[0]), C.c_int(depth.shape[1]), _ptr(cam)),
               "pslfe_frame_set_rgbd")
        self.n[slot] = len(kps)

    def set_from_orb_rgbd(self, orb, d_depth, width, height, cam):
        """Batched, HBM to HBM: d_depth is a device pointer to [nframes][height][width] float."""
        cam = np.ascontiguousarray(cam, CAMERA_DTYPE).reshape(1)
        _check(lib().pslfe_frame_set_from_orb_rgbd(self._h, orb._h, C.c_void_p(int(d_depth)), C.c_int(width), C.c_int(height),
                                                   _ptr(cam)), "pslfe_frame_set_from_orb_rgbd")

    def fetch(self, slot):
        """(mvKeysUn, mvDepth, mvuRight) of a slot."""
        kps = np.zeros(self.cap, KEYPOINT_DTYPE)
        dep = np.zeros(self.cap, np.float32)
        ur = np.zeros(self.cap, np.float32)
        n = C.c_int()
        _check(lib().pslfe_frame_fetch(self._h, C.c_int(slot), _ptr(kps), _ptr(dep), _ptr(ur), C.c_int(self.cap), C.byref(n)),
               "pslfe_frame_fetch")
        self.n[slot] = n.value
        return kps[:n.value], dep[:n.value], ur[:n.value]

    def debug_grid(self, slot):
        start = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(self.cap, np.int32)
        n = C.c_int()
        _check(lib().pslfe_frame_debug_grid(self._h, C.c_int(slot), _ptr(start), _ptr(idx), C.c_int(self.cap), C.byref(n)),
               "pslfe_frame_debug_grid")
        return start, idx[:n.value]

    def close(self):
        if self._h:
            lib().pslfe_frame_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBmatcher:
    """== ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:36-104), the per-frame projection searches."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri

    def _run(self, fn, frame, slot, queries, qdesc, taken, extra):
        queries = np.ascontiguousarray(queries, PROJQUERY_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        nq = len(queries)
        match = np.full(max(nq, 1), -1, np.int32)
        assigned = np.full(max(frame.n[slot], 1), -1, np.int32)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        nm = C.c_int()
        _check(fn(frame._h, C.c_int(slot), _ptr(queries), _ptr(qdesc), C.c_int(nq), _ptr(tk), extra, _ptr(match),
                  _ptr(assigned), C.byref(nm)), "pslfe_orb_search_by_projection")
        return nm.value, match[:nq], assigned[:frame.n[slot]]

    def SearchByProjectionLast(self, frame, slot, queries, qdesc, taken=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) src/ORBmatcher.cc:1328."""
        return self._run(lib().pslfe_orb_search_by_projection_last, frame, slot, queries, qdesc, taken,
                         C.c_int(1 if self.mbCheckOrientation else 0))

    def SearchByProjectionKF(self, frame, slot, queries, qdesc, taken=None, ORBdist=100):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) src/ORBmatcher.cc:1472 (relocalisation)."""
        queries = np.ascontiguousarray(queries, PROJQUERY_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        nq = len(queries)
        match = np.full(max(nq, 1), -1, np.int32)
        assigned = np.full(max(frame.n[slot], 1), -1, np.int32)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        nm = C.c_int()
        _check(lib().pslfe_orb_search_by_projection_kf(frame._h, C.c_int(slot), _ptr(queries), _ptr(qdesc), C.c_int(nq), _ptr(tk),
                                                       C.c_int(ORBdist), C.c_int(1 if self.mbCheckOrientation else 0), _ptr(match),
                                                       _ptr(assigned), C.byref(nm)), "pslfe_orb_search_by_projection_kf")
        return nm.value, match[:nq], assigned[:frame.n[slot]]

    def SearchByBoW(self, frame, slot, fidx, runs, qangle, qdesc):
        """SearchByBoW(pKF, F, vpMapPointMatches) src/ORBmatcher.cc:159 on host-provided FeatureVectors: fidx = F.mFeatVec
        flattened in node order; runs[i] = (start, len) of query i's node in fidx; qangle / qdesc per query."""
        fidx = np.ascontiguousarray(fidx, np.int32)
        q = np.zeros(len(runs), BOWQUERY_DTYPE)
        if len(runs):
            r = np.asarray(runs, np.int32).reshape(-1, 2)
            q["start"], q["len"], q["angle"] = r[:, 0], r[:, 1], np.asarray(qangle, np.float32)
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        nq = len(q)
        match = np.full(max(nq, 1), -1, np.int32)
        assigned = np.full(max(frame.n[slot], 1), -1, np.int32)
        nm = C.c_int()
        _check(lib().pslfe_orb_search_by_bow(frame._h, C.c_int(slot), _ptr(fidx), C.c_int(len(fidx)), _ptr(q), _ptr(qdesc), C.c_int(nq),
                                             C.c_float(self.mfNNratio), C.c_int(1 if self.mbCheckOrientation else 0), _ptr(match),
                                             _ptr(assigned), C.byref(nm)), "pslfe_orb_search_by_bow")
        return nm.value, match[:nq], assigned[:frame.n[slot]]

    def SearchByProjectionMap(self, frame, slot, queries, qdesc, taken=None):
        """SearchByProjection(F, vpMapPoints, th) src/ORBmatcher.cc:45."""
        return self._run(lib().pslfe_orb_search_by_projection_map, frame, slot, queries, qdesc, taken,
                         C.c_float(self.mfNNratio))


def hamming_knn2(q, t, ctx=None):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) -> (idx[nq,2], dist[nq,2])."""
    ctx = ctx or default_context()
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((max(len(q), 1), 2), np.int32)
    dist = np.zeros((max(len(q), 1), 2), np.int32)
    _check(lib().pslfe_hamming_knn2(ctx._h, _ptr(q), C.c_int(len(q)), _ptr(t), C.c_int(len(t)), _ptr(idx), _ptr(dist)),
           "pslfe_hamming_knn2")
    return idx[:len(q)], dist[:len(q)]


class LSDmatcher:
    """== ORB_SLAM2::LSDmatcher (add_inc/LSDmatcher.h:18-75), descriptor part."""

    TH_HIGH, TH_LOW = 80, 50

    def __init__(self, nnratio=0.95, checkOri=True, ctx=None):
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri
        self.ctx = ctx or default_context()

    def matchNNR(self, desc1, desc2, nnr):
        """add_src/LSDmatcher.cpp:354-376 -> (matches, matches_12)."""
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        m12 = np.full(max(len(d1), 1), -1, np.int32)
        nm = C.c_int()
        _check(lib().pslfe_line_match_nnr(self.ctx._h, _ptr(d1), C.c_int(len(d1)), _ptr(d2), C.c_int(len(d2)),
                                          C.c_float(nnr), _ptr(m12), C.byref(nm)), "pslfe_line_match_nnr")
        return nm.value, m12[:len(d1)]

    match = matchNNR  # LSDmatcher::match's live branch is matchNNR (:378-413)


class _DevArray:
    """Zero-copy view of library-owned HBM for frameworks that read __cuda_array_interface__
    (torch.as_tensor(view, device='cuda') aliases the memory; nothing is copied)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr,
                                         "version": 2, "strides": None}


def orb_results_as_arrays(orb, nframes):
    """(kps [F,cap,7] f32-view, desc [F,cap,32] u8, counts [F] i32) device views of the last batch."""
    k, d, c, cap = orb.results_device()
    return (_DevArray(k, (nframes, cap, 7), "<f4"), _DevArray(d, (nframes, cap, 32), "|u1"),
            _DevArray(c, (nframes,), "<i4"), cap)


def search_by_projection_last_device(frame, slot0, npairs, d_queries, d_qdesc, d_nq, qstride, check_orientation,
                                     d_match, d_nmatches):
    """Batched HBM-resident SearchByProjection(cur,last); all d_* are device addresses (ints)."""
    _check(lib().pslfe_orb_search_by_projection_last_device(
        frame._h, C.c_int(slot0), C.c_int(npairs), C.c_void_p(d_queries), C.c_void_p(d_qdesc), C.c_void_p(d_nq),
        C.c_int(qstride), C.c_int(1 if check_orientation else 0), C.c_void_p(d_match), C.c_void_p(d_nmatches)),
        "pslfe_orb_search_by_projection_last_device")


class LINEextractor:
    """== ORB_SLAM2::LINEextractor (add_inc/LineExtractor.h:160-255)."""

    def __init__(self, numOctaves=1, scale=1.2, nLSDFeature=200, min_line_length=0.0, ctx=None, max_batch=1):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        self.numOctaves = numOctaves
        _check(lib().pslfe_line_create(self.ctx._h, C.c_int(numOctaves), C.c_float(scale), C.c_int(nLSDFeature),
                                       C.c_double(min_line_length), C.c_int(max_batch), C.byref(self._h)), "pslfe_line_create")
        lib().pslfe_line_scale_factor.restype = C.c_float

    LSD_REFINE_STD, LSD_REFINE_ADV = 1, 2

    def set_refine(self, mode):
        """cv::createLineSegmentDetector(refine) behind the extractor: LSD_REFINE_ADV (default: rect_improve + NFA test, what
        the stock contrib LSDDetector constructs) or LSD_REFINE_STD (the vendored, never-called LSDDetectorC)."""
        _check(lib().pslfe_line_set_refine(self._h, C.c_int(mode)), "pslfe_line_set_refine")

    def GetLevels(self):
        return lib().pslfe_line_levels(self._h)

    def GetScaleFactor(self):
        return lib().pslfe_line_scale_factor(self._h)

    def GetScaleFactors(self):
        a = np.zeros(self.numOctaves, np.float32)
        _check(lib().pslfe_line_scale_factors(self._h, _ptr(a), None, None, None), "pslfe_line_scale_factors")
        return a

    def lsd_detect(self, image, cap=8192):
        """LSDDetector::detect up to the clamped segment list -> (n, 4) float32."""
        assert image.dtype == np.uint8 and image.ndim == 2
        h, w = image.shape
        seg = np.zeros((cap, 4), np.float32)
        n = C.c_int()
        _check(lib().pslfe_lsd_detect(self._h, _ptr(image), C.c_int(w), C.c_int(h), C.c_int(image.strides[0]), _ptr(seg),
                                      C.c_int(cap), C.byref(n)), "pslfe_lsd_detect")
        return seg[:n.value].copy()

    def debug_gradient(self, frame=0):
        W, H = C.c_int(), C.c_int()
        _check(lib().pslfe_line_debug_gradient(self._h, C.c_int(frame), C.byref(W), C.byref(H), None, None, None), "pslfe_line_debug_gradient")
        scaled = np.zeros((H.value, W.value), np.float64)
        ang = np.zeros((H.value, W.value), np.float32)
        mod = np.zeros((H.value, W.value), np.float64)
        _check(lib().pslfe_line_debug_gradient(self._h, C.c_int(frame), C.byref(W), C.byref(H), _ptr(scaled), _ptr(ang), _ptr(mod)),
               "pslfe_line_debug_gradient")
        return scaled, ang, mod

    def close(self):
        if self._h:
            lib().pslfe_line_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _line_call(self, image, mask=None):
    """operator()(image, mask, keylines, descriptors, lineVec2d) -> (keylines, descriptors, lineEq)."""
    if image is None or image.size == 0:
        return np.zeros(0, KEYLINE_DTYPE), np.zeros((0, 32), np.uint8), np.zeros((0, 3))
    assert image.dtype == np.uint8 and image.ndim == 2
    h, w = image.shape
    cap = 1024
    kls = np.zeros(cap, KEYLINE_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    eq = np.zeros((cap, 3), np.float64)
    n = C.c_int()
    _check(lib().pslfe_line_extract(self._h, _ptr(image), C.c_int(w), C.c_int(h), C.c_int(image.strides[0]), _ptr(kls), _ptr(desc),
                                    _ptr(eq), C.c_int(cap), C.byref(n)), "pslfe_line_extract")
    return kls[:n.value].copy(), desc[:n.value].copy(), eq[:n.value].copy()


def _line_extract_batch_device(self, d_ptr, nframes, w, h, stride, frame_stride):
    _check(lib().pslfe_line_extract_batch_device(self._h, C.c_void_p(d_ptr), C.c_int(nframes), C.c_int(w), C.c_int(h), C.c_int(stride),
                                                 C.c_size_t(frame_stride)), "pslfe_line_extract_batch_device")


def _line_fetch(self, frame, cap=1024):
    kls = np.zeros(cap, KEYLINE_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    eq = np.zeros((cap, 3), np.float64)
    n, st = C.c_int(), C.c_int()
    _check(lib().pslfe_line_fetch(self._h, C.c_int(frame), _ptr(kls), _ptr(desc), _ptr(eq), C.c_int(cap), C.byref(n), C.byref(st)),
           "pslfe_line_fetch")
    return kls[:n.value].copy(), desc[:n.value].copy(), eq[:n.value].copy(), st.value


def _line_results_device(self):
    k, d, e, c, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
    _check(lib().pslfe_line_results_device(self._h, C.byref(k), C.byref(d), C.byref(e), C.byref(c), C.byref(cap)), "pslfe_line_results_device")
    return k.value, d.value, e.value, c.value, cap.value


def _line_optimize_and_merge(self, segments, w, h, cap=1024):
    seg = np.ascontiguousarray(segments, np.float32).reshape(-1, 4)
    kls = np.zeros(cap, KEYLINE_DTYPE)
    n = C.c_int()
    _check(lib().pslfe_line_optimize_and_merge(self._h, _ptr(seg), C.c_int(len(seg)), C.c_int(w), C.c_int(h), _ptr(kls), C.c_int(cap),
                                               C.byref(n)), "pslfe_line_optimize_and_merge")
    return kls[:n.value].copy()


def _line_lbd(self, image, keylines, want_float=False):
    kls = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
    h, w = image.shape
    desc = np.zeros((max(len(kls), 1), 32), np.uint8)
    fdesc = np.zeros((max(len(kls), 1), 72), np.float32) if want_float else None
    _check(lib().pslfe_lbd_compute(self._h, _ptr(image), C.c_int(w), C.c_int(h), C.c_int(image.strides[0]), _ptr(kls), C.c_int(len(kls)),
                                   _ptr(desc), _ptr(fdesc)), "pslfe_lbd_compute")
    return (desc[:len(kls)], fdesc[:len(kls)]) if want_float else desc[:len(kls)]


def _line_debug_sobel(self, w, h, frame=0):
    dx = np.zeros((h, w), np.int16)
    dy = np.zeros((h, w), np.int16)
    _check(lib().pslfe_line_debug_sobel(self._h, C.c_int(frame), _ptr(dx), _ptr(dy)), "pslfe_line_debug_sobel")
    return dx, dy


def _line_pair(self, lines, radius, fanThr, cols, rows, cap=4096):
    """CPartiallyRecoverConnectivity(mLines, radius, fans, img, fanThr) -> fans (k, 4)."""
    L = np.ascontiguousarray(lines, np.float32).reshape(-1, 4)
    fans = np.zeros((cap, 4), np.float32)
    k = C.c_int()
    _check(lib().pslfe_lil_pair(self._h, _ptr(L), C.c_int(len(L)), C.c_float(radius), C.c_float(fanThr), C.c_int(cols), C.c_int(rows),
                                _ptr(fans), C.c_int(cap), C.byref(k)), "pslfe_lil_pair")
    return fans[:k.value].copy()


def _line_pair_batch_device(self, radius=20.0, fanThr=np.pi / 4):
    _check(lib().pslfe_line_pair_batch_device(self._h, C.c_float(radius), C.c_float(fanThr)), "pslfe_line_pair_batch_device")


def _line_fans_fetch(self, frame, cap=4096):
    fans = np.zeros((cap, 4), np.float32)
    k = C.c_int()
    _check(lib().pslfe_line_fans_fetch(self._h, C.c_int(frame), _ptr(fans), C.c_int(cap), C.byref(k)), "pslfe_line_fans_fetch")
    return fans[:k.value].copy()


def _line_match_batch_device(self, shift, nnr, d_matches12, d_nmatches):
    _check(lib().pslfe_line_match_batch_device(self._h, C.c_int(shift), C.c_float(nnr), C.c_void_p(d_matches12), C.c_void_p(d_nmatches)),
           "pslfe_line_match_batch_device")


LINEextractor.match_batch_device = _line_match_batch_device
LINEextractor.__call__ = _line_call
LINEextractor.extract_batch_device = _line_extract_batch_device
LINEextractor.fetch = _line_fetch
LINEextractor.results_device = _line_results_device
LINEextractor.optimize_and_merge = _line_optimize_and_merge
LINEextractor.lbd_compute = _line_lbd
LINEextractor.debug_sobel = _line_debug_sobel
LINEextractor.pair = _line_pair
LINEextractor.pair_batch_device = _line_pair_batch_device
def _line_fans_device(self):
    f, n, st = C.c_void_p(), C.c_void_p(), C.c_int()
    _check(lib().pslfe_line_fans_device(self._h, C.byref(f), C.byref(n), C.byref(st)), "pslfe_line_fans_device")
    return f.value, n.value


LINEextractor.fans_fetch = _line_fans_fetch
LINEextractor.fans_device = _line_fans_device


def _lsd_search_by_geom_appearance(self, kl_last, desc_last, kl_cur, desc_cur, has_mapline, desc_th, bounds):
    """LSDmatcher::SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th) -> (lmatches, matches12, assigned).
    bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    k1 = np.ascontiguousarray(kl_last, KEYLINE_DTYPE)
    k2 = np.ascontiguousarray(kl_cur, KEYLINE_DTYPE)
    d1 = np.ascontiguousarray(desc_last, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
    hm = np.ascontiguousarray(has_mapline, np.uint8)
    m12 = np.full(max(len(k1), 1), -1, np.int32)
    asg = np.full(max(len(k2), 1), -1, np.int32)
    n = C.c_int()
    _check(lib().pslfe_line_search_by_geom_appearance(self.ctx._h, _ptr(k1), _ptr(d1), C.c_int(len(k1)), _ptr(k2), _ptr(d2), C.c_int(len(k2)),
                                                      _ptr(hm), C.c_float(desc_th), *[C.c_float(b) for b in bounds], _ptr(m12), _ptr(asg),
                                                      C.byref(n)), "pslfe_line_search_by_geom_appearance")
    return n.value, m12[:len(k1)], asg[:len(k2)]


def _lsd_frame_bf_match(self, ldesc1, ldesc2, TH):
    """LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) -> LineMatches."""
    d1 = np.ascontiguousarray(ldesc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(ldesc2, np.uint8).reshape(-1, 32)
    lm = np.full(max(len(d1), 1), -1, np.int32)
    _check(lib().pslfe_line_frame_bf_match(self.ctx._h, _ptr(d1), C.c_int(len(d1)), _ptr(d2), C.c_int(len(d2)), C.c_float(self.mfNNratio),
                                           C.c_float(TH), _ptr(lm)), "pslfe_line_frame_bf_match")
    return lm[:len(d1)]


LSDmatcher.SearchByGeomNApearance = _lsd_search_by_geom_appearance
LSDmatcher.FrameBFMatch = _lsd_frame_bf_match


def associate_planes(planes, points, map_planes, dTh, aTh, live=True, map_bad=None, ctx=None):
    """Map::AssociatePlanesByBoundary (live) / InsectLineMatch::SearchMapInsectline (dead) -> (nmatches, assoc)."""
    ctx = ctx or default_context()
    p = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
    q = np.ascontiguousarray(points, np.float64).reshape(-1, 15)
    m = np.ascontiguousarray(map_planes, np.float32).reshape(-1, 4)
    b = None if map_bad is None else np.ascontiguousarray(map_bad, np.uint8)
    assoc = np.full(max(len(p), 1), -1, np.int32)
    n = C.c_int()
    _check(lib().pslfe_associate_planes(ctx._h, _ptr(p), _ptr(q), C.c_int(len(p)), _ptr(m), _ptr(b), C.c_int(len(m)), C.c_float(dTh),
                                        C.c_float(aTh), C.c_int(1 if live else 0), _ptr(assoc), C.byref(n)), "pslfe_associate_planes")
    return n.value, assoc[:len(p)]


LINEQUERY_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("radius", "<f4"), ("th_cos", "<f4"),
                            ("vx", "<f4"), ("vy", "<f4"), ("length", "<f4"), ("blocks", "<i4"), ("wdir", "<f8", (3,))])
assert LINEQUERY_DTYPE.itemsize == 64


def _lsd_search_by_projection(self, kls, desc, lineEq, bounds, queries, qdesc, mode=0, dir3d=None, taken=None, want_grid=False):
    """LSDmatcher::SearchByProjection: mode 0 = (CurrentFrame, LastFrame, th), mode 1 = (F, vpMapLines, ...).
    bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY) -> (nmatches, match, assigned[, grid_start, grid_idx])."""
    k = np.ascontiguousarray(kls, KEYLINE_DTYPE)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    eq = np.ascontiguousarray(lineEq, np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(queries, LINEQUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    d3 = None if dir3d is None else np.ascontiguousarray(dir3d, np.float64).reshape(-1, 3)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
    match = np.full(max(len(q), 1), -1, np.int32)
    asg = np.full(max(len(k), 1), -1, np.int32)
    nm, gn = C.c_int(), C.c_int()
    gs = np.zeros(64 * 48 + 1, np.int32) if want_grid else None
    gi = np.zeros(max(len(k), 1) * 112, np.int32) if want_grid else None
    _check(lib().pslfe_line_search_by_projection(self.ctx._h, _ptr(k), _ptr(d), _ptr(eq), _ptr(d3), C.c_int(len(k)),
                                                 *[C.c_float(b) for b in bounds], _ptr(q), _ptr(qd), C.c_int(len(q)), _ptr(tk),
                                                 C.c_int(mode), C.c_float(self.mfNNratio), _ptr(match), _ptr(asg), C.byref(nm),
                                                 _ptr(gs), _ptr(gi), C.c_int(0 if gi is None else len(gi)), C.byref(gn)),
           "pslfe_line_search_by_projection")
    out = (nm.value, match[:len(q)], asg[:len(k)])
    return out + (gs, gi[:gn.value]) if want_grid else out


LSDmatcher.SearchByProjection = _lsd_search_by_projection


class FrameGlue:
    """== the part of Frame::ExtractLSD after the extractor (src/Frame.cc:490-660): isLineGood (3-D RANSAC per keyline),
    convertFansToKeyLines (3-D crossing of paired lines) and the plane-from-pair loop.  `seed`: srand(seed) of the frame."""

    def __init__(self, max_lines=1024, max_fans=4096, max_batch=1, ctx=None):
        self.ctx = ctx or default_context()
        self.max_lines, self.max_fans, self.max_batch = max_lines, max_fans, max_batch
        self._h = C.c_void_p()
        _check(lib().pslfe_glue_create(self.ctx._h, C.c_int(max_lines), C.c_int(max_fans), C.c_int(max_batch), C.byref(self._h)),
               "pslfe_glue_create")

    def run(self, keylines, fans, depth, cam, seed=1):
        kls = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
        fans = np.ascontiguousarray(fans, np.float32).reshape(-1, 4)
        depth = np.ascontiguousarray(depth, np.float32)
        cam = np.ascontiguousarray(cam, CAMERA_DTYPE).reshape(1)
        self._n = len(kls)
        _check(lib().pslfe_glue_run(self._h, _ptr(kls), C.c_int(len(kls)), _ptr(fans), C.c_int(len(fans)), _ptr(depth),
                                    C.c_int(depth.shape[1]), C.c_int(depth.shape[0]), C.c_int(depth.shape[1]), _ptr(cam),
                                    C.c_uint32(seed)), "pslfe_glue_run")
        return self.fetch(0, self._n)

    def run_batch_device(self, nframes, d_kls, kl_stride, d_nkl, d_fans, fan_stride, d_nfans, d_depth, width, height, cam, seed0=1):
        cam = np.ascontiguousarray(cam, CAMERA_DTYPE).reshape(1)
        _check(lib().pslfe_glue_run_batch_device(self._h, C.c_int(nframes), C.c_void_p(int(d_kls)), C.c_int(kl_stride),
                                                 C.c_void_p(int(d_nkl)), C.c_void_p(int(d_fans)), C.c_int(fan_stride),
                                                 C.c_void_p(int(d_nfans)), C.c_void_p(int(d_depth)), C.c_int(width), C.c_int(height),
                                                 _ptr(cam), C.c_uint32(seed0)), "pslfe_glue_run_batch_device")

    def fetch(self, frame, nlines):
        """dict with mvLines3D, mvLineEq, the crossings (pair, xy, cross, le_l) and the planes."""
        I = P = self.max_fans
        out = dict(lines3d=np.zeros((nlines, 6), np.float64), lineEq=np.zeros((nlines, 3), np.float32),
                   pair=np.zeros((I, 2), np.int32), xy=np.zeros((I, 2), np.float32), cross=np.zeros((I, 3), np.float64),
                   le_l=np.zeros((I, 6), np.float64), planes=np.zeros((P, 4), np.float32), normals=np.zeros((P, 3), np.float64),
                   lineNo=np.zeros((P, 2), np.int32), cross3d=np.zeros((P, 3), np.float64), cross2d=np.zeros((P, 2), np.float64))
        ni, npl = C.c_int(), C.c_int()
        _check(lib().pslfe_glue_fetch(self._h, C.c_int(frame), C.c_int(nlines), _ptr(out["lines3d"]), _ptr(out["lineEq"]), _ptr(out["pair"]),
                                      _ptr(out["xy"]), _ptr(out["cross"]), _ptr(out["le_l"]), C.c_int(I), C.byref(ni), _ptr(out["planes"]),
                                      _ptr(out["normals"]), _ptr(out["lineNo"]), _ptr(out["cross3d"]), _ptr(out["cross2d"]), C.c_int(P),
                                      C.byref(npl)), "pslfe_glue_fetch")
        for k in ("pair", "xy", "cross", "le_l"):
            out[k] = out[k][:ni.value]
        for k in ("planes", "normals", "lineNo", "cross3d", "cross2d"):
            out[k] = out[k][:npl.value]
        return out

    def close(self):
        if self._h:
            lib().pslfe_glue_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBVocabulary:
    """== DBoW2 ORBVocabulary as far as Frame::ComputeBoW needs it: transform(descriptors, BowVector, FeatureVector, levelsup).
    Built from flat arrays (the host keeps parsing ORBvoc.txt): children[i] = list of child node ids of node i (node 0 = root),
    node_desc (nnodes x 32 u8), node_weight (f64), node_word (i32), L = depth."""

    def __init__(self, children, node_desc, node_weight, node_word, L, ctx=None):
        self.ctx = ctx or default_context()
        nn = len(children)
        cb = np.zeros(nn, np.int32)
        cc = np.array([len(c) for c in children], np.int32)
        cb[1:] = np.cumsum(cc)[:-1]
        ids = np.array([x for c in children for x in c], np.int32)
        self.arrays = (cb, cc, ids, np.ascontiguousarray(node_desc, np.uint8), np.ascontiguousarray(node_weight, np.float64),
                       np.ascontiguousarray(node_word, np.int32), int(L))
        self._h = C.c_void_p()
        _check(lib().pslfe_vocab_create(self.ctx._h, C.c_int(nn), _ptr(cb), _ptr(cc), _ptr(ids), C.c_int(len(ids)), _ptr(self.arrays[3]),
                                        _ptr(self.arrays[4]), _ptr(self.arrays[5]), C.c_int(L), C.byref(self._h)), "pslfe_vocab_create")

    def transform(self, desc, levelsup=4):
        """-> dict(word, weight, nid per feature; bow_id, bow_val; fv_node, fv_start, fv_idx)"""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        o = dict(word=np.zeros(m, np.int32), weight=np.zeros(m, np.float64), nid=np.zeros(m, np.int32), bow_id=np.zeros(m, np.int32),
                 bow_val=np.zeros(m, np.float64), fv_node=np.zeros(m, np.int32), fv_start=np.zeros(m + 1, np.int32), fv_idx=np.zeros(m, np.int32))
        nb, nf = C.c_int(), C.c_int()
        _check(lib().pslfe_compute_bow(self._h, _ptr(desc), C.c_int(n), C.c_int(levelsup), _ptr(o["word"]), _ptr(o["weight"]), _ptr(o["nid"]),
                                       _ptr(o["bow_id"]), _ptr(o["bow_val"]), C.byref(nb), _ptr(o["fv_node"]), _ptr(o["fv_start"]),
                                       _ptr(o["fv_idx"]), C.byref(nf)), "pslfe_compute_bow")
        for k in ("word", "weight", "nid"):
            o[k] = o[k][:n]
        o["bow_id"], o["bow_val"] = o["bow_id"][:nb.value], o["bow_val"][:nb.value]
        o["fv_node"], o["fv_start"] = o["fv_node"][:nf.value], o["fv_start"][:nf.value + 1]
        o["fv_idx"] = o["fv_idx"][:int(o["fv_start"][nf.value])] if nf.value else o["fv_idx"][:0]
        return o

    def close(self):
        if self._h:
            lib().pslfe_vocab_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TRIQUERY_DTYPE = np.dtype([("start", "<i4"), ("len", "<i4"), ("x", "<f4"), ("y", "<f4"), ("angle", "<f4"), ("stereo", "<i4")])
LINEFUSEQUERY_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("radius", "<f4"), ("level", "<i4")])
assert TRIQUERY_DTYPE.itemsize == 24 and LINEFUSEQUERY_DTYPE.itemsize == 24


class KeyFrameMatcher:
    """The KeyFrame-rate searches of LocalMapping / LoopClosing (pslfe_kf): ORBmatcher::Fuse (both), SearchBySim3,
    SearchForTriangulation (src/ORBmatcher.cc:657-1326), LSDmatcher::Fuse / SearchForTriangulation
    (add_src/LSDmatcher.cpp:705-984) and Map{Point,Line}::ComputeDistinctiveDescriptors, from the point where the host has
    projected its map points.  Give each host thread its own Context + KeyFrameMatcher."""

    TH_HIGH, TH_LOW = 100, 50

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        _check(lib().pslfe_kf_create(self.ctx._h, C.byref(self._h)), "pslfe_kf_create")

    def window_best(self, frame, slot, queries, qdesc, chi2=False, inv_level_sigma2=None):
        """Candidate loop of Fuse / SearchBySim3 -> (best_idx, best_dist) per projected map point."""
        q = np.ascontiguousarray(queries, PROJQUERY_DTYPE)
        qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        nq = len(q)
        bi = np.full(max(nq, 1), -1, np.int32)
        bd = np.full(max(nq, 1), 0x7fffffff, np.int32)
        s2 = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        _check(lib().pslfe_kf_window_best(self._h, frame._h, C.c_int(slot), _ptr(q), _ptr(qd), C.c_int(nq), C.c_int(1 if chi2 else 0),
                                          _ptr(s2), C.c_int(0 if s2 is None else len(s2)), _ptr(bi), _ptr(bd)), "pslfe_kf_window_best")
        return bi[:nq], bd[:nq]

    def Fuse(self, frame, slot, queries, qdesc, inv_level_sigma2):
        """ORBmatcher::Fuse(pKF, vpMapPoints, th) src/ORBmatcher.cc:825: -> (bestIdx, fused) with fused = bestDist <= TH_LOW;
        the caller replaces / adds map points (:950-964)."""
        bi, bd = self.window_best(frame, slot, queries, qdesc, True, inv_level_sigma2)
        return bi, bd <= self.TH_LOW

    def FuseSim3(self, frame, slot, queries, qdesc):
        """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) src/ORBmatcher.cc:968."""
        bi, bd = self.window_best(frame, slot, queries, qdesc, False)
        return bi, bd <= self.TH_LOW

    def SearchBySim3(self, frame1, slot1, frame2, slot2, q12, qdesc1, q21, qdesc2):
        """src/ORBmatcher.cc:1102 -> (nFound, match12)."""
        q1 = np.ascontiguousarray(q12, PROJQUERY_DTYPE)
        q2 = np.ascontiguousarray(q21, PROJQUERY_DTYPE)
        d1 = np.ascontiguousarray(qdesc1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(qdesc2, np.uint8).reshape(-1, 32)
        m = np.full(max(len(q1), 1), -1, np.int32)
        nf = C.c_int()
        _check(lib().pslfe_kf_search_by_sim3(self._h, frame1._h, C.c_int(slot1), frame2._h, C.c_int(slot2), _ptr(q1), _ptr(d1),
                                             C.c_int(len(q1)), _ptr(q2), _ptr(d2), C.c_int(len(q2)), _ptr(m), C.byref(nf)),
               "pslfe_kf_search_by_sim3")
        return nf.value, m[:len(q1)]

    def SearchForTriangulation(self, frame2, slot2, fidx2, taken2, queries, qdesc, F12, epipole, scale_factors, level_sigma2,
                               bOnlyStereo=False, checkOri=True):
        """src/ORBmatcher.cc:657 -> (nmatches, match per query); vMatchedPairs = the matched (idx1, idx2) sorted by idx1."""
        fidx = np.ascontiguousarray(fidx2, np.int32)
        tk = np.ascontiguousarray(taken2, np.uint8)
        q = np.ascontiguousarray(queries, TRIQUERY_DTYPE)
        qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        s2 = np.ascontiguousarray(level_sigma2, np.float32)
        match = np.full(max(len(q), 1), -1, np.int32)
        nm = C.c_int()
        _check(lib().pslfe_kf_search_for_triangulation(self._h, frame2._h, C.c_int(slot2), _ptr(fidx), C.c_int(len(fidx)), _ptr(tk), _ptr(q),
                                                       _ptr(qd), C.c_int(len(q)), _ptr(F), C.c_float(epipole[0]), C.c_float(epipole[1]),
                                                       C.c_int(1 if bOnlyStereo else 0), C.c_int(1 if checkOri else 0), _ptr(sf), _ptr(s2),
                                                       C.c_int(len(sf)), _ptr(match), C.byref(nm)), "pslfe_kf_search_for_triangulation")
        return nm.value, match[:len(q)]

    def LineFuse(self, keylines, desc, queries, qdesc):
        """Search of LSDmatcher::Fuse add_src/LSDmatcher.cpp:933-958 -> (bestIdx, bestDist)."""
        kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        q = np.ascontiguousarray(queries, LINEFUSEQUERY_DTYPE)
        qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        bi = np.full(max(len(q), 1), -1, np.int32)
        bd = np.full(max(len(q), 1), 256, np.int32)
        _check(lib().pslfe_kf_line_fuse_best(self._h, _ptr(kl), C.c_int(len(kl)), _ptr(d), C.c_int(len(d)), _ptr(q), _ptr(qd), C.c_int(len(q)),
                                             _ptr(bi), _ptr(bd)), "pslfe_kf_line_fuse_best")
        return bi[:len(q)], bd[:len(q)]

    def ComputeDistinctiveDescriptors(self, desc, offsets):
        """src/MapPoint.cc:242-304 for many map points / lines at once -> best row per point (relative to its run)."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(offsets, np.int32)
        npts = len(off) - 1
        best = np.full(max(npts, 1), -1, np.int32)
        _check(lib().pslfe_kf_distinctive_descriptors(self._h, _ptr(d), _ptr(off), C.c_int(npts), _ptr(best)),
               "pslfe_kf_distinctive_descriptors")
        return best[:npts]

    def close(self):
        if self._h:
            lib().pslfe_kf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _lsd_search_for_triangulation(self, ldesc1, ldesc2, has_mapline1, has_mapline2, TH=None, isDouble=True):
    """LSDmatcher::SearchForTriangulation add_src/LSDmatcher.cpp:705-781: FrameBFMatch both ways, mutual check, lines that
    already have a MapLine dropped.  TH = TH_LOW with the pair-vector overload (:705), TH_HIGH with the vector<int> one (:744).
    -> (nmatches, vMatchedPairs as an int array: index in KF2 or -1)."""
    TH = self.TH_LOW if TH is None else TH
    d1 = np.ascontiguousarray(ldesc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(ldesc2, np.uint8).reshape(-1, 32)
    out = np.full(len(d1), -1, np.int32)
    if len(d1) == 0 or len(d2) == 0:
        return 0, out
    m12 = self.FrameBFMatch(d1, d2, TH)
    m21 = self.FrameBFMatch(d2, d1, TH)
    n = 0
    for i, j in enumerate(m12):
        if j < 0 or (isDouble and m21[j] != i) or has_mapline1[i] or has_mapline2[j]:
            continue
        out[i] = j
        n += 1
    return n, out


LSDmatcher.SearchForTriangulation = _lsd_search_for_triangulation
