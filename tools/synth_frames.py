"""Synthetic 640x480 RGB-D streams standing in for the TUM / ICL-NUIM sequences BASELINE.json
names (none are available offline).  Test/bench infrastructure, numpy only.

Scene = random filled convex polygons (uniform grey in [30,225]: corners + long straight
edges) over a band-limited noise texture, viewed through a slowly drifting similarity/homography
so that frame-to-frame matching is meaningful.  Depth = tilted plane, z in [0.8, 4] m, x5000 (u16).

`style`: "desk"   polygons + noise sigma 6  (fr1_desk-like: many corners)
         "struct" polygons + noise sigma 2  (fr3_structure_notexture-like: few corners, lines); the painter's order hides most
                  edges: ~150 LSD segments and ~30 keylines of >= 50 px per frame with LSD_REFINE_ADV
         "sticks" structure scene at the line load the configuration names (200 lines): thin high-contrast bars (4 - 6 px wide,
                  58 - 80 px long, random orientation) placed WITHOUT overlap (2 px clearance), a 0.6 px point spread, noise
                  sigma 1: ~370 LSD segments and 160 - 185 keylines of >= 50 px per frame with LSD_REFINE_ADV, four corners
                  per bar for ORB (1000 keypoints)
"""
import numpy as np

SEED = 20250418


def orb_scale_factors(nlevels=8, scale_factor=1.2):
    """mvScaleFactor as ORBextractor::ORBextractor builds it (src/ORBextractor.cc:417-424): a vector<float> whose entry i is
    entry i-1 times the DOUBLE member scaleFactor (include/ORBextractor.h:98), itself the float argument 1.2f widened - the
    product is rounded to float at every level."""
    sf = np.float64(np.float32(scale_factor))
    s = [np.float32(1.0)]
    for _ in range(nlevels - 1):
        s.append(np.float32(np.float64(s[-1]) * sf))
    return np.array(s, np.float32)


def _smooth_noise(rng, h, w, sigma_px, amp):
    from scipy.ndimage import gaussian_filter
    n = rng.standard_normal((h, w)).astype(np.float32)
    n = gaussian_filter(n, sigma_px, mode="wrap")
    n *= amp / (n.std() + 1e-9)
    return n


class Scene:
    def __init__(self, w=640, h=480, style="desk", seed=SEED, n_poly=None):
        self.w, self.h, self.style = w, h, style
        rng = np.random.default_rng(seed)
        self.rng = rng
        scale = max(w / 640.0, h / 480.0)
        n_poly = n_poly or int(rng.integers(200, 401))
        # scene canvas is larger than the view so the drift never runs out of content
        self.cw, self.ch = int(w * 1.5), int(h * 1.5)
        polys = []
        if style == "sticks":
            polys = self._sticks(rng, scale)
            n_poly = 0
        for _ in range(n_poly):
            cx, cy = rng.uniform(0, self.cw), rng.uniform(0, self.ch)
            if rng.random() < 0.5:  # rectangle, random rotation
                a, b = rng.uniform(12, 90, 2) * scale
                th = rng.uniform(0, np.pi)
                c, s = np.cos(th), np.sin(th)
                pts = np.array([[-a, -b], [a, -b], [a, b], [-a, b]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
            else:  # convex polygon: sorted angles on an ellipse
                k = int(rng.integers(3, 7))
                ang = np.sort(rng.uniform(0, 2 * np.pi, k))
                ra, rb = rng.uniform(15, 80, 2) * scale
                pts = np.stack([cx + ra * np.cos(ang), cy + rb * np.sin(ang)], 1)
            polys.append((pts.astype(np.float64), float(rng.uniform(30, 225))))
        self.polys = polys
        sigma = {"desk": 6.0, "sticks": 1.0}.get(style, 2.0)
        self.noise = _smooth_noise(rng, self.ch, self.cw, 1.2, sigma)
        self.bg = float(rng.uniform(90, 160))
        self.tx, self.ty, self.rot = rng.uniform(-2, 2), rng.uniform(-2, 2), np.deg2rad(rng.uniform(-0.3, 0.3))

    def _sticks(self, rng, scale):
        """Non-overlapping thin bars over the canvas (rejection sampling against an occupancy mask; 2 px clearance)."""
        cw, ch = self.cw, self.ch
        occ = np.zeros((ch, cw), bool)
        want = int(400 * (cw * ch) / (640.0 * 480.0) / (scale * scale))
        polys, tries = [], 0
        while tries < want * 80 and len(polys) < want:
            tries += 1
            cx, cy = rng.uniform(0, cw), rng.uniform(0, ch)
            L, Wd = rng.uniform(58, 80) * scale / 2, rng.uniform(4, 6) * scale / 2
            th = rng.uniform(0, np.pi)
            c, s = np.cos(th), np.sin(th)
            r = int(np.ceil(L + Wd + 4 * scale))
            x0, y0, x1, y1 = max(int(cx) - r, 0), max(int(cy) - r, 0), min(int(cx) + r + 1, cw), min(int(cy) + r + 1, ch)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            u, v = (xx - cx) * c + (yy - cy) * s, -(xx - cx) * s + (yy - cy) * c
            if (occ[y0:y1, x0:x1] & (np.abs(u) < L + 2 * scale) & (np.abs(v) < Wd + 2 * scale)).any():
                continue
            occ[y0:y1, x0:x1] |= (np.abs(u) < L) & (np.abs(v) < Wd)
            pts = np.array([[-L, -Wd], [L, -Wd], [L, Wd], [-L, Wd]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
            g = rng.uniform(20, 70) if rng.random() < 0.5 else rng.uniform(190, 240)
            polys.append((pts.astype(np.float64), float(g)))
        return polys

    def _view_to_canvas(self, t):
        """3x3 map from view pixel coords to canvas coords at frame t (drift <=2 px, <=0.3 deg per frame)."""
        ang = self.rot * t
        c, s = np.cos(ang), np.sin(ang)
        cx, cy = self.w / 2.0, self.h / 2.0
        ox, oy = (self.cw - self.w) / 2.0 + self.tx * t, (self.ch - self.h) / 2.0 + self.ty * t
        A = np.array([[c, -s, cx - c * cx + s * cy + ox], [s, c, cy - s * cx - c * cy + oy], [0, 0, 1.0]])
        return A

    def canvas_to_view(self, t):
        return np.linalg.inv(self._view_to_canvas(t))

    def gray(self, t=0):
        w, h = self.w, self.h
        Ainv = self.canvas_to_view(t)
        img = np.full((h, w), self.bg, np.float32)
        for pts, g in self.polys:
            v = pts @ Ainv[:2, :2].T + Ainv[:2, 2]
            x0, y0 = np.floor(v.min(0)).astype(int)
            x1, y1 = np.ceil(v.max(0)).astype(int) + 1
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            inside = np.ones(xx.shape, bool)
            k = len(v)
            # orientation-agnostic convex test
            area = 0.0
            for i in range(k):
                p, q = v[i], v[(i + 1) % k]
                area += p[0] * q[1] - q[0] * p[1]
            sgn = 1.0 if area >= 0 else -1.0
            for i in range(k):
                p, q = v[i], v[(i + 1) % k]
                inside &= sgn * ((q[0] - p[0]) * (yy - p[1]) - (q[1] - p[1]) * (xx - p[0])) >= 0
            img[y0:y1, x0:x1][inside] = g
        # texture sampled from the canvas (nearest-neighbour is enough for a noise field)
        A = self._view_to_canvas(t)
        yy, xx = np.mgrid[0:h, 0:w]
        sx = np.clip(np.rint(A[0, 0] * xx + A[0, 1] * yy + A[0, 2]).astype(int), 0, self.cw - 1)
        sy = np.clip(np.rint(A[1, 0] * xx + A[1, 1] * yy + A[1, 2]).astype(int), 0, self.ch - 1)
        if self.style == "sticks":   # a camera's point spread: the rasteriser above is binary (LSD's NFA test rejects staircase edges)
            from scipy.ndimage import gaussian_filter
            img = gaussian_filter(img, 0.6)
        img += self.noise[sy, sx]
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def rgb(self, t=0):
        g = self.gray(t)
        return np.stack([g, g, g], -1)

    def depth_u16(self, t=0):
        yy, xx = np.mgrid[0:self.h, 0:self.w]
        z = 0.8 + 3.2 * (0.5 * xx / self.w + 0.5 * yy / self.h)
        return np.rint(z * 5000.0).astype(np.uint16)


def stream(n_frames, w=640, h=480, style="desk", seed=SEED):
    """(n_frames, h, w) uint8 gray frames of one drifting scene."""
    sc = Scene(w, h, style, seed)
    return np.stack([sc.gray(t) for t in range(n_frames)], 0)


def random_gray(w, h, seed, kind="noise"):
    """Small adversarial inputs for unit tests."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "flat":
        return np.full((h, w), 128, np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        return (((xx // 8 + yy // 8) & 1) * 200 + 20).astype(np.uint8)
    if kind == "blobs":
        from scipy.ndimage import gaussian_filter
        n = gaussian_filter(rng.standard_normal((h, w)), 2.0)
        n = (n - n.min()) / (n.max() - n.min())
        return (n * 255).astype(np.uint8)
    raise ValueError(kind)
