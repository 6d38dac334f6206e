"""Seeded synthetic inputs of the shapes SURVEY.md 8(d) names (no dataset ships with the reference:
config.json:3 points outside the repo).  Pure numpy; used by tests and bench.py for CPU and GPU alike.
"""
import numpy as np


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw), dtype=np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def world(seed, h, w, n_shapes=200):
    """Corner-rich u8 'scene': three octaves of value noise plus filled rectangles and discs."""
    rng = np.random.default_rng(seed)
    img = 0.5 * _value_noise(rng, h, w, 64) + 0.3 * _value_noise(rng, h, w, 16) + 0.2 * _value_noise(rng, h, w, 4)
    img = 40.0 + 150.0 * img
    yy, xx = np.mgrid[0:h, 0:w]
    n_shapes = int(n_shapes * (h * w) / (480 * 640) + 0.5)
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, w), rng.integers(0, h)
        val = float(rng.integers(0, 256))
        if rng.random() < 0.5:
            hw_, hh_ = rng.integers(4, 40), rng.integers(4, 40)
            x0, x1 = max(cx - hw_, 0), min(cx + hw_, w)
            y0, y1 = max(cy - hh_, 0), min(cy + hh_, h)
            img[y0:y1, x0:x1] = val
        else:
            r = int(rng.integers(4, 30))
            x0, x1 = max(cx - r, 0), min(cx + r + 1, w)
            y0, y1 = max(cy - r, 0), min(cy + r + 1, h)
            m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
            img[y0:y1, x0:x1][m] = val
    img += rng.normal(0.0, 2.0, size=img.shape).astype(np.float32)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8)


def _warp(src, h, w, tx, ty, rot_deg):
    """Bilinear sample of `src` (larger than h x w) under a small rotation about the centre + translation."""
    sh, sw = src.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    a = np.deg2rad(rot_deg)
    ca, sa = np.float32(np.cos(a)), np.float32(np.sin(a))
    xs = ca * (xx - cx) - sa * (yy - cy) + cx + (sw - w) / 2.0 + tx
    ys = sa * (xx - cx) + ca * (yy - cy) + cy + (sh - h) / 2.0 + ty
    xs = np.clip(xs, 0, sw - 1.001)
    ys = np.clip(ys, 0, sh - 1.001)
    x0 = xs.astype(np.int32)
    y0 = ys.astype(np.int32)
    fx = xs - x0
    fy = ys - y0
    s = src.astype(np.float32)
    v = (s[y0, x0] * (1 - fx) + s[y0, x0 + 1] * fx) * (1 - fy) + (s[y0 + 1, x0] * (1 - fx) + s[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(v + 0.5, 0, 255).astype(np.uint8)


def sequence(seed, n_frames, h=480, w=640):
    """'TUM-shaped' sequence: consecutive frames differ by <= +-3 px translation and <= 1 deg rotation."""
    rng = np.random.default_rng(seed)
    margin = 96
    scene = world(seed + 7919, h + 2 * margin, w + 2 * margin)
    frames = np.empty((n_frames, h, w), np.uint8)
    tx = ty = rot = 0.0
    for t in range(n_frames):
        frames[t] = _warp(scene, h, w, tx, ty, rot)
        tx = float(np.clip(tx + rng.uniform(-3, 3), -margin / 2, margin / 2))
        ty = float(np.clip(ty + rng.uniform(-3, 3), -margin / 2, margin / 2))
        rot = float(np.clip(rot + rng.uniform(-1, 1), -8, 8))
    return frames


def frame(seed, h=480, w=640):
    return world(seed, h, w)


class ProceduralSequence:
    """Frame `i` of an arbitrarily long 'TUM-shaped' sequence as a pure function of (seed, i): one seeded scene and a
    closed-form camera path (slow sinusoids: <= 2.2 px and <= 0.2 deg between consecutive frames), so any rank can produce
    any frame of its block without the frames before it and no dataset moves (SURVEY 8d input 5)."""

    def __init__(self, seed, h=480, w=640, margin=96):
        self.seed, self.h, self.w, self.margin = int(seed), int(h), int(w), int(margin)
        self.scene = world(self.seed + 7919, self.h + 2 * margin, self.w + 2 * margin)
        rng = np.random.default_rng(self.seed)
        self.phase = rng.uniform(0, 2 * np.pi, 3)

    def pose(self, i):
        a = self.margin * 0.45
        tx = a * np.sin(0.047 * i + self.phase[0])
        ty = a * np.sin(0.031 * i + self.phase[1])
        rot = 7.0 * np.sin(0.027 * i + self.phase[2])
        return float(tx), float(ty), float(rot)

    def frame(self, i):
        return _warp(self.scene, self.h, self.w, *self.pose(int(i)))


def descriptors(seed, n, planted_from=None, planted_frac=0.3, max_flips=40):
    """n x 32 uniform random descriptors; optionally a fraction are noisy copies of rows of `planted_from`."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if planted_from is not None and len(planted_from):
        k = int(n * planted_frac)
        rows = rng.choice(n, size=k, replace=False)
        src = rng.integers(0, len(planted_from), size=k)
        d[rows] = planted_from[src]
        for r in rows:
            nf = int(rng.integers(0, max_flips + 1))
            bits = rng.choice(256, size=nf, replace=False)
            for b in bits:
                d[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return d
