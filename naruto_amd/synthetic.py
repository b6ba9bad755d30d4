"""Synthetic, seed-reproducible workloads shaped like the reference's.

There is no dataset or simulator on the benchmark box, so rays / targets / parameters are generated
here with numpy ``RandomState`` (identical on every host) or in closed form.  Used by bench.py, by
the parity tests and by the golden-vector generator.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np


def closed_form_table(n_params: int, amp: float = 1e-4, freq: float = 0.37) -> np.ndarray:
    """Hash-table values ``amp * sin(freq * i)`` -- lets fixtures avoid storing multi-MB tables."""
    i = np.arange(n_params, dtype=np.float64)
    return (amp * np.sin(freq * i)).astype(np.float32)


def closed_form_uncert_grid(dims: Sequence[int]) -> np.ndarray:
    """Smooth, non-constant, strictly positive-ish grid so that the x<->z axis quirk is observable."""
    nx, ny, nz = dims
    i = np.arange(nx, dtype=np.float64)[:, None, None]
    j = np.arange(ny, dtype=np.float64)[None, :, None]
    k = np.arange(nz, dtype=np.float64)[None, None, :]
    g = 3.0 + 0.8 * np.sin(0.31 * i + 0.1) * np.cos(0.23 * j) + 0.5 * np.sin(0.17 * k + 0.05 * i)
    return g.astype(np.float32)


def mlp_weights(seed: int, in_sdf: int = 80, hidden: int = 32, geo: int = 15, in_col: int = 63,
                hidden_col: int = 32) -> Dict[str, np.ndarray]:
    """nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights, numpy-seeded."""
    rs = np.random.RandomState(seed)

    def lin(o, i):
        b = 1.0 / np.sqrt(i)
        return rs.uniform(-b, b, size=(o, i)).astype(np.float32)

    return {"sdf_w0": lin(hidden, in_sdf), "sdf_w1": lin(1 + geo, hidden),
            "col_w0": lin(hidden_col, in_col), "col_w1": lin(3, hidden_col)}


def random_rays(n_rays: int, bound, seed: int = 0, zero_depth_frac: float = 0.05,
                depth_range=(0.5, 2.5)) -> Dict[str, np.ndarray]:
    """rays_o ~ U(bbox shrunk 20 %), rays_d ~ uniform on the sphere, target_d ~ U(depth_range) with a
    fraction of invalid (zero) depths, target_rgb ~ U(0,1)."""
    rs = np.random.RandomState(seed)
    bound = np.asarray(bound, dtype=np.float64)
    lo, hi = bound[:, 0], bound[:, 1]
    c, h = 0.5 * (lo + hi), 0.5 * (hi - lo) * 0.8
    rays_o = rs.uniform(-1, 1, size=(n_rays, 3)) * h + c
    d = rs.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    target_d = rs.uniform(depth_range[0], depth_range[1], size=(n_rays, 1))
    target_d[rs.uniform(size=n_rays) < zero_depth_frac] = 0.0
    target_rgb = rs.uniform(0, 1, size=(n_rays, 3))
    return {"rays_o": rays_o.astype(np.float32), "rays_d": d.astype(np.float32),
            "target_d": target_d.astype(np.float32), "target_rgb": target_rgb.astype(np.float32)}


def pinhole_rays(H: int, W: int, fx: float, fy: float, bound, seed: int = 0) -> Dict[str, np.ndarray]:
    """Camera at the bbox centre, H x W pinhole fan rotated by a fixed rotation (BASELINE.json
    configs[0]: 64x64 rays)."""
    rs = np.random.RandomState(seed)
    bound = np.asarray(bound, dtype=np.float64)
    centre = bound.mean(axis=1)
    i, j = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    dirs = np.stack([(i - cx) / fx, -(j - cy) / fy, -np.ones_like(i)], -1).reshape(-1, 3)
    a, b = 0.4, -0.25
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    rot = rz @ rx
    rays_d = dirs @ rot.T
    n = H * W
    target_d = rs.uniform(0.5, 2.5, size=(n, 1))
    target_d[rs.uniform(size=n) < 0.05] = 0.0
    return {"rays_o": np.broadcast_to(centre, (n, 3)).astype(np.float32).copy(),
            "rays_d": rays_d.astype(np.float32), "target_d": target_d.astype(np.float32),
            "target_rgb": rs.uniform(0, 1, size=(n, 3)).astype(np.float32)}


def lattice_points(dims: Sequence[int]) -> np.ndarray:
    """Normalised [0,1]^3 lattice [X,Y,Z,3] including both faces (what get_map_volumes queries)."""
    axes = [np.linspace(0.0, 1.0, n, dtype=np.float64) for n in dims]
    g = np.stack(np.meshgrid(*axes, indexing="ij"), -1)
    return g.astype(np.float32)


class AnalyticRoom:
    """A closed-form scene inside a bounding box for end-to-end accuracy experiments (tests/accuracy_study.py): an axis-aligned
    box room (the volume's box shrunk by ``wall_margin``) with a sphere in it, seen by pinhole cameras on a ring around the sphere.
    Ray casting, surface colour and the true distance field are analytic, so RGB-D frames from any number of poses are mutually
    consistent -- what the random rays / random targets of the parity tests are not.  Directions are unit vectors and depths are
    distances along the ray (the field is trained on o + t d either way)."""

    def __init__(self, bound, wall_margin: float = 0.45, sphere_radius: float = 0.6, max_depth: float = 4.8):
        b = np.asarray(bound, dtype=np.float64)
        self.lo, self.hi = b[:, 0] + wall_margin, b[:, 1] - wall_margin
        self.centre = 0.5 * (self.lo + self.hi) + np.array([0.25, -0.35, -0.15])
        self.radius = float(sphere_radius)
        self.max_depth = float(max_depth)

    # ---- geometry -------------------------------------------------------------------------------------------------
    def cast(self, o: np.ndarray, d: np.ndarray):
        """First hit of rays o + t d (o inside the room, |d| = 1): distance t [N], hit point [N,3], kind [N] (0..5 wall faces
        -x,+x,-y,+y,-z,+z; 6 sphere)."""
        o, d = np.asarray(o, np.float64), np.asarray(d, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            t_ax = np.where(d > 0, (self.hi - o) / d, np.where(d < 0, (self.lo - o) / d, np.inf))
        axis = np.argmin(t_ax, axis=1)
        t_box = t_ax[np.arange(len(o)), axis]
        kind = 2 * axis + (d[np.arange(len(o)), axis] > 0)
        oc = o - self.centre
        bq = (oc * d).sum(1)
        disc = bq * bq - ((oc * oc).sum(1) - self.radius ** 2)
        t_s = np.where(disc > 0, -bq - np.sqrt(np.maximum(disc, 0.0)), np.inf)
        t_s = np.where(t_s > 1e-6, t_s, np.inf)
        hit_s = t_s < t_box
        t = np.where(hit_s, t_s, t_box)
        kind = np.where(hit_s, 6, kind)
        return t, o + t[:, None] * d, kind

    def colour(self, p: np.ndarray, kind: np.ndarray) -> np.ndarray:
        """Smooth position-dependent albedo in [0.1, 0.9]; every wall face and the sphere have their own phase."""
        ph = kind[:, None].astype(np.float64) * np.array([0.9, 1.7, 2.3])
        f = np.stack([np.sin(1.9 * p[:, 0] + 0.7 * p[:, 1] + ph[:, 0]), np.sin(1.3 * p[:, 1] - 1.1 * p[:, 2] + ph[:, 1]),
                      np.sin(1.6 * p[:, 2] + 0.9 * p[:, 0] + ph[:, 2])], 1)
        return 0.5 + 0.4 * f

    def sdf(self, p: np.ndarray) -> np.ndarray:
        """True distance to the nearest surface for points in the room's free space (negative inside the sphere / behind a wall)."""
        p = np.asarray(p, np.float64)
        d_wall = np.minimum(p - self.lo, self.hi - p).min(1)
        d_sph = np.linalg.norm(p - self.centre, axis=1) - self.radius
        return np.minimum(d_wall, d_sph)

    # ---- cameras ---------------------------------------------------------------------------------------------------
    def pose(self, k: int, n: int):
        """Camera k of n on a ring around the sphere (radius 1.45 m, height varying), looking at the sphere's centre with a small
        offset: position [3] and rotation [3,3] (camera x right, y up, looking along -z)."""
        a = 2.0 * np.pi * (k + 0.37) / n
        pos = self.centre + np.array([1.45 * np.cos(a), 1.45 * np.sin(a), 0.55 * np.sin(2.3 * a + 0.4)])
        pos = np.minimum(np.maximum(pos, self.lo + 0.3), self.hi - 0.3)
        target = self.centre + 0.25 * np.array([np.sin(3.1 * a), np.cos(2.2 * a), np.sin(1.3 * a)])
        fwd = target - pos
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        return pos, np.stack([right, up, -fwd], 1)

    def rays(self, k: int, n: int, H: int = 60, W: int = 80, f: float = 60.0, jitter: Optional[np.random.RandomState] = None, count: Optional[int] = None):
        """The frame of camera k: one ray per pixel centre, or ``count`` rays at continuous random pixel positions (``jitter``).
        Returns dict rays_o, rays_d, target_rgb [.,3], target_d [.,1] (0 = no measurement: surface beyond ``max_depth``), hit [.,3]."""
        pos, R = self.pose(k, n)
        if jitter is None:
            u, v = np.meshgrid(np.arange(W, dtype=np.float64) + 0.5, np.arange(H, dtype=np.float64) + 0.5)
            u, v = u.reshape(-1), v.reshape(-1)
        else:
            u, v = jitter.uniform(0, W, count), jitter.uniform(0, H, count)
        dc = np.stack([(u - W / 2) / f, -(v - H / 2) / f, -np.ones_like(u)], 1)
        dw = dc @ R.T
        dw /= np.linalg.norm(dw, axis=1, keepdims=True)
        o = np.broadcast_to(pos, dw.shape).copy()
        t, hit, kind = self.cast(o, dw)
        depth = np.where(t <= self.max_depth, t, 0.0)
        return {"rays_o": o.astype(np.float32), "rays_d": dw.astype(np.float32), "target_rgb": self.colour(hit, kind).astype(np.float32),
                "target_d": depth[:, None].astype(np.float32), "hit": hit.astype(np.float32)}
