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
