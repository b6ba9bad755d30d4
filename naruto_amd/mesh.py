"""N4 ("next" row of SURVEY.md section 8f): the dense volume -> mesh path of the mapper on the device.

Mirror of ``extract_mesh`` (reference src/slam/coslam/coslam_utils.py:100-226; callers ``save_mesh`` /
``save_uncert_mesh``, coslam.py:421-492): same arguments, same lattice, same vertex transforms, same two colour
branches the shipped configs use (``mesh.render_color: False`` -> ``query_color`` at the vertices; ``color_func=None``
with ``render_uncert`` -> the uncertainty at the vertices through matplotlib's ``jet``).

What is different:
  * one query over the whole lattice (the lattice is expanded on the device from its three axis vectors, the SDF volume
    never leaves HBM) instead of 65 536-point chunks with a copy each;
  * marching cubes runs on the device (``naruto_mesh_count`` / ``naruto_mesh_emit``).  The reference uses the third-party
    ``marching_cubes`` module (coslam_utils.py:26,145), which is not available; same algorithm (one vertex per crossed
    lattice edge, linear interpolation in float64, cells beyond ``truncation`` skipped), this repo's own vertex /
    triangle order and case table (tools/gen_mc_table.py) -- the surface is the same, the index order is not;
  * the result is a plain :class:`Mesh` (vertices float64 [V,3], faces int64 [F,3], vertex_colors uint8 [V,4] like
    trimesh's) with a binary-PLY ``export``; ``trimesh`` is not a dependency.  ``config['mesh']['render_color'] = True``
    (no shipped config sets it) colours the vertices through ``render_surface_color`` along trimesh-style angle-weighted
    vertex normals (:func:`vertex_normals`).
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def get_voxels(x_max, x_min, y_max, y_min, z_max, z_min, voxel_size=None, resolution=None):
    """Co-SLAM utils.getVoxels (third_parties/coslam/utils.py; call sites coslam_utils.py:78,124)."""
    x_max, x_min, y_max, y_min, z_max, z_min = (float(v) for v in (x_max, x_min, y_max, y_min, z_max, z_min))
    if voxel_size is not None:
        nx = round((x_max - x_min) / voxel_size + 0.0005)
        ny = round((y_max - y_min) / voxel_size + 0.0005)
        nz = round((z_max - z_min) / voxel_size + 0.0005)
        return torch.linspace(x_min, x_max, nx + 1), torch.linspace(y_min, y_max, ny + 1), torch.linspace(z_min, z_max, nz + 1)
    return torch.linspace(x_min, x_max, resolution), torch.linspace(y_min, y_max, resolution), torch.linspace(z_min, z_max, resolution)


def jet_lut() -> torch.Tensor:
    """matplotlib's 'jet' as its 256-entry lookup table (piecewise-linear segments of matplotlib/_cm.py)."""
    seg = {
        0: ([0.00, 0.35, 0.66, 0.89, 1.00], [0.0, 0.0, 1.0, 1.0, 0.5]),
        1: ([0.000, 0.125, 0.375, 0.640, 0.910, 1.000], [0.0, 0.0, 1.0, 1.0, 0.0, 0.0]),
        2: ([0.00, 0.11, 0.34, 0.65, 1.00], [0.5, 1.0, 1.0, 0.0, 0.0]),
    }
    xind = 255.0 * np.linspace(0.0, 1.0, 256)
    cols = []
    for c in range(3):                                          # matplotlib.colors._create_lookup_table, continuous segments
        x, y = np.asarray(seg[c][0]) * 255.0, np.asarray(seg[c][1])
        ind = np.searchsorted(x, xind)[1:-1]
        distance = (xind[1:-1] - x[ind - 1]) / (x[ind] - x[ind - 1])
        cols.append(np.clip(np.concatenate([[y[0]], distance * (y[ind] - y[ind - 1]) + y[ind - 1], [y[-1]]]), 0.0, 1.0))
    return torch.from_numpy(np.stack(cols, -1))


@dataclass
class Mesh:
    vertices: np.ndarray                 # float64 [V,3], metric world coordinates
    faces: np.ndarray                    # int64 [F,3]
    vertex_colors: Optional[np.ndarray] = None       # uint8 [V,4] (trimesh's representation of float colours in [0,1])

    def export(self, path: str) -> None:
        """binary little-endian PLY: x y z float32 (+ red green blue alpha uchar), faces as uchar-counted int32 lists."""
        v = np.asarray(self.vertices, dtype=np.float32)
        head = ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}", "property float x", "property float y", "property float z"]
        fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
        if self.vertex_colors is not None:
            head += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
            fields += [("r", "u1"), ("g", "u1"), ("b", "u1"), ("a", "u1")]
        head += [f"element face {len(self.faces)}", "property list uchar int vertex_indices", "end_header"]
        vert = np.empty(len(v), dtype=fields)
        vert["x"], vert["y"], vert["z"] = v[:, 0], v[:, 1], v[:, 2]
        if self.vertex_colors is not None:
            for i, name in enumerate("rgba"):
                vert[name] = self.vertex_colors[:, i]
        face = np.empty(len(self.faces), dtype=[("n", "u1"), ("i", "<i4", (3,))])
        face["n"] = 3
        face["i"] = self.faces
        with open(path, "wb") as f:
            f.write(("\n".join(head) + "\n").encode("ascii"))
            f.write(vert.tobytes())
            f.write(face.tobytes())


def vertex_normals(vertices: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """trimesh's ``Trimesh.vertex_normals`` (third-party, unpinned; used by the reference at coslam_utils.py:180-181) restated:
    unit face normals summed into their vertices with the triangle's corner ANGLE as weight (degenerate triangles -- a corner angle
    below 1e-8 or a zero-area normal -- contribute nothing), then normalised; vertices without a contribution get a zero normal.
    vertices [V,3] float64, faces [F,3] int64 -> [V,3] float64, on the tensors' device."""
    v = vertices.to(torch.float64)
    f = faces.to(torch.int64)
    tri = v[f]                                                   # [F,3,3]

    def unit(a):
        n = torch.linalg.norm(a, dim=-1, keepdim=True)
        ok = n > 2.220446049250313e-14                           # trimesh.util.unitize's tolerance (100 * float64 eps)
        return torch.where(ok, a / torch.where(ok, n, torch.ones_like(n)), torch.zeros_like(a)), ok[..., 0]

    e01, e02, e12 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 2] - tri[:, 1]
    fn, fn_ok = unit(torch.cross(e01, e02, dim=-1))
    u, _ = unit(e01)
    w2, _ = unit(e02)
    w, _ = unit(e12)
    a0 = torch.arccos(torch.clamp((u * w2).sum(-1), -1.0, 1.0))
    a1 = torch.arccos(torch.clamp((-u * w).sum(-1), -1.0, 1.0))
    ang = torch.stack([a0, a1, np.pi - a0 - a1], dim=-1)         # [F,3]
    ang = torch.where(((ang < 1e-8).any(-1) | ~fn_ok)[:, None], torch.zeros_like(ang), ang)
    out = torch.zeros_like(v)
    for k in range(3):
        out.index_add_(0, f[:, k], fn * ang[:, k:k + 1])
    return unit(out)[0]


def _float_colors_to_rgba8(color) -> np.ndarray:
    """trimesh.visual.color.to_rgba for float input in [0,1]: round(c * 255) (half to even), opaque alpha.
    Device tensors are converted on the device: only the bytes cross PCIe."""
    c = torch.as_tensor(color)
    rgba = torch.full((c.shape[0], 4), 255, dtype=torch.uint8, device=c.device)
    rgba[:, :c.shape[1]] = torch.round(c.to(torch.float64).clamp(0.0, 1.0) * 255.0).to(torch.uint8)
    return rgba.cpu().numpy()


def lattice_points(tx: torch.Tensor, ty: torch.Tensor, tz: torch.Tensor) -> torch.Tensor:
    """[X*Y*Z, 3] = stack(meshgrid(tx, ty, tz, indexing='ij')) flattened, expanded on the device."""
    lib = _lib.load()
    dims = (C.c_uint32 * 3)(tx.numel(), ty.numel(), tz.numel())
    x = torch.empty(tx.numel() * ty.numel() * tz.numel(), 3, dtype=torch.float32, device=tx.device)
    with torch.cuda.device(tx.device):
        check(lib.naruto_lattice_points(dims, tx.data_ptr(), ty.data_ptr(), tz.data_ptr(), x.data_ptr(), _stream()), "naruto_lattice_points")
    return x


def marching_cubes(sdf_vol: torch.Tensor, isolevel: float = 0.0, truncation: float = 3.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """sdf_vol float32 [X,Y,Z] on the device -> (vertices float64 [V,3] in lattice-index coordinates, triangles int32 [F,3])."""
    if sdf_vol.dim() != 3 or sdf_vol.dtype != torch.float32 or not sdf_vol.is_cuda:
        raise ValueError("marching_cubes: need a float32 [X,Y,Z] volume on the GPU")
    lib = _lib.load()
    vol = sdf_vol.contiguous()
    dims = (C.c_uint32 * 3)(*vol.shape)
    ws_bytes = lib.naruto_mesh_workspace(dims)
    if ws_bytes == 0:
        raise ValueError(f"marching_cubes: unsupported volume shape {tuple(vol.shape)}: " + lib.naruto_last_error().decode("utf-8", "replace"))
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=vol.device)
    counts = torch.zeros(2, dtype=torch.int64, device=vol.device)
    with torch.cuda.device(vol.device):
        check(lib.naruto_mesh_count(dims, vol.data_ptr(), float(isolevel), float(truncation), ws.data_ptr(), counts.data_ptr(), _stream()), "naruto_mesh_count")
        n_v, n_f = (int(c) for c in counts.cpu())                                  # the one host sync: output sizes
        vertices = torch.empty(n_v, 3, dtype=torch.float64, device=vol.device)
        triangles = torch.empty(n_f, 3, dtype=torch.int32, device=vol.device)
        check(lib.naruto_mesh_emit(dims, vol.data_ptr(), float(isolevel), ws.data_ptr(), n_v, n_f, vertices.data_ptr(), triangles.data_ptr(), _stream()),
              "naruto_mesh_emit")
    return vertices, triangles


@torch.no_grad()
def extract_mesh(query_fn: Callable, config, bounding_box: torch.Tensor, marching_cube_bound=None, color_func: Optional[Callable] = None, voxel_size=None,
                 resolution=None, isolevel: float = 0.0, scene_name: str = "", mesh_savepath: str = "", render_uncert: bool = True) -> Mesh:
    """coslam_utils.py:100-226.  ``query_fn`` = ``model.query_sdf``, ``color_func`` = ``model.query_color`` or None."""
    device = bounding_box.device
    if marching_cube_bound is None:
        marching_cube_bound = bounding_box
    mcb = torch.as_tensor(marching_cube_bound)
    x_min, y_min, z_min = mcb[:, 0]
    x_max, y_max, z_max = mcb[:, 1]
    tx, ty, tz = get_voxels(x_max, x_min, y_max, y_min, z_max, z_min, voxel_size, resolution)            # float32, host (tiny)

    # the lattice is separable: normalise the three axis vectors exactly as the reference normalises every point (:131-133)
    bb_cpu = bounding_box.detach().cpu()
    axes = [tx, ty, tz]
    if config["grid"]["tcnn_encoding"]:
        axes = [(t - bb_cpu[i, 0]) / (bb_cpu[i, 1] - bb_cpu[i, 0]) for i, t in enumerate(axes)]
    axes = [t.to(torch.float32).to(device).contiguous() for t in axes]
    flat = lattice_points(*axes)
    sdf = query_fn(flat[:, None, :])                                                                         # [N,1] (:135-138)
    vol = torch.reshape(sdf, (tx.numel(), ty.numel(), tz.numel())).to(torch.float32)
    del flat

    verts_idx, triangles = marching_cubes(vol, isolevel, truncation=3.0)                                    # :145

    # :148-162 in float64 with numpy's promotion of the float32 axis values
    n_axis = torch.tensor([tx.numel() - 1, ty.numel() - 1, tz.numel() - 1], dtype=torch.float64, device=device)
    scale = torch.stack([tx[-1] - tx[0], ty[-1] - ty[0], tz[-1] - tz[0]]).to(torch.float64).to(device)        # float32 differences, then widened
    offset = torch.stack([tx[0], ty[0], tz[0]]).to(torch.float64).to(device)
    vertices = verts_idx / n_axis
    vertices = scale[None, :] * vertices + offset
    vertices = vertices / config["data"]["sc_factor"] - config["data"]["translation"]

    colors = None
    if color_func is not None or render_uncert:
        vert_flat = vertices.to(bounding_box.dtype)
        if config["grid"]["tcnn_encoding"]:
            vert_flat = (vert_flat - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])         # :165-166 / :198-199
    if color_func is not None and not config["mesh"]["render_color"]:
        color = color_func(vert_flat[:, None, :]) if len(vert_flat) else torch.zeros(0, 3, device=device)    # :169-176
        colors = _float_colors_to_rgba8(torch.reshape(color, (len(vert_flat), -1)))
    elif color_func is not None:
        # :178-186: colour rendered along the vertex normals; the reference hands over the METRIC vertices (not vert_flat)
        if len(vertices):
            normals = vertex_normals(vertices, triangles)
            color = color_func(vertices.to(bounding_box.dtype), normals.to(bounding_box.dtype))
        else:
            color = torch.zeros(0, 3, device=device)
        colors = _float_colors_to_rgba8(torch.reshape(color, (len(vertices), -1)))
    elif render_uncert:
        if len(vert_flat):
            raw_uncert = query_fn(vert_flat[:, None, :], return_uncert=True)[:, 0, 1].to(torch.float32)     # :202-206
            un = (raw_uncert - raw_uncert.min()) / (raw_uncert.max() - raw_uncert.min())                     # :210
            x = un * 256.0                                                                                    # Colormap.__call__
            idx = torch.where(torch.isnan(x), torch.zeros_like(x), x).clamp(0.0, 255.0).to(torch.int64)
            rgb = jet_lut().to(device)[idx]
            rgb = torch.where(torch.isnan(x)[:, None], torch.zeros_like(rgb), rgb)
            colors = _float_colors_to_rgba8(rgb)
        else:
            colors = np.zeros((0, 4), dtype=np.uint8)

    mesh = Mesh(vertices.cpu().numpy(), triangles.to(torch.int64).cpu().numpy(), colors)
    if mesh_savepath:
        os.makedirs(os.path.split(mesh_savepath)[0] or ".", exist_ok=True)
        mesh.export(mesh_savepath)
    return mesh
