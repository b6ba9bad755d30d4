"""CPU restatement (numpy, float64) of the mesh path of NARUTO's mapper -- row N4 of SURVEY.md section 8(f).

TEST INFRASTRUCTURE ONLY.  Nothing under ``naruto_amd/`` may import this file; only ``tests/`` does, as the checker.

What is restated (paths under /root/reference):

* In tree: ``extract_mesh`` -- the lattice, the chunked ``query_sdf``, the vertex transforms, the vertex colour /
  uncertainty-colour branches: src/slam/coslam/coslam_utils.py:100-226 (callers: coslam.py:421-492).
* NOT in tree -- **parity unpinned**:
    - ``marching_cubes`` (``import marching_cubes as mcubes``, coslam_utils.py:26,145): the NumpyMarchingCubes
      extension that Co-SLAM @ 3bb904e vendors under external/ (from NeuralRGBD); absent here and not installable.
      Restated as the published algorithm: classic marching cubes on the cell lattice, bit c of the case set when
      corner value < isolevel, one vertex per crossed lattice edge placed by linear interpolation
      ``t = (iso - v0) / (v1 - v0)`` in float64, vertices shared between the cells around an edge, cells with a corner
      beyond ``truncation`` skipped.  The vertex SET and the surface are implementation independent; the ORDER of
      vertices / triangles and the triangulation of the ambiguous cases are not, so they follow this repo's own
      conventions (tools/gen_mc_table.py): vertices ordered by (owner voxel linear index, axis), triangles by
      (cell linear index, table order).
    - ``getVoxels`` / ``get_batch_query_fn`` (third_parties/coslam/utils.py, same Co-SLAM commit): restated in
      oracle/spec_torch.py (get_voxels) and inline below.
    - matplotlib's ``jet`` colormap (coslam_utils.py:211): matplotlib IS importable in the build container, the
      fixture tests/golden/g10_extract_mesh.npz holds its 256-entry lookup table (key jet_lut, oracle/make_golden.py).
"""

from __future__ import annotations

import numpy as np


def corner_offsets():
    return np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)


def edge_geometry():
    """edge id -> (axis, offset of the owner voxel relative to the cell's lower corner)."""
    out = []
    for e in range(12):
        a, q = e >> 2, e & 3
        u, w = [i for i in range(3) if i != a]
        off = [0, 0, 0]
        off[u], off[w] = q & 1, q >> 1
        out.append((a, tuple(off)))
    return out


def marching_cubes(vol: np.ndarray, isolevel: float, truncation: float, table: dict):
    """vol [X,Y,Z] -> (vertices float64 [V,3] in lattice-index coordinates, triangles int64 [F,3])."""
    vol = np.asarray(vol, dtype=np.float64)
    X, Y, Z = vol.shape
    n_tris, tris = table["n_tris"], table["tris"]
    inside = vol < isolevel
    beyond = np.abs(vol) > truncation
    case = np.zeros((X, Y, Z), dtype=np.int64)                 # per voxel; voxels that are no cell's lower corner stay 0
    skip = np.zeros((X - 1, Y - 1, Z - 1), dtype=bool)
    cell_case = np.zeros((X - 1, Y - 1, Z - 1), dtype=np.int64)
    for c, (dx, dy, dz) in enumerate(corner_offsets()):
        sl = (slice(dx, X - 1 + dx), slice(dy, Y - 1 + dy), slice(dz, Z - 1 + dz))
        cell_case |= inside[sl].astype(np.int64) << c
        skip |= beyond[sl]
    cell_case[skip] = 0
    case[:X - 1, :Y - 1, :Z - 1] = cell_case
    emits = (case != 0) & (case != 255)

    # a lattice edge owns a vertex when its end points differ and one of the (up to four) cells around it emits
    flags = np.zeros((X, Y, Z, 3), dtype=bool)
    dims = (X, Y, Z)
    for a in range(3):
        u, w = [i for i in range(3) if i != a]
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[a], hi[a] = slice(0, dims[a] - 1), slice(1, dims[a])
        change = inside[tuple(lo)] != inside[tuple(hi)]
        near = np.zeros_like(change)
        for du in (0, 1):
            for dw in (0, 1):
                shifted = np.zeros((X, Y, Z), dtype=bool)
                src = [slice(None)] * 3
                dst = [slice(None)] * 3
                src[u], dst[u] = (slice(0, dims[u] - du), slice(du, dims[u]))
                src[w], dst[w] = (slice(0, dims[w] - dw), slice(dw, dims[w]))
                shifted[tuple(dst)] = emits[tuple(src)]            # shifted[v] = emits[v - du e_u - dw e_w]
                near |= shifted[tuple(lo)]
        flags[tuple(lo) + (a,)] = change & near

    vert_id = np.cumsum(flags.reshape(-1)) - 1                     # order: (voxel linear index, axis)
    vert_id = vert_id.reshape(X, Y, Z, 3)
    owners = np.argwhere(flags)                                    # rows (i, j, k, a) in exactly that order
    verts = owners[:, :3].astype(np.float64)
    if len(owners):
        i, j, k, a = owners.T
        v0 = vol[i, j, k]
        nb = owners[:, :3].copy()
        nb[np.arange(len(owners)), a] += 1
        v1 = vol[nb[:, 0], nb[:, 1], nb[:, 2]]
        t = (isolevel - v0) / (v1 - v0)
        verts[np.arange(len(owners)), a] += t

    geo = edge_geometry()
    out = []
    for (i, j, k) in np.argwhere(emits):
        c = case[i, j, k]
        for tnum in range(n_tris[c]):
            tri = []
            for e in tris[c, tnum]:
                a, off = geo[int(e)]
                tri.append(vert_id[i + off[0], j + off[1], k + off[2], a])
            out.append(tri)
    faces = np.array(out, dtype=np.int64).reshape(-1, 3)
    return verts, faces


def mesh_vertex_transform(verts_index: np.ndarray, tx: np.ndarray, ty: np.ndarray, tz: np.ndarray, sc_factor: float, translation):
    """coslam_utils.py:148-162 -- index coordinates -> metric world coordinates (float64, numpy promotion rules)."""
    v = np.array(verts_index, dtype=np.float64, copy=True)
    v[:, :3] /= np.array([[tx.shape[0] - 1, ty.shape[0] - 1, tz.shape[0] - 1]])
    scale = np.array([tx[-1] - tx[0], ty[-1] - ty[0], tz[-1] - tz[0]])              # float32 when tx is float32
    offset = np.array([tx[0], ty[0], tz[0]])
    v[:, :3] = scale[np.newaxis, :] * v[:, :3] + offset
    v[:, :3] = v[:, :3] / sc_factor - translation
    return v


def jet_lut():
    """matplotlib's 'jet' as a 256-entry table (matplotlib/_cm.py _jet_data + colors._create_lookup_table)."""
    data = {
        "red": [(0.00, 0), (0.35, 0), (0.66, 1), (0.89, 1), (1.00, 0.5)],
        "green": [(0.000, 0), (0.125, 0), (0.375, 1), (0.640, 1), (0.910, 0), (1.000, 0)],
        "blue": [(0.00, 0.5), (0.11, 1), (0.34, 1), (0.65, 0), (1.00, 0)],
    }
    xind = 255.0 * np.linspace(0.0, 1.0, 256)
    cols = []
    for ch in ("red", "green", "blue"):
        x = np.array([p[0] for p in data[ch]], dtype=np.float64) * 255.0
        y = np.array([p[1] for p in data[ch]], dtype=np.float64)
        ind = np.searchsorted(x, xind)[1:-1]
        distance = (xind[1:-1] - x[ind - 1]) / (x[ind] - x[ind - 1])
        lut = np.concatenate([[y[0]], distance * (y[ind] - y[ind - 1]) + y[ind - 1], [y[-1]]])
        cols.append(np.clip(lut, 0.0, 1.0))
    return np.stack(cols, -1)


def jet_colors(u: np.ndarray, lut: np.ndarray):
    """colormap(x)[:, :3] for float x (matplotlib Colormap.__call__: x*N truncated, clipped to [0, N-1]; NaN -> 'bad')."""
    x = np.asarray(u, dtype=np.float64) * 256.0
    idx = np.clip(x, -1, 256).astype(np.int64)
    idx = np.clip(idx, 0, 255)
    return lut[idx]


def check_closed(faces: np.ndarray) -> bool:
    """every directed edge (a, b) has exactly one opposite (b, a): closed, consistently oriented surface."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    fwd = {}
    for a, b in e:
        if (a, b) in fwd:
            return False
        fwd[(a, b)] = 1
    return all((b, a) in fwd for (a, b) in fwd)


def vertex_normals(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """trimesh.Trimesh.vertex_normals restated (third-party, unpinned, absent here -- "parity unpinned"; the reference uses it at
    coslam_utils.py:180-181): trimesh.geometry.weighted_vertex_normals = unit face normals summed into their three vertices weighted
    by the triangle's corner angle (trimesh.triangles.angles: triangles with any corner angle below tol.merge = 1e-8 count as
    degenerate, all three weights zero), then unitized (zero vectors stay zero).  A plain loop over the faces."""
    v = np.asarray(vertices, dtype=np.float64)
    out = np.zeros_like(v)
    for a, b, c in np.asarray(faces, dtype=np.int64):
        n = np.cross(v[b] - v[a], v[c] - v[a])
        ln = np.linalg.norm(n)
        if not ln > 100 * np.finfo(np.float64).eps:
            continue
        n = n / ln

        def unit(e):
            le = np.linalg.norm(e)
            return e / le if le > 100 * np.finfo(np.float64).eps else np.zeros(3)
        u, w2, w = unit(v[b] - v[a]), unit(v[c] - v[a]), unit(v[c] - v[b])
        a0 = np.arccos(np.clip(np.dot(u, w2), -1, 1))
        a1 = np.arccos(np.clip(np.dot(-u, w), -1, 1))
        ang = np.array([a0, a1, np.pi - a0 - a1])
        if (ang < 1e-8).any():
            continue
        out[a] += n * ang[0]; out[b] += n * ang[1]; out[c] += n * ang[2]
    ln = np.linalg.norm(out, axis=1)
    ok = ln > 100 * np.finfo(np.float64).eps
    out[ok] /= ln[ok, None]
    out[~ok] = 0.0
    return out


def extract_mesh(query_fn, config, bounding_box, table, marching_cube_bound=None, color_func=None, voxel_size=None, isolevel=0.0,
                 render_uncert=True, lut=None):
    """coslam_utils.py:100-226 with ``marching_cubes`` above in place of the third-party module.
    -> dict(vol, verts_index, faces, vertices, colors) (colors: float [V,3] or None)."""
    import torch
    from oracle import spec_torch as S
    if marching_cube_bound is None:
        marching_cube_bound = bounding_box
    mcb = torch.as_tensor(marching_cube_bound, dtype=torch.float64)
    tx, ty, tz = S.get_voxels(mcb, voxel_size)                                                              # :124
    query_pts = torch.stack(torch.meshgrid(tx, ty, tz, indexing="ij"), -1).to(torch.float32)                 # :125
    sh = query_pts.shape
    flat = query_pts.reshape([-1, 3])
    bb = bounding_box.cpu()
    if config["grid"]["tcnn_encoding"]:
        flat = (flat - bb[:, 0]) / (bb[:, 1] - bb[:, 0])                                                      # :131-133
    chunk = 1024 * 64
    with torch.no_grad():
        raw = [query_fn(flat[i:i + chunk, None, :]).cpu().numpy() for i in range(0, flat.shape[0], chunk)]   # :137-138
    raw = np.concatenate(raw, 0).astype(np.float32)
    raw = np.reshape(raw, list(sh[:-1]) + [-1])
    vol = raw.squeeze()
    verts_index, faces = marching_cubes(vol, isolevel, 3.0, table)                                            # :145
    vertices = mesh_vertex_transform(verts_index, tx.numpy(), ty.numpy(), tz.numpy(), config["data"]["sc_factor"], config["data"]["translation"])
    colors = None
    if (color_func is not None and not config["mesh"]["render_color"]) or (color_func is None and render_uncert):
        vert_flat = torch.from_numpy(vertices).to(bounding_box)
        if config["grid"]["tcnn_encoding"]:
            vert_flat = (vert_flat - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])         # :165-166
        with torch.no_grad():
            if color_func is not None:
                c = [color_func(vert_flat[i:i + chunk, None, :]).cpu().numpy() for i in range(0, vert_flat.shape[0], chunk)]
                colors = np.reshape(np.concatenate(c, 0).astype(np.float32), [vert_flat.shape[0], -1])       # :172-176
            else:
                u = [query_fn(vert_flat[i:i + chunk, None, :], return_uncert=True)[:, 0, 1].cpu().numpy()
                     for i in range(0, vert_flat.shape[0], chunk)]
                u = np.concatenate(u, 0).astype(np.float32)
                un = (u - u.min()) / (u.max() - u.min())                                                      # :210
                colors = jet_colors(un.flatten(), jet_lut() if lut is None else lut)                          # :213-214
    elif color_func is not None and config["mesh"]["render_color"]:                                          # :178-186
        normals = vertex_normals(vertices, faces)
        with torch.no_grad():
            c = [color_func(torch.from_numpy(vertices[i:i + chunk]).to(bounding_box), torch.from_numpy(normals[i:i + chunk]).to(bounding_box)).cpu().numpy()
                 for i in range(0, vertices.shape[0], chunk)]
        colors = np.reshape(np.concatenate(c, 0).astype(np.float32), [vertices.shape[0], -1]) if len(c) else np.zeros((0, 3), np.float32)
    return {"vol": vol, "verts_index": verts_index, "faces": faces, "vertices": vertices, "colors": colors}
