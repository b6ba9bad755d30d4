#!/usr/bin/env python3
"""Generate the marching-cubes case table used by naruto_amd/csrc/naruto_mesh.hip (row N4 of SURVEY.md section 8f).

The reference extracts meshes with the third-party `marching_cubes` module of Co-SLAM / NeuralRGBD
(coslam_utils.py:26,145), which is not part of /root/reference and cannot be installed here.  Rather than typing a
256-row table from memory, the table is DERIVED: for every sign configuration of the 8 cube corners the iso-surface
crosses the cube edges whose end points differ; on every cube face the crossings are joined by segments (two
crossings: one segment; four crossings -- the ambiguous face -- the segments cut off the two inside corners, the same
rule from both cubes sharing the face, so the surface stays watertight), the directed segments chain into closed
loops, every loop is triangulated (a fan where possible; never with a chord lying in a cube face, which would overlap the
neighbouring cell there).  Triangles are wound so that their normal points to the outside (larger values).

Conventions (shared by the kernel, the oracle and the tests):
  corner c          offset (c & 1, (c >> 1) & 1, (c >> 2) & 1) along (x, y, z); bit c of the case = value < isolevel
  edge  e = 4*a + q axis a, q = du + 2*dw where (u, w) are the other two axes in increasing order; the edge runs from
                    the corner with offsets (a: 0, u: du, w: dw) along +a

Writes naruto_amd/csrc/naruto_mc_table.inc (product) and tests/golden/mc_table.npz (the same data for the oracle).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corner_offset(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def corner_id(off):
    return off[0] | (off[1] << 1) | (off[2] << 2)


def edge_id(ca, cb):
    oa, ob = corner_offset(ca), corner_offset(cb)
    diff = [i for i in range(3) if oa[i] != ob[i]]
    assert len(diff) == 1
    a = diff[0]
    u, w = [i for i in range(3) if i != a]
    lo = oa if oa[a] == 0 else ob
    return 4 * a + lo[u] + 2 * lo[w]


def faces():
    """Six faces, corners counter-clockwise as seen from outside the cube."""
    out = []
    for a in range(3):
        u, w = (a + 1) % 3, (a + 2) % 3            # e_u x e_w = e_a
        for s in (0, 1):
            ring = []
            for du, dw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                off = [0, 0, 0]
                off[a], off[u], off[w] = s, du, dw
                ring.append(corner_id(off))
            out.append(ring if s == 1 else ring[::-1])
    return out


def case_triangles(case):
    inside = [(case >> c) & 1 for c in range(8)]
    nxt = {}
    for ring in faces():
        starts, ends = {}, {}                      # position i on the ring -> edge id of (ring[i], ring[i+1])
        for i in range(4):
            ca, cb = ring[i], ring[(i + 1) % 4]
            if inside[ca] != inside[cb]:
                (starts if inside[ca] else ends)[i] = edge_id(ca, cb)
        if len(starts) == 1:
            (p,), (q,) = starts.values(), ends.values()
            assert p not in nxt
            nxt[p] = q
        elif len(starts) == 2:                     # ambiguous face: cut off each inside corner ring[i]
            for i, p in starts.items():
                assert p not in nxt
                nxt[p] = ends[(i - 1) % 4]
    tris = []
    todo = set(nxt)
    assert sorted(nxt.values()) == sorted(nxt.keys())
    while todo:
        e0 = min(todo)
        loop = [e0]
        todo.remove(e0)
        e = nxt[e0]
        while e != e0:
            loop.append(e)
            todo.remove(e)
            e = nxt[e]
        assert len(loop) >= 3
        for (i, j, k) in triangulate(loop):
            tris.append((loop[i], loop[k], loop[j]))          # reversed: normals towards the outside
    return tris


def edge_faces(e):
    a, q = e >> 2, e & 3
    u, w = [i for i in range(3) if i != a]
    return {(u, q & 1), (w, q >> 1)}


def triangulate(loop):
    """Triangles (i < j < k, positions on the loop) of the first triangulation -- fans first -- none of whose chords
    lies in a cube face: a chord in a face would overlap (or cross) the neighbouring cell's segments on that face."""
    n = len(loop)

    def chord_ok(i, j):
        if (j - i) % n in (1, n - 1):
            return True                                        # a side of the polygon, not a chord
        return not (edge_faces(loop[i]) & edge_faces(loop[j]))

    def solve(i, j):
        """all triangulations of the sub-polygon i..j (positions), chord (i, j) already accepted"""
        if j - i < 2:
            return [[]]
        out = []
        for k in range(i + 1, j):
            if chord_ok(i, k) and chord_ok(k, j):
                for left in solve(i, k):
                    for right in solve(k, j):
                        out.append(left + [(i, k, j)] + right)
        return out

    options = solve(0, n - 1)
    assert options, f"no face-chord-free triangulation for loop {loop}"
    fan = [(0, k, k + 1) for k in range(1, n - 1)]
    for o in options:
        if sorted(o) == fan:
            return fan
    return sorted(options[0])


def build():
    table = [case_triangles(c) for c in range(256)]
    max_t = max(len(t) for t in table)
    n_tris = np.array([len(t) for t in table], dtype=np.uint8)
    tris = np.full((256, max_t, 3), -1, dtype=np.int8)
    edge_mask = np.zeros(256, dtype=np.uint16)
    for c, t in enumerate(table):
        for k, tri in enumerate(t):
            tris[c, k] = tri
            for e in tri:
                edge_mask[c] |= 1 << e
    return n_tris, tris, edge_mask


def main():
    n_tris, tris, edge_mask = build()
    max_t = tris.shape[1]
    inc = os.path.join(ROOT, "naruto_amd", "csrc", "naruto_mc_table.inc")
    with open(inc, "w") as f:
        f.write("// Generated by tools/gen_mc_table.py -- do not edit.  Marching-cubes case table, conventions in that file.\n")
        f.write(f"constexpr int kMcMaxTris = {max_t};\n")
        f.write("__device__ const uint8_t kMcNumTris[256] = {\n")
        for r in range(0, 256, 32):
            f.write("    " + ", ".join(str(int(v)) for v in n_tris[r:r + 32]) + ",\n")
        f.write("};\n")
        f.write(f"__device__ const int8_t kMcTris[256][{max_t * 3}] = {{\n")
        for c in range(256):
            f.write("    {" + ", ".join(str(int(v)) for v in tris[c].reshape(-1)) + "},\n")
        f.write("};\n")
    npz = os.path.join(ROOT, "tests", "golden", "mc_table.npz")
    np.savez_compressed(npz, n_tris=n_tris, tris=tris, edge_mask=edge_mask)
    print(f"max triangles per case: {max_t}; total triangles: {int(n_tris.sum())}; wrote {inc} and {npz}")


if __name__ == "__main__":
    sys.exit(main())
