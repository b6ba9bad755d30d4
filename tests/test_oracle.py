"""CPU tests of the oracle: pinned against the golden vectors generated from the reference
(oracle/make_golden.py), cross-checked against the independent C restatement, and property tests that pin
the intended semantics of the un-vendored ("parity unpinned") arithmetic."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from naruto_amd import synthetic as syn
from oracle import spec_torch as S

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libnaruto_oracle.so"))
    return lib


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------ golden fixtures (reference-pinned rows)
@pytest.mark.parametrize("name", ["g1_render_train_t12", "g1_render_train_t16", "g1_render_train_init",
                                  "g6_render_train_perturb", "g1_render_train_s128"])
def test_oracle_render_train_golden(name):
    g = H.load_golden(name)
    cfg = H.office_cfg(int(g["hash_size"]), perturb=float(g["perturb"]), n_samples_d=int(g["n_samples_d"]))
    w = {k: g[k] for k in ("sdf_w0", "sdf_w1", "col_w0", "col_w1")}
    ora = H.make_oracle(cfg, float(g["table_amp"]), int(g["seed"]), weights=w)
    t = {k: torch.from_numpy(g[k]) for k in ("rays_o", "rays_d", "target_rgb", "target_d")}
    rand = torch.from_numpy(g["rand"]) if "rand" in g else None
    ora.eval()
    with torch.no_grad():
        rend = ora.forward(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"], rand=rand)
    for k in ("z_vals", "raw", "rgb", "depth", "disp_map", "acc_map", "depth_var", "uncert_map", "weights"):
        H.assert_close(rend[k], g["out_" + k], 3e-6, f"{name}.{k}", rel=1e-5)
    ora.train()
    ret = ora.forward(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"], rand=rand)
    for k in ("rgb_loss", "depth_loss", "sdf_loss", "fs_loss", "psnr", "uncert_loss"):
        H.assert_close(ret[k].reshape(-1), g["loss_" + k], 1e-6, f"{name}.{k}", rel=1e-5)
    S.total_loss(ret, cfg["training"]).backward()
    for k in ("sdf_w0", "sdf_w1", "col_w0", "col_w1"):
        scale = np.abs(g["grad_" + k]).max()
        H.assert_close(H.ora_grads(ora)[k], g["grad_" + k], 1e-5 * scale, f"{name}.grad.{k}", rel=1e-4)
    tg = ora.table.grad.numpy()
    H.assert_close(tg[g["grad_table_idx"]], g["grad_table_val"], 1e-5 * np.abs(g["grad_table_val"]).max(), f"{name}.grad.table", rel=1e-4)


@pytest.mark.parametrize("hash_size", [12, 16])
def test_oracle_query_golden(hash_size):
    g = H.load_golden(f"g3_query_volume_t{hash_size}")
    cfg = H.office_cfg(hash_size)
    w = {k: g[k] for k in ("sdf_w0", "sdf_w1", "col_w0", "col_w1")}
    ora = H.make_oracle(cfg, float(g["table_amp"]), int(g["seed"]), weights=w).eval()
    with torch.no_grad():
        for tag in ("pts", "oob"):
            p = torch.from_numpy(g[tag])
            H.assert_close(ora.query_sdf(p, embed=True), g[f"{tag}_embed"], 1e-6, f"{tag}.embed")
            H.assert_close(ora.query_sdf(p, return_uncert=True), g[f"{tag}_sdf_uncert"], 3e-6, f"{tag}.sdf_uncert")
            H.assert_close(ora.query_color_sdf(p), g[f"{tag}_raw"].reshape(-1, 5), 3e-6, f"{tag}.raw")
        um, sv = S.get_map_volumes(ora.query_sdf, ora.bounding_box, float(g["map_voxel"]))
    H.assert_close(sv, g["map_sdf"], 3e-6, "map.sdf")
    H.assert_close(um, g["map_uncert"], 3e-6, "map.uncert")


def test_oracle_composite_edges_golden():
    g = H.load_golden("g5_composite_edges")
    outs = S.raw2outputs(torch.from_numpy(g["raw"]), torch.from_numpy(g["z_vals"]), 0.1, 1.0, False)
    for k, o in zip(("rgb", "disp_map", "acc_map", "weights", "depth", "depth_var", "uncert_map"), outs):
        H.assert_close(o, g["out_" + k], 2e-6, f"composite.{k}", rel=1e-5)
    # empty ray (row 5): weights underflow to exactly 0, disp is 0/0 = NaN exactly as in the reference
    assert float(outs[2][5]) == 0.0 and np.isnan(float(outs[1][5]))


# ------------------------------------------------------------------------------ C restatement vs torch restatement
def test_c_oracle_levels_and_hash(c_oracle):
    for cfg_kind, res, T in (("office", 275, 16), ("mp3d", 1015, 16), ("unit", 1024, 19), ("office12", 275, 12)):
        meta = S.HashGridMeta.from_desired_resolution(res, log2_hashmap_size=T)
        scale = np.zeros(16, np.float32)
        r, sz, off = np.zeros(16, np.uint32), np.zeros(16, np.uint32), np.zeros(17, np.uint32)
        c_oracle.oracle_levels(C.c_uint32(T), C.c_uint32(16), C.c_float(float(meta.per_level_scale)), fptr(scale), fptr(r), fptr(sz), fptr(off))
        assert np.array_equal(scale, meta.scale) and np.array_equal(r, meta.resolution)
        assert np.array_equal(sz, meta.size) and np.array_equal(off, meta.offset)
    meta = S.HashGridMeta.from_desired_resolution(275, log2_hashmap_size=14)
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(0, 1, (500, 3)), rs.uniform(-0.8, 1.8, (300, 3)), [[0, 0, 0], [1, 1, 1]]]).astype(np.float32)
    table = syn.closed_form_table(meta.n_params, 0.5)
    feat = np.zeros((x.shape[0], 32), np.float32)
    c_oracle.oracle_hash_encode(C.c_uint32(14), C.c_uint32(16), C.c_float(float(meta.per_level_scale)), C.c_uint32(x.shape[0]), fptr(x), fptr(table), fptr(feat))
    want = S.hash_encode(torch.from_numpy(x), torch.from_numpy(table), meta).numpy()
    np.testing.assert_allclose(feat, want, atol=2e-6, rtol=0)


def test_c_oracle_oneblob_query_composite(c_oracle):
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.uniform(0, 1, (400, 3)), rs.uniform(-1.3, 2.3, (400, 3)), [[0, 1, 0.5], [1.0, 0.0, 0.999]]]).astype(np.float32)
    pos = np.zeros((x.shape[0], 48), np.float32)
    c_oracle.oracle_oneblob(C.c_uint32(x.shape[0]), fptr(x), fptr(pos))
    np.testing.assert_allclose(pos, S.oneblob_encode(torch.from_numpy(x), 16).numpy(), atol=3e-6, rtol=0)
    cfg = H.office_cfg(12)
    ora = H.make_oracle(cfg, 0.3, 4).eval()
    meta = ora.meta
    xq = x[:600]
    raw, geo = np.zeros((600, 5), np.float32), np.zeros((600, 15), np.float32)
    dims = np.asarray(ora.uncert_grid.shape, np.int32)
    arrs = [ora.table, ora.uncert_grid, ora.sdf_w0, ora.sdf_w1, ora.col_w0, ora.col_w1]
    arrs = [np.ascontiguousarray(a.detach().numpy()) for a in arrs]
    c_oracle.oracle_query(C.c_uint32(12), C.c_uint32(16), C.c_float(float(meta.per_level_scale)), C.c_uint32(600), fptr(xq), fptr(arrs[0]),
                          fptr(arrs[1]), fptr(dims), fptr(arrs[2]), fptr(arrs[3]), fptr(arrs[4]), fptr(arrs[5]), fptr(raw), fptr(geo))
    with torch.no_grad():
        want = ora.query_color_sdf(torch.from_numpy(xq)).numpy()
        _, wgeo = ora.query_sdf(torch.from_numpy(xq), return_geo=True)
    np.testing.assert_allclose(raw, want, atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(geo, wgeo.numpy(), atol=5e-6, rtol=1e-5)
    g = H.load_golden("g5_composite_edges")
    n, s = g["z_vals"].shape
    outs = {k: np.zeros(n, np.float32) for k in ("disp", "acc", "depth", "var", "um")}
    rgb, wts = np.zeros((n, 3), np.float32), np.zeros((n, s), np.float32)
    c_oracle.oracle_composite(C.c_uint32(n), C.c_uint32(s), fptr(g["raw"]), fptr(g["z_vals"]), C.c_float(0.1), C.c_float(1.0), C.c_int(0), fptr(rgb),
                              fptr(outs["disp"]), fptr(outs["acc"]), fptr(wts), fptr(outs["depth"]), fptr(outs["var"]), fptr(outs["um"]))
    H.assert_close(rgb, g["out_rgb"], 3e-6, "c.rgb")
    H.assert_close(wts, g["out_weights"], 3e-6, "c.weights")
    H.assert_close(outs["depth"], g["out_depth"], 3e-6, "c.depth")
    H.assert_close(outs["var"], g["out_depth_var"], 3e-6, "c.depth_var")
    H.assert_close(outs["um"], g["out_uncert_map"], 3e-6, "c.uncert_map", rel=1e-5)
    H.assert_close(outs["disp"], g["out_disp_map"], 3e-6, "c.disp", rel=1e-5)


# ------------------------------------------------------------------------------ properties of the unpinned rows
def test_hash_grid_properties():
    meta = S.HashGridMeta.from_desired_resolution(275, log2_hashmap_size=16)
    # shipped office0 numbers: 814 088 entries, levels 0..4 dense
    assert meta.n_entries == 814088 and meta.n_params == 1628176
    assert list(meta.resolution[:6]) == [16, 20, 24, 29, 35, 42]
    assert list(meta.size[:5]) == [4096, 8000, 13824, 24392, 42880] and all(meta.size[5:] == 65536)
    assert all(meta.size % 8 == 0)
    # dense levels: the index is a bijection of the lattice; hashed levels: always < table size
    for lvl in (0, 3):
        r = int(meta.resolution[lvl])
        g = torch.arange(r)
        gx, gy, gz = torch.meshgrid(g, g, g, indexing="ij")
        idx = S.hash_grid_index(meta, lvl, gx.reshape(-1), gy.reshape(-1), gz.reshape(-1))
        assert idx.unique().numel() == r ** 3
    big = torch.randint(0, 2 ** 32, (10000, 3), dtype=torch.int64)
    for lvl in (5, 15):
        idx = S.hash_grid_index(meta, lvl, big[:, 0], big[:, 1], big[:, 2])
        assert int(idx.min()) >= 0 and int(idx.max()) < int(meta.size[lvl])
    # trilinear weights form a partition of unity: encoding a constant table returns the constant
    x = torch.rand(2000, 3) * 1.6 - 0.3
    e = S.hash_encode(x, torch.full((meta.n_params,), 0.75), meta)
    assert torch.allclose(e, torch.full_like(e, 0.75), atol=2e-6)
    # interpolation: at lattice nodes of a dense level the feature equals the stored entry
    lvl, r = 1, int(meta.resolution[1])
    tab = torch.arange(meta.n_params, dtype=torch.float32) * 1e-3
    node = torch.tensor([[3, 7, 11]], dtype=torch.float32)
    xn = (node - 0.5) / float(meta.scale[lvl]) + 1e-7
    e = S.hash_encode(xn, tab, meta)
    entry = int(meta.offset[lvl]) + 3 + 7 * r + 11 * r * r
    assert abs(float(e[0, 2 * lvl]) - float(tab[2 * entry])) < 2e-3


def test_oneblob_properties():
    x = torch.cat([torch.rand(4000, 3), torch.tensor([[0.0, 1.0, 0.5]])])
    e = S.oneblob_encode(x, 16).reshape(-1, 3, 16)
    assert torch.allclose(e.sum(-1), torch.ones(e.shape[0], 3), atol=1e-5)      # every dim's 16 bins sum to 1
    assert float(e.min()) >= -1e-6
    assert int((e > 1e-7).sum(-1).max()) <= 3                                     # at most 3 active bins per dim
    # periodic: x and x + 1 encode identically
    a = S.oneblob_encode(torch.tensor([[0.2, 0.7, 0.95]]), 16)
    b = S.oneblob_encode(torch.tensor([[0.2, 0.7, 0.95]]) - 1.0, 16)
    assert torch.allclose(a, b, atol=2e-6)


def test_sdf2weights_properties():
    rs = np.random.RandomState(0)
    sdf = torch.from_numpy(np.sort(rs.normal(size=(64, 43)).astype(np.float32) * 0.4, axis=1)[:, ::-1].copy())
    z = torch.from_numpy(np.sort(rs.uniform(0, 5, size=(64, 43)).astype(np.float32), axis=1))
    w = S.sdf2weights(sdf, z, 0.1, 1.0)
    s = w.sum(-1)
    assert bool(((s - 1).abs() < 1e-5).all())
    # zero past (first sign change depth + truncation)
    signs = sdf[:, 1:] * sdf[:, :-1] < 0
    first = torch.argmax(signs.float(), dim=1)
    zmin = z.gather(1, first[:, None])
    assert float((w * (z >= zmin + 0.1)).abs().max()) == 0.0
    # no sign change: truncated at z[0] + trunc
    w2 = S.sdf2weights(torch.full((1, 43), 0.3), z[:1], 0.1, 1.0)
    assert float((w2 * (z[:1] >= z[0, 0] + 0.1)).abs().max()) == 0.0


def test_uncert_grid_axis_quirk_and_manual_restatement():
    grid = torch.from_numpy(syn.closed_form_uncert_grid((7, 9, 5)))
    x = torch.rand(3000, 3) * 1.4 - 0.2
    a = S.sample_uncert_grid_ref(grid, x)
    b = S.sample_uncert_grid_manual(grid, x)
    assert torch.allclose(a, b, atol=2e-6)
    # coordinate 0 walks the LAST axis: centre of voxel (i,j,k) is x = ((k+.5)/Nz, (j+.5)/Ny, (i+.5)/Nx)
    p = torch.tensor([[(3 + 0.5) / 5, (4 + 0.5) / 9, (2 + 0.5) / 7]])
    assert abs(float(S.sample_uncert_grid_ref(grid, p)) - float(grid[2, 4, 3])) < 1e-5


def test_loss_quirks():
    """rgb_missing is written into a BOOL mask (no effect unless 0), and the uncertainty NLL is the mean of an
    OUTER product (scene_rep.py:249-250, :284) -- both reproduced from the reference and pinned by the goldens."""
    cfg = H.office_cfg(12)
    rays = syn.random_rays(40, cfg["mapping"]["bound"], seed=3, zero_depth_frac=0.3)
    t = {k: torch.from_numpy(v) for k, v in rays.items()}
    ora = H.make_oracle(cfg, 0.2, 3).train()
    ret = ora.forward(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"])
    ora.eval()
    with torch.no_grad():
        rend = ora.forward(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"])
    plain = torch.mean((rend["rgb"] - t["target_rgb"]) ** 2)
    assert abs(float(ret["rgb_loss"]) - float(plain)) < 1e-7
    valid = t["target_d"].squeeze() > 0
    um, x, y = rend["uncert_map"][valid], rend["depth"][valid], t["target_d"].squeeze()[valid]
    outer = torch.mean(1 / (2 * (um + 1e-9))) * torch.mean((x - y) ** 2) + 0.5 * torch.mean(torch.log(um + 1e-9))
    assert abs(float(ret["uncert_loss"]) - float(outer)) < 1e-5 * max(1.0, abs(float(outer)))


# ------------------------------------------------------------------------------ N1 / N2 ("next" rows)
def _active_inputs(g):
    t = {k: torch.from_numpy(g[k]) for k in ("rays_o", "rays_d", "target_rgb", "target_d")}
    return t, int(g["n_cur"]), int(g["base"]), int(g["K"]), int(g["mul"]), [list(map(float, b)) for b in g["bound"]]


def test_oracle_active_ray_sampler_golden():
    """The numpy restatement reproduces the reference's ActiveRaySampler output exactly; its deterministic variant
    (what the HIP kernel implements) selects the same multiset of uncertainty values -- only ties may differ."""
    g = H.load_golden("g8_active_ray")
    t, n_cur, base, K, mul, bound = _active_inputs(g)
    out, vals, sel = S.active_ray_sample(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"], n_cur, g["vol"], bound, base, K, mul)
    for k, a in zip(("out_rays_o", "out_rays_d", "out_target_rgb", "out_target_d"), out):
        assert np.array_equal(a.numpy(), g[k]), k
    assert np.array_equal(vals, g["cand_vals"])
    out_d, _, sel_d = S.active_ray_sample(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"], n_cur, g["vol"], bound, base, K, mul,
                                          deterministic=True)
    assert np.array_equal(np.sort(vals[sel]), np.sort(vals[sel_d]))                  # same K smallest values
    assert vals[sel_d].max() <= np.partition(vals, K)[K]                             # nothing above the (K+1)-th smallest
    for a, b in zip(out, out_d):
        assert torch.equal(a[K:], b[K:])                                             # the untouched parts are identical
    assert (np.diff(sel_d) > 0).all()


def test_oracle_rays_to_world_golden():
    g = H.load_golden("g8_active_ray")
    o, d = S.rays_to_world(torch.from_numpy(g["dcam"]), torch.from_numpy(g["ids"]), torch.from_numpy(g["poses"]))
    assert np.array_equal(o.numpy(), g["world_o"]) and np.allclose(d.numpy(), g["world_d"], atol=1e-7)


def test_oracle_planner_aggregation_golden():
    """N3: the restatement of NarutoPlanner.uncertainty_aggregation_v2 reproduces the reference's outputs exactly for the
    reference's own target selection; the deterministic selection (what the HIP path does) picks from the same top_k."""
    g = H.load_golden("g9_planner_aggregation")
    bbox = [list(map(float, b)) for b in g["bbox"]]
    dims, ranges, goal_idx = S.goal_space(bbox, 0.1, list(g["gs_z_levels"]))
    assert dims == g["uncert"].shape
    coll, agg, valid = S.uncert_aggregation(g["uncert"], g["sdf"], g["targets"], goal_idx, dims, 0.1, (0.5, 2.0), 0.8)
    assert np.array_equal(coll.numpy(), g["collections"])
    assert np.array_equal(agg.numpy().reshape(g["aggregated"].shape), g["aggregated"])
    top_k, sub = int(g["top_k"]), int(g["top_k_subset"])
    assert np.array_equal(S.topk_targets_reference(g["uncert"], top_k, sub), g["targets"])
    det = S.topk_targets_deterministic(g["uncert"], top_k, sub)
    flat = g["uncert"].reshape(-1)
    kth = np.sort(flat)[-top_k]
    vals = g["uncert"][det[:, 0], det[:, 1], det[:, 2]]
    assert det.shape == (sub, 3) and (vals >= kth).all() and len({tuple(r) for r in det}) == sub
    ref_vals = g["uncert"][g["targets"][:, 0], g["targets"][:, 1], g["targets"][:, 2]]
    assert (ref_vals >= kth).all()                       # both selections are subsets of the top_k


# --------------------------------------------------------------------------------------------- N4: mesh path
def _mesh_cfg(g):
    cfg = H.office_cfg(int(g["hash_size"]))
    cfg["data"]["sc_factor"], cfg["data"]["translation"] = float(g["sc_factor"]), float(g["translation"])
    return cfg


def test_mc_table_is_what_the_generator_derives():
    """tests/golden/mc_table.npz == tools/gen_mc_table.py's derivation == the table compiled into the HIP library."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_mc_table", os.path.join(root, "tools", "gen_mc_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    n_tris, tris, edge_mask = gen.build()
    t = H.load_golden("mc_table")
    assert np.array_equal(t["n_tris"], n_tris) and np.array_equal(t["tris"], tris) and np.array_equal(t["edge_mask"], edge_mask)
    inc = open(os.path.join(root, "naruto_amd", "csrc", "naruto_mc_table.inc")).read()
    rows = [r for r in inc.split("kMcTris")[1].split("\n") if r.strip().startswith("{")]
    compiled = np.array([[int(v) for v in r.strip().strip("{},").split(",")] for r in rows], dtype=np.int8)
    assert np.array_equal(compiled.reshape(256, -1, 3), tris)
    # every case: triangles use exactly the crossed edges (corner states differ), no triangle without a crossing
    for c in range(256):
        crossed = 0
        for e in range(12):
            a, q = e >> 2, e & 3
            u, w = [i for i in range(3) if i != a]
            off = [0, 0, 0]
            off[u], off[w] = q & 1, q >> 1
            lo = off[0] | (off[1] << 1) | (off[2] << 2)
            if ((c >> lo) & 1) != ((c >> (lo | (1 << a))) & 1):
                crossed |= 1 << e
        assert crossed == int(edge_mask[c]), c
    assert n_tris[0] == 0 and n_tris[255] == 0 and int(n_tris.max()) == 5


def test_oracle_marching_cubes_properties():
    from oracle import mesh_numpy as MN
    table = H.load_golden("mc_table")
    n = 21
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    c0 = np.array([0.05, -0.03, 0.02])
    vol = (np.linalg.norm(g - c0, axis=-1) - 0.6).astype(np.float32)
    v, f = MN.marching_cubes(vol, 0.0, 3.0, table)
    assert MN.check_closed(f) and len(v) - len(f) * 3 // 2 + len(f) == 2             # closed, Euler characteristic of a sphere
    p = v / (n - 1) * 2 - 1 - c0
    nrm = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    assert ((nrm * p[f].mean(1)).sum(-1) > 0).all()                                   # normals towards larger values (outside)
    assert np.abs(np.linalg.norm(p, axis=-1) - 0.6).max() < 0.01                      # vertices on the surface
    # the vertex set is exactly the set of crossed lattice edges
    vol64 = vol.astype(np.float64)
    n_cross = sum(int(((np.take(vol64, range(0, n - 1), a) < 0) != (np.take(vol64, range(1, n), a) < 0)).sum()) for a in range(3))
    assert n_cross == len(v)
    # a noisy volume (all 256 cases, ambiguous faces everywhere): still no directed edge twice, holes only on the boundary
    rs = np.random.RandomState(0)
    vol = rs.standard_normal((11, 12, 13)).astype(np.float32)
    v, f = MN.marching_cubes(vol, 0.0, 1e9, table)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    es = set(map(tuple, e))
    assert len(es) == len(e)
    open_edges = np.array([(a, b) for (a, b) in es if (b, a) not in es])
    vv = v[open_edges]
    assert ((vv <= 0) | (vv >= np.array(vol.shape) - 1)).any(-1).all()
    # truncation removes cells, never adds
    v2, f2 = MN.marching_cubes(vol, 0.0, 1.5, table)
    assert 0 < len(f2) < len(f) and len(v2) < len(v)


def test_oracle_extract_mesh_golden():
    """N4: the restatement of extract_mesh reproduces what the reference's own function returned (oracle/make_golden.py
    case_extract_mesh), both colour branches."""
    from oracle import mesh_numpy as MN
    g = H.load_golden("g10_extract_mesh")
    table = H.load_golden("mc_table")
    cfg = _mesh_cfg(g)
    w = {k: g[k] for k in ("sdf_w0", "sdf_w1", "col_w0", "col_w1")}
    ora = H.make_oracle(cfg, float(g["table_amp"]), int(g["seed"]), weights=w).eval()
    assert np.array_equal(MN.jet_lut(), g["jet_lut"])
    for tag, color_func in (("color", ora.query_color), ("uncert", None)):
        o = MN.extract_mesh(ora.query_sdf, cfg, ora.bounding_box, table, marching_cube_bound=torch.from_numpy(g["mcb"]), color_func=color_func,
                            voxel_size=float(g["voxel"]), isolevel=float(g["isolevel"]))
        H.assert_close(o["vol"], g["vol"], 3e-6, f"{tag}.vol")
        assert np.array_equal(o["faces"], g["faces"])
        H.assert_close(o["vertices"], g[f"{tag}_vertices"], 1e-5, f"{tag}.vertices")
        if tag == "color":
            H.assert_close(o["colors"], g["color_colors"], 3e-6, "colors")
        else:
            assert (np.abs(o["colors"] - g["uncert_colors"]).max(-1) > 1e-6).mean() < 0.01       # a bin edge may flip with 1-ulp noise
    v, f = MN.marching_cubes(g["vol"], float(g["isolevel"]), 3.0, table)
    assert np.array_equal(v, g["verts_index"]) and np.array_equal(f, g["faces"])


def test_bf16_restatement_is_pinned_to_the_exact_network():
    """oracle/spec_bf16.py (what the GPU tests hold the bf16-MFMA kernels to) against an fp64 evaluation of the EXACT network (the reference's
    decoder, decoder.py:29-41,99-116): the distance stays inside the bound derived from the half-ulp rounding of every bf16 operand
    (bf16_forward_bound: absolute-value propagation, no cancellation assumed), elementwise -- and is, as random signs make it, an order of
    magnitude inside; the uncertainty channel, which does not pass through the MLPs, is untouched.  The restatement's gradients are finite
    and close to the exact network's (bounded relative distance: the weight gradients see rounded cotangents and rounded inputs)."""
    from oracle import spec_bf16 as BF
    cfg = H.office_cfg(12)
    ora = H.make_oracle(cfg, 0.3, 61)
    rs = np.random.RandomState(61)
    x = torch.from_numpy(np.concatenate([rs.uniform(0, 1, (1500, 3)), rs.uniform(-0.4, 1.4, (250, 3))]).astype(np.float32))
    raw, out = BF.query_color_sdf_bf16(ora, x)
    bound, exact = BF.bf16_forward_bound(ora, x)
    err = (raw[:, :4].detach().double() - exact).abs()
    slack = 1e-6 * exact.abs() + 1e-7                                     # fp32 accumulation of the restatement against the fp64 evaluation
    assert bool((err <= bound + slack).all()), f"outside the derived bound: worst ratio {float((err / (bound + slack)).max()):.3f}"
    assert float((err / (bound + slack)).max()) < 0.5, "the worst-case bound should be loose by the cancellation of ~80 random-sign terms"
    assert float(err.max()) <= 1e-2 * float(exact.abs().max())
    assert torch.equal(raw[:, 4], ora.query_color_sdf(x)[:, 4]), "the uncertainty channel does not pass through the MLPs"
    cot = torch.from_numpy(rs.normal(size=(x.shape[0], 5)).astype(np.float32))
    (raw * cot).sum().backward()
    g_bf = {k: v.grad.clone() for k, v in (("sdf_w0", ora.sdf_w0), ("sdf_w1", ora.sdf_w1), ("col_w0", ora.col_w0), ("col_w1", ora.col_w1), ("table", ora.table))}
    ora.zero_grad()
    (ora.query_color_sdf(x) * cot).sum().backward()
    for k, p_ in (("sdf_w0", ora.sdf_w0), ("sdf_w1", ora.sdf_w1), ("col_w0", ora.col_w0), ("col_w1", ora.col_w1), ("table", ora.table)):
        a, b = g_bf[k].double(), p_.grad.double()
        assert bool(torch.isfinite(a).all())
        rel = float((a - b).norm() / b.norm())
        assert rel < 0.1, f"bf16 restatement, gradient {k}: relative distance {rel:.3e} to the exact network"      # measured 0.5 - 5 % (ReLU masks of the bf16 forward, rounded cotangents)
