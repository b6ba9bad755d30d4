"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the level tables it
derives equal the oracle's, argument validation, config loading, synthetic workloads."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from naruto_amd import config as cfgmod
from naruto_amd import synthetic as syn
from oracle import spec_torch as S

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    from naruto_amd import _lib
    header = open(os.path.join(ROOT, "include", "naruto_hip.h")).read()
    declared = set(re.findall(r"\b(naruto_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(built_lib, name), f"libnaruto_hip.so does not export {name}"
    assert built_lib.naruto_version() >= 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from naruto_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _lib.load()


@pytest.mark.parametrize("res,T", [(275, 16), (275, 12), (1015, 16), (1024, 22), (340, 16)])
def test_field_levels_match_oracle(built_lib, res, T):
    from naruto_amd import ops
    meta = S.HashGridMeta.from_desired_resolution(res, log2_hashmap_size=T)
    h = ops.FieldHandle(log2_hashmap_size=T, per_level_scale=float(np.exp2(np.log2(res / 16) / 15)), uncert_dims=(4, 5, 6),
                        bbox_min=(0, 0, 0), bbox_max=(1, 1, 1), trunc=0.1, sc_factor=1.0)
    scale, r, size, off = h.levels()
    assert np.array_equal(np.asarray(scale, np.float32), meta.scale)
    assert np.array_equal(np.asarray(r), meta.resolution) and np.array_equal(np.asarray(size), meta.size)
    assert np.array_equal(np.asarray(off), meta.offset) and h.n_entries == meta.n_entries


def test_create_rejects_unsupported_configurations(built_lib):
    from naruto_amd import ops, _lib
    kw = dict(per_level_scale=1.2, uncert_dims=(4, 5, 6), bbox_min=(0, 0, 0), bbox_max=(1, 1, 1), trunc=0.1, sc_factor=1.0)
    for bad in (dict(n_levels=8), dict(n_features=4), dict(hidden_dim=64), dict(n_bins=8), dict(geo_feat_dim=7), dict(log2_hashmap_size=40),
                dict(trunc=0.0), dict(uncert_dims=(0, 1, 1))):
        with pytest.raises(_lib.NarutoError):
            ops.FieldHandle(**{**kw, **bad})
    # NULL arguments come back as error codes + message, never a crash
    rc = built_lib.naruto_sample_z(4, None, 0.0, 5.0, 0, 0, 0.0, 64, None, None, None)
    assert rc != 0 and b"NULL" in built_lib.naruto_last_error()


def test_hot_path_refuses_cpu_tensors(built_lib):
    from naruto_amd.field import NarutoFieldHIP
    cfg = H.office_cfg(12)
    m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32))
    m.get_uncert_grid(0.1)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        m.query_sdf(torch.rand(10, 3))
    # state_dict keeps the REFERENCE MODEL's own key set and shapes -- recorded from JointEncodingNaruto.state_dict() by oracle/make_golden.py
    # (fixture g11: what CoSLAMNaruto.save_ckpt / load_ckpt exchange, coslam.py:494-517) --, incl. the aliased top-level nets
    g11 = H.load_golden("g11_state_dict_t12")
    sd = m.state_dict()
    assert set(sd.keys()) == set(g11.keys()), set(sd.keys()) ^ set(g11.keys())
    for k, shape in g11.items():
        assert tuple(sd[k].shape) == tuple(int(v) for v in shape), (k, tuple(sd[k].shape), shape)
    keys = set(sd.keys())
    want = {"uncert_grid", "embed_fn.params", "embedpos_fn.params",
            "decoder.sdf_net.model.0.weight", "decoder.sdf_net.model.2.weight",
            "decoder.color_net.model.0.weight", "decoder.color_net.model.2.weight",
            "sdf_net.model.0.weight", "sdf_net.model.2.weight", "color_net.model.0.weight", "color_net.model.2.weight"}
    assert keys == want, keys ^ want
    assert m.embed_fn.params.numel() == 2 * 4096 * 16 and tuple(m.uncert_grid.shape) == (49, 56, 35)
    assert m.decoder.sdf_net.model[0].weight.shape == (32, 80) and m.decoder.color_net.model[0].weight.shape == (32, 63)


def test_unsupported_model_configs_raise():
    from naruto_amd.field import NarutoFieldHIP
    bb = torch.tensor([[0.0, 1.0]] * 3)
    for path, val in ((("grid", "oneGrid"), False), (("decoder", "tcnn_network"), True), (("decoder", "uncert_grid"), False),
                      (("training", "n_importance"), 8)):
        cfg = H.office_cfg(12)
        cfg[path[0]][path[1]] = val
        with pytest.raises(NotImplementedError):
            NarutoFieldHIP(cfg, bb)


def test_config_inherit_from(tmp_path):
    base = tmp_path / "base.yaml"
    base.write_text("mapping:\n  sample: 2048\n  iters: 10\ngrid:\n  hash_size: 16\ntraining:\n  trunc: 0.1\n")
    scene = tmp_path / "scene.yaml"
    scene.write_text(f"inherit_from: {base}\nmapping:\n  bound: [[-1,1],[-2,2],[0,3]]\n  iters: 20\n")
    cfg = cfgmod.load_config(str(scene))
    assert cfg["mapping"]["sample"] == 2048 and cfg["mapping"]["iters"] == 20 and cfg["grid"]["hash_size"] == 16
    assert cfg["mapping"]["bound"][1] == [-2, 2]
    o = cfgmod.office0_config()
    assert o["mapping"]["bound"] == [[-2.2, 2.6], [-3.4, 2.1], [-1.4, 2.0]]
    assert o["training"]["n_samples_d"] == 32 and o["training"]["n_range_d"] == 11 and o["grid"]["hash_size"] == 16
    bb = torch.tensor(o["mapping"]["bound"])
    assert S.get_resolution(bb, o["grid"]["voxel_sdf"]) == 275
    assert S.get_resolution(torch.tensor(cfgmod.mp3d_large_config()["mapping"]["bound"]), 0.02) == 1015


def test_synthetic_workloads_are_reproducible():
    a = syn.random_rays(100, cfgmod.office0_config()["mapping"]["bound"], seed=5)
    b = syn.random_rays(100, cfgmod.office0_config()["mapping"]["bound"], seed=5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.allclose(np.linalg.norm(a["rays_d"], axis=1), 1.0, atol=1e-6)
    assert (a["target_d"] == 0).sum() > 0
    p = syn.pinhole_rays(64, 64, 32.0, 32.0, cfgmod.office0_config()["mapping"]["bound"])
    assert p["rays_o"].shape == (4096, 3) and p["rays_d"].shape == (4096, 3)
    assert syn.closed_form_table(10, 0.5).dtype == np.float32


def test_reference_workload_config0_on_oracle():
    """BASELINE.json configs[0]: office_0, 64x64 rays, 32 samples/ray, CPU path -- runs on the oracle."""
    cfg = H.office_cfg(16, n_samples_d=21)          # 21 + 11 near-surface = 32 samples per ray
    rays = syn.pinhole_rays(64, 64, 32.0, 32.0, cfg["mapping"]["bound"])
    ora = H.make_oracle(cfg, 1e-4, 0).train()
    t = {k: torch.from_numpy(v) for k, v in rays.items()}
    ret = ora.forward(t["rays_o"], t["rays_d"], t["target_rgb"], t["target_d"])
    loss = S.total_loss(ret, cfg["training"])
    loss.backward()
    assert torch.isfinite(loss) and ret["rgb"].shape == (4096, 3)
    assert float(ora.table.grad.abs().sum()) > 0


def test_feistel_permutation_is_a_bijection(built_lib):
    """naruto_perm_index (host build of the device function behind naruto_assemble_rays / naruto_sample_distinct): a bijection
    of [0, n) for every n, different for different keys -- i.e. the first k values are k DISTINCT indices, which is what
    python's random.sample guarantees in the reference."""
    from naruto_amd import _lib
    lib = _lib.load()
    for n in (1, 2, 3, 7, 64, 100, 1000, 4097, 40800):
        p = np.array([lib.naruto_perm_index(i, n, 11, 5, 2) for i in range(n)], dtype=np.int64)
        assert p.min() == 0 and p.max() == n - 1 and len(np.unique(p)) == n, n
    a = [lib.naruto_perm_index(i, 40800, 11, 5, 2) for i in range(64)]
    b = [lib.naruto_perm_index(i, 40800, 11, 6, 2) for i in range(64)]
    c = [lib.naruto_perm_index(i, 40800, 11, 5, 3) for i in range(64)]
    assert a != b and a != c and len(set(a)) == 64
    # no gross bias: the first 4096 draws of 163 200 spread over all 16 sixteenths of the range
    q = np.array([lib.naruto_perm_index(i, 163200, 3, 1, 2) for i in range(4096)]) * 16 // 163200
    counts = np.bincount(q, minlength=16)
    assert counts.min() > 180 and counts.max() < 340, counts


# --------------------------------------------------------------------------------------------- N4: mesh host side
def test_mesh_host_helpers(tmp_path):
    from naruto_amd import mesh as M
    g = H.load_golden("g10_extract_mesh")
    assert np.array_equal(M.jet_lut().numpy(), g["jet_lut"])                       # == matplotlib's table (oracle/make_golden.py)
    mcb = g["mcb"]
    tx, ty, tz = M.get_voxels(mcb[0, 1], mcb[0, 0], mcb[1, 1], mcb[1, 0], mcb[2, 1], mcb[2, 0], float(g["voxel"]))
    assert (tx.numel(), ty.numel(), tz.numel()) == g["vol"].shape
    ox, oy, oz = S.get_voxels(torch.from_numpy(mcb).double(), float(g["voxel"]))
    assert torch.equal(tx, ox) and torch.equal(ty, oy) and torch.equal(tz, oz)
    rx, _, _ = M.get_voxels(1.0, 0.0, 1.0, 0.0, 1.0, 0.0, None, 9)
    assert rx.numel() == 9
    # PLY round trip
    col = M._float_colors_to_rgba8(g["color_colors"])
    assert col.shape == (len(g["color_colors"]), 4) and (col[:, 3] == 255).all()
    mesh = M.Mesh(g["color_vertices"], g["faces"], col)
    path = tmp_path / "m.ply"
    mesh.export(str(path))
    blob = open(path, "rb").read()
    head, body = blob.split(b"end_header\n", 1)
    assert b"element vertex %d" % len(mesh.vertices) in head and b"element face %d" % len(mesh.faces) in head
    vdt = np.dtype([("p", "<f4", (3,)), ("c", "u1", (4,))])
    v = np.frombuffer(body, dtype=vdt, count=len(mesh.vertices))
    assert np.array_equal(v["p"], mesh.vertices.astype(np.float32)) and np.array_equal(v["c"], col)
    f = np.frombuffer(body, dtype=np.dtype([("n", "u1"), ("i", "<i4", (3,))]), count=len(mesh.faces), offset=vdt.itemsize * len(mesh.vertices))
    assert (f["n"] == 3).all() and np.array_equal(f["i"], mesh.faces)
    assert len(body) == vdt.itemsize * len(mesh.vertices) + 13 * len(mesh.faces)


def test_mesh_entry_points_validate_arguments(built_lib):
    lib = built_lib
    dims = (C.c_uint32 * 3)(0, 4, 4)
    assert lib.naruto_mesh_workspace(dims) == 0
    assert lib.naruto_mesh_count(dims, None, 0.0, 3.0, None, None, None) < 0
    dims = (C.c_uint32 * 3)(4, 4, 4)
    assert lib.naruto_mesh_workspace(dims) >= 64 * 10
    assert lib.naruto_mesh_count(dims, None, 0.0, 3.0, None, None, None) < 0
    assert b"NULL" in lib.naruto_last_error()
    assert lib.naruto_mesh_emit(dims, None, 0.0, None, 0, 0, None, None, None) < 0
    assert lib.naruto_lattice_points(dims, None, None, None, None, None) < 0
    big = (C.c_uint32 * 3)(2048, 2048, 2048)
    assert lib.naruto_mesh_workspace(big) == 0


def test_vertex_normals_match_the_loop_restatement():
    """naruto_amd.mesh.vertex_normals (vectorised torch) against oracle.mesh_numpy.vertex_normals (a loop over the faces of
    trimesh's angle-weighted rule) on a closed surface with a degenerate triangle and an unreferenced vertex thrown in; on a cube
    the angle weights make every corner normal the (1,1,1)/sqrt(3) diagonal whichever way its faces are split."""
    import numpy as np
    import torch
    from naruto_amd import mesh as M
    from oracle import mesh_numpy as MN
    corners = np.array([[x, y, z] for x in (0., 1.) for y in (0., 1.) for z in (0., 1.)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]       # outward-facing
    faces = []
    for q, (a, b, c, d) in enumerate(quads):
        faces += [(a, b, c), (a, c, d)] if q % 2 == 0 else [(a, b, d), (b, c, d)]
    faces = np.array(faces)
    n = M.vertex_normals(torch.from_numpy(corners), torch.from_numpy(faces)).numpy()
    want = (corners * 2 - 1) / np.sqrt(3.0)
    assert np.abs(n - want).max() < 1e-12
    rs = np.random.RandomState(3)
    v = np.concatenate([corners + rs.normal(scale=0.05, size=corners.shape), [[5., 5., 5.]]])           # vertex 8: unreferenced
    f = np.concatenate([faces, [[0, 0, 3]], [[1, 2, 2]]])                                                # degenerate triangles
    got = M.vertex_normals(torch.from_numpy(v), torch.from_numpy(f)).numpy()
    ref = MN.vertex_normals(v, f)
    assert np.abs(got - ref).max() < 1e-12
    assert np.all(got[8] == 0.0) and np.allclose(np.linalg.norm(got[:8], axis=1), 1.0)


def test_bench_host_helpers():
    """bench.py's host-side pieces that need no GPU: every named workload resolves to a config with the sampling its name states,
    and the committed PMC passes are found per workload and kernel (roofline.traffic)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in bench.WORKLOADS:
        cfg, n_rays = bench.workload(name)
        tr = cfg["training"]
        assert n_rays > 0 and tr["n_samples_d"] + tr["n_range_d"] > 1
        if "x" in name.split("_")[-1] or "x" in name.split("_")[-2]:
            tok = [t for t in name.split("_") if "x" in t and t[0].isdigit()][0]
            n, s_ = (int(v) for v in tok.split("x"))
            assert n == n_rays and s_ == tr["n_samples_d"] + tr["n_range_d"], name
    prof, src = bench.pmc_profile("office0_2048x128", "k_query_fwd<color>")
    assert src is not None and prof["traffic_bytes"] > 0 and prof["fetch_bytes"] > 0
    prof, src = bench.pmc_profile("unit1024_T22_131072x43", "k_query_fwd<color>")
    assert src is not None and prof["traffic_bytes"] > 1e9            # the 2^22-entry table is HBM resident
    assert bench.pmc_profile("office0_2048x128", "no such kernel") == ({}, None)
    assert bench.physical_cores() >= 1


def test_no_kernel_spills_to_scratch(built_lib):
    """Every kernel of the built device code object: no vector register spilled, no private (scratch) segment -- read from the code
    object's own metadata notes, so a change that pushes a hot kernel over its register budget fails HERE, before anyone times it.
    (Spilled SCALAR registers live in lanes of a vector register, not in memory; they are listed by tools/kernel_resources.sh.)"""
    from naruto_amd import _lib
    res = _lib.kernel_resources()
    hot = [k for k in res if re.match(r"k_(query_|render_|hash_scatter_|bin_|loss_|sample_|composite_|bwd_finish|adam_)", k)]
    for must in ("k_query_fwd<true,256,false>", "k_query_fwd<true,128,false>", "k_query_fwd<true,256,true>", "k_query_fwd_loss<false,true>", "k_query_fwd_loss<true,true>", "k_query_fwd_loss<false,false>", "k_query_fwd_loss<true,false>",
                 "k_query_fwd_loss_packed<false,8>", "k_query_fwd_loss_packed<true,8>", "k_query_bwd", "k_query_bwd_bf", "k_render_fwd<false>",
                 "k_render_fwd_packed<false,256>", "k_render_fwd_packed<true,256>", "k_render_fwd_packed<false,512>", "k_render_fwd_packed<true,512>", "k_query_fwd_list<false>", "k_query_fwd_list<true>", "k_sort_count", "k_sort_fill", "k_sort_more",
                 "k_hash_scatter_lds", "k_bin_count", "k_bin_fill<1024>", "k_bin_fill<512>", "k_bin_apply",
                 "k_loss_bwd_fused", "k_bwd_finish"):
        assert must in res, f"{must} not found in the code object (have: {sorted(res)[:8]} ...)"
    assert len(hot) >= 30
    bad = {k: v for k, v in res.items() if v.get("vgpr_spill_count", 0) != 0 or v.get("private_segment_fixed_size", 0) != 0}
    assert not bad, f"kernels with spilled vector registers / scratch: {bad}"
    # the occupancy each hot kernel was written for: registers per lane within the budget of its waves per SIMD (512 / waves)
    budget = {"k_query_fwd<true,256,false>": 256, "k_query_fwd<true,128,false>": 256, "k_query_fwd<true,256,true>": 256, "k_query_fwd_loss<false,true>": 256, "k_query_fwd_loss<true,true>": 256, "k_query_fwd_loss<false,false>": 256, "k_query_fwd_loss<true,false>": 256,
              "k_query_fwd_loss_packed<false,8>": 256, "k_query_fwd_loss_packed<true,8>": 256, "k_bin_fill<1024>": 128, "k_query_fwd_bf<true,256,false>": 256,
              "k_render_fwd_packed<false,256>": 256, "k_render_fwd_packed<true,256>": 256, "k_render_fwd_packed<false,512>": 256, "k_render_fwd_packed<true,512>": 256, "k_query_fwd_list<false>": 256, "k_query_fwd_list<true>": 256,
              "k_hash_scatter_lds": 128, "k_query_bwd": 512, "k_query_bwd_bf": 512}
    for k, b in budget.items():
        assert res[k]["vgpr_count"] <= b, (k, res[k])            # .vgpr_count is the unified total (architectural + accumulation registers)
    # Spilled SCALAR registers (round 5): they live in lanes of a vector register -- v_writelane / v_readlane around the code that needs the
    # scalar file -- not in memory, so they cost issue slots, not traffic; the kernels of the mapping iteration are held to a budget so that
    # a change that doubles them is seen.  (End of round 4: k_query_fwd_loss<false,true> 214, k_hash_scatter_lds 186, k_query_bwd 18; the
    # walk's loss-stage / sampling arguments now go through LDS instead of being held in scalar registers across the tile loop.)
    # k_hash_scatter_lds: 127 until the uncertainty units' scan + compaction went in, 307 with it -- static counts over 177 000 instructions (the
    # level loop is unrolled per level; a workgroup runs one sixteenth of it), spread evenly over the level units' code whatever the new role's
    # own form (inlined, its own function, its uniform values through LDS, its queue in static or dynamic LDS: all measured, 307 each time, 435 as
    # a function); the level units' workgroups did not slow down (hashed 49 -> 46 us, profiles/r05_scatter_timeline.txt).
    # Round 6, the evidence the review asked for (profiles/r06_scatter_sgpr_spill_roles.txt, tools/scatter_sgpr_ab.sh: both builds, per workgroup type, three
    # batch sizes): dense-level units 2 - 3 % slower with 307 than with 127, hashed units 5 - 6 % faster, the launch 51.2 vs 57.7 / 95.2 vs 91.8 / 1302 vs 1390 us.
    # k_query_fwd_loss<false,true> (round 6): 104 -> 132 with the HALF tile -- a second instantiation of the gather and of the x3 matrix phase in the tile
    # loop (the count is static spill SITES, per instantiation) -- while the kernel went 62.7 -> 60.4 us and every phase of its per-wave timeline
    # (tools/walk_timeline.py, profiles/r06_walk_timeline_*.txt) is as long or shorter than before; the ray constants and PointSrc went to LDS to
    # keep it there (138 with them in scalar registers).
    # k_query_fwd_loss<true,true> (the bf16 walk): 115 -> 143 with round 6's per-ray prefetch (ray constants through LDS, depths from the image); its step time
    # did not move (0.167 ms before and after).
    sgpr_budget = {"k_query_fwd_loss<false,true>": 140, "k_query_fwd_loss<true,true>": 150, "k_query_fwd_loss_short<false>": 128, "k_query_fwd_loss_short<true>": 128,
                   "k_hash_scatter_lds": 320, "k_query_bwd": 32, "k_query_bwd_bf": 32, "k_loss_bwd_fused": 0, "k_bwd_finish": 0,
                   "k_query_fwd<true,512,false>": 16, "k_query_fwd<true,256,false>": 16}
    for k, b in sgpr_budget.items():
        assert k in res, f"{k} not found in the code object"
        assert res[k].get("sgpr_spill_count", 0) <= b, (k, res[k])
