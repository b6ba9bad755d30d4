#!/usr/bin/env python3
"""Timings of the "next" rows (N1 active ray sampler, N2 ray transform, N3 planner aggregation, A9 map volumes) at the
office_0 sizes, HIP events on the launch stream.  Run on the GPU box:  python tools/time_next_rows.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naruto_amd import config as C                     # noqa: E402
from naruto_amd import synthetic as syn                # noqa: E402
from naruto_amd.active_ray_sampler import ActiveRaySamplerHIP, rays_to_world   # noqa: E402
from naruto_amd.planner_aggregation import GoalSpaceAggregatorHIP              # noqa: E402


def events_ms(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda", 0)
    cfg = C.office0_config()
    bound = cfg["mapping"]["bound"]
    rs = np.random.RandomState(0)
    out = {}
    # N3: 49 x 56 x 35 volumes, goal space 25 x 28 x 3, top_k 4000 -> 300 targets (configs/default.py:93-98)
    ag = GoalSpaceAggregatorHIP(bound, 0.1, device=dev)
    dims = (ag.Nx, ag.Ny, ag.Nz)
    X, Y, Z = np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")
    sdf = (np.minimum.reduce([X - 1.5, dims[0] - 2.5 - X, Y - 1.5, dims[1] - 2.5 - Y, Z - 1.5, dims[2] - 2.5 - Z]) * 0.5).astype(np.float32)
    unc = (rs.uniform(0.01, 3.0, dims) * ((sdf >= 0) & (sdf < 0.5))).astype(np.float32)
    u_d, s_d = torch.from_numpy(unc).to(dev), torch.from_numpy(sdf).to(dev)
    out["N3 uncertainty_aggregation_v2 (2100 goals x 300 targets, device volumes)"] = events_ms(lambda: ag.uncertainty_aggregation_v2([u_d, s_d]))
    out["N3 from numpy volumes (incl. 2 x 384 KB H2D)"] = events_ms(lambda: ag.uncertainty_aggregation_v2([unc, sdf]))
    # N1: 2048 + 2048*4 oversampled + 100 current rays, K = 500
    cfg["mapping"]["sample"] = 2048
    smp = ActiveRaySamplerHIP(config=cfg, num_uncert_sample=500, oversample_mul=4)
    n_cur = 100
    n = smp.oversample_num + n_cur
    rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(n, bound, seed=0).items()}
    out[f"N1 ActiveRaySamplerHIP.sample_rays ({n} rays -> {2048 + n_cur // 4})"] = events_ms(
        lambda: smp.sample_rays(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"], list(range(n_cur)), u_d, bound))
    # N2
    poses = torch.eye(4, device=dev).repeat(64, 1, 1)
    ids = torch.randint(0, 64, (n,), device=dev)
    dcam = torch.randn(n, 3, device=dev)
    out[f"N2 rays_to_world ({n} rays)"] = events_ms(lambda: rays_to_world(dcam, ids, poses))
    # N4: save_mesh at mesh.voxel_final = 0.02 m over the office_0 bound (241 x 276 x 171 lattice points)
    import time
    from naruto_amd import mesh as M
    from naruto_amd.field import NarutoFieldHIP
    bbox = torch.tensor(bound, dtype=torch.float32, device=dev)
    model = NarutoFieldHIP(cfg, bbox).to(dev).eval()
    model.get_uncert_grid(0.1)
    w = syn.mlp_weights(0)
    with torch.no_grad():
        model.embed_fn.params.copy_(torch.from_numpy(syn.closed_form_table(model.embed_fn.params.numel(), 0.02)))
        model.decoder.sdf_net.model[0].weight.copy_(torch.from_numpy(w["sdf_w0"]))
        model.decoder.sdf_net.model[2].weight.copy_(torch.from_numpy(w["sdf_w1"]))
        model.decoder.color_net.model[0].weight.copy_(torch.from_numpy(w["col_w0"]))
        model.decoder.color_net.model[2].weight.copy_(torch.from_numpy(w["col_w1"]))
    tx, ty, tz = M.get_voxels(bound[0][1], bound[0][0], bound[1][1], bound[1][0], bound[2][1], bound[2][0], 0.02)
    axes = [((t - bbox[i, 0].cpu()) / (bbox[i, 1].cpu() - bbox[i, 0].cpu())).to(dev) for i, t in enumerate((tx, ty, tz))]
    n_lat = tx.numel() * ty.numel() * tz.numel()
    out[f"N4 lattice_points ({tx.numel()} x {ty.numel()} x {tz.numel()} = {n_lat})"] = events_ms(lambda: M.lattice_points(*axes), 5)
    flat = M.lattice_points(*axes)
    out["N4 query_sdf over the lattice"] = events_ms(lambda: model.query_sdf(flat[:, None, :]), 3)
    with torch.no_grad():
        vol = model.query_sdf(flat[:, None, :]).reshape(tx.numel(), ty.numel(), tz.numel()).contiguous()
    iso = float(vol.median())
    del flat
    v, f = M.marching_cubes(vol, iso, 3.0)
    out[f"N4 marching cubes -> {len(v)} vertices, {len(f)} triangles (incl. the size read-back)"] = events_ms(lambda: M.marching_cubes(vol, iso, 3.0), 5)
    # a smooth surface of the size a room has (the closed-form field above is far busier than a trained map)
    g = torch.stack(torch.meshgrid(torch.linspace(-1, 1, tx.numel(), device=dev), torch.linspace(-1, 1, ty.numel(), device=dev),
                                   torch.linspace(-1, 1, tz.numel(), device=dev), indexing="ij"), -1)
    box = (g.abs() - 0.9).amax(-1).float().contiguous()
    vb, fb = M.marching_cubes(box, 0.0, 3.0)
    out[f"N4 marching cubes, box surface -> {len(vb)} vertices, {len(fb)} triangles"] = events_ms(lambda: M.marching_cubes(box, 0.0, 3.0), 5)
    del g, box
    for tag, cf in (("query_color", model.query_color), ("uncertainty", None)):
        for rep in range(3):                                   # the first call also pays the allocator's hipMallocs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mesh = M.extract_mesh(model.query_sdf, cfg, bbox, color_func=cf, voxel_size=0.02, isolevel=iso)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if rep == 0:
                out[f"N4 extract_mesh end to end, {tag} colours, first call, wall clock"] = dt
        out[f"N4 extract_mesh end to end, {tag} colours, third call, wall clock incl. D2H of {len(mesh.vertices)} vertices"] = dt
    print(json.dumps({k: round(v * 1e3, 2) for k, v in out.items()}, indent=1), "(microseconds)")


if __name__ == "__main__":
    main()
