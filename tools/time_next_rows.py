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
    print(json.dumps({k: round(v * 1e3, 2) for k, v in out.items()}, indent=1), "(microseconds)")


if __name__ == "__main__":
    main()
