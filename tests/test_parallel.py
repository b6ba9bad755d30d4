"""world_size-2 gloo tests (CPU) of the data-parallel protocol: ray sharding + loss-sum all-reduce + flat
gradient all-reduce reproduce the single-process gradient of the whole batch.

The GPU kernels cannot run here, so the per-rank compute is done with the oracle, arranged exactly like the
product path: per-rank loss SUMS (the 16-slot vector of naruto_loss_sums), all-reduce, normalisation by the
global counts, backward, gradient all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from naruto_amd import parallel, synthetic as syn
from oracle import spec_torch as S

import helpers as H


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 2048, 2049):
        for w in (1, 2, 3, 8):
            pieces = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


def _loss_sums(rend, target_rgb, target_d, cfg):
    """The oracle's version of naruto_loss_sums (slots as in include/naruto_hip.h)."""
    tr = cfg["training"]
    td = target_d.squeeze(-1)
    valid = (td > 0) & (td < cfg["cam"]["depth_trunc"])
    z, sdf = rend["z_vals"], rend["raw"][..., 3]
    trunc = tr["trunc"] * cfg["data"]["sc_factor"]
    front = (z < (target_d - trunc)).float()
    back = (z > (target_d + trunc)).float()
    sm = (1 - front) * (1 - back) * (target_d > 0).float()
    um = rend["uncert_map"]
    s = torch.zeros(16, dtype=torch.float64)
    s[0] = ((rend["rgb"] - target_rgb) ** 2).sum()
    s[1] = ((rend["depth"] - td)[valid] ** 2).sum()
    s[2] = valid.sum()
    s[3] = ((sdf * front - front) ** 2).sum()
    s[4] = front.sum()
    s[5] = (((z + sdf * trunc) * sm - target_d * sm) ** 2).sum()
    s[6] = (sm != 0).sum()
    s[7] = (1 / (2 * (um[valid] + 1e-9))).sum()
    s[8] = torch.log(um[valid] + 1e-9).sum()
    s[9] = um.min()
    return s


def _total_from_sums(s, n_total, S_tot, tr):
    nv = s[2]
    ns = s[4] + s[6]
    rgb = s[0] / (3 * n_total)
    depth = s[1] / nv
    fs = s[3] / (n_total * S_tot) * (1 - s[4] / ns)
    sdf = s[5] / (n_total * S_tot) * (1 - s[6] / ns)
    unc = (s[7] / nv) * (s[1] / nv) + 0.5 * s[8] / nv
    return tr["rgb_weight"] * rgb + tr["depth_weight"] * depth + tr["sdf_weight"] * sdf + tr["fs_weight"] * fs + tr["uncert_weight"] * unc


def _worker(rank, world, port, n_rays, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = H.office_cfg(12)
    ora = H.make_oracle(cfg, 0.2, 8)
    rays = syn.random_rays(n_rays, cfg["mapping"]["bound"], seed=77, zero_depth_frac=0.2)
    t = [torch.from_numpy(rays[k]) for k in ("rays_o", "rays_d", "target_rgb", "target_d")]
    ro, rd, trgb, tdep = parallel.shard_rays(t, rank, world)
    ora.eval()
    rend = ora.forward(ro, rd, trgb, tdep)                      # differentiable render of this rank's shard
    sums = _loss_sums(rend, trgb, tdep, cfg)
    # forward/backward split exactly as in the product: the all-reduced sums are constants of the backward
    # for the COUNT slots, and carry gradient for the value slots -- autograd through a differentiable all-reduce
    # is emulated by d(total)/d(local sums) evaluated at the global sums
    g = sums.detach().clone()
    parallel.allreduce_loss_sums(g, None)
    g.requires_grad_(True)
    S_tot = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    total = _total_from_sums(g, n_rays, S_tot, cfg["training"])
    (d_sums,) = torch.autograd.grad(total, g)
    d_sums[[2, 4, 6, 9]] = 0                                      # counts / min are not differentiable
    sums.backward(d_sums)
    params = [ora.table, ora.sdf_w0, ora.sdf_w1, ora.col_w0, ora.col_w1, ora.uncert_grid]
    parallel.allreduce_grads(params, None)
    if rank == 0:
        torch.save({"total": total.detach(), "grads": [p.grad.clone() for p in params], "sums": g.detach()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol_equals_single_process(tmp_path):
    n_rays = 96
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, n_rays, out), nprocs=2, join=True)
    got = torch.load(out)
    cfg = H.office_cfg(12)
    ora = H.make_oracle(cfg, 0.2, 8).train()
    rays = syn.random_rays(n_rays, cfg["mapping"]["bound"], seed=77, zero_depth_frac=0.2)
    t = [torch.from_numpy(rays[k]) for k in ("rays_o", "rays_d", "target_rgb", "target_d")]
    ret = ora.forward(*t)
    total = S.total_loss(ret, cfg["training"])
    total.backward()
    assert abs(float(got["total"]) - float(total)) < 1e-5 * max(1.0, abs(float(total)))
    want = [ora.table.grad, ora.sdf_w0.grad, ora.sdf_w1.grad, ora.col_w0.grad, ora.col_w1.grad, ora.uncert_grid.grad]
    for a, b in zip(got["grads"], want):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9


def _shard_opt_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    n = 1003                                             # not a multiple of the world size: the bucket is padded
    c = -(-n // (4 * world)) * 4
    p_full = torch.randn(n)
    g_local = torch.randn(n, generator=torch.Generator().manual_seed(10 + rank))
    # replicated: all-reduce, the identical Adam step everywhere
    g_sum = g_local.clone()
    parallel.all_reduce_sum(g_sum)
    pa = torch.nn.Parameter(p_full.clone())
    oa = torch.optim.Adam([pa], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    pa.grad = g_sum
    oa.step()
    # sharded: reduce-scatter, Adam on this rank's slice, all-gather
    store = torch.zeros(c * world)
    store[:n] = p_full
    bucket = torch.zeros(c * world)
    bucket[:n] = g_local
    sl = torch.nn.Parameter(store[rank * c:(rank + 1) * c].clone())
    g_slice = torch.empty(c)
    parallel.reduce_scatter_sum(bucket, g_slice)
    ob = torch.optim.Adam([sl], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    sl.grad = g_slice
    ob.step()
    parallel.all_gather_into(store, sl.detach().clone())
    want_slice = torch.cat([g_sum, torch.zeros(c * world - n)])[rank * c:(rank + 1) * c]
    if world == 2:           # two operands: one order of summation -> the same bits
        ok = torch.equal(store[:n], pa.detach()) and torch.equal(g_slice, want_slice)
    else:                    # three or more: the two collectives may add in different orders
        # a first Adam step moves every parameter by ~lr whatever its gradient (eps = 1e-15), so a loose bound on the parameters would
        # pass with every update missing or of the wrong sign: compare the elements whose summed gradient is well above the summation
        # noise (|g| > 1e-3: the two summation orders differ by ~1e-7 there, the update lr * g / (|g| + eps') by ~1e-6 relative) tightly,
        # and require the rest to have moved by lr in SOME direction
        big = g_sum.abs() > 1e-3
        moved = (store[:n] - p_full).abs()
        ok = (torch.allclose(g_slice, want_slice, rtol=1e-6, atol=1e-6) and bool(big.sum() > n // 2)
              and float((store[:n] - pa.detach())[big].abs().max()) <= 1e-6
              and float((moved - 0.01).abs().max()) <= 1e-4)
    if rank == 0:
        torch.save({"ok": bool(ok), "pad_untouched": bool((store[n:] == 0).all())}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_protocol(tmp_path, world):
    """reduce_scatter_sum + Adam on the rank's slice + all_gather_into (the sharded table optimiser of MappingTrainer for large tables)
    equals all_reduce_sum + the full Adam step on every rank (bit for bit over two ranks; to summation order beyond), over a padded
    bucket and a world size that does not divide the table."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "shard.pt")
    mp.spawn(_shard_opt_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out)
    assert r["ok"] and r["pad_untouched"]
