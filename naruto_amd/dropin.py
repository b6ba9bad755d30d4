"""The reference's OWN mapping loop body, restated against any model with the reference's attribute surface.

This is what the reference's ``CoSLAMNaruto`` executes per iteration after INTEGRATION.md's two-line swap
(``self.model = NarutoFieldHIP(...)`` at reference src/slam/coslam/coslam.py:65) -- caller code this repository does NOT replace:

    ret  = self.model.forward(rays_o, rays_d, target_s, target_d)          coslam.py:364
    loss = self.get_loss_from_ret(ret, smooth=True)                         coslam.py:366, 154-174 (ten scalar torch ops)
           ... self.smoothness(...) through self.model.query_sdf(embed=True)   Co-SLAM CoSLAM.smoothness [not in tree]
    loss.backward(retain_graph=True)                                        coslam.py:368
    self.map_optimizer.step(); self.map_optimizer.zero_grad()               coslam.py:370-376 (torch.optim.Adam, create_optimizer :409-419)
    every 5th: self.uncert_optim.step(); self.uncert_optim.zero_grad()      coslam.py:397-399

``bench.py --path dropin`` times it and the GPU tests check it against the oracle driven the same way; it is also the template for
the three optional one-line changes INTEGRATION.md lists after the swap (``optimizer="fused"``: ``optim.Adam`` -> ``FusedAdam``;
``smoothness="fused"``: Co-SLAM's torch smoothness -> ``naruto_amd.trainer.smoothness``).  Nothing here is on the product's own fast
path (``MappingTrainer``)."""

from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.optim as optim


def coordinates(voxel_dim: int, device, flatten: bool = False) -> torch.Tensor:
    """Co-SLAM utils.coordinates [not in tree]: integer lattice [n,n,n,3]."""
    ax = torch.arange(0, voxel_dim, dtype=torch.long, device=device)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    if not flatten:
        return torch.stack([x, y, z], dim=-1)
    return torch.stack((x.flatten(), y.flatten(), z.flatten()))


class DropInCaller:
    """``CoSLAMNaruto``'s optimisation state and loop body (create_optimizer, init_uncert_grid_optim, get_loss_from_ret, smoothness,
    the global_BA iteration) around ``model`` -- a ``NarutoFieldHIP`` here, the reference's ``JointEncodingNaruto`` there."""

    def __init__(self, model, config: Dict, uncert_voxel: float = 0.1, optimizer: str = "torch", smoothness: str = "reference"):
        assert optimizer in ("torch", "fused") and smoothness in ("reference", "fused")
        self.model, self.config = model, config
        self.bounding_box = model.bounding_box
        self.smoothness_mode = smoothness
        adam = optim.Adam
        if optimizer == "fused":
            from .trainer import FusedAdam
            adam = FusedAdam
        # init_uncert_grid_optim, coslam.py:240-243
        self.uncert_optim = adam(params=[model.get_uncert_grid(uncert_voxel)], lr=1)
        # create_optimizer, coslam.py:409-419
        trainable_parameters = [{'params': model.decoder.parameters(), 'weight_decay': 1e-6, 'lr': config['mapping']['lr_decoder']},
                                {'params': model.embed_fn.parameters(), 'eps': 1e-15, 'lr': config['mapping']['lr_embed']}]
        self.map_optimizer = adam(trainable_parameters, betas=(0.9, 0.99))

    # Co-SLAM CoSLAM.smoothness [not in tree]: the lattice is built on the HOST and copied per call, the TV term is six sliced torch ops
    def smoothness(self, sample_points=256, voxel_size=0.1, margin=0.05, color=False):
        if self.smoothness_mode == "fused":
            from .trainer import smoothness as fused_smoothness
            return fused_smoothness(self.model, self.config, sample_points, voxel_size, margin)
        bb = self.bounding_box
        volume = bb[:, 1] - bb[:, 0]
        grid_size = (sample_points - 1) * voxel_size
        offset_max = bb[:, 1] - bb[:, 0] - grid_size - 2 * margin
        offset = torch.rand(3).to(offset_max) * offset_max + margin
        coords = coordinates(sample_points - 1, 'cpu', flatten=False).float().to(volume)
        pts = (coords + torch.rand((1, 1, 1, 3)).to(volume)) * voxel_size + bb[:, 0] + offset
        pts_tcnn = (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        sdf = self.model.query_sdf(pts_tcnn, embed=True)
        tv_x = torch.pow(sdf[1:, ...] - sdf[:-1, ...], 2).sum()
        tv_y = torch.pow(sdf[:, 1:, ...] - sdf[:, :-1, ...], 2).sum()
        tv_z = torch.pow(sdf[:, :, 1:, ...] - sdf[:, :, :-1, ...], 2).sum()
        return (tv_x + tv_y + tv_z) / (sample_points ** 3)

    def get_loss_from_ret(self, ret, rgb=True, sdf=True, depth=True, fs=True, uncert=True, smooth=False):
        """coslam.py:154-174."""
        tr = self.config['training']
        loss = 0
        if rgb:
            loss += tr['rgb_weight'] * ret['rgb_loss']
        if depth:
            loss += tr['depth_weight'] * ret['depth_loss']
        if sdf:
            loss += tr['sdf_weight'] * ret["sdf_loss"]
        if fs:
            loss += tr['fs_weight'] * ret["fs_loss"]
        if smooth and tr['smooth_weight'] > 0:
            loss += tr['smooth_weight'] * self.smoothness(tr['smooth_pts'], tr['smooth_vox'], margin=tr['smooth_margin'])
        if uncert and (self.config['decoder']['pred_uncert'] or self.config['decoder']['uncert_grid']):
            loss += tr['uncert_weight'] * ret['uncert_loss']
        return loss

    def first_frame_iteration(self, rays_o, rays_d, target_s, target_d):
        """One iteration of first_frame_mapping's loop, coslam.py:200-217."""
        self.map_optimizer.zero_grad()
        ret = self.model.forward(rays_o, rays_d, target_s, target_d)
        loss = self.get_loss_from_ret(ret)
        loss.backward()
        self.map_optimizer.step()
        return ret, loss

    def ba_iteration(self, i: int, rays_o, rays_d, target_s, target_d):
        """Iteration ``i`` (0-based) of global_BA's loop, coslam.py:361-399 (tracking is off in every shipped config: no pose optimiser)."""
        mp = self.config['mapping']
        ret = self.model.forward(rays_o, rays_d, target_s, target_d)
        loss = self.get_loss_from_ret(ret, smooth=True)
        loss.backward(retain_graph=True)
        if (i + 1) % mp["map_accum_step"] == 0:
            if (i + 1) > mp["map_wait_step"]:
                self.map_optimizer.step()
            self.map_optimizer.zero_grad()
        if self.config['decoder']['uncert_grid'] and (i + 1) % 5 == 0:
            self.uncert_optim.step()
            self.uncert_optim.zero_grad()
        return ret, loss


class GraphedIteration:
    """The caller's OWN loop body (``DropInCaller.ba_iteration``: model.forward -> get_loss_from_ret -> loss.backward -> optimiser
    steps) recorded into hipGraphs with torch's whole-iteration capture and replayed: what is left of the unchanged caller's cost
    once the host stops issuing ~60 small launches per iteration.  Needs a loop body without host work: ``optimizer="fused"`` (the
    device-side step count) and ``smoothness="fused"`` (Co-SLAM's smoothness builds its lattice on the CPU); the autograd nodes of
    this library launch on the capturing stream and draw their random numbers from device-side counters, so they record as they are.

        step = GraphedIteration(caller, n_rays)            # once per ray count: warm-up + capture, training state restored afterwards
        ret, loss = step(i, rays_o, rays_d, target_s, target_d)   # instead of caller.ba_iteration(i, ...)

    Two graphs (with / without the uncertainty grid's Adam step of every 5th iteration, coslam.py:397-399); inputs are copied into
    static buffers, ``ret`` / ``loss`` are the graphs' static outputs (overwritten by the next replay)."""

    def __init__(self, caller: "DropInCaller", n_rays: int, warmup: int = 3):
        from .trainer import FusedAdam
        assert isinstance(caller.map_optimizer, FusedAdam) and caller.smoothness_mode == "fused", \
            "whole-iteration capture needs optimizer='fused' and smoothness='fused' (no host work inside the loop body)"
        self.caller = caller
        m = caller.model
        dev = m.embed_fn.params.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.rays_o, self.rays_d, self.target_s = torch.zeros(n_rays, 3, **f32), torch.zeros(n_rays, 3, **f32), torch.zeros(n_rays, 3, **f32)
        self.target_d = torch.ones(n_rays, 1, **f32)
        self.rays_d[:, 2] = 1.0
        params = list(m.parameters())
        opts = (caller.map_optimizer, caller.uncert_optim)
        snap_p = [p.detach().clone() for p in params]
        snap_o = [(o.step_dev.clone(), [(s_['exp_avg'].clone(), s_['exp_avg_sq'].clone(), s_['lag']) for s_ in o.state.values()], o._n_steps) for o in opts]
        rng = None if m._rng_state is None else m._rng_state.clone()
        args = (self.rays_o, self.rays_d, self.target_s, self.target_d)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for k in range(warmup):
                caller.ba_iteration(4 if k == warmup - 1 else 0, *args)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        if rng is None and m._rng_state is not None:
            rng = m._rng_state.clone()
            rng[1] = 0
        self.graphs, self.out, pool = [], [], None
        # Inside a capture nothing may depend on host-side state that changes between replays: every .grad has to be a PERSISTENT tensor
        # that autograd adds into and the optimiser zeroes in place (FusedAdam.zero_grad does so while a stream is capturing) -- the
        # uncertainty grid's gradient really accumulates over five replays (coslam.py:397-399).
        for p in params:
            p.grad = torch.zeros_like(p)
        for variant in (0, 4):                       # iteration index with (i + 1) % 5 != 0 / == 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                ret, loss = caller.ba_iteration(variant, *args)
            pool = g.pool()
            self.graphs.append(g)
            self.out.append((ret, loss))
        with torch.no_grad():
            for p, q in zip(params, snap_p):
                p.copy_(q)
                p.grad.zero_()
            for o, (sd, mv, ns) in zip(opts, snap_o):
                o.step_dev.copy_(sd)
                o._n_steps = ns
                for s_, (m0, v0, lag) in zip(o.state.values(), mv):
                    s_['exp_avg'].copy_(m0)
                    s_['exp_avg_sq'].copy_(v0)
                    s_['lag'] = lag
            if rng is not None:
                m._rng_state.copy_(rng)

    def __call__(self, i: int, rays_o, rays_d, target_s, target_d):
        self.rays_o.copy_(rays_o, non_blocking=True)
        self.rays_d.copy_(rays_d, non_blocking=True)
        self.target_s.copy_(target_s, non_blocking=True)
        self.target_d.copy_(target_d.reshape(self.target_d.shape), non_blocking=True)
        k = 1 if (i + 1) % 5 == 0 else 0
        self.graphs[k].replay()
        # the reference's per-forward assertion (scene_rep.py:280): the replayed kernels fold every iteration's minimum into one device word;
        # read it back every few replays, asynchronously
        m = self.caller.model
        self._n = getattr(self, "_n", 0) + 1
        if self._n % m.assert_every == 0:
            m.note_min_uncert(m.min_uncert_running())
            m.check_asserts()
        return self.out[k]
