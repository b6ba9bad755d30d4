"""Whole-iteration hipGraph capture of a caller's OWN mapping loop body.

``GraphedIteration`` records one iteration of a caller object's ``ba_iteration`` (the reference's ``global_BA`` loop body,
coslam.py:361-399: model.forward -> get_loss_from_ret -> loss.backward -> optimiser steps) with torch's graph capture and replays it.
It is the third of the opt-in changes INTEGRATION.md lists for an unchanged caller; the caller object needs ``model`` (a
``NarutoFieldHIP``), ``config``, ``map_optimizer`` / ``uncert_optim`` (``FusedAdam``), ``smoothness_mode == "fused"`` and
``ba_iteration(i, rays_o, rays_d, target_s, target_d)`` -- tools/dropin_caller.py::DropInCaller is the restatement of the reference's
caller the tests and bench.py drive it with."""

from __future__ import annotations

import torch


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

    def __init__(self, caller, n_rays: int, warmup: int = 3):
        from .trainer import FusedAdam
        mp = caller.config['mapping']
        # the two recorded variants are "i + 1 not a multiple of 5" and "a multiple of 5" with the mapping optimiser stepping in BOTH: any other
        # stepping pattern (coslam.py:370-376) would replay the wrong host decisions for most values of i
        if int(mp.get('map_accum_step', 1)) != 1 or int(mp.get('map_wait_step', 0)) != 0:
            raise NotImplementedError("GraphedIteration records a loop body whose mapping optimiser steps every iteration: "
                                      "mapping.map_accum_step must be 1 and mapping.map_wait_step 0 (every shipped config)")
        assert isinstance(caller.map_optimizer, FusedAdam) and isinstance(caller.uncert_optim, FusedAdam) and caller.smoothness_mode == "fused", \
            "whole-iteration capture needs optimizer='fused' (BOTH optimisers: the mapping one and the uncertainty grid's -- their step counts " \
            "live on the device and are snapshotted around the warm-up) and smoothness='fused' (no host work inside the loop body)"
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
