"""The mapping-iteration harness around the field (the "train step" of the headline metric).

Mirrors the parts of the reference's SLAM driver that sit directly on the hot path (reference
src/slam/coslam/coslam.py): ``create_optimizer`` (:409-419), ``init_uncert_grid_optim`` (:240-243),
``get_loss_from_ret`` (:154-174), Co-SLAM's ``smoothness``, and the iteration body of
``first_frame_mapping`` (:200-219) / ``global_BA`` (:361-399):

    zero_grad -> model.forward -> weighted loss -> backward -> Adam.step   (+ uncert Adam every 5th iter)

Ray assembly (keyframe database, active ray sampler) stays with the caller, as in the reference.
"""

from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.optim as optim

from . import parallel
from .field import NarutoFieldHIP


def create_optimizer(model: NarutoFieldHIP, config: Dict) -> optim.Adam:
    """coslam.py:409-419 (oneGrid: no colour-grid group)."""
    groups = [{'params': model.decoder.parameters(), 'weight_decay': 1e-6, 'lr': config['mapping']['lr_decoder']},
              {'params': model.embed_fn.parameters(), 'eps': 1e-15, 'lr': config['mapping']['lr_embed']}]
    return optim.Adam(groups, betas=(0.9, 0.99))


def init_uncert_grid_optim(model: NarutoFieldHIP, voxel_size: float = 0.1) -> optim.Adam:
    """coslam.py:240-243."""
    return optim.Adam(params=[model.get_uncert_grid(voxel_size)], lr=1)


def smoothness(model: NarutoFieldHIP, config: Dict, sample_points: int = 256, voxel_size: float = 0.1, margin: float = 0.05,
               offset_rand: Optional[torch.Tensor] = None, jitter_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Co-SLAM CoSLAM.smoothness: total variation of the hash features on a (sample_points-1)^3 lattice placed at a
    random offset (fused: lattice points + gather, TV loss + its feature gradient, table scatter in the backward)."""
    from . import ops
    dev = model.embed_fn.params.device
    r = torch.rand(6, device=dev)                     # drawn on the device: no host sync per iteration
    if offset_rand is not None:
        r[:3] = offset_rand.to(dev).reshape(3)
    if jitter_rand is not None:
        r[3:] = jitter_rand.to(dev).reshape(3)
    return ops.smoothness(model._handle(), model.embed_fn.params, sample_points, voxel_size, margin, r)


def get_loss_from_ret(model: NarutoFieldHIP, config: Dict, ret: Dict, rgb=True, sdf=True, depth=True, fs=True, uncert=True,
                      smooth=False) -> torch.Tensor:
    """coslam.py:154-174."""
    tr = config['training']
    loss = 0
    if rgb:
        loss = loss + tr['rgb_weight'] * ret['rgb_loss']
    if depth:
        loss = loss + tr['depth_weight'] * ret['depth_loss']
    if sdf:
        loss = loss + tr['sdf_weight'] * ret["sdf_loss"]
    if fs:
        loss = loss + tr['fs_weight'] * ret["fs_loss"]
    if smooth and tr['smooth_weight'] > 0:
        loss = loss + tr['smooth_weight'] * smoothness(model, config, tr['smooth_pts'], tr['smooth_vox'], margin=tr['smooth_margin'])
    if uncert and (config['decoder']['pred_uncert'] or config['decoder']['uncert_grid']):
        loss = loss + tr['uncert_weight'] * ret['uncert_loss']
    return loss


def unpack_rays(flat: torch.Tensor, n_rays: int):
    """Views (rays_o [N,3], rays_d [N,3], target_rgb [N,3], target_d [N,1]) into one flat [10 N] buffer."""
    n = n_rays
    return (flat[0:3 * n].view(n, 3), flat[3 * n:6 * n].view(n, 3), flat[6 * n:9 * n].view(n, 3), flat[9 * n:10 * n].view(n, 1))


def pack_rays(rays_o, rays_d, target_rgb, target_d):
    """One contiguous buffer holding a ray batch; returns the four views.  A batch built this way reaches a
    graph-captured trainer with a single device copy."""
    n = rays_o.shape[0]
    flat = torch.cat([rays_o.reshape(-1), rays_d.reshape(-1), target_rgb.reshape(-1), target_d.reshape(-1)]).contiguous()
    return unpack_rays(flat, n)


class FusedAdam(optim.Optimizer):
    """``torch.optim.Adam`` (amsgrad off, L2 weight decay, not maximize) as ONE HIP launch over all parameter tensors -- a
    ``torch.optim.Optimizer`` subclass with Adam's constructor, so the reference's ``create_optimizer`` / ``init_uncert_grid_optim``
    (coslam.py:240-243, 409-419) take it by changing ``optim.Adam`` to ``naruto_amd.FusedAdam`` and nothing else: same param-group keys,
    ``zero_grad`` / ``step`` / ``state_dict`` / ``add_param_group``.  The step count lives on the device, so a captured launch stays
    valid under hipGraph replay.  State per parameter: ``exp_avg``, ``exp_avg_sq`` (created with the optimiser)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = self.param_groups[0]['params'][0].device
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev)       # {completed steps, ticket word of k_adam_multi}
        # int32[1] device word that already holds THIS step's 1-based number when step() runs (MappingTrainer's iteration
        # counter, advanced by the forward): no counting in the optimiser at all
        self.external_step = None
        self._n_steps = 0                     # host mirror of the launches issued (only DIFFERENCES are used: the per-parameter lag)
        for g in self.param_groups:
            for p in g['params']:
                self._init_state(p)

    def _init_state(self, p):
        if 'exp_avg' not in self.state[p]:
            self.state[p]['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            self.state[p]['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            # torch.optim.Adam counts steps per parameter: one added after k steps, or skipped for want of a gradient, lags by that much
            self.state[p]['lag'] = getattr(self, '_n_steps', 0)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, 'step_dev'):
            for p in self.param_groups[-1]['params']:
                self._init_state(p)

    def zero_grad(self, set_to_none: bool = True):
        """As torch's; while the current stream is CAPTURING (whole-iteration hipGraph capture of a caller's loop body,
        naruto_amd.graphed.GraphedIteration) gradients are zeroed IN PLACE whatever ``set_to_none`` says: dropping the tensor is a host-side
        act a replay cannot repeat, and the next backward would then overwrite instead of accumulate."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            grads = [p.grad for g in self.param_groups for p in g['params'] if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)
            return
        super().zero_grad(set_to_none=set_to_none)

    def moments(self, p):
        st = self.state[p]
        return st['exp_avg'], st['exp_avg_sq']

    def _steps_done(self) -> int:
        """Optimiser steps taken so far: the device counter, or -- driven by an external iteration counter (MappingTrainer) -- that word."""
        src = self.external_step if self.external_step is not None else self.step_dev
        return int(src.reshape(-1)[0].item())

    def state_dict(self):
        """torch.optim.Adam's layout: every parameter's state also carries ``step`` (a float32 scalar tensor, steps taken BY THAT
        parameter = global count - its lag), so that a ``torch.optim.Adam`` can load this dictionary; ``naruto_step`` is the global count."""
        done = self._steps_done()
        sd = super().state_dict()
        sd['state'] = {k: dict(v) for k, v in sd['state'].items()}
        for st in sd['state'].values():
            st['step'] = torch.tensor(float(max(done - int(st.get('lag', 0)), 0)), dtype=torch.float32)
        sd['naruto_step'] = done
        return sd

    def load_state_dict(self, state_dict):
        """Accepts its own dictionaries and ``torch.optim.Adam``'s (no ``naruto_step`` / ``lag``: the global count is the largest
        per-parameter ``step``, a parameter's lag the difference to it).  With an external step word the count is NOT written there
        (it is the trainer's iteration counter, restored by the trainer): the lags are set against the count it holds."""
        state_dict = dict(state_dict)
        step = state_dict.pop('naruto_step', None)
        super().load_state_dict(state_dict)
        self._plan_cache = None                 # the moments are new tensors now
        per_param = {}
        for g in self.param_groups:
            for p in g['params']:
                st = self.state[p]
                if 'step' in st:
                    per_param[p] = int(float(st.pop('step')))
        if step is None:
            step = max(per_param.values()) if per_param else 0
        step = int(step)
        for g in self.param_groups:
            for p in g['params']:
                st = self.state[p]
                fresh = 'exp_avg' not in st               # no state in the loaded dictionary (torch Adam: the parameter never saw a gradient)
                if fresh:
                    self._init_state(p)                   # (sets the lag against the PRE-load step count: overwritten below)
                if fresh or 'lag' not in st or p in per_param:
                    st['lag'] = step - per_param.get(p, 0 if fresh else step)
        if self.external_step is None:
            self.step_dev[0] = step
        self._n_steps = step

    def _plan(self):
        """Per (betas) batch of <= 8 tensors: a reusable NarutoAdamSeg array with everything but the gradient pointers filled in."""
        from . import _lib
        plan, by_betas = [], {}
        for g in self.param_groups:
            for p in g['params']:
                by_betas.setdefault(tuple(g['betas']), []).append((p, g))
        for betas, items in by_betas.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                segs = (_lib.NarutoAdamSeg * len(chunk))()
                for k, (p, g) in enumerate(chunk):
                    m, v = self.moments(p)
                    assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), "FusedAdam: contiguous fp32 parameters on the GPU"
                    segs[k].param, segs[k].exp_avg, segs[k].exp_avg_sq, segs[k].n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                plan.append((betas, chunk, segs))
        self._plan_key = tuple(p.data_ptr() for g in self.param_groups for p in g['params'])
        self._plan_cache = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        """``zero_grad``: zero the gradients inside the same launch, once consumed."""
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plan = getattr(self, "_plan_cache", None)
        if plan is None or self._plan_key != tuple(p.data_ptr() for g in self.param_groups for p in g['params']):
            plan = self._plan()
        launches, skipped = [], []
        for betas, chunk, segs in plan:
            keep, n = [], 0
            for p, g in chunk:
                grad = p.grad
                if grad is None:
                    skipped.append(p)
                    continue
                if not grad.is_contiguous() or grad.dtype != torch.float32:
                    grad = grad.contiguous().float()
                    keep.append(grad)
                sg = segs[n]
                if sg.param != p.data_ptr():                  # a tensor without gradient before this one: re-pack (and re-plan next time)
                    m, v = self.moments(p)
                    sg.param, sg.exp_avg, sg.exp_avg_sq, sg.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    self._plan_cache = None
                sg.grad, sg.lr, sg.eps, sg.weight_decay, sg.step_lag = grad.data_ptr(), g['lr'], g['eps'], g['weight_decay'], self.state[p]['lag']
                n += 1
            if n:
                launches.append((betas, segs, n, chunk[0][0].device))
        if not launches:
            return loss
        self._n_steps += 1
        for p in skipped:
            self.state[p]['lag'] += 1                      # sat this step out: its own step count does not advance
        if self.external_step is not None:
            for betas, segs, n, dev in launches:
                ops.adam_multi_segs(segs, n, dev, betas=betas, step_dev=self.external_step, zero_grad=zero_grad)
            return loss
        if len(launches) == 1:
            # the common case: the kernel itself advances the device-side step count (no "step += 1" launch)
            betas, segs, n, dev = launches[0]
            ops.adam_multi_segs(segs, n, dev, betas=betas, step_dev=self.step_dev, advance=True, zero_grad=zero_grad)
            return loss
        self.step_dev[:1].add_(1)
        for betas, segs, n, dev in launches:
            ops.adam_multi_segs(segs, n, dev, betas=betas, step_dev=self.step_dev, zero_grad=zero_grad)
        return loss


class MappingTrainer:
    """Owns the model + the two Adam instances and runs mapping iterations on ray batches.

    ``fused_adam``: torch.optim.Adam (what the reference's driver builds) or the one-kernel-per-tensor HIP Adam.
    ``capture(n_rays)`` records the whole iteration (forward, losses, backward, optimisers) into two hipGraphs
    (with / without the every-5th-iteration uncertainty-grid step); ``step`` then replays them."""

    def __init__(self, config: Dict, bounding_box: torch.Tensor, device, uncert_voxel: float = 0.1, group=None,
                 fused_adam: bool = False, shard_table_optimizer: Optional[bool] = None):
        self.config = config
        self.device = torch.device(device)
        mp = config.get('mapping', {})
        if int(mp.get('map_accum_step', 1)) != 1 or int(mp.get('map_wait_step', 0)) != 0:
            # coslam.py:370-376 steps the mapping Adam every map_accum_step-th iteration, and not before map_wait_step; every shipped
            # config has 1 / 0 and the fused iteration steps every time
            raise NotImplementedError("MappingTrainer steps the mapping optimiser every iteration: mapping.map_accum_step must be 1 and "
                                      "mapping.map_wait_step 0 (drive NarutoFieldHIP from your own loop for gradient accumulation)")
        self.model = NarutoFieldHIP(config, bounding_box.to(self.device)).to(self.device)
        # Data parallel with a LARGE table (T > 2^17: 281 MB at T = 2^22): the table's optimiser is sharded over the ranks (ZeRO-1 shape) --
        # reduce-scatter of the table gradient, Adam on this rank's 1/world slice with moments for that slice only, all-gather of the
        # updated parameters -- instead of all-reduce + the identical full-size Adam on every rank: the same bytes on the wire
        # (a ring all-reduce IS reduce-scatter + all-gather), 1/world of the optimiser's traffic (1.97 GB per step at T = 2^22) and of
        # its moments' memory.  None: on when a group is given and the table has more than 2^22 parameters.
        self.table_shard = None
        n_table = int(self.model.embed_fn.params.numel())
        if shard_table_optimizer is None:
            shard_table_optimizer = bool(group is not None and fused_adam and n_table > (1 << 22))
        if shard_table_optimizer:
            assert fused_adam and group is not None, "the sharded table optimiser belongs to the data-parallel fused trainer"
            world, rank_ = parallel.world_size(group), parallel.rank(group)
            c = -(-n_table // (4 * world)) * 4                   # floats per rank, a multiple of 4
            store = torch.zeros(c * world, dtype=torch.float32, device=self.device)
            with torch.no_grad():
                store[:n_table].copy_(self.model.embed_fn.params.reshape(-1))
                self.model.embed_fn.params.data = store[:n_table]          # same Parameter, storage padded to world x c
            p_slice = store[rank_ * c:(rank_ + 1) * c]
            p_slice.grad = torch.zeros(c, dtype=torch.float32, device=self.device)       # the reduce-scatter's output
            self.table_shard = {"c": c, "n_pad": c * world, "store": store, "p_slice": p_slice, "send": torch.empty(c, dtype=torch.float32, device=self.device)}
        if fused_adam:
            groups = [{'params': self.model.decoder.parameters(), 'weight_decay': 1e-6, 'lr': config['mapping']['lr_decoder']}]
            if self.table_shard is None:
                groups.append({'params': self.model.embed_fn.parameters(), 'eps': 1e-15, 'lr': config['mapping']['lr_embed']})
            self.map_optimizer = FusedAdam(groups, betas=(0.9, 0.99))
            if self.table_shard is not None:
                self.table_optimizer = FusedAdam([{'params': [self.table_shard["p_slice"]], 'eps': 1e-15, 'lr': config['mapping']['lr_embed']}], betas=(0.9, 0.99))
            self.uncert_optim = FusedAdam([self.model.get_uncert_grid(uncert_voxel)], lr=1)
        else:
            self.map_optimizer = create_optimizer(self.model, config)
            self.uncert_optim = init_uncert_grid_optim(self.model, uncert_voxel)
        # the uncertainty grid's gradient accumulates over 5 iterations (coslam.py:397-399): keep it as a
        # persistent tensor that autograd adds into, zeroed after each uncert step
        self.model.uncert_grid.grad = torch.zeros_like(self.model.uncert_grid)
        self.group = group
        self.iter = 0
        self._graphs = None
        self._static = None
        # fast path: forward + backward as two C calls on persistent buffers (ops.TrainStep) instead of the autograd node
        self.direct = bool(fused_adam)
        self.fuse_optimizer = bool(fused_adam)          # single process: apply the mapping Adam inside the backward
        self._train_steps = {}
        self.max_cached_steps = 2
        self.assert_every = 8           # graph replay: deferred ``uncert_map.min() > 0`` check every this many iterations
        if self.direct:
            # {seed, iteration counter}: the kernels' own random numbers are keyed by it, the forward advances the counter,
            # and the mapping Adam reads its step number from it (one iteration = one step)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            self.iter_state = torch.tensor([seed, 0], dtype=torch.int64, device=self.device)
            self.map_optimizer.external_step = self.iter_state.view(torch.int32)[2:3]
            if self.table_shard is not None:
                self.table_optimizer.external_step = self.map_optimizer.external_step
        tr = config['training']
        # get_loss_from_ret's weights laid out like the node's loss vector (slot 8 = smoothness term)
        self._loss_w = torch.tensor([tr['rgb_weight'], tr['depth_weight'], tr['sdf_weight'], tr['fs_weight'], 0.0,
                                     tr['uncert_weight'], 0.0, 0.0, tr['smooth_weight'], 0.0], dtype=torch.float32, device=self.device)
        if group is not None:
            self.model.enable_data_parallel(group)

    def parameters(self):
        return list(self.model.decoder.parameters()) + list(self.model.embed_fn.parameters()) + [self.model.uncert_grid]

    def _train_step(self, n_rays: int, use_smooth: bool):
        from . import ops
        key = (n_rays, use_smooth, self.model.n_rays_total)
        ts = self._train_steps.pop(key, None)
        if ts is not None:
            self._train_steps[key] = ts                 # most recently used last
        if ts is None:
            # the batch size of the reference's BA loop moves with the keyframe count (sample // n_kf, ceil(len(idx_cur) / 4)):
            # keep the persistent buffers of the last few sizes only (each set is ~80 MB at 2048 x 128).  A captured graph
            # holds its TrainStep through self._static, so eviction here never invalidates a replay.
            while len(self._train_steps) >= self.max_cached_steps:
                self._train_steps.pop(next(iter(self._train_steps)))
            tr, cam = self.config['training'], self.config['cam']
            m = self.model
            ts = ops.TrainStep(m._handle(), m._params(), m.uncert_grid.grad, n_rays, n_samples_d=tr['n_samples_d'], n_range_d=tr['n_range_d'],
                               near=cam['near'], far=cam['far'], range_d=tr['range_d'], depth_trunc=cam['depth_trunc'],
                               rgb_missing=tr['rgb_missing'], perturb=tr['perturb'] > 0., loss_weights=self._loss_w,
                               smooth=(tr['smooth_pts'], tr['smooth_vox'], tr['smooth_margin']) if use_smooth else None,
                               group=self.group, n_rays_total=self.model.n_rays_total, rng_state=self.iter_state,
                               min_uncert_running=self.model.min_uncert_running(),
                               table_grad_pad=self.table_shard["n_pad"] if self.table_shard is not None else 0)
            if self.fuse_optimizer and self.group is None and ops.handle_supports_overwrite(m._handle()):
                # optimiser in the backward: the mapping Adam is applied by the launch that finishes the gradients
                names = {id(p): n for n, p in m._params().items()}
                entries = {}
                for gp in self.map_optimizer.param_groups:
                    for p in gp['params']:
                        mm, vv = self.map_optimizer.moments(p)
                        entries[names[id(p)]] = (mm, vv, gp['lr'], gp['eps'], gp['weight_decay'])
                betas = self.map_optimizer.param_groups[0]['betas']
                assert all(tuple(gp['betas']) == tuple(betas) for gp in self.map_optimizer.param_groups)
                ts.fuse_adam(entries, betas, self.map_optimizer.external_step)
            self._train_steps[key] = ts
        return ts

    def _iteration_direct(self, rays_o, rays_d, target_rgb, target_d, smooth: bool, uncert_step: bool, check: bool = True):
        """Same iteration as _iteration, without autograd: ops.TrainStep + the fused Adams (9 launches single-process, 14 + two collectives data-parallel)."""
        model = self.model
        model.train()
        if check:
            model.check_asserts()
        tr = self.config['training']
        use_smooth = bool(smooth and tr['smooth_weight'] > 0)
        ts = self._train_step(rays_o.shape[0], use_smooth)
        with torch.no_grad():
            if self.group is not None and ts.opt is None:
                # data parallel: forward | all-reduce of the loss sums | backward up to the MLP weight gradients | their (20 KB)
                # all-reduce, issued asynchronously so that it runs under the table scatter | scatter | all-reduce of the table
                # gradient | identical Adam steps on every rank
                ts.run_forward(rays_o, rays_d, target_rgb, target_d.reshape(-1))
                parallel.allreduce_loss_sums(ts.sums, self.group)
                ts.run_backward(phase=1)
                pending = parallel.all_reduce_sum(ts.grad_bucket_mlp, self.group, async_op=True)
                ts.run_backward(phase=2)
                if self.table_shard is not None:
                    parallel.reduce_scatter_sum(ts.grad_bucket_table, self.table_shard["p_slice"].grad, self.group)
                else:
                    parallel.all_reduce_sum(ts.grad_bucket_table, self.group)
                pending.wait()
                losses = ts.losses
            else:
                losses = ts.run(rays_o, rays_d, target_rgb, target_d.reshape(-1))
            if ts.opt is None:
                for name, p in model._params().items():
                    p.grad = ts.grads[name]
                self.map_optimizer.step()
                self._table_shard_step()
            else:
                model.uncert_grid.grad = ts.grads["uncert_grid"]          # table / weight gradients are consumed inside the backward
            if uncert_step:
                if self.group is not None:
                    parallel.allreduce_grads([model.uncert_grid], self.group)
                self.uncert_optim.step(zero_grad=True)
        model.note_min_uncert(losses[6])
        if model.strict_assert:
            model.check_asserts(block=True)
        ret = {"rgb": ts.rgb, "depth": ts.depth, "rgb_loss": losses[0], "depth_loss": losses[1], "sdf_loss": losses[2], "fs_loss": losses[3],
               "psnr": losses[4], "uncert_loss": losses[5], "_losses": losses, "_smooth_loss": losses[8]}
        return ret, losses[9]

    def _table_shard_step(self, adam: bool = True, gather: bool = True):
        """Sharded table optimiser: Adam on this rank's slice (its gradient sits in p_slice.grad after the reduce-scatter), then the
        all-gather of the updated slices into every rank's table."""
        sh = self.table_shard
        if sh is None:
            return
        if adam:
            self.table_optimizer.step()
        if gather:
            sh["send"].copy_(sh["p_slice"])
            parallel.all_gather_into(sh["store"], sh["send"], self.group)

    def _iteration(self, rays_o, rays_d, target_rgb, target_d, smooth: bool, uncert_step: bool, check: bool = True):
        if self.direct:
            rays_o, rays_d, target_rgb, target_d = (t.to(self.device, torch.float32).contiguous() for t in (rays_o, rays_d, target_rgb, target_d))
            return self._iteration_direct(rays_o, rays_d, target_rgb, target_d, smooth, uncert_step, check)
        model = self.model
        model.train()
        self.map_optimizer.zero_grad(set_to_none=True)
        tr = self.config['training']
        use_smooth = smooth and tr['smooth_weight'] > 0
        # one RNG launch per iteration: the depth jitter [N,S] and the six numbers placing the smoothness lattice
        n_rays = rays_o.shape[0]
        n_z = n_rays * (tr['n_samples_d'] + tr['n_range_d']) if tr['perturb'] > 0. else 0
        r = torch.rand(n_z + 6, device=self.device)
        rand = r[:n_z].view(n_rays, -1) if n_z else None
        sm = (tr['smooth_pts'], tr['smooth_vox'], tr['smooth_margin'], r[n_z:]) if use_smooth else None
        ret = model.forward(rays_o, rays_d, target_rgb, target_d, rand=rand, _check=check, _smooth=sm)
        # get_loss_from_ret (coslam.py:154-174) as ONE dot product over the node's loss vector (the smoothness term sits
        # in slot 8; the fused node computed it alongside so that its table gradient shares the scatter pass)
        w = self._loss_w
        if self.group is not None and use_smooth:
            # every rank differentiates the same lattice: scale the term so that the sum over ranks is ONE smoothness term
            if getattr(self, '_loss_w_dp', None) is None:
                self._loss_w_dp = self._loss_w.clone()
                self._loss_w_dp[8] /= parallel.world_size(self.group)
            w = self._loss_w_dp
        loss = torch.dot(ret['_losses'], w)
        loss.backward()
        if self.group is not None:
            # one collective over the flat (table + MLP weights) gradient; the uncertainty grid's gradient keeps
            # accumulating locally and is reduced only when its optimiser steps (every 5th iteration)
            parallel.allreduce_grads(list(self.model.decoder.parameters()) + list(self.model.embed_fn.parameters()), self.group)
        self.map_optimizer.step()
        if uncert_step:
            if self.group is not None:
                parallel.allreduce_grads([self.model.uncert_grid], self.group)
            self.uncert_optim.step()
            self.model.uncert_grid.grad.zero_()
        return ret, loss

    def step(self, rays_o, rays_d, target_rgb, target_d, smooth: bool = False, n_rays_total: int = 0,
             uncert_step: Optional[bool] = None, first: bool = False):
        """One mapping iteration (global_BA body, coslam.py:361-399).  With a process group the rays passed
        in are THIS RANK's shard; gradients are summed over ranks before the (identical) Adam steps.

        ``uncert_step``: whether the uncertainty-grid Adam steps (and its accumulated gradient is zeroed) after this
        iteration.  Default (None): every 5th call of ``step`` over the trainer's lifetime.  The reference's drivers count
        differently -- ``global_BA`` restarts its ``(i + 1) % 5`` counter on every call (coslam.py:397-399) and
        ``first_frame_mapping`` never steps inside the loop (coslam.py:197-217) -- so ``global_BA`` / ``first_frame_mapping``
        below pass it explicitly."""
        self.model.n_rays_total = n_rays_total
        self.iter += 1
        if uncert_step is None:
            uncert_step = self.iter % 5 == 0
        uncert_step = bool(uncert_step)
        if self._graphs is not None:
            st = self._static
            assert smooth == st['smooth'] and rays_o.shape[0] == st['rays_o'].shape[0], "captured for another configuration"
            src = getattr(rays_o, "_base", None)
            if rays_o.data_ptr() == st['rays_o'].data_ptr() and rays_d.data_ptr() == st['rays_d'].data_ptr() and \
                    target_rgb.data_ptr() == st['target_rgb'].data_ptr() and target_d.data_ptr() == st['target_d'].data_ptr():
                pass                                                      # the batch was assembled in ray_buffers(): nothing to copy
            elif src is not None and src.numel() == st['flat'].numel() and src.is_contiguous() and all(
                    t._base is src for t in (rays_d, target_rgb, target_d)) and rays_o.data_ptr() == src.data_ptr():
                st['flat'].copy_(src.reshape(-1), non_blocking=True)      # rays packed by pack_rays(): one copy
            else:
                st['rays_o'].copy_(rays_o, non_blocking=True)
                st['rays_d'].copy_(rays_d, non_blocking=True)
                st['target_rgb'].copy_(target_rgb, non_blocking=True)
                st['target_d'].copy_(target_d.reshape(st['target_d'].shape), non_blocking=True)
            if st.get('segments') is not None:
                # data parallel: three graph segments with the collectives as ordinary eager RCCL calls in between
                seg, ts = st['segments'], st['ts']
                seg['fwd'].replay()
                parallel.allreduce_loss_sums(ts.sums, self.group)
                seg['bwd_mlp'].replay()
                pending = parallel.all_reduce_sum(ts.grad_bucket_mlp, self.group, async_op=True)      # under the scatter
                seg['bwd_table'].replay()
                if self.table_shard is not None:
                    parallel.reduce_scatter_sum(ts.grad_bucket_table, self.table_shard["p_slice"].grad, self.group)
                else:
                    parallel.all_reduce_sum(ts.grad_bucket_table, self.group)
                if pending is not None:
                    pending.wait()
                if uncert_step:
                    parallel.allreduce_grads([self.model.uncert_grid], self.group)
                seg['opt'][1 if uncert_step else 0].replay()
                self._table_shard_step(adam=False)                       # the slice's Adam is part of the opt segment; the all-gather is eager
                if self.iter % self.assert_every == 0:
                    self.model.note_min_uncert(self.model.min_uncert_running())
                    self.model.check_asserts()
                return st['ret'][0], st['loss'][0]
            if first and uncert_step and len(self._graphs) > 2:
                # (advisor, round 5) the graph with the FIRST iteration's prologue has no uncertainty-step twin: replaying graph 1 here would train on
                # whatever the previous call's last launch prefetched.  The reference never asks for it (global_BA steps the grid after iterations
                # 5, 10, ...: coslam.py:397-399), so it is an error rather than a fourth captured graph.
                raise RuntimeError("MappingTrainer.step(first=True, uncert_step=True): no graph was captured for a call's first iteration with an "
                                   "uncertainty-grid step; the first iteration of a global_BA call is never one")
            which = 2 if (first and len(self._graphs) > 2) else (1 if uncert_step else 0)      # (2: capture(first_prologue=...))
            self._graphs[which].replay()
            ret = st['ret'][which]
            # the reference's in-line ``assert uncert_map.min() > 0`` (scene_rep.py:280) as a deferred check: every replay folds its
            # minimum into one device word (the loss tail does, NarutoTrainStep.min_uncert_running), and every ``assert_every``-th
            # replay queues an asynchronous copy of that RUNNING minimum and tests whatever has landed -- no host sync, and no
            # iteration goes unchecked
            if self.iter % self.assert_every == 0:
                self.model.note_min_uncert(self.model.min_uncert_running() if self.direct else ret['_losses'][6])
                self.model.check_asserts()
            return ret, st['loss'][which]
        return self._iteration(rays_o, rays_d, target_rgb, target_d, smooth, uncert_step)

    def chain_length(self) -> int:
        """Iterations the chained graph of capture(chain=...) holds (0: none)."""
        ch = self._static.get('chain') if self._graphs is not None else None
        return ch[3] if ch else 0

    def step_chain(self, n_rays_total: int = 0):
        """Replay the chained graph of ``capture(chain=...)``: len(chain) iterations, the batches drawn by the captured prologues."""
        g, ret, loss, n = self._static['chain']
        self.model.n_rays_total = n_rays_total
        before = self.iter
        self.iter += n
        g.replay()
        if self.iter // self.assert_every != before // self.assert_every:
            self.model.note_min_uncert(self.model.min_uncert_running() if self.direct else ret['_losses'][6])
            self.model.check_asserts()
        return ret, loss

    def ray_buffers(self):
        """After capture(): the graph's own input buffers (rays_o [N,3], rays_d [N,3], target_rgb [N,3], target_d [N,1]).  A batch
        written straight into them (``KeyframeRayStore.assemble_batch(..., out=trainer.ray_buffers())``) and passed to ``step``
        reaches the replay without the device copy an outside batch needs.  None when nothing is captured."""
        if self._graphs is None:
            return None
        st = self._static
        return st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d']

    def global_BA(self, batches, smooth: bool = True, n_rays_total: int = 0):
        """The optimisation loop of one ``global_BA`` call (coslam.py:361-399) over an iterable of ray batches
        ``(rays_o, rays_d, target_rgb, target_d)``: the uncertainty grid steps after iterations 5, 10, ... OF THIS CALL (the
        reference's ``(i + 1) % 5`` restarts with every call), its gradient accumulating in between."""
        out = None
        for i, b in enumerate(batches):
            out = self.step(*b, smooth=smooth, n_rays_total=n_rays_total, uncert_step=(i + 1) % 5 == 0)
        return out

    def first_frame_mapping(self, batches, n_rays_total: int = 0):
        """The optimisation loop of ``first_frame_mapping`` (coslam.py:197-217): the uncertainty grid's gradient is zeroed
        once, accumulates over ALL iterations, and its Adam steps once at the end -- without zeroing the gradient afterwards
        (the first ``global_BA`` call keeps adding to it, as in the reference)."""
        with torch.no_grad():
            self.model.uncert_grid.grad.zero_()
        out = None
        for b in batches:
            out = self.step(*b, smooth=False, n_rays_total=n_rays_total, uncert_step=False)
        with torch.no_grad():
            if self.group is not None:
                # every rank holds its shard's accumulated gradient; the sum stays in .grad on all ranks (what the next
                # all-reduce would see is then world x too large, so rescale the kept copy)
                parallel.allreduce_grads([self.model.uncert_grid], self.group)
                self.uncert_optim.step()
                self.model.uncert_grid.grad.div_(parallel.world_size(self.group))
            else:
                self.uncert_optim.step()
        return out

    def capture(self, n_rays: int, smooth: bool = False, n_rays_total: int = 0, warmup: int = 3, prologue=None, first_prologue=None, on_buffers=None, chain=None):
        """Record the iteration into hipGraphs (static shapes: n_rays rays per call).
        ``prologue(rays_o, rays_d, target_rgb, target_d)``: launches recorded IN FRONT of the iteration inside the same graphs -- the
        ray assembly / active ray selection that fill the iteration's input buffers (naruto_amd.ba_loop.FusedBA); single process.
        ``first_prologue``: a third graph (no uncertainty-grid step) with THIS prologue instead, replayed by ``step(first=True)`` -- the
        first iteration of a ``global_BA`` call, which has to assemble its own batch where later ones find theirs prepared by the
        previous iteration's last launch (``NarutoFusedAdam.next_batch``).  ``on_buffers(rays_o, rays_d, target_rgb, target_d, train_step)``
        is called once the graph's input buffers and the persistent TrainStep exist, before anything is launched.
        ``chain``: a list of uncert_step flags -- ONE more graph that holds len(chain) iterations back to back (the first with ``first_prologue``
        when given), replayed by ``step_chain()``: a whole ``global_BA`` call as one graph launch (the launch-to-launch gap between graphs, 5 - 8 us
        on MI355X, is paid once per call instead of once per iteration)."""
        dev = self.device
        self.model.n_rays_total = n_rays_total
        flat = torch.zeros(n_rays * 10, device=dev)
        st = {'flat': flat, 'smooth': smooth, 'ret': [None, None], 'loss': [None, None]}
        st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'] = unpack_rays(flat, n_rays)
        st['rays_d'][:, 2] = 1.0
        st['target_d'].fill_(1.0)
        # snapshot: the warm-up / capture iterations below must not change the training state -- parameters, both optimisers'
        # moments and step counts, the accumulated uncertainty-grid gradient and the iteration counter are all put back
        params = self.parameters()
        snap = [p.detach().clone() for p in params]
        iter_snap = self.iter_state.clone() if self.direct else None
        ugrad_snap = self.model.uncert_grid.grad.detach().clone()
        opt_snap = []
        opts_all = (self.map_optimizer, self.uncert_optim) + ((self.table_optimizer,) if self.table_shard is not None else ())
        for opt in opts_all:
            if isinstance(opt, FusedAdam):
                opt_snap.append((opt.step_dev.clone(), [(st_['exp_avg'].clone(), st_['exp_avg_sq'].clone()) for st_ in opt.state.values()]))
            else:
                import copy
                opt_snap.append(copy.deepcopy(opt.state_dict()))
        if on_buffers is not None:
            tr_cfg0 = self.config['training']
            on_buffers(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'], self._train_step(n_rays, bool(smooth and tr_cfg0['smooth_weight'] > 0)))
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for i in range(warmup):
                if first_prologue is not None and i == 0:
                    first_prologue(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'])
                elif prologue is not None:
                    prologue(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'])
                self._iteration(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'], smooth, i == warmup - 1, check=False)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        graphs = []
        pool = None
        import os
        segmented = self.group is not None and self.direct and os.environ.get("NARUTO_GRAPH_DIST", "segmented") != "whole"
        # with a process group its watchdog thread polls events while this thread captures: only this thread's calls may end the capture
        cap_mode = "thread_local" if self.group is not None else "global"
        assert prologue is None or not segmented, "a captured prologue belongs to the single-process graph"
        if segmented:
            # Data parallel: forward | backward | optimiser as three graph segments; the two all-reduces (loss sums, flat
            # gradient) run between them as eager RCCL calls on the same stream.  (Capturing the collectives inside one graph
            # also works -- NARUTO_GRAPH_DIST=whole, 0.47 ms at world size 1 -- but a multi-rank capture cannot be
            # exercised on the single-GPU test boxes, so the default does not depend on it.)
            tr_cfg = self.config['training']
            ts = self._train_step(n_rays, bool(smooth and tr_cfg['smooth_weight'] > 0))
            args = (st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'].reshape(-1))
            seg = {'opt': [None, None]}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode), torch.no_grad():
                ts.run_forward(*args)
            pool = g.pool()
            seg['fwd'] = g
            # the backward as TWO segments split at the MLP / table boundary (as the eager path does): the 20 KB bucket of MLP weight
            # gradients is complete after the first and is all-reduced asynchronously WHILE the second -- the table scatter, the longest
            # kernel of the step -- replays; the table bucket follows it.  Same kernels in the same order as the one-piece backward:
            # same bits (tests/test_gpu_parity.py::test_two_rank_data_parallel_training compares against the single-process trajectory).
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode), torch.no_grad():
                ts.run_backward(phase=1)
            seg['bwd_mlp'] = g
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode), torch.no_grad():
                ts.run_backward(phase=2)
            seg['bwd_table'] = g
            for name, p in self.model._params().items():
                p.grad = ts.grads[name]
            for variant in (False, True):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode), torch.no_grad():
                    self.map_optimizer.step()
                    self._table_shard_step(gather=False)
                    if variant:
                        self.uncert_optim.step(zero_grad=True)
                seg['opt'][1 if variant else 0] = g
            losses = ts.losses
            self.model._pending_min_uncert = losses[6]
            st['segments'], st['ts'] = seg, ts
            st['ret'][0] = {"rgb": ts.rgb, "depth": ts.depth, "rgb_loss": losses[0], "depth_loss": losses[1], "sdf_loss": losses[2],
                            "fs_loss": losses[3], "psnr": losses[4], "uncert_loss": losses[5], "_losses": losses, "_smooth_loss": losses[8]}
            st['loss'][0] = losses[9]
            graphs = [seg['fwd']]
        for variant in (() if segmented else (False, True)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode):
                if prologue is not None:
                    prologue(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'])
                ret, loss = self._iteration(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'], smooth, variant, check=False)
            pool = g.pool()
            st['ret'][1 if variant else 0] = ret
            st['loss'][1 if variant else 0] = loss
            graphs.append(g)
        if first_prologue is not None and not segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode):
                first_prologue(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'])
                ret, loss = self._iteration(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'], smooth, False, check=False)
            st['ret'].append(ret)
            st['loss'].append(loss)
            graphs.append(g)
        st['chain'] = None
        if chain and not segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=cap_mode):
                for i, u in enumerate(chain):
                    pro = first_prologue if (i == 0 and first_prologue is not None) else prologue
                    if pro is not None:
                        pro(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'])
                    ret, loss = self._iteration(st['rays_o'], st['rays_d'], st['target_rgb'], st['target_d'], smooth, bool(u), check=False)
            st['chain'] = (g, ret, loss, len(chain))
        if self.direct and not segmented:
            # the graphs hold the ADDRESSES of this TrainStep's buffers: keep it alive with them, whatever the LRU cache below evicts
            tr_cfg = self.config['training']
            st['ts'] = self._train_step(n_rays, bool(smooth and tr_cfg['smooth_weight'] > 0))
        # restore parameters and optimiser state to "before capture"
        with torch.no_grad():
            for p, q in zip(params, snap):
                p.copy_(q)
            if iter_snap is not None:
                self.iter_state.copy_(iter_snap)
            self.model.uncert_grid.grad.copy_(ugrad_snap)
            for opt, sn in zip(opts_all, opt_snap):
                if isinstance(opt, FusedAdam):
                    opt.step_dev.copy_(sn[0])
                    for st_, (m0, v0) in zip(opt.state.values(), sn[1]):
                        st_['exp_avg'].copy_(m0)
                        st_['exp_avg_sq'].copy_(v0)
                else:
                    # torch.optim.Adam creates its state lazily: a fresh optimiser has none before capture.  The captured graphs
                    # hold the state tensors' ADDRESSES, so restore by value into the existing tensors.
                    had = sn['state']
                    index = {id(q): i for i, q in enumerate(q for gp in opt.param_groups for q in gp['params'])}
                    for p_, stt in opt.state.items():
                        idx = index[id(p_)]
                        for k, val in stt.items():
                            if torch.is_tensor(val):
                                if idx in had and k in had[idx]:
                                    val.copy_(had[idx][k])
                                else:
                                    val.zero_()
        self._graphs, self._static = graphs, st
