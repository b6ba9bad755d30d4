"""One mapping (bundle-adjustment) iteration end to end on the device: what the reference's ``CoSLAMNaruto.global_BA`` does per
iteration (reference src/slam/coslam/coslam.py:310-399) --

    rays, ids = keyframeDatabase.sample_global_rays(sample_num)                :325        Co-SLAM KeyFrameDatabase [not in tree]
    idx_cur   = random.sample(valid pixels of the current frame, n_cur)         :332-340
    rays_o, rays_d = poses_all[ids] applied to the camera-frame directions      :342-347
    rays_* = active_ray_sampler.sample_rays(...)        (mapping.active_ray)     :349-359    src/slam/coslam/active_ray_sampler.py:77-149
    ret = model.forward(...); loss = get_loss_from_ret(ret, smooth=True); loss.backward(); Adam     :361-399

-- as ONE stream of launches: ``naruto_assemble_rays`` (N2) -> ``naruto_active_ray_select`` (N1) -> the fused training iteration
(``MappingTrainer``), the first two writing straight into the iteration's input buffers, all of it recorded in one hipGraph.  What
changes between replays lives in device memory: the draws are keyed by the trainer's {seed, iteration counter} (advanced by every
forward), the keyframe / pose / valid-pixel counts sit in a three-word tensor the host refreshes once per ``global_BA`` call, the
planner's uncertainty volume is refreshed in place.  The graph is re-captured only when the ray COUNT changes (n_cur = max(sample_num
// n_kf, min_pixels_cur): constant once n_kf exceeds sample_num / min_pixels_cur, i.e. for all but the first ~20 keyframes).

The reference draws with Python's ``random`` on the host and round-trips through numpy for the active rays; the drawn sets differ by
construction (tests: same distribution properties, and the chained launches equal the three operators run one by one)."""

from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .active_ray_sampler import ActiveRaySamplerHIP
from .keyframe_store import KeyFrameStoreHIP
from .trainer import MappingTrainer


class FusedBA:
    def __init__(self, trainer: MappingTrainer, store: KeyFrameStoreHIP, sampler: Optional[ActiveRaySamplerHIP] = None,
                 max_poses: int = 4096, use_graph: bool = True, one_launch_prologue: Optional[bool] = None, prefetch: Optional[bool] = None):
        assert trainer.direct and trainer.group is None, "FusedBA drives the single-process fused trainer (MappingTrainer(fused_adam=True))"
        self.trainer, self.store, self.sampler = trainer, store, sampler
        self.config = trainer.config
        self.device = trainer.device
        self.use_graph = use_graph
        # assembly + selection as ONE launch (naruto_assemble_select) instead of two.  Measured (round 5, profiles/r05_ba_prologue_ab.txt): the
        # one launch takes 43 us against 20 + 7.5 -- the selecting workgroup has to draw all 6 444 candidates itself (Feistel walks: integer
        # multiplies on ONE CU) where k_assemble_rays spreads them over 34 -- 0.210 against 0.193 ms per iteration: OFF by default
        # (NARUTO_BA_ONE_LAUNCH_PROLOGUE=1 or the argument switch it on; same rays either way, tested)
        self.one_launch_prologue = (os.environ.get("NARUTO_BA_ONE_LAUNCH_PROLOGUE", "0") == "1") if one_launch_prologue is None else bool(one_launch_prologue)
        # PREFETCH (round 5, on by default; NARUTO_BA_PREFETCH=0 or the argument switch it off): the ray assembly of iteration i + 1 rides in the
        # LAST launch of iteration i (k_bwd_finish_next: NarutoFusedAdam.next_batch) -- by then nothing reads the ray buffers any more and the
        # iteration counter that keys the draw has been advanced -- so only the FIRST iteration of a global_BA call launches k_assemble_rays
        # itself.  Same batches, same trajectory (tests: the graph twin prefetches, the eager twin does not).
        self.prefetch = (os.environ.get("NARUTO_BA_PREFETCH", "1") != "0") if prefetch is None else bool(prefetch)
        self._armed = None            # the TrainStep whose finishing launch draws the next batch (it holds the struct and its keep-alives)
        # ... and with active rays that assembly also looks the candidates' keys up (NarutoRayBatch.keys_out): the selection in front of the
        # next forward starts from the keys (naruto_active_ray_select_keyed) instead of two dependent trips to memory per candidate.
        # NARUTO_BA_KEYED_SELECT=0 switches it off.
        self.keyed = sampler is not None and os.environ.get("NARUTO_BA_KEYED_SELECT", "1") != "0"
        self._keys = None
        # CALL GRAPH (round 5; NARUTO_BA_CALL_GRAPH=0 switches it off): the mapping.iters iterations of a global_BA call recorded as ONE graph
        # next to the per-iteration graphs -- a call of the configured length is one graph launch
        self.call_graph = use_graph and os.environ.get("NARUTO_BA_CALL_GRAPH", "1") != "0"
        mp = self.config['mapping']
        self.active = sampler is not None
        self.sample_num = sampler.oversample_num if self.active else int(mp['sample'])
        self.min_pixels_cur = sampler.min_pixels_cur if self.active else int(mp['min_pixels_cur'])
        self.filter_depth = bool(mp.get('filter_depth', False))
        dev = self.device
        self.current = torch.zeros(store.total_pixels, 7, dtype=torch.float32, device=dev)        # the current frame's rays, refreshed per call
        self.poses = torch.zeros(int(max_poses), 4, 4, dtype=torch.float32, device=dev)
        self.dyn = torch.zeros(3, dtype=torch.int64, device=dev)                                   # {n_kf, n_poses, n_cur_pop}
        # pinned staging for the asynchronous refresh of ``dyn``: TWO buffers used in turn, each guarded by the event recorded behind
        # its last copy -- a second prepare() must not overwrite a buffer whose host-to-device copy is still queued behind replays
        self._dyn_host = [torch.zeros(3, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._dyn_done = [None, None]
        self._dyn_turn = 0
        self._n_cur_pop = 1
        self._vol_ptr = None          # data_ptr of the sampler's volume the graph was captured with
        self._shape = None            # (n_cur, n_train) the graph was captured for
        self._stage = None            # the oversampled batch between assembly and selection (active ray only)
        self._ws = None
        self.bbox = [[float(v) for v in row] for row in self.config['mapping']['bound']]

    # ---------------------------------------------------------------------------------------------
    def sizes(self, n_kf: int, n_valid_cur: int):
        """(n_cur, n_train): current-frame rays drawn (coslam.py:332-340) and rays the training step sees."""
        n_cur = max(self.sample_num // n_kf, self.min_pixels_cur)
        if self.filter_depth:
            n_cur = min(n_valid_cur, n_cur)
        n_train = self.sampler.n_out(n_cur) if self.active else self.sample_num + n_cur
        return n_cur, n_train

    def _later_prologue(self, n_cur: int):
        """With prefetch: what an iteration that finds its batch assembled still has to launch (the selection), or None."""
        if not self.active:
            return None
        sampler = self.sampler

        def prologue(rays_o, rays_d, target_rgb, target_d):
            # (the candidates' keys were looked up by the assembly that rode in the previous iteration's finishing launch)
            sampler.sample_rays(*self._stage, n_cur, None, self.bbox, out=(rays_o, rays_d, target_rgb, target_d), workspace=self._ws,
                                keys=self._keys if self._keyed(n_cur) else None)
        return prologue

    def _keyed(self, n_cur: int) -> bool:
        return self.keyed and self.sample_num + n_cur - self.sampler.n_out(n_cur) <= 8192

    def _arm_prefetch(self, n_cur: int, bufs, train_step):
        """Point the fused optimiser's next_batch at the draw of the NEXT iteration (into the stage when active rays select from it, else
        straight into the iteration's own input buffers)."""
        out = self._stage if self.active else bufs
        keys = self.sampler.key_lookup(self.sample_num + n_cur, n_cur, self.bbox, self._keys) if (self.active and self._keyed(n_cur)) else None
        b, keep = self.store.next_batch_struct(self.sample_num, self.current, self.poses, self.min_pixels_cur, out, filter_depth=self.filter_depth,
                                               rng=self.trainer.iter_state, dyn=self.dyn, n_cur=n_cur, n_cur_pop=self._n_cur_pop, keys=keys)
        assert train_step.opt is not None, "prefetch needs the optimiser in the backward (MappingTrainer(fused_adam=True))"
        import ctypes as C
        self._disarm_prefetch()
        # the struct and everything it points into live ON the TrainStep whose backward reads them (the trainer's cache and a captured graph's
        # _static['ts'] outlive this object): the host pointer in opt.next_batch can never dangle
        train_step._next_batch_keep = (b, keep)
        train_step.opt.next_batch = C.cast(C.pointer(b), C.c_void_p)
        self._armed = train_step

    def _disarm_prefetch(self):
        """Take the next-batch draw off the TrainStep it rides on: its backward no longer assembles anything (and no longer overwrites the stage /
        ray buffers of this object).  Called on re-arm, on close() and when this object is dropped."""
        ts = getattr(self, "_armed", None)
        self._armed = None
        if ts is not None and getattr(ts, "opt", None) is not None:
            ts.opt.next_batch = None
            ts._next_batch_keep = None

    def close(self):
        """Detach from the trainer: the TrainStep this object armed keeps working as a plain training step (trainer.step, first_frame_mapping).
        A graph captured with the prefetch inside is dropped (its finishing launch would go on drawing batches)."""
        armed = getattr(self, "_armed", None)
        st = getattr(self.trainer, "_static", None)
        if armed is not None and self.use_graph and st is not None and st.get('ts') is armed:
            self.trainer._graphs = None
        self._disarm_prefetch()
        self._shape = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prologue(self, n_cur: int):
        store, sampler = self.store, self.sampler
        rng = self.trainer.iter_state

        def prologue(rays_o, rays_d, target_rgb, target_d):
            kw = dict(filter_depth=self.filter_depth, rng=rng, dyn=self.dyn, n_cur=n_cur, n_cur_pop=self._n_cur_pop)
            if not self.active:
                store.assemble_batch(self.sample_num, self.current, self.poses, self.min_pixels_cur, out=(rays_o, rays_d, target_rgb, target_d), **kw)
                return
            if self.one_launch_prologue and self.sample_num + n_cur - sampler.n_out(n_cur) <= 8192:
                # assembly + selection in one launch (naruto_assemble_select): the oversampled batch is never written
                store.assemble_select(sampler, self.sample_num, self.current, self.poses, self.min_pixels_cur, self.bbox,
                                      out=(rays_o, rays_d, target_rgb, target_d), **kw)
                return
            if self._keyed(n_cur):         # the assembly looks the candidates' keys up while the rows are in registers (NarutoRayBatch.keys_out)
                store.assemble_batch(self.sample_num, self.current, self.poses, self.min_pixels_cur, out=self._stage,
                                     keys=sampler.key_lookup(self.sample_num + n_cur, n_cur, self.bbox, self._keys), **kw)
                sampler.sample_rays(*self._stage, n_cur, None, self.bbox, out=(rays_o, rays_d, target_rgb, target_d), workspace=self._ws, keys=self._keys)
                return
            store.assemble_batch(self.sample_num, self.current, self.poses, self.min_pixels_cur, out=self._stage, **kw)
            sampler.sample_rays(*self._stage, n_cur, None, self.bbox, out=(rays_o, rays_d, target_rgb, target_d), workspace=self._ws)
        return prologue

    def prepare(self, current_rays: torch.Tensor, poses_all: torch.Tensor, uncert_vol=None, smooth: bool = True):
        """Per ``global_BA`` call: the current frame's rays [H*W,7], all poses [P,4,4] (the current frame's LAST), optionally the
        planner's refreshed uncertainty volume.  One count of the valid-depth pixels is read back (the reference does the same
        filtering on the host, coslam.py:332-337); everything else is asynchronous."""
        dev = self.device
        cur = current_rays.to(dev, torch.float32).reshape(-1, 7)
        assert cur.shape[0] == self.current.shape[0], "current_rays: one row per pixel of the frame the store was built for"
        self.current.copy_(cur, non_blocking=True)
        P = poses_all.shape[0]
        assert P <= self.poses.shape[0], "more poses than FusedBA(max_poses=...)"
        self.poses[:P].copy_(poses_all.to(dev, torch.float32), non_blocking=True)
        n_kf = len(self.store)
        assert n_kf > 0, "no keyframe stored yet"
        n_valid = cur.shape[0]
        if self.filter_depth:
            n_valid = int(((cur[:, -1] > 0.0) & (cur[:, -1] <= self.config["cam"]["depth_trunc"])).sum().item())
        turn = self._dyn_turn
        self._dyn_turn ^= 1
        if self._dyn_done[turn] is not None:
            self._dyn_done[turn].synchronize()            # the copy that last read this staging buffer has landed (two calls back: normally long ago)
        host = self._dyn_host[turn]
        host[0], host[1], host[2] = n_kf, P, max(n_valid, 1)
        self._n_cur_pop = max(n_valid, 1)
        self.dyn.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._dyn_done[turn] = ev
        vol_moved = False
        if self.active and uncert_vol is not None:
            vol = self.sampler.set_volume(uncert_vol, dev)
            vol_moved = self._vol_ptr is not None and vol.data_ptr() != self._vol_ptr      # a new tensor (other shape): the captured launch reads the old one
            self._vol_ptr = vol.data_ptr()
        n_cur, n_train = self.sizes(n_kf, n_valid)
        if self._shape != (n_cur, n_train, smooth) or (vol_moved and (self.use_graph or self.keyed)):
            f32 = dict(dtype=torch.float32, device=dev)
            n_stage = self.sample_num + n_cur
            if self.active:
                self._stage = (torch.empty(n_stage, 3, **f32), torch.empty(n_stage, 3, **f32), torch.empty(n_stage, 3, **f32), torch.empty(n_stage, 1, **f32))
                self._ws = torch.empty(self.sampler.workspace_elems(n_stage), dtype=torch.int32, device=dev)
                self._keys = torch.zeros(n_stage, dtype=torch.int32, device=dev)
            self._pro = self._prologue(n_cur)
            self._pro_later = self._later_prologue(n_cur) if self.prefetch else self._pro
            if self.use_graph:
                chain = [(i + 1) % 5 == 0 for i in range(int(self.config['mapping']['iters']))] if self.call_graph else None
                if self.prefetch:
                    self.trainer.capture(n_train, smooth=smooth, prologue=self._pro_later, first_prologue=self._pro, chain=chain,
                                         on_buffers=lambda ro, rd, tc, td, ts: self._arm_prefetch(n_cur, (ro, rd, tc, td), ts))
                else:
                    self.trainer.capture(n_train, smooth=smooth, prologue=self._pro, chain=chain)
            else:
                f = torch.zeros(n_train * 10, **f32)
                from .trainer import unpack_rays
                self._eager_bufs = unpack_rays(f, n_train)
                self.trainer._graphs = None
                self._disarm_prefetch()
                if self.prefetch:
                    tr_cfg = self.config['training']
                    self._arm_prefetch(n_cur, self._eager_bufs, self.trainer._train_step(n_train, bool(smooth and tr_cfg['smooth_weight'] > 0)))
            self._shape = (n_cur, n_train, smooth)
        return n_cur, n_train

    def iteration(self, i: int, smooth: bool = True):
        """Iteration ``i`` (0-based) of the current ``global_BA`` call: the uncertainty grid's Adam steps after iterations 5, 10, ...
        (coslam.py:397-399)."""
        tr = self.trainer
        if self.use_graph:
            bufs = tr.ray_buffers()
            return tr.step(*bufs, smooth=smooth, uncert_step=(i + 1) % 5 == 0, first=(i == 0))           # the replay starts with the prologue's launches
        bufs = self._eager_bufs
        full = i == 0 or not self.prefetch
        if self.prefetch:
            # the TrainStep this iteration WILL run on (the trainer's cache is keyed on (n_rays, smooth, n_rays_total) and evicts): if it is not
            # the one whose finishing launch draws the batches, nothing drew this iteration's batch and nothing would draw the next one --
            # arm it and assemble this batch with the full prologue (the draw is keyed by the iteration counter: assembling twice is idempotent)
            tr_cfg = self.config['training']
            ts = tr._train_step(bufs[0].shape[0], bool(smooth and tr_cfg['smooth_weight'] > 0))
            if ts is not self._armed or ts.opt is None or not ts.opt.next_batch:
                self._arm_prefetch(self._shape[0], bufs, ts)
                full = True
        pro = self._pro if full else self._pro_later
        if pro is not None:
            pro(*bufs)
        return tr.step(*bufs, smooth=smooth, uncert_step=(i + 1) % 5 == 0)

    def global_BA(self, current_rays: torch.Tensor, poses_all: torch.Tensor, n_iters: Optional[int] = None, uncert_vol=None, smooth: bool = True):
        """The optimisation loop of one ``global_BA`` call (coslam.py:293-399 without pose optimisation: tracking is off in every
        shipped config)."""
        self.prepare(current_rays, poses_all, uncert_vol, smooth)
        return self.call_iterations(n_iters, smooth)

    def call_iterations(self, n_iters: Optional[int] = None, smooth: bool = True):
        """The iterations of the current ``global_BA`` call (after ``prepare``): one launch of the call graph when the call has the
        configured length, iteration by iteration otherwise.  Returns the last iteration's (ret, loss)."""
        n_iters = int(self.config['mapping']['iters']) if n_iters is None else int(n_iters)
        if self.use_graph and self.trainer.chain_length() == n_iters and self._shape is not None and self._shape[2] == smooth:
            return self.trainer.step_chain()
        out = None
        for i in range(n_iters):
            out = self.iteration(i, smooth)
        return out
