"""N2, store side ("next" row of SURVEY.md section 8f): a device-resident keyframe ray store and the BA batch assembly.

Mirror of what the mapping loop needs from ``KeyFrameDatabaseNaruto`` (reference src/slam/coslam/model/keyframe.py:16-60,
on top of Co-SLAM's ``KeyFrameDatabase`` [not in tree]) and of the ray assembly in ``CoSLAMNaruto.global_BA``
(src/slam/coslam/coslam.py:293-344):

  * ``add_keyframe(batch, filter_depth)``  -- keep ``num_rays_to_save`` distinct (valid-depth) pixels of the frame as
    [direction 3 | rgb 3 | depth 1] rows, tiled when the frame has fewer (keyframe.py:38-60);
  * ``sample_global_rays(bs)``             -- ``bs`` distinct stored rays + their frame ids (Co-SLAM; ``random.sample``);
  * ``assemble_batch(...)``                -- the fused form of coslam.py:310-344: global rays + distinct current-frame
    pixels, rotated to world with the pose of their keyframe, in ONE kernel; the result feeds ``ActiveRaySamplerHIP`` /
    ``MappingTrainer.step`` without leaving the device.

Everything lives on the GPU; the "random.sample" draws are keyed Feistel permutations (``naruto_assemble_rays``), so a
batch is reproducible from (seed, counter) and the whole BA iteration can be captured in a hipGraph.  The reference keeps
the store on the host and draws with Python's ``random`` -- the drawn SETS differ, their distribution (uniform, without
replacement) does not.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import check


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KeyFrameStoreHIP:
    def __init__(self, config: Dict, H: int, W: int, num_kf: int, num_rays_to_save: int, device, seed: int = 0,
                 filter_depth_mode: str = "reference"):
        """``filter_depth_mode`` -- what ``filter_depth=True`` selects from:
          "reference": the reference's behaviour, quirk included: the indices are drawn from ``range(num_valid)`` but then
                       index the UNFILTERED pixel list (keyframe.py:27-36 ``rays[:, idxs]``, coslam.py:318-327
                       ``current_rays[idx_cur]``), i.e. distinct pixels among the FIRST ``num_valid`` pixels of the frame,
                       invalid-depth ones included;
          "valid_only": what the option's name promises: distinct pixels among those with a valid depth."""
        assert filter_depth_mode in ("reference", "valid_only")
        self.filter_depth_mode = filter_depth_mode
        self.config = config
        self.H, self.W = int(H), int(W)
        self.total_pixels = self.H * self.W
        self.num_rays_to_save = int(num_rays_to_save)
        self.device = torch.device(device)
        self.rays = torch.zeros(int(num_kf), self.num_rays_to_save, 7, dtype=torch.float32, device=self.device)   # Co-SLAM: self.rays
        # int64 ids of the stored keyframes on the device (Co-SLAM keeps them on the host), in a buffer of the store's capacity: its
        # ADDRESS never changes, so a captured batch-assembly launch stays valid while keyframes are added
        self._ids = torch.zeros(int(num_kf), dtype=torch.int64, device=self.device)
        self._n_ids = 0
        self.seed, self.counter = int(seed), 0

    def __len__(self):
        return self._n_ids

    @property
    def frame_ids(self) -> Optional[torch.Tensor]:
        return self._ids[:self._n_ids] if self._n_ids else None

    def attach_ids(self, frame_ids: torch.Tensor):
        """Co-SLAM KeyFrameDatabase.attach_ids."""
        frame_ids = frame_ids.to(self.device, torch.int64).reshape(-1)
        k = int(frame_ids.shape[0])
        if self._n_ids + k > self._ids.shape[0]:
            raise RuntimeError(f"keyframe store is full ({self._ids.shape[0]} keyframes): construct it with a larger num_kf")
        self._ids[self._n_ids:self._n_ids + k] = frame_ids
        self._n_ids += k

    def _distinct(self, n: int, count: int) -> torch.Tensor:
        lib = _lib.load()
        out = torch.empty(count, dtype=torch.int64, device=self.device)
        self.counter += 1
        with torch.cuda.device(self.device):
            check(lib.naruto_sample_distinct(n, count, self.seed, self.counter, out.data_ptr(), _stream()), "naruto_sample_distinct")
        return out

    def add_keyframe(self, batch: Dict, filter_depth: bool = False):
        """keyframe.py:38-60.  batch: 'direction' [1,H,W,3] (or [H*W,3]), 'rgb' likewise, 'depth' [1,H,W], 'frame_id'."""
        rays = torch.cat([batch['direction'], batch['rgb'], batch['depth'][..., None]], dim=-1).to(self.device, torch.float32)
        rays = rays.reshape(-1, rays.shape[-1])                                         # [H*W, 7]
        if filter_depth:
            valid = (rays[:, -1] > 0.0) & (rays[:, -1] <= self.config["cam"]["depth_trunc"])
            pool = torch.nonzero(valid).reshape(-1)
            n_take = min(int(pool.shape[0]), self.num_rays_to_save)
            sel = self._distinct(int(pool.shape[0]), n_take) if n_take > 0 else pool[:0]
            if self.filter_depth_mode == "valid_only":
                sel = pool[sel]
        else:
            sel = self._distinct(rays.shape[0], self.num_rays_to_save)
        fid = batch['frame_id']
        self.attach_ids(fid if isinstance(fid, torch.Tensor) else torch.tensor([fid]))
        if sel.shape[0] == 0:                        # as in the reference: the id is attached, no rays are stored
            return
        kept = rays[sel]
        if kept.shape[0] < self.num_rays_to_save:    # "while rays.shape[1] < n: rays = cat([rays, rays])" then truncate = periodic tiling
            reps = -(-self.num_rays_to_save // kept.shape[0])
            kept = kept.repeat(reps, 1)[:self.num_rays_to_save]
        self.rays[len(self) - 1] = kept

    def sample_global_rays(self, bs: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Co-SLAM KeyFrameDatabase.sample_global_rays: bs distinct stored rays [bs,7] and their frame ids [bs]."""
        n_kf = len(self)
        idx = self._distinct(n_kf * self.num_rays_to_save, bs)
        rays = self.rays[:n_kf].reshape(-1, 7)[idx]
        return rays, self.frame_ids[idx // self.num_rays_to_save]

    @staticmethod
    def _set_keys(b, keys):
        """NarutoRayBatch.keys_out from ``ActiveRaySamplerHIP.key_lookup``'s tuple (None: off)."""
        if keys is None:
            return
        k, base, n_tail, vol, bmin = keys
        b.keys_out, b.key_base, b.key_tail = k.data_ptr(), int(base), int(n_tail)
        b.key_vol = vol.data_ptr()
        b.key_dims = (C.c_uint32 * 3)(*vol.shape)
        b.key_bbox_min = (C.c_float * 3)(*bmin)
        b.key_voxel_scale = 10.0

    def _draw(self, sample_num, current_rays, poses_all, min_pixels_cur, filter_depth, rng, dyn, n_cur, n_cur_pop):
        """The NarutoRayBatch of one draw (everything but the output buffers) + what it keeps alive."""
        n_kf = len(self)
        assert n_kf > 0, "no keyframe stored yet"
        cur = current_rays.to(self.device, torch.float32).reshape(-1, 7).contiguous()
        poses = poses_all.to(self.device, torch.float32).contiguous()
        fixed = n_cur is not None
        if fixed:
            assert n_cur_pop is not None and not (filter_depth and self.filter_depth_mode == "valid_only"), "fixed sizes: pass n_cur_pop (reference mode)"
        else:
            n_cur = max(sample_num // n_kf, int(min_pixels_cur))
            n_cur_pop = cur.shape[0]
        cur_list = None
        if filter_depth and not fixed:
            valid = (cur[:, -1] > 0.0) & (cur[:, -1] <= self.config["cam"]["depth_trunc"])
            cur_list = torch.nonzero(valid).reshape(-1).to(torch.int32).contiguous()
            n_cur_pop = int(cur_list.shape[0])
            n_cur = min(n_cur_pop, n_cur)
            if self.filter_depth_mode == "reference":
                cur_list = None                   # pixels 0 .. n_valid-1 of the unfiltered frame (coslam.py:318-327)
        if rng is None:
            self.counter += 1                  # the host-keyed draw; with ``rng`` the device-side {seed, counter} keys it and this one rests
        b = _lib.NarutoRayBatch()
        b.store, b.n_kf, b.rays_per_kf = self.rays.data_ptr(), n_kf, self.num_rays_to_save
        b.frame_ids, b.keyframe_every, b.n_global = self._ids.data_ptr(), int(self.config['mapping']['keyframe_every']), int(sample_num)
        b.current, b.cur_list = cur.data_ptr(), (cur_list.data_ptr() if cur_list is not None and n_cur_pop > 0 else None)
        b.n_cur_pop, b.n_cur = max(n_cur_pop, 1), n_cur
        b.poses, b.n_poses, b.seed, b.counter = poses.data_ptr(), poses.shape[0], self.seed, self.counter
        if rng is not None:
            assert rng.is_cuda and rng.dtype == torch.int64 and rng.numel() >= 2
            b.rng, b.seed, b.counter = rng.data_ptr(), 0, 0
        if dyn is not None:
            assert dyn.is_cuda and dyn.dtype == torch.int64 and dyn.numel() >= 3
            b.dyn = dyn.data_ptr()
        return b, n_cur, (cur, poses, cur_list)

    def assemble_batch(self, sample_num: int, current_rays: torch.Tensor, poses_all: torch.Tensor, min_pixels_cur: int,
                       filter_depth: bool = False, return_ids: bool = False, out=None, rng: Optional[torch.Tensor] = None,
                       dyn: Optional[torch.Tensor] = None, n_cur: Optional[int] = None, n_cur_pop: Optional[int] = None, keys=None):
        """coslam.py:310-344 fused: -> rays_o [N,3], rays_d [N,3], target_s [N,3], target_d [N,1], n_cur (and ids_all).
        N = sample_num + n_cur, n_cur = max(sample_num // n_kf, min_pixels_cur) (capped by the valid pixels).
        current_rays [H*W,7]; poses_all [P,4,4] camera-to-world with the current frame's pose LAST (index -1).
        ``out``: (rays_o, rays_d, target_s, target_d) to write into -- contiguous fp32 device tensors of exactly N rows, e.g. a
        captured trainer's ``ray_buffers()``.
        For a launch captured in a hipGraph (naruto_amd.ba_loop.FusedBA): ``rng`` int64[2] device {seed, counter} keys the draws
        instead of this store's host counter, ``dyn`` int64[3] device {n_kf, n_poses, n_cur_pop} replaces the host values at replay
        time; ``n_cur`` / ``n_cur_pop`` then fix the current-frame draw's size and population up front (no device read-back; with
        ``filter_depth`` in the reference's mode the population is the number of valid-depth pixels, counted by the caller).
        ``keys``: ``ActiveRaySamplerHIP.key_lookup(...)`` -- the active ray sampler's candidate keys are looked up here, while the rows are
        in registers, for ``sample_rays(..., keys=...)``."""
        lib = _lib.load()
        b, n_cur, keep = self._draw(sample_num, current_rays, poses_all, min_pixels_cur, filter_depth, rng, dyn, n_cur, n_cur_pop)
        n = sample_num + n_cur
        f32 = dict(dtype=torch.float32, device=self.device)
        if out is not None:
            rays_o, rays_d, target_s, target_d = out
            for a, c in ((rays_o, 3), (rays_d, 3), (target_s, 3), (target_d, 1)):
                if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == n * c):
                    raise RuntimeError(f"assemble_batch: out tensors must be contiguous fp32 device tensors of {n} rows")
        else:
            rays_o, rays_d, target_s = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
            target_d = torch.empty(n, 1, **f32)
        ids = torch.empty(n, dtype=torch.int64, device=self.device) if return_ids else None
        b.rays_o, b.rays_d, b.target_s, b.target_d = rays_o.data_ptr(), rays_d.data_ptr(), target_s.data_ptr(), target_d.data_ptr()
        b.ids_out = ids.data_ptr() if ids is not None else None
        self._set_keys(b, keys)
        with torch.cuda.device(self.device):
            check(lib.naruto_assemble_rays(C.byref(b), _stream()), "naruto_assemble_rays")
        out = (rays_o, rays_d, target_s, target_d, n_cur)
        return out + (ids,) if return_ids else out

    def next_batch_struct(self, sample_num: int, current_rays: torch.Tensor, poses_all: torch.Tensor, min_pixels_cur: int, out, filter_depth: bool = False,
                          rng: Optional[torch.Tensor] = None, dyn: Optional[torch.Tensor] = None, n_cur: Optional[int] = None, n_cur_pop: Optional[int] = None,
                          keys=None):
        """The ``NarutoRayBatch`` of ``assemble_batch(..., out=out)`` WITHOUT launching anything: for ``NarutoFusedAdam.next_batch`` (the next
        iteration's batch assembled by the launch that finishes this iteration's gradients).  Device-keyed draws only (``rng``).  Returns the
        struct and the tensors it points into (keep both alive as long as the struct is in use)."""
        assert rng is not None, "a prefetched batch is keyed by the device-side iteration state"
        b, n_cur, keep = self._draw(sample_num, current_rays, poses_all, min_pixels_cur, filter_depth, rng, dyn, n_cur, n_cur_pop)
        n = sample_num + n_cur
        rays_o, rays_d, target_s, target_d = out
        for a, c in ((rays_o, 3), (rays_d, 3), (target_s, 3), (target_d, 1)):
            if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == n * c):
                raise RuntimeError(f"next_batch_struct: out tensors must be contiguous fp32 device tensors of {n} rows")
        b.rays_o, b.rays_d, b.target_s, b.target_d = rays_o.data_ptr(), rays_d.data_ptr(), target_s.data_ptr(), target_d.data_ptr()
        b.ids_out = None
        self._set_keys(b, keys)
        return b, (keep, out, rng, dyn, keys)

    def assemble_select(self, sampler, sample_num: int, current_rays: torch.Tensor, poses_all: torch.Tensor, min_pixels_cur: int, bbox,
                        uncert_vol=None, filter_depth: bool = False, out=None, rng: Optional[torch.Tensor] = None,
                        dyn: Optional[torch.Tensor] = None, n_cur: Optional[int] = None, n_cur_pop: Optional[int] = None):
        """``assemble_batch`` followed by ``sampler.sample_rays`` (coslam.py:310-359) in ONE launch, without the oversampled batch in
        between (``naruto_assemble_select``): row r of the virtual batch is what ``assemble_batch`` draws for the same keys, the selection
        is ``ActiveRaySamplerHIP.sample_rays``'s.  -> rays_o, rays_d, target_s [n_out,3], target_d [n_out,1], n_cur."""
        lib = _lib.load()
        b, n_cur, keep = self._draw(sample_num, current_rays, poses_all, min_pixels_cur, filter_depth, rng, dyn, n_cur, n_cur_pop)
        vol = sampler._volume(uncert_vol, self.device)
        base, K = sampler.base_sample_num, sampler.num_uncert_sample
        n_tail = -((-int(n_cur)) // sampler.oversample_mul)
        n_out = base + n_tail
        f32 = dict(dtype=torch.float32, device=self.device)
        if out is not None:
            o_out, d_out, s_out, t_out = out
            for a, c in ((o_out, 3), (d_out, 3), (s_out, 3), (t_out, 1)):
                if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == n_out * c):
                    raise RuntimeError(f"assemble_select: out tensors must be contiguous fp32 device tensors of {n_out} rows")
        else:
            o_out, d_out, s_out = torch.empty(n_out, 3, **f32), torch.empty(n_out, 3, **f32), torch.empty(n_out, 3, **f32)
            t_out = torch.empty(n_out, 1, **f32)
        dims = (C.c_uint32 * 3)(*vol.shape)
        bmin = (C.c_float * 3)(*(float(r[0]) for r in bbox))
        with torch.cuda.device(self.device):
            check(lib.naruto_assemble_select(C.byref(b), base, K, n_tail, vol.data_ptr(), dims, bmin, 10.0, o_out.data_ptr(), d_out.data_ptr(),
                                             s_out.data_ptr(), t_out.data_ptr(), _stream()), "naruto_assemble_select")
        return o_out, d_out, s_out, t_out, n_cur
