"""ActiveRaySamplerHIP: drop-in for the reference's ``ActiveRaySampler`` (reference
src/slam/coslam/active_ray_sampler.py:31-149) with the uncertainty lookup, the K-smallest selection and the batch
re-assembly on the device.  The reference does them in numpy on the host inside every
mapping iteration (GPU -> CPU -> GPU); here nothing leaves the device and nothing synchronises.

Quirks kept: the K rays with the SMALLEST cached value are chosen (``np.argpartition(...)[:K]``, :127); the voxel scale
is the hard-coded ``* 10`` (:112); ``-len(idx_cur) // oversample_mul`` floor-divides a negative number (:111,:131), i.e.
ceil(len/mul) rays are kept from the current frame.  Not reproducible by construction: the ORDER of the K selected rays
(numpy's introselect order is unspecified); here they come out by ascending candidate index, ties at the threshold
value go to the lower index.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .ops import _f32c, _p, _stream


def rays_to_world(rays_d_cam: torch.Tensor, pose_ids: torch.Tensor, poses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """coslam.py:342-344: (rays_o, rays_d) in world coordinates from camera-frame directions and per-ray pose ids."""
    lib = _lib.load()
    d = _f32c(rays_d_cam, "rays_d_cam").reshape(-1, 3)
    ids = pose_ids.to(device=d.device, dtype=torch.int64).contiguous()
    P = _f32c(poses, "poses").reshape(-1, 4, 4)
    n = d.shape[0]
    o_out, d_out = torch.empty(n, 3, device=d.device), torch.empty(n, 3, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(lib.naruto_rays_to_world(n, _p(d), _p(ids), _p(P), _p(o_out), _p(d_out), _stream()), "naruto_rays_to_world")
    return o_out, d_out


class ActiveRaySamplerHIP:
    def __init__(self, config: Dict = None, num_uncert_sample: int = 500, oversample_mul: int = 4) -> None:
        self.num_uncert_sample = num_uncert_sample
        self.oversample_mul = oversample_mul
        self.base_sample_num = config['mapping']['sample']
        self.oversample_num = self.base_sample_num * self.oversample_mul
        self.min_pixels_cur = config['mapping']['min_pixels_cur'] * self.oversample_mul
        self._vol_dev = None          # set_volume(): an explicitly pinned device copy of the planner's uncertainty volume

    def set_volume(self, uncert_vol, device) -> torch.Tensor:
        """Upload the planner's cached uncertainty volume once (call again whenever the planner refreshes it, every 5 frames in
        the reference); ``sample_rays(..., uncert_vol=None, ...)`` then uses this copy.  A volume of the same shape is copied INTO
        the existing device tensor, so a captured launch that reads it (naruto_amd.ba_loop.FusedBA) sees the refresh."""
        src = uncert_vol if torch.is_tensor(uncert_vol) else torch.from_numpy(np.ascontiguousarray(uncert_vol, dtype=np.float32))
        want = torch.device(device)
        # ("cuda" and "cuda:0" name the same device but compare unequal: resolve the index before comparing)
        want_idx = want.index if want.index is not None else (torch.cuda.current_device() if want.type == "cuda" else None)
        have = self._vol_dev.device if self._vol_dev is not None else None
        same_dev = have is not None and have.type == want.type and (have.index if have.index is not None else want_idx) == want_idx
        if self._vol_dev is not None and tuple(self._vol_dev.shape) == tuple(src.shape) and same_dev:
            self._vol_dev.copy_(src, non_blocking=True)
        else:
            self._vol_dev = _f32c(src.to(device), "uncert_vol")
        return self._vol_dev

    def _volume(self, uncert_vol, device) -> torch.Tensor:
        if uncert_vol is None:
            if self._vol_dev is None:
                raise RuntimeError("sample_rays(uncert_vol=None) needs a volume pinned with set_volume() first")
            return self._vol_dev
        if torch.is_tensor(uncert_vol):
            return _f32c(uncert_vol.to(device), "uncert_vol")
        # a host array is uploaded on every call (~100 k floats): the planner mutates / replaces its cached volume, and neither
        # id() nor the shape can tell a refreshed array from the old one
        return torch.from_numpy(np.ascontiguousarray(uncert_vol, dtype=np.float32)).to(device)

    def n_out(self, n_idx_cur: int) -> int:
        """Rows of the selected batch: base + ceil(len(idx_cur) / oversample_mul)."""
        return self.base_sample_num + (-((-int(n_idx_cur)) // self.oversample_mul))

    def workspace_elems(self, n_total: int) -> int:
        return _lib.load().naruto_active_ray_workspace(int(n_total), self.num_uncert_sample) // 4 + 4

    def key_lookup(self, n_total: int, idx_cur, bbox: List, keys: torch.Tensor):
        """What ``KeyFrameStoreHIP.assemble_batch(..., keys=...)`` needs to look the candidates' keys up while it assembles a batch of
        ``n_total`` rows (``NarutoRayBatch.keys_out``): (keys tensor, base, n_tail, pinned volume, bbox_min); ``sample_rays(..., keys=keys)``
        then skips its own lookup."""
        vol = self._volume(None, keys.device)
        n_idx = int(idx_cur) if isinstance(idx_cur, int) else len(idx_cur)
        n_tail = -((-n_idx) // self.oversample_mul)
        assert keys.is_cuda and keys.dtype == torch.int32 and keys.numel() >= n_total - n_tail - self.base_sample_num
        return keys, self.base_sample_num, n_tail, vol, tuple(float(b[0]) for b in bbox)

    def sample_rays(self, rays_o, rays_d, target_s, target_d, idx_cur, uncert_vol, bbox: List, out=None, workspace=None, keys=None):
        """active_ray_sampler.py:77-149.  ``idx_cur``: the current-frame indices (only their NUMBER is used, as in the reference) or
        that number.  ``out`` = (rays_o, rays_d, target_s, target_d) to write into (e.g. a captured trainer's ``ray_buffers()``) and
        ``workspace`` (int32, ``workspace_elems`` long) make the call allocation-free.  ``keys`` (int32 [candidates]): the candidates' keys as
        the batch's assembly left them (``key_lookup``): the lookup is skipped (``uncert_vol`` is then not read)."""
        lib = _lib.load()
        rays_o, rays_d, target_s = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d"), _f32c(target_s, "target_s")
        td = _f32c(target_d, "target_d").reshape(-1)
        dev = rays_o.device
        vol = self._volume(uncert_vol, dev) if keys is None else None
        n_total, base, K = rays_o.shape[0], self.base_sample_num, self.num_uncert_sample
        n_idx = int(idx_cur) if isinstance(idx_cur, int) else len(idx_cur)
        n_tail = -((-n_idx) // self.oversample_mul)                 # ceil(len / mul), as the reference's -len//mul slice
        n_out = base + n_tail
        if out is not None:
            o_out, d_out, s_out, t_out = out
            for a, c in ((o_out, 3), (d_out, 3), (s_out, 3), (t_out, 1)):
                if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.numel() == n_out * c):
                    raise RuntimeError(f"sample_rays: out tensors must be contiguous fp32 device tensors of {n_out} rows")
        else:
            o_out, d_out, s_out = (torch.empty(n_out, 3, device=dev) for _ in range(3))
            t_out = torch.empty(n_out, 1, device=dev)
        if keys is not None:
            assert keys.is_cuda and keys.dtype == torch.int32 and keys.numel() >= n_total - n_tail - base
            with torch.cuda.device(dev):
                _lib.check(lib.naruto_active_ray_select_keyed(n_total, base, K, n_tail, _p(rays_o), _p(rays_d), _p(target_s), _p(td), _p(keys),
                                                              _p(o_out), _p(d_out), _p(s_out), _p(t_out), _stream()), "naruto_active_ray_select_keyed")
            return o_out, d_out, s_out, t_out
        dims = (C.c_uint32 * 3)(*vol.shape)
        bmin = (C.c_float * 3)(*(float(b[0]) for b in bbox))
        with torch.cuda.device(dev):
            ws = workspace if workspace is not None else torch.empty(self.workspace_elems(n_total), dtype=torch.int32, device=dev)
            assert ws.dtype == torch.int32 and ws.numel() >= self.workspace_elems(n_total)
            _lib.check(lib.naruto_active_ray_select(n_total, base, K, n_tail, _p(rays_o), _p(rays_d), _p(target_s), _p(td), _p(vol), dims, bmin, 10.0,
                                                    _p(o_out), _p(d_out), _p(s_out), _p(t_out), _p(ws), _stream()), "naruto_active_ray_select")
        return o_out, d_out, s_out, t_out
