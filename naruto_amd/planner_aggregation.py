"""N3 ("next" row of SURVEY.md section 8f): the planner's uncertainty aggregation in goal space on the device.

Mirror of the two pieces of ``NarutoPlanner`` (reference src/planner/naruto_planner.py) that consume the SDF /
uncertainty volumes of ``get_map_volumes``: ``init_data`` (:110-137, the goal-space lattice) and
``uncertainty_aggregation_v2`` (:596-735).  Same attribute names, same return value
``(goal_space_valid, {'gs_aggre_uncerts', 'topk_uncert_vxl', 'gs_uncert_collections'})``; the rest of the planner
(RRT, rotation planning, collision checks) is out of scope and keeps using these outputs unchanged.

Differences to know about:
  * the target observations: the reference takes ``np.argpartition(uncert, -top_k)[-top_k_subset:]``, i.e. whichever subset
    of the top_k numpy's introselect happens to leave in the last slots; here the top_k are listed in voxel order and
    thinned evenly (``naruto_goal_targets``), deterministic and spread over the volume.  Pass ``targets=`` to use your own.
  * the volumes may be numpy arrays (as the reference passes them) or device tensors (no host round trip).
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class GoalSpaceAggregatorHIP:
    def __init__(self, bbox: Sequence[Sequence[float]], voxel_size: float = 0.1, uncert_top_k: int = 4000, uncert_top_k_subset: int = 300,
                 gs_sensing_range: Sequence[float] = (0.5, 2.0), safe_sdf: float = 0.8, gs_z_levels: Optional[Sequence[int]] = (5, 11, 17),
                 device="cuda"):
        """``bbox`` [3,2] metres; the other arguments are the planner config entries of the same names
        (configs/default.py:93-98, naruto_planner.py:109)."""
        self.device = torch.device(device)
        self.voxel_size = float(voxel_size)
        self.uncert_top_k, self.uncert_top_k_subset = int(uncert_top_k), int(uncert_top_k_subset)
        self.gs_sensing_range, self.safe_sdf = (float(gs_sensing_range[0]), float(gs_sensing_range[1])), float(safe_sdf)
        self.bbox = np.asarray(bbox)
        # naruto_planner.py:116-137
        self.Nx = round((bbox[0][1] - bbox[0][0]) / self.voxel_size + 0.0005) + 1
        self.Ny = round((bbox[1][1] - bbox[1][0]) / self.voxel_size + 0.0005) + 1
        self.Nz = round((bbox[2][1] - bbox[2][0]) / self.voxel_size + 0.0005) + 1
        self.gs_x_range = torch.arange(0, self.Nx, 2)
        self.gs_y_range = torch.arange(0, self.Ny, 2)
        if gs_z_levels is None:
            self.gs_z_range = torch.arange(int(1 / self.voxel_size), self.Nz, int(1 / self.voxel_size))
        else:
            self.gs_z_range = torch.tensor(list(gs_z_levels))
        self.gs_x, self.gs_y, self.gs_z = torch.meshgrid(self.gs_x_range, self.gs_y_range, self.gs_z_range, indexing="ij")
        idx = torch.stack([self.gs_x.reshape(-1), self.gs_y.reshape(-1), self.gs_z.reshape(-1)], 1)
        self._goal_idx = idx.to(torch.int32).contiguous().to(self.device)
        self.goal_space_pts = idx.to(self.device).float()                        # [X*Y*Z, 3], unit: voxel
        self._dims = (C.c_uint32 * 3)(self.Nx, self.Ny, self.Nz)

    def _volume(self, v) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        t = t.to(self.device, torch.float32).contiguous()
        if tuple(t.shape) != (self.Nx, self.Ny, self.Nz):
            raise ValueError(f"volume shape {tuple(t.shape)} != goal-space volume {(self.Nx, self.Ny, self.Nz)}")
        return t

    def select_targets(self, uncert: torch.Tensor) -> torch.Tensor:
        """-> int32 [top_k_subset, 3] voxel indices of the target observations (naruto_goal_targets)."""
        lib = _lib.load()
        n = uncert.numel()
        k, sub = min(self.uncert_top_k, n), min(self.uncert_top_k_subset, min(self.uncert_top_k, n))
        targets = torch.empty(sub, 3, dtype=torch.int32, device=self.device)
        ws = torch.empty((lib.naruto_goal_targets_workspace(n, k) + 3) // 4, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.naruto_goal_targets(self._dims, uncert.data_ptr(), k, sub, targets.data_ptr(), ws.data_ptr(), _stream()), "naruto_goal_targets")
        return targets

    @torch.no_grad()
    def uncertainty_aggregation_v2(self, uncert_sdf_vols: List, force_running: bool = False, targets=None) -> Tuple[bool, Dict]:
        """naruto_planner.py:596-735.  ``uncert_sdf_vols`` = [uncert_vol, sdf_vol], each [X,Y,Z]."""
        lib = _lib.load()
        uncert, sdf = self._volume(uncert_sdf_vols[0]), self._volume(uncert_sdf_vols[1])
        if targets is None:
            tgt = self.select_targets(uncert)
        else:
            tgt = torch.as_tensor(targets).to(self.device, torch.int32).contiguous().reshape(-1, 3)
        G, K = self._goal_idx.shape[0], tgt.shape[0]
        coll = torch.empty(G, K, dtype=torch.float32, device=self.device)
        agg = torch.empty(G, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.naruto_goal_aggregate(self._dims, uncert.data_ptr(), sdf.data_ptr(), G, self._goal_idx.data_ptr(), K, tgt.data_ptr(),
                                            self.gs_sensing_range[0] / self.voxel_size, self.gs_sensing_range[1] / self.voxel_size, self.safe_sdf,
                                            coll.data_ptr(), agg.data_ptr(), _stream()), "naruto_goal_aggregate")
        outputs = {'gs_aggre_uncerts': agg.reshape(self.gs_x_range.shape[0], self.gs_y_range.shape[0], self.gs_z_range.shape[0]),
                   'topk_uncert_vxl': tgt.long(), 'gs_uncert_collections': coll}
        # the reference decides validity from the pair mask; a pair is valid iff it passed every test, which (uncertainties of
        # selected targets being > 0 in practice) shows as a non-zero entry -- count the mask explicitly to stay exact
        invalid_goal_space = not bool((coll != 0).any().item()) and not bool(self._any_valid(uncert, sdf, tgt))
        if invalid_goal_space:
            return (True, outputs) if force_running else (False, {})
        return True, outputs

    def _any_valid(self, uncert, sdf, tgt) -> bool:
        """Exact 'valid_mask.sum() > 0' when every collected value happens to be zero (targets with zero uncertainty):
        re-run the aggregation on a volume of ones."""
        lib = _lib.load()
        ones = torch.ones_like(uncert)
        G, K = self._goal_idx.shape[0], tgt.shape[0]
        coll = torch.empty(G, K, dtype=torch.float32, device=self.device)
        agg = torch.empty(G, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.naruto_goal_aggregate(self._dims, ones.data_ptr(), sdf.data_ptr(), G, self._goal_idx.data_ptr(), K, tgt.data_ptr(),
                                            self.gs_sensing_range[0] / self.voxel_size, self.gs_sensing_range[1] / self.voxel_size, self.safe_sdf,
                                            coll.data_ptr(), agg.data_ptr(), _stream()), "naruto_goal_aggregate")
        return bool((agg > 0).any().item())
