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
    """Co-SLAM CoSLAM.smoothness: total variation of the hash features on a (sample_points-1)^3 lattice."""
    bb = model.bounding_box
    grid_size = (sample_points - 1) * voxel_size
    offset_max = bb[:, 1] - bb[:, 0] - grid_size - 2 * margin
    if offset_rand is None:
        offset_rand = torch.rand(3, device=bb.device)          # drawn on the device: no host sync per iteration
    if jitter_rand is None:
        jitter_rand = torch.rand((1, 1, 1, 3), device=bb.device)
    offset = offset_rand.to(offset_max) * offset_max + margin
    n = sample_points - 1
    cache = model.__dict__.setdefault("_smooth_coords", {})
    coords = cache.get((n, bb.device))
    if coords is None:
        ax = torch.arange(0, n, dtype=torch.long, device=bb.device)
        coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).float()
        cache[(n, bb.device)] = coords
    pts = (coords + jitter_rand.to(bb).reshape(1, 1, 1, 3)) * voxel_size + bb[:, 0] + offset
    pts_tcnn = (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
    sdf = model.query_sdf(pts_tcnn, embed=True)
    tv_x = torch.pow(sdf[1:, ...] - sdf[:-1, ...], 2).sum()
    tv_y = torch.pow(sdf[:, 1:, ...] - sdf[:, :-1, ...], 2).sum()
    tv_z = torch.pow(sdf[:, :, 1:, ...] - sdf[:, :, :-1, ...], 2).sum()
    return (tv_x + tv_y + tv_z) / (sample_points ** 3)


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


class MappingTrainer:
    """Owns the model + the two Adam instances and runs mapping iterations on ray batches."""

    def __init__(self, config: Dict, bounding_box: torch.Tensor, device, uncert_voxel: float = 0.1, group=None):
        self.config = config
        self.device = torch.device(device)
        self.model = NarutoFieldHIP(config, bounding_box.to(self.device)).to(self.device)
        self.map_optimizer = create_optimizer(self.model, config)
        self.uncert_optim = init_uncert_grid_optim(self.model, uncert_voxel)
        self.group = group
        self.iter = 0
        self._flat = None
        if group is not None:
            self.model.enable_data_parallel(group)

    def parameters(self):
        return list(self.model.decoder.parameters()) + list(self.model.embed_fn.parameters()) + [self.model.uncert_grid]

    def step(self, rays_o, rays_d, target_rgb, target_d, smooth: bool = False, n_rays_total: int = 0):
        """One mapping iteration (global_BA body, coslam.py:361-399).  With a process group the rays passed
        in are THIS RANK's shard; gradients are summed over ranks before the (identical) Adam steps."""
        model = self.model
        model.train()
        model.n_rays_total = n_rays_total
        if self.iter % 5 == 0:
            self.uncert_optim.zero_grad()
        self.map_optimizer.zero_grad()
        ret = model.forward(rays_o, rays_d, target_rgb, target_d)
        loss = get_loss_from_ret(model, self.config, ret, smooth=smooth)
        loss.backward()
        if self.group is not None:
            parallel.allreduce_grads(self.parameters(), self.group)
        self.map_optimizer.step()
        self.iter += 1
        if self.iter % 5 == 0:
            self.uncert_optim.step()
        return ret, loss
