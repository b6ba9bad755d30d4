"""BENCH / TEST HARNESS (not product code): the reference's OWN mapping loop body, restated against any model with the reference's
attribute surface.

This is what the reference's ``CoSLAMNaruto`` executes per iteration after INTEGRATION.md's two-line swap
(``self.model = NarutoFieldHIP(...)`` at reference src/slam/coslam/coslam.py:65) -- caller code this repository does NOT replace:

    ret  = self.model.forward(rays_o, rays_d, target_s, target_d)          coslam.py:364
    loss = self.get_loss_from_ret(ret, smooth=True)                         coslam.py:366, 154-174 (ten scalar torch ops)
           ... self.smoothness(...) through self.model.query_sdf(embed=True)   Co-SLAM CoSLAM.smoothness [not in tree]
    loss.backward(retain_graph=True)                                        coslam.py:368
    self.map_optimizer.step(); self.map_optimizer.zero_grad()               coslam.py:370-376 (torch.optim.Adam, create_optimizer :409-419)
    every 5th: self.uncert_optim.step(); self.uncert_optim.zero_grad()      coslam.py:397-399

``bench.py --path dropin`` times it and the GPU tests check it against the oracle driven the same way (both put tools/ on sys.path); it is also the template for
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
            from naruto_amd.trainer import FusedAdam
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
            from naruto_amd.trainer import smoothness as fused_smoothness
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
        """coslam.py:154-174: naruto_amd.trainer.get_loss_from_ret is that function (the same scalar torch ops in the same order); the
        smoothness term is added here because this caller can also run Co-SLAM's own torch smoothness."""
        from naruto_amd.trainer import get_loss_from_ret
        tr = self.config['training']
        if not (smooth and tr['smooth_weight'] > 0):
            return get_loss_from_ret(self.model, self.config, ret, rgb, sdf, depth, fs, uncert, smooth=False)
        # the reference adds the smoothness term BEFORE the uncertainty term (coslam.py:166-172): keep its order of additions
        loss = get_loss_from_ret(self.model, self.config, ret, rgb, sdf, depth, fs, uncert=False, smooth=False)
        loss = loss + tr['smooth_weight'] * self.smoothness(tr['smooth_pts'], tr['smooth_vox'], margin=tr['smooth_margin'])
        if uncert and (self.config['decoder']['pred_uncert'] or self.config['decoder']['uncert_grid']):
            loss = loss + tr['uncert_weight'] * ret['uncert_loss']
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
