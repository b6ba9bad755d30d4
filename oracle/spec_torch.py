"""CPU restatement (pure PyTorch, fp32) of NARUTO's neural-implicit mapping hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``naruto_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the timed CPU baseline -- never as the
product path.

What is restated, and where it comes from (paths under /root/reference):

* In-tree (pinned by importing the reference itself, see oracle/make_golden.py):
    - z sampling, render_rays ............ src/slam/coslam/model/scene_rep.py:150-225
    - calc_embedding / uncert grid ....... src/slam/coslam/model/scene_rep.py:49-64
    - raw2outputs ........................ src/slam/coslam/model/scene_rep.py:66-96
    - query_sdf / query_color_sdf ........ src/slam/coslam/model/scene_rep.py:98-148
    - forward (losses) ................... src/slam/coslam/model/scene_rep.py:227-287
    - SDFNetNaruto / ColorSDFNet_v2_Naruto  src/slam/coslam/model/decoder.py:18-116
    - get_loss_from_ret .................. src/slam/coslam/coslam.py:154-174
    - get_map_volumes .................... src/slam/coslam/coslam_utils.py:58-97
* NOT in tree -- **parity unpinned** (restated from the published algorithms; the
  reference has no tests / golden vectors and the dependencies are absent):
    - tiny-cuda-nn (unpinned git HEAD, README.md:171-173) ``HashGrid`` and
      ``OneBlob`` encodings: include/tiny-cuda-nn/encodings/grid.h
      (grid_scale, grid_resolution, pos_fract, grid_index, coherent_prime_hash,
      kernel_grid) and encodings/oneblob.h + common_device.h (quartic_cdf).
    - HengyiWang/Co-SLAM @ 3bb904e (scripts/installation/conda_env/build.sh:22-23):
      model/scene_rep.py (get_resolution, get_encoding, sdf2weights, run_network,
      query_color), model/decoder.py (ColorNet), model/encodings.py (get_encoder),
      model/utils.py (get_masks, get_sdf_loss, compute_loss, mse2psnr,
      coordinates), utils.py (getVoxels), coslam.py (smoothness).
  Call sites that anchor them: scene_rep.py:20-22,59,80,110,114,144,184,
  decoder.py:11-15, coslam.py:29-30,168, coslam_utils.py:33.
"""

from __future__ import annotations

import ctypes
import ctypes.util
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# tcnn derives the level tables with the C float functions std::log2(float) / exp2f / ceilf; use the very
# same libm entry points so that the restatement does not depend on numpy's own exp2 rounding.
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _fn in ("exp2f", "log2f", "ceilf"):
    getattr(_libm, _fn).restype = ctypes.c_float
    getattr(_libm, _fn).argtypes = [ctypes.c_float]

# tiny-cuda-nn coherent_prime_hash primes (grid.h) -- first is 1 for memory coherence
HASH_PRIMES = (1, 2654435761, 805459861)
U32 = 0xFFFFFFFF


# ----------------------------------------------------------------------------------
# A3: multi-resolution hash grid (tcnn HashGrid, linear interpolation)  [parity unpinned]
# ----------------------------------------------------------------------------------
class HashGridMeta:
    """Level tables exactly as tcnn's GridEncodingTemplated constructor derives them."""

    def __init__(self, n_levels=16, n_features=2, log2_hashmap_size=16,
                 base_resolution=16, per_level_scale=2.0):
        self.n_levels = int(n_levels)
        self.n_features = int(n_features)
        self.log2_hashmap_size = int(log2_hashmap_size)
        self.base_resolution = int(base_resolution)
        # tcnn reads per_level_scale into a float and takes std::log2 of that float
        self.per_level_scale = np.float32(per_level_scale)
        self.log2_per_level_scale = np.float32(_libm.log2f(float(self.per_level_scale)))
        scales, ress, sizes, offsets = [], [], [], [0]
        for lvl in range(self.n_levels):
            # grid_scale(): exp2f(level * log2_pls) * base - 1
            scale = np.float32(np.float32(_libm.exp2f(float(np.float32(lvl) * self.log2_per_level_scale)))
                               * np.float32(self.base_resolution) - np.float32(1.0))
            res = int(_libm.ceilf(float(scale))) + 1          # grid_resolution()
            max_params = (2 ** 32 - 1) // 2
            dense = res ** 3
            params = max_params if float(res) ** 3 > float(max_params) else dense
            params = (params + 7) // 8 * 8                    # next_multiple(.., 8)
            params = min(params, 1 << self.log2_hashmap_size)  # GridType::Hash
            scales.append(scale)
            ress.append(res)
            sizes.append(params)
            offsets.append(offsets[-1] + params)
        self.scale = np.asarray(scales, dtype=np.float32)
        self.resolution = np.asarray(ress, dtype=np.int64)
        self.size = np.asarray(sizes, dtype=np.int64)
        self.offset = np.asarray(offsets, dtype=np.int64)
        self.n_entries = int(offsets[-1])
        self.n_params = self.n_entries * self.n_features
        self.n_output_dims = self.n_levels * self.n_features

    @staticmethod
    def from_desired_resolution(desired_resolution, n_levels=16, n_features=2,
                                log2_hashmap_size=16, base_resolution=16):
        # Co-SLAM model/encodings.py get_encoder():
        #   per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        pls = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        return HashGridMeta(n_levels, n_features, log2_hashmap_size, base_resolution, pls)


def hash_grid_index(meta: HashGridMeta, lvl: int, gx, gy, gz):
    """grid_index<3, CoherentPrime>() on uint32 coordinates held in int64 tensors."""
    res = int(meta.resolution[lvl])
    size = int(meta.size[lvl])
    # loop "dim < N_DIMS && stride <= hashmap_size" then hash iff hashmap_size < stride
    stride, dims_used = 1, 0
    for _ in range(3):
        if stride > size:
            break
        dims_used += 1
        stride *= res
    if size < stride:
        idx = ((gx * HASH_PRIMES[0]) & U32) ^ ((gy * HASH_PRIMES[1]) & U32) ^ ((gz * HASH_PRIMES[2]) & U32)
    else:
        assert dims_used == 3
        idx = (gx + ((gy * res) & U32) + ((gz * (res * res)) & U32)) & U32
    return idx % size


def hash_encode(x: torch.Tensor, table: torch.Tensor, meta: HashGridMeta) -> torch.Tensor:
    """x [M,3] (any range; no clamping) , table [n_entries*F] -> [M, L*F], level-major."""
    M = x.shape[0]
    Fd = meta.n_features
    tab = table.view(-1, Fd)
    outs = []
    for lvl in range(meta.n_levels):
        scale = float(meta.scale[lvl])
        # fmaf(scale, x, 0.5f): product and sum are exact in fp64, one rounding to fp32
        pos = (x.double() * scale + 0.5).to(x.dtype)
        g = torch.floor(pos)
        w = pos - g                                 # fractional part, identity interpolation
        gi = g.to(torch.int64) & U32                # (uint32_t)(int)floorf
        off = int(meta.offset[lvl])
        res = torch.zeros(M, Fd, dtype=x.dtype, device=x.device)
        for corner in range(8):
            wgt = torch.ones(M, dtype=x.dtype, device=x.device)
            c = []
            for dim in range(3):
                if (corner >> dim) & 1:
                    wgt = wgt * w[:, dim]
                    c.append((gi[:, dim] + 1) & U32)
                else:
                    wgt = wgt * (1 - w[:, dim])
                    c.append(gi[:, dim])
            idx = hash_grid_index(meta, lvl, c[0], c[1], c[2]) + off
            res = res + wgt[:, None] * tab[idx]
        outs.append(res)
    return torch.cat(outs, dim=-1)


# ----------------------------------------------------------------------------------
# A4: OneBlob (tcnn, quartic kernel, n_bins = 2^k)  [parity unpinned]
# ----------------------------------------------------------------------------------
def quartic_cdf(x: torch.Tensor, inv_radius: float) -> torch.Tensor:
    u = x * inv_radius
    u2 = u * u
    u4 = u2 * u2
    return torch.clamp((15.0 / 16.0) * u * (1 - (2.0 / 3.0) * u2 + (1.0 / 5.0) * u4) + 0.5, 0.0, 1.0)


def oneblob_encode(x: torch.Tensor, n_bins: int = 16) -> torch.Tensor:
    """x [M,D] -> [M, D*n_bins], dim-major (column d*n_bins + b)."""
    M, D = x.shape
    b = torch.arange(n_bins, dtype=x.dtype, device=x.device) / n_bins            # left boundaries
    t = b[None, None, :] - x[:, :, None]                                          # [M,D,B]
    left = quartic_cdf(t, n_bins) + quartic_cdf(t - 1.0, n_bins) + quartic_cdf(t + 1.0, n_bins)
    right = torch.roll(left, shifts=-1, dims=-1).clone()                          # __shfl width=n_bins
    right[..., -1] = right[..., -1] + 1.0
    return (right - left).reshape(M, D * n_bins)


# ----------------------------------------------------------------------------------
# A2: uncertainty voxel grid, trilinear (scene_rep.py:58-64) -- via torch's grid_sample, as the
# reference does, and a hand-written restatement the C oracle / HIP kernel follow.
# ----------------------------------------------------------------------------------
def sample_uncert_grid_ref(uncert_grid: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    grid = (x * 2 - 1)[None, None, None, ...]
    u = F.grid_sample(uncert_grid[None, None, ...], grid, align_corners=False)
    return u.reshape(-1)


def sample_uncert_grid_manual(uncert_grid: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Same thing spelled out.  uncert_grid is [Nx,Ny,Nz]; grid_sample's (x,y,z) index (W,H,D) =
    (Nz,Ny,Nx), so input coordinate 0 walks the LAST axis and coordinate 2 the FIRST (the x<->z quirk)."""
    D, H, W = uncert_grid.shape
    g = x * 2 - 1
    ix = ((g[:, 0] + 1) * W - 1) / 2
    iy = ((g[:, 1] + 1) * H - 1) / 2
    iz = ((g[:, 2] + 1) * D - 1) / 2
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    fx, fy, fz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = torch.zeros_like(ix)
    flat = uncert_grid.reshape(-1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                wgt = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                idx = (zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1)
                out = out + torch.where(ok, flat[idx] * wgt, torch.zeros_like(wgt))
    return out


# ----------------------------------------------------------------------------------
# A6: sdf2weights (Co-SLAM model/scene_rep.py)  [parity unpinned]
# ----------------------------------------------------------------------------------
def sdf2weights(sdf: torch.Tensor, z_vals: torch.Tensor, trunc: float, sc_factor: float) -> torch.Tensor:
    weights = torch.sigmoid(sdf / trunc) * torch.sigmoid(-sdf / trunc)
    signs = sdf[:, 1:] * sdf[:, :-1]
    mask = torch.where(signs < 0.0, torch.ones_like(signs), torch.zeros_like(signs))
    inds = torch.argmax(mask, dim=1)            # first sign change; 0 if none
    z_min = torch.gather(z_vals, 1, inds[..., None])
    mask = torch.where(z_vals < z_min + sc_factor * trunc, torch.ones_like(z_vals), torch.zeros_like(z_vals))
    weights = weights * mask
    return weights / (torch.sum(weights, dim=-1, keepdim=True) + 1e-8)


# ----------------------------------------------------------------------------------
# A7: raw2outputs (scene_rep.py:66-96)
# ----------------------------------------------------------------------------------
def raw2outputs(raw, z_vals, trunc, sc_factor, white_bkgd=False):
    rgb = torch.sigmoid(raw[..., :3])
    weights = sdf2weights(raw[..., 3], z_vals, trunc, sc_factor)
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    depth_var = torch.sum(weights * torch.square(z_vals - depth_map.unsqueeze(-1)), dim=-1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(weights, -1))
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    uncert = F.softplus(raw[..., 4]) + 0.01
    uncert_map = torch.sum(weights * weights * uncert, -1)
    return rgb_map, disp_map, acc_map, weights, depth_map, depth_var, uncert_map


# ----------------------------------------------------------------------------------
# A1: z sampling (scene_rep.py:158-180)
# ----------------------------------------------------------------------------------
def sample_z(n_rays, target_d, near, far, n_samples_d, n_range_d, range_d, perturb, rand=None,
             n_samples=None, device=None, dtype=torch.float32):
    if target_d is not None:
        z_samples = torch.linspace(-range_d, range_d, steps=n_range_d).to(target_d)
        z_samples = z_samples[None, :].repeat(n_rays, 1) + target_d
        z_samples[target_d.squeeze(-1) <= 0] = torch.linspace(near, far, steps=n_range_d).to(target_d)
        if n_samples_d > 0:
            z_vals = torch.linspace(near, far, n_samples_d)[None, :].repeat(n_rays, 1).to(target_d)
            z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        else:
            z_vals = z_samples
    else:
        z_vals = torch.linspace(near, far, n_samples).to(device=device, dtype=dtype)
        z_vals = z_vals[None, :].repeat(n_rays, 1)
    if perturb > 0.:
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        if rand is None:
            rand = torch.rand(z_vals.shape).to(z_vals)
        z_vals = lower + (upper - lower) * rand
    return z_vals


# ----------------------------------------------------------------------------------
# A8: losses (scene_rep.py:248-285 + Co-SLAM model/utils.py get_masks/get_sdf_loss [unpinned])
# ----------------------------------------------------------------------------------
def get_masks(z_vals, target_d, truncation):
    front_mask = torch.where(z_vals < (target_d - truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    back_mask = torch.where(z_vals > (target_d + truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    depth_mask = torch.where(target_d > 0.0, torch.ones_like(target_d), torch.zeros_like(target_d))
    sdf_mask = (1.0 - front_mask) * (1.0 - back_mask) * depth_mask
    num_fs_samples = torch.count_nonzero(front_mask)
    num_sdf_samples = torch.count_nonzero(sdf_mask)
    num_samples = num_sdf_samples + num_fs_samples
    fs_weight = 1.0 - num_fs_samples / num_samples
    sdf_weight = 1.0 - num_sdf_samples / num_samples
    return front_mask, sdf_mask, fs_weight, sdf_weight


def get_sdf_loss(z_vals, target_d, predicted_sdf, truncation):
    front_mask, sdf_mask, fs_weight, sdf_weight = get_masks(z_vals, target_d, truncation)
    fs_loss = F.mse_loss(predicted_sdf * front_mask, torch.ones_like(predicted_sdf) * front_mask) * fs_weight
    sdf_loss = F.mse_loss((z_vals + predicted_sdf * truncation) * sdf_mask, target_d * sdf_mask) * sdf_weight
    return fs_loss, sdf_loss


def mse2psnr(x):
    return -10. * torch.log(x) / torch.log(torch.Tensor([10.])).to(x)


def render_losses(rend, target_rgb, target_d, depth_trunc, rgb_missing, trunc, sc_factor):
    td = target_d.squeeze(-1)
    valid = (td > 0.) * (td < depth_trunc)
    # NB (scene_rep.py:249-250): rgb_weight stays a BOOL tensor, so writing rgb_missing (0.05) into it
    # stores True -- rays with invalid depth keep colour weight 1 (0 only if rgb_missing == 0).
    rgb_weight = valid.clone().unsqueeze(-1)
    rgb_weight[rgb_weight == 0] = rgb_missing
    rgb_loss = F.mse_loss(rend["rgb"] * rgb_weight, target_rgb * rgb_weight)
    psnr = mse2psnr(rgb_loss)
    depth_loss = F.mse_loss(rend["depth"][valid], td[valid])
    fs_loss, sdf_loss = get_sdf_loss(rend["z_vals"], target_d, rend["raw"][..., 3], trunc * sc_factor)
    um = rend["uncert_map"][valid]
    x = rend["depth"][valid]
    y = td[valid]
    # NB (scene_rep.py:284): [Nv,1] * [Nv] broadcasts to [Nv,Nv] -- mean of an outer product.
    uncert_loss = torch.mean((1 / (2 * (um + 1e-9).unsqueeze(-1))) * ((x - y) ** 2)) + 0.5 * torch.mean(torch.log(um + 1e-9))
    return {"rgb": rend["rgb"], "depth": rend["depth"], "rgb_loss": rgb_loss, "depth_loss": depth_loss,
            "sdf_loss": sdf_loss, "fs_loss": fs_loss, "psnr": psnr, "uncert_loss": uncert_loss}


def total_loss(ret, tr_cfg, smooth_term=None):
    """coslam.py:154-174 get_loss_from_ret (uncert_grid on)."""
    loss = tr_cfg['rgb_weight'] * ret['rgb_loss'] + tr_cfg['depth_weight'] * ret['depth_loss'] \
        + tr_cfg['sdf_weight'] * ret['sdf_loss'] + tr_cfg['fs_weight'] * ret['fs_loss']
    if smooth_term is not None:
        loss = loss + tr_cfg['smooth_weight'] * smooth_term
    loss = loss + tr_cfg['uncert_weight'] * ret['uncert_loss']
    return loss


# ----------------------------------------------------------------------------------
# The field: mirror of JointEncodingNaruto + JointEncoding (+ decoder) as one torch module.
# ----------------------------------------------------------------------------------
def get_resolution(bounding_box: torch.Tensor, voxel: float) -> int:
    """Co-SLAM JointEncoding.get_resolution."""
    dim_max = (bounding_box[:, 1] - bounding_box[:, 0]).max()
    if voxel > 10:
        return int(voxel)
    return int(dim_max / voxel)


def uncert_grid_dims(bounding_box: torch.Tensor, voxel_size: float):
    """scene_rep.py:50-52."""
    return tuple(round((bounding_box[i, 1] - bounding_box[i, 0]).item() / voxel_size + 0.0005) + 1 for i in range(3))


class OracleField(nn.Module):
    def __init__(self, config: Dict, bounding_box: torch.Tensor, uncert_voxel: float = 0.1,
                 n_levels: int = 16, n_features: int = 2, base_resolution: int = 16):
        super().__init__()
        self.config = config
        self.register_buffer("bounding_box", bounding_box.clone().float(), persistent=False)
        self.resolution_sdf = get_resolution(self.bounding_box, config['grid']['voxel_sdf'])
        self.meta = HashGridMeta.from_desired_resolution(
            self.resolution_sdf, n_levels, n_features, config['grid']['hash_size'], base_resolution)
        self.n_bins = config['pos']['n_bins']
        dec = config['decoder']
        self.geo = dec['geo_feat_dim']
        in_sdf = self.meta.n_output_dims + 3 * self.n_bins
        in_col = 3 * self.n_bins + self.geo
        self.table = nn.Parameter((torch.rand(self.meta.n_params) * 2 - 1) * 1e-4)   # tcnn init U(-1e-4,1e-4)
        self.sdf_w0 = nn.Parameter(torch.empty(dec['hidden_dim'], in_sdf))
        self.sdf_w1 = nn.Parameter(torch.empty(1 + self.geo, dec['hidden_dim']))
        self.col_w0 = nn.Parameter(torch.empty(dec['hidden_dim_color'], in_col))
        self.col_w1 = nn.Parameter(torch.empty(3, dec['hidden_dim_color']))
        for w in (self.sdf_w0, self.sdf_w1, self.col_w0, self.col_w1):
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))                               # nn.Linear default
        nx, ny, nz = uncert_grid_dims(self.bounding_box, uncert_voxel)
        self.uncert_grid = nn.Parameter(torch.ones(nx, ny, nz) * 3)

    # -- A2..A5 ------------------------------------------------------------------
    def calc_embedding(self, x):
        embed = hash_encode(x, self.table, self.meta)
        u = sample_uncert_grid_ref(self.uncert_grid, x)
        return torch.cat([u[:, None], embed], dim=1)

    def sdf_net(self, embed33, pos48):
        h = F.relu(F.linear(torch.cat([embed33[:, 1:], pos48], -1), self.sdf_w0))
        out = F.linear(h, self.sdf_w1)
        return torch.cat([out, embed33[:, :1]], dim=1)                # [M, 1+geo+1]

    def query_color_sdf(self, query_points):
        x = query_points.reshape(-1, query_points.shape[-1])
        pos = oneblob_encode(x, self.n_bins)
        h = self.sdf_net(self.calc_embedding(x), pos)
        sdf, geo, unc = h[:, :1], h[:, 1:1 + self.geo], h[:, 1 + self.geo:]
        c = F.relu(F.linear(torch.cat([pos, geo], -1), self.col_w0))
        rgb = F.linear(c, self.col_w1)
        return torch.cat([rgb, sdf, unc], -1)

    def query_color(self, query_points):
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def query_sdf(self, query_points, return_geo=False, embed=False, return_uncert=False):
        x = query_points.reshape(-1, query_points.shape[-1])
        if embed:
            e = hash_encode(x, self.table, self.meta)
            return e.reshape(list(query_points.shape[:-1]) + [e.shape[-1]])
        h = self.sdf_net(self.calc_embedding(x), oneblob_encode(x, self.n_bins))
        sdf, geo, unc = h[:, :1], h[:, 1:1 + self.geo], h[:, 1 + self.geo:]
        sdf = sdf.reshape(list(query_points.shape[:-1]))
        if return_uncert:
            sdf = torch.stack([sdf, unc.reshape(list(query_points.shape[:-1]))], -1)
        if not return_geo:
            return sdf
        return sdf, geo.reshape(list(query_points.shape[:-1]) + [geo.shape[-1]])

    def run_network(self, pts):
        flat = pts.reshape(-1, pts.shape[-1])
        flat = (flat - self.bounding_box[:, 0]) / (self.bounding_box[:, 1] - self.bounding_box[:, 0])
        out = self.query_color_sdf(flat)
        return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])

    # -- A1, A6, A7 --------------------------------------------------------------
    def render_rays(self, rays_o, rays_d, target_d=None, rand=None):
        tr, cam = self.config['training'], self.config['cam']
        z_vals = sample_z(rays_o.shape[0], target_d, cam['near'], cam['far'], tr['n_samples_d'],
                          tr['n_range_d'], tr['range_d'], tr['perturb'], rand=rand,
                          n_samples=tr.get('n_samples'), device=rays_o.device, dtype=rays_o.dtype)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        raw = self.run_network(pts)
        rgb_map, disp_map, acc_map, weights, depth_map, depth_var, uncert_map = raw2outputs(
            raw, z_vals, tr['trunc'], self.config['data']['sc_factor'], tr['white_bkgd'])
        return {'rgb': rgb_map, 'depth': depth_map, 'disp_map': disp_map, 'acc_map': acc_map,
                'depth_var': depth_var, 'z_vals': z_vals, 'raw': raw, 'uncert_map': uncert_map,
                'weights': weights}

    def render_surface_color(self, rays_o, normal):
        """Co-SLAM JointEncoding.render_surface_color [third_parties/coslam/model/scene_rep.py, not in tree; call site reference
        coslam.py:446-447 via extract_mesh, coslam_utils.py:178-186]: n_range_d samples at linspace(-trunc, trunc) along the normal."""
        tr = self.config['training']
        z_vals = torch.linspace(-tr['trunc'], tr['trunc'], steps=tr['n_range_d']).to(rays_o).repeat(rays_o.shape[0], 1)
        pts = rays_o[..., None, :] + normal[..., None, :] * z_vals[..., :, None]
        raw = self.run_network(pts)
        return raw2outputs(raw, z_vals, tr['trunc'], self.config['data']['sc_factor'], tr['white_bkgd'])[0]

    # -- A8 ----------------------------------------------------------------------
    def forward(self, rays_o, rays_d, target_rgb, target_d, global_step=0, rand=None):
        rend = self.render_rays(rays_o, rays_d, target_d=target_d, rand=rand)
        if not self.training:
            return rend
        tr = self.config['training']
        return render_losses(rend, target_rgb, target_d, self.config['cam']['depth_trunc'],
                             tr['rgb_missing'], tr['trunc'], self.config['data']['sc_factor'])

    def param_groups(self):
        """coslam.py:409-419 create_optimizer + :240-243 init_uncert_grid_optim."""
        return ([{'params': [self.sdf_w0, self.sdf_w1, self.col_w0, self.col_w1], 'weight_decay': 1e-6,
                  'lr': self.config['mapping']['lr_decoder']},
                 {'params': [self.table], 'eps': 1e-15, 'lr': self.config['mapping']['lr_embed']}],
                [self.uncert_grid])


# ----------------------------------------------------------------------------------
# Callers either side of the field (A9/A10)
# ----------------------------------------------------------------------------------
def get_voxels(bbox: torch.Tensor, voxel_size: float):
    """Co-SLAM utils.py getVoxels (voxel_size branch)."""
    ts = []
    for i in range(3):
        lo, hi = float(bbox[i, 0]), float(bbox[i, 1])
        n = round((hi - lo) / voxel_size + 0.0005)
        ts.append(torch.linspace(lo, hi, n + 1))
    return ts


def get_map_volumes(query_fn, bounding_box, voxel_size):
    """coslam_utils.py:58-97 (torch tensors returned instead of numpy)."""
    tx, ty, tz = get_voxels(bounding_box, voxel_size)
    q = torch.stack(torch.meshgrid(tx, ty, tz, indexing='ij'), -1).to(torch.float32).to(bounding_box.device)
    q = (q - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])
    sdf = query_fn(q, embed=False, return_uncert=True)
    sdf, unc = sdf[..., 0], sdf[..., 1]
    um = F.softplus(unc) + 0.01
    mask = (sdf >= 0) * (sdf < 0.5)
    um = torch.where(mask, um, torch.zeros_like(um))
    return um, sdf


def smoothness(field, sample_points, voxel_size, margin, offset_rand, jitter_rand):
    """Co-SLAM coslam.py smoothness(): TV of hash features on a (sample_points-1)^3 lattice."""
    bb = field.bounding_box
    grid_size = (sample_points - 1) * voxel_size
    offset_max = bb[:, 1] - bb[:, 0] - grid_size - 2 * margin
    offset = offset_rand.to(bb) * offset_max + margin
    n = sample_points - 1
    ax = torch.arange(0, n, dtype=torch.long)
    coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).float().to(bb)
    pts = (coords + jitter_rand.to(bb).reshape(1, 1, 1, 3)) * voxel_size + bb[:, 0] + offset
    pts_tcnn = (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
    sdf = field.query_sdf(pts_tcnn, embed=True)
    tv_x = torch.pow(sdf[1:, ...] - sdf[:-1, ...], 2).sum()
    tv_y = torch.pow(sdf[:, 1:, ...] - sdf[:, :-1, ...], 2).sum()
    tv_z = torch.pow(sdf[:, :, 1:, ...] - sdf[:, :, :-1, ...], 2).sum()
    return (tv_x + tv_y + tv_z) / (sample_points ** 3)


# ----------------------------------------------------------------------------------
# N1 / N2 ("next" rows): active ray sampler (src/slam/coslam/active_ray_sampler.py:77-149) and the
# camera->world ray transform (src/slam/coslam/coslam.py:342-344), restated in numpy / torch.
# ----------------------------------------------------------------------------------
def active_ray_lookup(rays_o, rays_d, target_d, n_cur, uncert_vol, bbox, base, mul):
    """Cached-uncertainty value of every candidate ray (reference :108-122)."""
    pts = rays_o + rays_d * target_d
    n_tail = -((-n_cur) // mul)
    pts = pts[base:-n_tail]
    loc = ((pts - torch.tensor(bbox, dtype=torch.float32)[:, 0]) * 10).detach().cpu().numpy()
    idx = loc.round().astype(int)
    for a in range(3):
        idx[:, a] = np.clip(idx[:, a], 0, uncert_vol.shape[a] - 1)
    return uncert_vol[idx[:, 0], idx[:, 1], idx[:, 2]], n_tail


def active_ray_sample(rays_o, rays_d, target_s, target_d, n_cur, uncert_vol, bbox, base, K, mul, deterministic=False):
    """Reference :124-147.  deterministic=True fixes what numpy leaves open: the K smallest by (value, index), listed
    by ascending candidate index -- the order the HIP kernel produces."""
    vals, n_tail = active_ray_lookup(rays_o, rays_d, target_d, n_cur, uncert_vol, bbox, base, mul)
    if deterministic:
        order = np.lexsort((np.arange(vals.size), vals))[:K]
        sel = np.sort(order)
    else:
        sel = np.argpartition(vals, K, axis=None)[:K]
    out = []
    for t in (rays_o, rays_d, target_s, target_d):
        out.append(torch.cat([t[sel + base], t[:base - K], t[-n_tail:]]))
    return out, vals, sel


def rays_to_world(rays_d_cam, ids_all, poses_all):
    """coslam.py:342-344."""
    rays_d = torch.sum(rays_d_cam[..., None, None, :] * poses_all[ids_all, None, :3, :3], -1)
    rays_o = poses_all[ids_all, None, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
    return rays_o, rays_d.reshape(-1, 3)


# ---------------------------------------------------------------------------------------------------
# N3 ("next" row): the planner's uncertainty aggregation in goal space
# (reference src/planner/naruto_planner.py: init_data :110-137, uncertainty_aggregation_v2 :596-735)
# ---------------------------------------------------------------------------------------------------
def goal_space(bbox, voxel_size: float = 0.1, gs_z_levels=(5, 11, 17)):
    """naruto_planner.py:116-137 -> (dims (Nx,Ny,Nz), ranges (gs_x_range, gs_y_range, gs_z_range), goal_idx int64 [G,3])."""
    Nx = round((bbox[0][1] - bbox[0][0]) / voxel_size + 0.0005) + 1
    Ny = round((bbox[1][1] - bbox[1][0]) / voxel_size + 0.0005) + 1
    Nz = round((bbox[2][1] - bbox[2][0]) / voxel_size + 0.0005) + 1
    gx = torch.arange(0, Nx, 2)
    gy = torch.arange(0, Ny, 2)
    gz = torch.arange(int(1 / voxel_size), Nz, int(1 / voxel_size)) if gs_z_levels is None else torch.tensor(list(gs_z_levels))
    X, Y, Z = torch.meshgrid(gx, gy, gz, indexing="ij")
    return (Nx, Ny, Nz), (gx, gy, gz), torch.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], 1)


def topk_targets_reference(uncert: np.ndarray, top_k: int, top_k_subset: int) -> np.ndarray:
    """naruto_planner.py:629-632: np.argpartition(uncert, -top_k)[-top_k_subset:] -> voxel indices int64 [k,3].  WHICH subset
    of the top_k comes out is an artefact of numpy's introselect; only 'a subset of the top_k largest' is specified."""
    idx = np.argpartition(uncert, -top_k, axis=None)[-top_k_subset:]
    return np.column_stack(np.unravel_index(idx, uncert.shape)).astype(np.int64)


def topk_targets_deterministic(uncert: np.ndarray, top_k: int, top_k_subset: int) -> np.ndarray:
    """What the HIP path selects: the top_k largest values (ties: lower flat index first), listed in flat-index order,
    thinned to top_k_subset entries at positions floor(i * top_k / top_k_subset) -- spread over the volume, which is what
    the reference's comment asks of the subset ("avoid Uncertainty Point Concentration", configs/default.py:94)."""
    flat = uncert.reshape(-1)
    order = np.lexsort((np.arange(flat.size), -flat.astype(np.float64)))      # by value descending, then index ascending
    top = np.sort(order[:top_k])
    pick = top[(np.arange(top_k_subset, dtype=np.int64) * top_k) // top_k_subset]
    return np.column_stack(np.unravel_index(pick, uncert.shape)).astype(np.int64)


def uncert_aggregation(uncert: np.ndarray, sdf: np.ndarray, targets: np.ndarray, goal_idx: torch.Tensor, dims, voxel_size: float,
                       sensing_range=(0.5, 2.0), safe_sdf: float = 0.8):
    """naruto_planner.py:637-710 for given target voxels -> (collections [G,k] fp32, aggregated [G] fp32, valid mask [G,k]).
    A (goal, target) pair counts iff min < |goal - target| < max (voxels), the goal is not on the volume border and the sdf
    at the goal and its six neighbours is >= safe_sdf, and the sdf at 30 points of the segment goal -> target (truncated
    to voxel indices) is > 0 everywhere."""
    Nx, Ny, Nz = dims
    unc = torch.from_numpy(np.ascontiguousarray(uncert))
    sd = torch.from_numpy(np.ascontiguousarray(sdf))
    tgt = torch.from_numpy(np.asarray(targets)).float()                       # [k,3]
    gp = goal_idx.float()                                                     # [G,3]
    k = tgt.shape[0]
    goal_pts = gp[:, None, :].repeat(1, k, 1)
    view = goal_pts - tgt
    dist = torch.norm(view, dim=2)
    valid = (dist < sensing_range[1] / voxel_size) * (dist > sensing_range[0] / voxel_size)
    x, y, z = goal_idx[:, 0], goal_idx[:, 1], goal_idx[:, 2]
    unsafe = (x < 1) + (x + 1 >= Nx) + (y < 1) + (y + 1 >= Ny) + (z < 1) + (z + 1 >= Nz)
    for dx, dy, dz in ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        unsafe = unsafe + (sd[(x + dx).clamp(0, Nx - 1), (y + dy).clamp(0, Ny - 1), (z + dz).clamp(0, Nz - 1)] < safe_sdf)
    valid[unsafe.reshape(-1) > 0, :] = False
    near = view[valid]
    t = torch.linspace(0, 1, 30)
    pts = (goal_pts[valid][..., None] - t * near[..., None]).permute(0, 2, 1).long()
    vis = sd[pts[:, :, 0], pts[:, :, 1], pts[:, :, 2]].min(dim=1)[0] > 0 if near.shape[0] else torch.zeros(0, dtype=torch.bool)
    valid = valid.masked_scatter(valid.clone(), vis)
    ti = tgt.long()
    ku = unc[ti[:, 0], ti[:, 1], ti[:, 2]][None, :].repeat(gp.shape[0], 1)
    coll = torch.zeros_like(ku)
    coll[valid] = ku[valid]
    return coll, coll.sum(dim=1), valid
