"""NarutoFieldHIP: drop-in for the reference's ``JointEncodingNaruto`` (reference
src/slam/coslam/model/scene_rep.py:25-287), the ``nn.Module`` that ``CoSLAMNaruto`` instantiates at
src/slam/coslam/coslam.py:65 -- same constructor, same method names / argument meaning / return
dicts, same parameter and state_dict names, but every operator runs as hand-written HIP kernels from
libnaruto_hip.so.  There is no PyTorch fallback: tensors must live on the GPU.

Attribute surface kept: forward, render_rays, raw2outputs, sdf2weights,
query_sdf, query_color, query_color_sdf, run_network, calc_embedding, render_surface_color,
get_uncert_grid, embed_fn / embedpos_fn / decoder / color_net / sdf_net sub-modules, uncert_grid.
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops


class _HashGridEncoding(nn.Module):
    """Holds ``params`` like ``tcnn.Encoding`` does (state_dict key ``embed_fn.params``); callable."""

    def __init__(self, owner: "NarutoFieldHIP", n_params: int, n_output_dims: int):
        super().__init__()
        self.params = nn.Parameter((torch.rand(n_params) * 2 - 1) * 1e-4)      # tcnn: U(-1e-4, 1e-4)
        self.n_output_dims = n_output_dims
        object.__setattr__(self, "_owner", owner)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.hash_encode(self._owner._handle(), x.reshape(-1, 3), self.params)


class _OneBlobEncoding(nn.Module):
    """Parameter-free; ``embedpos_fn.params`` (empty) appears in the state_dict as with tcnn.  The fused query kernels evaluate the
    encoding in registers; called on its own (forward only) it runs ``naruto_oneblob_fwd``."""

    def __init__(self, owner: "NarutoFieldHIP", n_bins: int):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(0))
        self.n_bins = n_bins
        self.n_output_dims = 3 * n_bins
        object.__setattr__(self, "_owner", owner)

    def forward(self, x):
        return ops.oneblob_encode(self._owner._handle(), x)


class _Mlp(nn.Module):
    """Weight holder with the reference's module path ``.model.{0,2}.weight`` (bias-free Linear, ReLU, Linear).  The fused query
    kernels evaluate the MLPs in registers; called on its own (forward only, reference decoder.py:29-41 for the sdf net) it runs
    ``naruto_decoder_fwd``."""

    def __init__(self, owner: "NarutoFieldHIP", part: int, d_in: int, d_hidden: int, d_out: int):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(d_in, d_hidden, bias=False), nn.ReLU(inplace=True),
                                   nn.Linear(d_hidden, d_out, bias=False))
        self._part = part
        object.__setattr__(self, "_owner", owner)

    def forward(self, x, return_geo=True):
        o = self._owner
        ops._forward_only(x, *o._params().values(), what="sdf_net" if self._part == 1 else "color_net")
        out = ops.decoder_part(o._handle(), o._params(), self._part, x)
        if self._part == 1 and not return_geo:
            return out[..., :1]
        return out


class _Decoder(nn.Module):
    """``ColorSDFNet_v2_Naruto`` (reference decoder.py:81-116): holds the two nets; ``decoder(embed, embed_pos)`` on its own (forward
    only) runs ``naruto_decoder_fwd``."""

    def __init__(self, owner: "NarutoFieldHIP", in_sdf: int, in_col: int, hidden: int, hidden_col: int, geo: int):
        super().__init__()
        self.color_net = _Mlp(owner, 2, in_col, hidden_col, 3)
        self.sdf_net = _Mlp(owner, 1, in_sdf, hidden, 1 + geo)
        object.__setattr__(self, "_owner", owner)

    def forward(self, embed, embed_pos):
        o = self._owner
        ops._forward_only(embed, embed_pos, *o._params().values(), what="decoder")
        return ops.decoder_part(o._handle(), o._params(), 0, embed, embed_pos)


class NarutoFieldHIP(nn.Module):
    N_LEVELS, N_FEATURES, BASE_RESOLUTION = 16, 2, 16        # Co-SLAM get_encoder defaults

    def __init__(self, config: Dict, bound_box: torch.Tensor):
        super().__init__()
        self.config = config
        self.bounding_box = bound_box
        dec, grid = config['decoder'], config['grid']
        if not grid.get('oneGrid', True):
            raise NotImplementedError("oneGrid=False (separate colour grid) is not shipped by the reference configs")
        if dec.get('tcnn_network', False) or dec.get('pred_uncert', False) or not dec.get('uncert_grid', False):
            raise NotImplementedError("supported decoder: tcnn_network=False, pred_uncert=False, uncert_grid=True")
        if config['training'].get('n_importance', 0) > 0:
            raise NotImplementedError("n_importance > 0 cannot run in the reference either (scene_rep.py:204)")
        if 'hash' not in grid['enc'].lower() or 'blob' not in config['pos']['enc'].lower():
            raise NotImplementedError("supported encodings: HashGrid + OneBlob")
        self.get_resolution()
        self.per_level_scale = float(np.exp2(np.log2(self.resolution_sdf / self.BASE_RESOLUTION) / (self.N_LEVELS - 1)))
        self._uncert_dims = None
        self._handles: Dict = {}
        n_params = self._make_handle((1, 1, 1)).n_params
        self.input_ch = self.N_LEVELS * self.N_FEATURES
        self.input_ch_pos = 3 * config['pos']['n_bins']
        self.embedpos_fn = _OneBlobEncoding(self, config['pos']['n_bins'])
        self.embed_fn = _HashGridEncoding(self, n_params, self.input_ch)
        self.decoder = _Decoder(self, self.input_ch + self.input_ch_pos, self.input_ch_pos + dec['geo_feat_dim'],
                                dec['hidden_dim'], dec['hidden_dim_color'], dec['geo_feat_dim'])
        # the reference re-registers the two nets at top level (batchify(fn, None) returns the module itself)
        self.color_net = self.decoder.color_net
        self.sdf_net = self.decoder.sdf_net
        self.act_uncertainty = nn.Softplus()
        self.process_group = None            # set by enable_data_parallel()
        self.n_rays_total = 0
        self.strict_assert = False
        self._pending_min_uncert = None
        self._min_uncert_host, self._min_uncert_queue, self._min_uncert_slot = None, [], 0
        # training forward for an unchanged caller (coslam.py:361-399): ONE autograd node over the fused training kernels
        # (naruto_train_forward / naruto_train_backward).  False: the modular operators (field query | composite + losses), which
        # also differentiate the rendered rgb / depth.
        self.fused_train = True
        self._node_states: Dict = {}
        self.max_node_states = 3               # the BA batch size moves with the keyframe count: keep the last few sizes' buffers
        self._rng_state = None                 # int64 {seed, iteration counter} keying the kernels' own depth jitter
        self._min_uncert_run = None            # float32[1]: running minimum of uncert_map.min() over every fused forward
        self.assert_every = 8                  # fused forward: read the running minimum back every this many calls (asynchronously)
        self._n_fused_forwards = 0

    # ------------------------------------------------------------------ construction helpers
    def get_resolution(self):
        """Co-SLAM JointEncoding.get_resolution."""
        dim_max = (self.bounding_box[:, 1] - self.bounding_box[:, 0]).max()
        g = self.config['grid']
        self.resolution_sdf = g['voxel_sdf'] if g['voxel_sdf'] > 10 else int(dim_max / g['voxel_sdf'])
        self.resolution_color = g['voxel_color'] if g['voxel_color'] > 10 else int(dim_max / g['voxel_color'])

    def _make_handle(self, uncert_dims) -> ops.FieldHandle:
        key = tuple(int(v) for v in uncert_dims)
        h = self._handles.get(key)
        if h is None:
            bb = self.bounding_box.detach().float().cpu()
            tr = self.config['training']
            h = ops.FieldHandle(n_levels=self.N_LEVELS, n_features=self.N_FEATURES,
                                log2_hashmap_size=self.config['grid']['hash_size'], base_resolution=self.BASE_RESOLUTION,
                                per_level_scale=self.per_level_scale, n_bins=self.config['pos']['n_bins'],
                                hidden_dim=self.config['decoder']['hidden_dim'], geo_feat_dim=self.config['decoder']['geo_feat_dim'],
                                hidden_dim_color=self.config['decoder']['hidden_dim_color'], uncert_dims=key,
                                bbox_min=bb[:, 0].tolist(), bbox_max=bb[:, 1].tolist(), trunc=tr['trunc'],
                                sc_factor=self.config['data']['sc_factor'], white_bkgd=tr['white_bkgd'],
                                mlp_mode=self.config['decoder'].get('mlp_precision', 'fp32'))
            self._handles[key] = h
        return h

    def _handle(self) -> ops.FieldHandle:
        if getattr(self, "uncert_grid", None) is None:
            raise RuntimeError("uncert_grid is not initialised: call get_uncert_grid(voxel_size) first "
                               "(reference coslam.py:240-243)")
        return self._make_handle(tuple(self.uncert_grid.shape))

    def get_uncert_grid(self, voxel_size):
        """scene_rep.py:49-56 (on the module's device instead of the hard-coded "cuda")."""
        bb = self.bounding_box
        Nx = round((bb[0, 1] - bb[0, 0]).item() / voxel_size + 0.0005) + 1
        Ny = round((bb[1, 1] - bb[1, 0]).item() / voxel_size + 0.0005) + 1
        Nz = round((bb[2, 1] - bb[2, 0]).item() / voxel_size + 0.0005) + 1
        dev = self.embed_fn.params.device
        self.uncert_grid = torch.nn.parameter.Parameter(torch.ones([Nx, Ny, Nz], device=dev).float() * 3)
        self.cache_uncert = np.zeros([Nx, Ny, Nz], dtype=np.float32)
        return self.uncert_grid

    def enable_data_parallel(self, group, n_rays_total: int = 0):
        """Ray-sharded data parallelism (naruto_amd.parallel): loss sums are all-reduced over ``group``."""
        self.process_group = group
        self.n_rays_total = n_rays_total

    def _params(self) -> Dict[str, torch.Tensor]:
        return {"table": self.embed_fn.params, "uncert_grid": self.uncert_grid,
                "sdf_w0": self.decoder.sdf_net.model[0].weight, "sdf_w1": self.decoder.sdf_net.model[2].weight,
                "col_w0": self.decoder.color_net.model[0].weight, "col_w1": self.decoder.color_net.model[2].weight}

    # ------------------------------------------------------------------ A6/A7
    def sdf2weights(self, sdf, z_vals, args=None):
        raw = torch.zeros(*sdf.shape, 5, dtype=torch.float32, device=sdf.device)
        raw[..., 3] = sdf
        return ops.composite(self._handle(), raw, z_vals)[3]

    def raw2outputs(self, raw, z_vals, white_bkgd=False):
        """-> rgb_map, disp_map, acc_map, weights, depth_map, depth_var, uncert_map (scene_rep.py:66-96)."""
        if bool(white_bkgd) != bool(self.config['training']['white_bkgd']):
            raise NotImplementedError("white_bkgd is fixed by config['training']['white_bkgd'] in this build")
        return ops.composite(self._handle(), raw, z_vals)

    # ------------------------------------------------------------------ A9
    def query_sdf(self, query_points, return_geo=False, embed=False, return_uncert=False):
        """scene_rep.py:98-130.  query_points [..., 3], already normalised to the unit cube."""
        lead = list(query_points.shape[:-1])
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]])
        if embed:
            e = ops.hash_encode(self._handle(), flat, self.embed_fn.params)
            return torch.reshape(e, lead + [e.shape[-1]])
        out = ops.field_query(self._handle(), self._params(), x=flat, color=False, want_geo=return_geo)
        su, geo = (out if return_geo else (out, None))
        sdf = su if return_uncert else su[:, 0]
        sdf = torch.reshape(sdf, lead + ([2] if return_uncert else []))
        if not return_geo:
            return sdf
        return sdf, torch.reshape(geo, lead + [geo.shape[-1]])

    def calc_embedding(self, x):
        """scene_rep.py:58-64: embed_fn(x) with the trilinear sample of the uncertainty grid in front -> [M,33] (channel 0 = uncertainty,
        x<->z quirk of the reference's grid_sample call included).  The query functions fuse this; on its own it is two launches
        (differentiable w.r.t. the table through ``embed_fn``; the uncertainty channel comes out detached)."""
        flat = torch.reshape(x, [-1, x.shape[-1]])
        u = ops.uncert_sample(self._handle(), flat, self.uncert_grid)
        return torch.cat([u, self.embed_fn(flat)], dim=1)

    def query_color_sdf(self, query_points):
        """scene_rep.py:132-148: -> raw [M,5] = (rgb pre-sigmoid, sdf, uncert_raw)."""
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]])
        return ops.field_query(self._handle(), self._params(), x=flat, color=True)

    def query_color(self, query_points):
        """Co-SLAM JointEncoding.query_color."""
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def run_network(self, inputs):
        """Co-SLAM JointEncoding.run_network: world-space points [..., 3] -> raw [..., 5]."""
        flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        bb = self.bounding_box.to(flat)
        flat = (flat - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        out = ops.field_query(self._handle(), self._params(), x=flat, color=True)
        return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])

    # ------------------------------------------------------------------ A1 + render
    def _sample_z(self, rays_o, target_d, rand=None):
        tr, cam = self.config['training'], self.config['cam']
        n_rays = rays_o.shape[0]
        if target_d is not None:
            n_range, n_unif, n_samples = tr['n_range_d'], tr['n_samples_d'], 0
        else:
            n_range, n_unif, n_samples = 0, 0, tr['n_samples']        # KeyError with shipped configs, as in the reference
        S = n_unif + n_range if target_d is not None else n_samples
        if tr['perturb'] > 0. and rand is None:
            rand = torch.rand(n_rays, S, device=rays_o.device)
        if not tr['perturb'] > 0.:
            rand = None
        return ops.sample_z(n_rays, target_d, float(cam['near']), float(cam['far']), n_unif, n_range, float(tr['range_d']),
                            n_samples, rand, device=rays_o.device)

    def render_rays(self, rays_o, rays_d, target_d=None, rand=None, want_raw=True, fused=None):
        """scene_rep.py:150-225.  ``rand`` (optional, [N,S]) replaces the jitter draw for reproducible tests.

        Without autograd (eval renders, planner-side queries: ``torch.no_grad()``) the whole call is ONE launch
        (``naruto_render_fwd``: sampling + field query + compositing per ray, raw kept on chip); ``want_raw=False`` then also
        skips writing ``raw`` / ``z_vals`` (the returned dict has no such keys).  With autograd enabled the three
        differentiable operators are used."""
        if fused is None:
            fused = not torch.is_grad_enabled()
        if fused:
            tr, cam = self.config['training'], self.config['cam']
            n_rays = rays_o.shape[0]
            if target_d is not None:
                n_range, n_unif, n_samples = tr['n_range_d'], tr['n_samples_d'], 0
            else:
                n_range, n_unif, n_samples = 0, 0, tr['n_samples']        # KeyError with shipped configs, as in the reference
            S = n_unif + n_range if target_d is not None else n_samples
            if tr['perturb'] > 0. and rand is None:
                rand = torch.rand(n_rays, S, device=rays_o.device)
            if not tr['perturb'] > 0.:
                rand = None
            return ops.render_fused(self._handle(), self._params(), rays_o, rays_d, target_d, near=cam['near'], far=cam['far'],
                                    n_samples_d=n_unif, n_range_d=n_range, range_d=tr['range_d'], n_samples=n_samples, rand=rand,
                                    want_raw=want_raw)
        z_vals = self._sample_z(rays_o, target_d, rand)
        raw = ops.field_query(self._handle(), self._params(), rays_o=rays_o, rays_d=rays_d, z_vals=z_vals, color=True)
        raw = raw.reshape(z_vals.shape[0], z_vals.shape[1], 5)
        rgb_map, disp_map, acc_map, weights, depth_map, depth_var, uncert_map = ops.composite(self._handle(), raw, z_vals)
        return {'rgb': rgb_map, 'depth': depth_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_var': depth_var,
                'z_vals': z_vals, 'raw': raw, 'uncert_map': uncert_map}

    def render_surface_color(self, rays_o, normal):
        """Co-SLAM ``JointEncoding.render_surface_color`` [third_parties/coslam/model/scene_rep.py, not in tree; the mesh colour
        function when ``mesh.render_color`` is set, reference coslam.py:446-447]: colour of surface points rendered along their
        normals -- ``n_range_d`` samples at ``linspace(-trunc, trunc)`` around each point, through ``run_network`` and
        ``raw2outputs``.  rays_o [N,3] world points, normal [N,3] -> rgb [N,3].  Without autograd: one launch."""
        tr = self.config['training']
        trunc, S = float(tr['trunc']), int(tr['n_range_d'])
        rays_o = rays_o.to(self.bounding_box.device, torch.float32)
        normal = normal.to(self.bounding_box.device, torch.float32)
        if not torch.is_grad_enabled():
            return ops.render_fused(self._handle(), self._params(), rays_o, normal, None, near=-trunc, far=trunc, n_samples_d=0, n_range_d=0,
                                    range_d=0.0, n_samples=S, rand=None, want_raw=False)['rgb']
        z_vals = torch.linspace(-trunc, trunc, steps=S).to(rays_o).repeat(rays_o.shape[0], 1)
        raw = ops.field_query(self._handle(), self._params(), rays_o=rays_o, rays_d=normal, z_vals=z_vals, color=True)
        raw = raw.reshape(z_vals.shape[0], z_vals.shape[1], 5)
        return ops.composite(self._handle(), raw, z_vals)[0]

    # ------------------------------------------------------------------ A8
    def note_min_uncert(self, value: torch.Tensor):
        """Queue this iteration's ``uncert_map.min()`` (a device scalar) for the deferred check: an asynchronous copy into
        pinned host memory and an event; nothing waits.  (Skipped while a hipGraph is being captured.)"""
        if not value.is_cuda or torch.cuda.is_current_stream_capturing():
            self._pending_min_uncert = value
            return
        if self._min_uncert_host is None:
            self._min_uncert_host = torch.empty(16, dtype=torch.float32, pin_memory=True)
        if len(self._min_uncert_queue) >= 16:
            self.check_asserts(block=True)
        slot = self._min_uncert_slot
        self._min_uncert_slot = (slot + 1) % 16
        self._min_uncert_host[slot:slot + 1].copy_(value.detach().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._min_uncert_queue.append((slot, ev))

    def min_uncert_running(self) -> torch.Tensor:
        """float32[1] device word that every fused training forward (and every replay of a captured iteration) folds its
        ``uncert_map.min()`` into: the reference's per-forward assertion (scene_rep.py:280) covers every iteration, read back late."""
        if self._min_uncert_run is None:
            self._min_uncert_run = torch.full((1,), float("inf"), dtype=torch.float32, device=self.embed_fn.params.device)
        return self._min_uncert_run

    def _node_state(self, n_rays: int, explicit_rand: bool) -> "ops.TrainNodeState":
        tr, cam = self.config['training'], self.config['cam']
        h = self._handle()
        key = (int(n_rays), bool(explicit_rand), id(h), tr['n_samples_d'], tr['n_range_d'], bool(tr['perturb'] > 0.))
        st = self._node_states.pop(key, None)
        if st is None:
            while len(self._node_states) >= self.max_node_states:
                self._node_states.pop(next(iter(self._node_states)))
            dev = self.embed_fn.params.device
            if self._rng_state is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # follows torch.manual_seed
                self._rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
            st = ops.TrainNodeState(h, self._params(), None, n_rays, n_samples_d=tr['n_samples_d'], n_range_d=tr['n_range_d'],
                                    near=cam['near'], far=cam['far'], range_d=tr['range_d'], depth_trunc=cam['depth_trunc'],
                                    rgb_missing=tr['rgb_missing'], perturb=tr['perturb'] > 0.,
                                    loss_weights=torch.zeros(10, dtype=torch.float32, device=dev), smooth=None,
                                    device_rng=not explicit_rand, rng_state=self._rng_state, min_uncert_running=self.min_uncert_running())
        self._node_states[key] = st                     # most recently used last
        return st

    def check_asserts(self, block: bool = False):
        """The reference asserts ``uncert_map.min() > 0`` inside forward (scene_rep.py:280), which costs a device sync
        per iteration -- with eager launches the host then never runs ahead of the GPU.  Here the value is produced on
        the device, copied to the host asynchronously and checked when its copy has landed (a few iterations late);
        ``block=True`` (or ``strict_assert = True``: the reference's in-line behaviour) waits for everything queued."""
        if self._pending_min_uncert is not None:
            v = float(self._pending_min_uncert.item())
            self._pending_min_uncert = None
            assert v > 0, "uncert_map.min() > 0 violated (scene_rep.py:280)"
        while self._min_uncert_queue and (block or self._min_uncert_queue[0][1].query()):
            slot, ev = self._min_uncert_queue.pop(0)
            ev.synchronize()
            v = float(self._min_uncert_host[slot])
            assert v > 0, "uncert_map.min() > 0 violated (scene_rep.py:280)"

    def forward(self, rays_o, rays_d, target_rgb, target_d, global_step=0, rand=None, _check=True, _smooth=None):
        """scene_rep.py:227-287."""
        if not self.training:
            return self.render_rays(rays_o, rays_d, target_d=target_d, rand=rand)
        # inside a hipGraph capture (MappingTrainer.capture, graphed.GraphedIteration) nothing may touch the host: the running minimum is
        # still folded in by the kernels, and whoever replays the graph reads it back (note_min_uncert / check_asserts) outside
        capturing = rays_o.is_cuda and torch.cuda.is_current_stream_capturing()
        if _check and not capturing:
            self.check_asserts()
        cfg = self.config
        if self.fused_train and _smooth is None and self.process_group is None and rays_o.is_cuda:
            # the unchanged caller's route: sampling + field query + loss stage + tail as the fused training launches, the backward
            # as naruto_train_backward with the caller's loss weights read from the cotangents of the scalar losses
            if not cfg['training']['perturb'] > 0.:
                rand = None
            st = self._node_state(rays_o.shape[0], rand is not None)
            rgb, depth, l0, l1, l2, l3, psnr, l5, losses = ops.train_forward_node(st, self._params(), rays_o, rays_d, target_rgb, target_d, rand)
            self._n_fused_forwards += 1
            if not capturing:
                if self.strict_assert or self._n_fused_forwards % self.assert_every == 0:
                    self.note_min_uncert(self._min_uncert_run)
                if self.strict_assert:
                    self.check_asserts(block=True)
            return {"rgb": rgb, "depth": depth, "rgb_loss": l0, "depth_loss": l1, "sdf_loss": l2, "fs_loss": l3, "psnr": psnr.detach(),
                    "uncert_loss": l5, "_losses": losses, "_smooth_loss": losses[8]}
        z_vals = self._sample_z(rays_o, target_d, rand)
        rgb, depth, _disp, _acc, _var, _um, _raw, losses = ops.render_train(
            self._handle(), self._params(), rays_o, rays_d, z_vals, target_rgb, target_d, cfg['cam']['depth_trunc'],
            cfg['training']['rgb_missing'], group=self.process_group, n_rays_total=self.n_rays_total, smooth=_smooth)
        self.note_min_uncert(losses[6])
        if self.strict_assert:
            self.check_asserts(block=True)
        return {"rgb": rgb, "depth": depth, "rgb_loss": losses[0], "depth_loss": losses[1], "sdf_loss": losses[2],
                "fs_loss": losses[3], "psnr": losses[4].detach(), "uncert_loss": losses[5], "_losses": losses, "_smooth_loss": losses[8]}


_LATTICE_CACHE: Dict = {}


def _map_lattice(bounding_box: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """Normalised query lattice of get_map_volumes (Co-SLAM getVoxels + meshgrid), cached on the device."""
    key = (tuple(float(v) for v in bounding_box.detach().cpu().reshape(-1)), float(voxel_size), str(bounding_box.device))
    q = _LATTICE_CACHE.get(key)
    if q is None:
        ts = []
        for i in range(3):
            lo, hi = float(bounding_box[i, 0]), float(bounding_box[i, 1])
            n = round((hi - lo) / voxel_size + 0.0005)                        # Co-SLAM getVoxels
            ts.append(torch.linspace(lo, hi, n + 1))
        q = torch.stack(torch.meshgrid(*ts, indexing='ij'), -1).to(torch.float32).to(bounding_box.device)
        q = ((q - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])).contiguous()
        _LATTICE_CACHE[key] = q
    return q


def get_map_volumes(query_fn, bounding_box: torch.Tensor, voxel_size: float):
    """Planner query path (reference src/slam/coslam/coslam_utils.py:58-97): dense lattice -> [uncert_vol,
    sdf_vol] as numpy.  The lattice is cached on the device, the reference's discarded ``embed=True`` pass
    (coslam_utils.py:86-87) is skipped, the post-processing is one kernel and both volumes come back in one
    device-to-host copy."""
    import ctypes as C
    from . import _lib
    q = _map_lattice(bounding_box, voxel_size)
    with torch.no_grad():
        su = query_fn(q, embed=False, return_uncert=True).contiguous()
        M = su.numel() // 2
        out = torch.empty(2, M, dtype=torch.float32, device=su.device)
        with torch.cuda.device(su.device):
            _lib.check(_lib.load().naruto_map_volumes(M, su.data_ptr(), out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "naruto_map_volumes")
        # one device-to-host copy into a cached PINNED buffer (a pageable destination costs a staging copy and ~0.1 ms more)
        key = ("pinned", 2 * M)
        pin = _LATTICE_CACHE.get(key)
        if pin is None:
            pin = torch.empty(2, M, dtype=torch.float32, pin_memory=True)
            _LATTICE_CACHE[key] = pin
        pin.copy_(out, non_blocking=True)
        torch.cuda.current_stream(su.device).synchronize()
        host = pin.numpy()
    shape = tuple(q.shape[:-1])
    return [host[0].reshape(shape).copy(), host[1].reshape(shape).copy()]
