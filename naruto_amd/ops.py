"""torch-facing operators over the C ABI: tensor checks, stream plumbing and autograd wiring.

Every operator launches hand-written HIP kernels from libnaruto_hip.so on torch's CURRENT stream; torch
is used for memory, streams and autograd bookkeeping only.  Mapping to the reference (paths under
/root/reference): see include/naruto_hip.h and DESIGN.md.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import NarutoFieldDesc, NarutoGrads, NarutoParams, NarutoPoints, check

PARAM_NAMES = ("table", "uncert_grid", "sdf_w0", "sdf_w1", "col_w0", "col_w1")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> C.c_void_p:
    """torch's CURRENT stream on the current device as a hipStream_t."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _on_device:
    """``with torch.cuda.device(dev)`` without its cost when ``dev`` already is the current device (every call of a single-GPU process)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.idx is not None and self.idx != self.prev:
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.idx is not None and self.idx != self.prev:
            torch.cuda.set_device(self.prev)
        return False


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (the hot path has no CPU implementation), got {t.device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class FieldHandle:
    """Owns one NarutoField* (immutable description of the scene representation)."""

    def __init__(self, *, n_levels=16, n_features=2, log2_hashmap_size=16, base_resolution=16, per_level_scale,
                 n_bins=16, hidden_dim=32, geo_feat_dim=15, hidden_dim_color=32, uncert_dims, bbox_min, bbox_max,
                 trunc, sc_factor, white_bkgd=False, mlp_mode: str = "fp32"):
        lib = _lib.load()
        d = NarutoFieldDesc()
        d.n_levels, d.n_features, d.log2_hashmap_size = n_levels, n_features, log2_hashmap_size
        d.base_resolution, d.per_level_scale, d.n_bins = base_resolution, float(per_level_scale), n_bins
        d.hidden_dim, d.geo_feat_dim, d.hidden_dim_color = hidden_dim, geo_feat_dim, hidden_dim_color
        for i in range(3):
            d.uncert_dims[i] = int(uncert_dims[i])
            d.bbox_min[i] = float(bbox_min[i])
            d.bbox_max[i] = float(bbox_max[i])
        d.trunc, d.sc_factor, d.white_bkgd = float(trunc), float(sc_factor), int(bool(white_bkgd))
        if mlp_mode not in ("fp32", "bf16"):
            raise ValueError(f"mlp_mode must be 'fp32' (exact, the parity mode) or 'bf16' (speed mode), got {mlp_mode!r}")
        d.mlp_mode = _lib.MLP_BF16 if mlp_mode == "bf16" else _lib.MLP_FP32
        self.mlp_mode = mlp_mode
        self.desc = d
        self._h = C.c_void_p()
        check(lib.naruto_field_create(C.byref(d), C.byref(self._h)), "naruto_field_create")
        self.n_levels = n_levels
        self.n_entries = int(lib.naruto_field_n_entries(self._h))
        self.n_params = self.n_entries * n_features
        self.uncert_dims = tuple(int(v) for v in uncert_dims)

    @property
    def ptr(self) -> C.c_void_p:
        return self._h

    def levels(self):
        lib = _lib.load()
        L = self.n_levels
        scale = (C.c_float * L)()
        res = (C.c_uint32 * L)()
        size = (C.c_uint32 * L)()
        off = (C.c_uint32 * (L + 1))()
        check(lib.naruto_field_levels(self._h, scale, res, size, off), "naruto_field_levels")
        return list(scale), list(res), list(size), list(off)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.load().naruto_field_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def _params_struct(params: Dict[str, torch.Tensor]) -> NarutoParams:
    s = NarutoParams()
    for n in PARAM_NAMES:
        setattr(s, n, _p(params.get(n)))
    return s


def _points_struct(x, rays_o, rays_d, z_vals) -> Tuple[NarutoPoints, int]:
    s = NarutoPoints()
    if x is not None:
        s.x = _p(x)
        return s, x.shape[0]
    s.rays_o, s.rays_d, s.z_vals = _p(rays_o), _p(rays_d), _p(z_vals)
    s.n_samples = z_vals.shape[1]
    return s, z_vals.shape[0] * z_vals.shape[1]


# ---------------------------------------------------------------------------------------------------
# A1
# ---------------------------------------------------------------------------------------------------
def sample_z(n_rays: int, target_d: Optional[torch.Tensor], near: float, far: float, n_samples_d: int, n_range_d: int,
             range_d: float, n_samples: int = 0, rand: Optional[torch.Tensor] = None, device=None) -> torch.Tensor:
    lib = _lib.load()
    if target_d is not None:
        target_d = _f32c(target_d, "target_d").reshape(-1)
        device = target_d.device
        S = n_samples_d + n_range_d
    else:
        S = n_samples
    z = torch.empty(n_rays, S, dtype=torch.float32, device=device)
    if rand is not None:
        rand = _f32c(rand, "rand")
        assert rand.shape == (n_rays, S)
    with _on_device(z.device):
        check(lib.naruto_sample_z(n_rays, _p(target_d), near, far, n_samples_d, n_range_d, range_d, n_samples, _p(rand),
                                  _p(z), _stream()), "naruto_sample_z")
    return z


def render_fused(handle: "FieldHandle", params: Dict[str, torch.Tensor], rays_o, rays_d, target_d, *, near: float, far: float, n_samples_d: int,
                 n_range_d: int, range_d: float, n_samples: int = 0, rand: Optional[torch.Tensor] = None, want_raw: bool = True,
                 want_weights: bool = False) -> Dict[str, torch.Tensor]:
    """render_rays as ONE launch (naruto_render_fwd): depth sampling + field query + compositing per ray.  Inference only (no
    autograd).  ``want_raw``: also write raw [N,S,5] and z_vals [N,S] (the reference's render dict carries them)."""
    lib = _lib.load()
    rays_o, rays_d = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d")
    dev = rays_o.device
    N = rays_o.shape[0]
    if target_d is not None:
        target_d = _f32c(target_d, "target_d").reshape(-1)
        S = n_samples_d + n_range_d
    else:
        S = n_samples
    if rand is not None:
        rand = _f32c(rand, "rand")
        assert rand.shape == (N, S)
    f32 = dict(dtype=torch.float32, device=dev)
    out = {"rgb": torch.empty(N, 3, **f32)}
    for k in ("depth", "disp_map", "acc_map", "depth_var", "uncert_map"):
        out[k] = torch.empty(N, **f32)
    if want_raw:
        out["raw"], out["z_vals"] = torch.empty(N, S, 5, **f32), torch.empty(N, S, **f32)
    if want_weights:
        out["weights"] = torch.empty(N, S, **f32)
    r = _lib.NarutoRender()
    r.n_rays, r.rays_o, r.rays_d, r.target_d = N, _p(rays_o), _p(rays_d), _p(target_d)
    r.near_, r.far_, r.n_samples_d, r.n_range_d, r.range_d, r.n_samples = float(near), float(far), int(n_samples_d), int(n_range_d), float(range_d), int(n_samples)
    r.rand, r.rng = _p(rand), None
    r.rgb, r.depth, r.disp, r.acc, r.depth_var, r.uncert_map = (_p(out[k]) for k in ("rgb", "depth", "disp_map", "acc_map", "depth_var", "uncert_map"))
    r.weights, r.raw, r.z_vals = _p(out.get("weights")), _p(out.get("raw")), _p(out.get("z_vals"))
    ps = _params_struct({k: _f32c(v.detach(), k) for k, v in params.items()})
    with _on_device(dev):
        check(lib.naruto_render_fwd(handle.ptr, C.byref(ps), C.byref(r), _stream()), "naruto_render_fwd")
    return out


# ---------------------------------------------------------------------------------------------------
# The model's sub-modules called on their own (forward only): embedpos_fn(x), decoder(embed, embed_pos), sdf_net(x), color_net(x)
# ---------------------------------------------------------------------------------------------------
def _forward_only(*tensors, what: str):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(f"{what} on its own is forward only (call it under torch.no_grad()); gradients flow through "
                                  "forward / render_rays / query_sdf / query_color_sdf, which evaluate it inside the fused kernels")


def oneblob_encode(handle: "FieldHandle", x: torch.Tensor) -> torch.Tensor:
    """embedpos_fn(x): tcnn OneBlob, 16 bins per coordinate.  x [..., 3] (normalised) -> [M, 48]."""
    _forward_only(x, what="embedpos_fn")
    lib = _lib.load()
    x = _f32c(x.detach().reshape(-1, 3), "x")
    out = torch.empty(x.shape[0], 48, dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        check(lib.naruto_oneblob_fwd(handle.ptr, x.shape[0], _p(x), _p(out), _stream()), "naruto_oneblob_fwd")
    return out


def uncert_sample(handle: "FieldHandle", x: torch.Tensor, uncert_grid: torch.Tensor) -> torch.Tensor:
    """calc_embedding's channel 0 (scene_rep.py:58-64): trilinear sample of the uncertainty grid at normalised points x [M,3] -> [M,1]
    (forward only; gradients to the grid flow through the fused query operators)."""
    lib = _lib.load()
    x = _f32c(x.detach().reshape(-1, 3), "x")
    g = _f32c(uncert_grid.detach(), "uncert_grid")
    out = torch.empty(x.shape[0], 1, dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        check(lib.naruto_uncert_sample(handle.ptr, x.shape[0], _p(x), _p(g), _p(out), _stream()), "naruto_uncert_sample")
    return out


def decoder_part(handle: "FieldHandle", params: Dict[str, torch.Tensor], part: int, a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """part 0: decoder(embed [M,33], embed_pos [M,48]) -> [M,5]; 1: sdf_net([M,81]) -> [M,17]; 2: color_net([M,63]) -> [M,3]."""
    lib = _lib.load()
    width_a, width_out = {0: (33, 5), 1: (81, 17), 2: (63, 3)}[part]
    a = _f32c(a.detach().reshape(-1, a.shape[-1]), "input")
    if a.shape[1] != width_a:
        raise ValueError(f"expected {width_a} input channels, got {a.shape[1]}")
    if b is not None:
        b = _f32c(b.detach().reshape(-1, b.shape[-1]), "embed_pos")
        if b.shape != (a.shape[0], 48):
            raise ValueError(f"embed_pos: expected [{a.shape[0]}, 48], got {list(b.shape)}")
    out = torch.empty(a.shape[0], width_out, dtype=torch.float32, device=a.device)
    ps = _params_struct({k: _f32c(v.detach(), k) for k, v in params.items()})
    with _on_device(a.device):
        check(lib.naruto_decoder_fwd(handle.ptr, C.byref(ps), a.shape[0], part, _p(a), _p(b), _p(out), _stream()), "naruto_decoder_fwd")
    return out


# ---------------------------------------------------------------------------------------------------
# A3 alone: embed_fn(x)
# ---------------------------------------------------------------------------------------------------
class _HashEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle: FieldHandle, x: torch.Tensor, table: torch.Tensor):
        lib = _lib.load()
        x = _f32c(x, "x")
        table = _f32c(table, "table")
        M = x.shape[0]
        feat = torch.empty(M, 32, dtype=torch.float32, device=x.device)
        with _on_device(x.device):
            check(lib.naruto_hash_encode_fwd(handle.ptr, M, _p(x), _p(table), _p(feat), _stream()), "naruto_hash_encode_fwd")
        ctx.handle = handle
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, table)
        return feat

    @staticmethod
    def backward(ctx, d_feat):
        lib = _lib.load()
        x, table = ctx.saved_tensors
        d_table = None
        if ctx.needs_input_grad[2] and d_feat is not None:
            d_feat = _f32c(d_feat, "d_feat")
            d_table = torch.zeros_like(table)
            with _on_device(x.device):
                ws = torch.empty((lib.naruto_scatter_workspace(ctx.handle.ptr, x.shape[0]) + 3) // 4, dtype=torch.float32, device=x.device)
                check(lib.naruto_hash_encode_bwd(ctx.handle.ptr, x.shape[0], _p(x), _p(d_feat), None, _p(d_table), _p(ws), _stream()),
                      "naruto_hash_encode_bwd")
        return None, None, d_table


def hash_encode(handle: FieldHandle, x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    return _HashEncode.apply(handle, x, table)


class _Smoothness(torch.autograd.Function):
    """Co-SLAM's feature-grid smoothness term as three small kernels forward + the table scatter backward."""

    @staticmethod
    def forward(ctx, handle, table, sample_points, voxel_size, margin, rand6):
        lib = _lib.load()
        table = _f32c(table, "table")
        rand6 = _f32c(rand6, "rand6").reshape(-1)
        assert rand6.numel() == 6
        n = sample_points - 1
        dev = table.device
        x = torch.empty(n ** 3, 3, dtype=torch.float32, device=dev)
        d_feat = torch.empty(n ** 3, 32, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with _on_device(dev):
            ws = torch.empty((lib.naruto_smoothness_workspace(sample_points) + 3) // 4, dtype=torch.float32, device=dev)
            check(lib.naruto_smoothness_fwd(handle.ptr, _p(table), sample_points, voxel_size, margin, _p(rand6), _p(x), _p(d_feat), _p(loss),
                                            _p(ws), _stream()), "naruto_smoothness_fwd")
        ctx.handle = handle
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, d_feat, table)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, d_feat, table = ctx.saved_tensors
        if g is None or not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        g = _f32c(g, "grad").reshape(1)
        d_table = torch.zeros_like(table)
        with _on_device(x.device):
            ws = torch.empty((lib.naruto_scatter_workspace(ctx.handle.ptr, x.shape[0]) + 3) // 4, dtype=torch.float32, device=x.device)
            check(lib.naruto_hash_encode_bwd(ctx.handle.ptr, x.shape[0], _p(x), _p(d_feat), _p(g), _p(d_table), _p(ws), _stream()),
                  "naruto_hash_encode_bwd")
        return None, d_table, None, None, None, None


def smoothness(handle: FieldHandle, table: torch.Tensor, sample_points: int, voxel_size: float, margin: float, rand6: torch.Tensor):
    return _Smoothness.apply(handle, table, int(sample_points), float(voxel_size), float(margin), rand6)


# ---------------------------------------------------------------------------------------------------
# A2-A5 fused
# ---------------------------------------------------------------------------------------------------
class _FieldQuery(torch.autograd.Function):
    """raw (or sdf_uncert) [, geo] = field(points).  Gradients flow to the six parameter tensors only
    (the reference never differentiates w.r.t. the query points: tracking is disabled in every config)."""

    @staticmethod
    def forward(ctx, handle, color, want_geo, x, rays_o, rays_d, z_vals, table, uncert_grid, sdf_w0, sdf_w1, col_w0, col_w1):
        lib = _lib.load()
        params = {"table": table, "uncert_grid": uncert_grid, "sdf_w0": sdf_w0, "sdf_w1": sdf_w1, "col_w0": col_w0,
                  "col_w1": col_w1}
        params = {k: _f32c(v, k) for k, v in params.items()}
        if x is not None:
            x = _f32c(x, "x")
            dev = x.device
        else:
            rays_o, rays_d, z_vals = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d"), _f32c(z_vals, "z_vals")
            dev = z_vals.device
        pts, M = _points_struct(x, rays_o, rays_d, z_vals)
        need_grad = any(ctx.needs_input_grad[7:])
        out = torch.empty(M, 5 if color else 2, dtype=torch.float32, device=dev)
        geo = torch.empty(M, 15, dtype=torch.float32, device=dev) if want_geo else None
        feat = torch.empty(16, M, 2, dtype=torch.float32, device=dev) if need_grad else None
        ps = _params_struct(params)
        with _on_device(dev):
            check(lib.naruto_query_fwd(handle.ptr, C.byref(ps), M, C.byref(pts), _p(out) if color else None,
                                       None if color else _p(out), _p(geo), _p(feat), _stream()), "naruto_query_fwd")
        ctx.handle, ctx.color, ctx.want_geo, ctx.M = handle, color, want_geo, M
        ctx.set_materialize_grads(False)
        ctx.has_x = x is not None
        if need_grad:
            ctx.save_for_backward(feat, *(t for t in (x, rays_o, rays_d, z_vals) if t is not None),
                                  *(params[k] for k in PARAM_NAMES))
        if want_geo:
            return out, geo
        return out

    @staticmethod
    def backward(ctx, d_out, d_geo=None):
        lib = _lib.load()
        saved = ctx.saved_tensors
        feat = saved[0]
        if ctx.has_x:
            x, rays_o, rays_d, z_vals = saved[1], None, None, None
            rest = saved[2:]
        else:
            x = None
            rays_o, rays_d, z_vals = saved[1:4]
            rest = saved[4:]
        params = dict(zip(PARAM_NAMES, rest))
        M = ctx.M
        dev = feat.device
        d_out = _f32c(d_out, "d_out") if d_out is not None else None
        if ctx.color:
            d_raw = d_out if d_out is not None else torch.zeros(M, 5, dtype=torch.float32, device=dev)
        else:
            d_raw = torch.zeros(M, 5, dtype=torch.float32, device=dev)
            if d_out is not None:
                d_raw[:, 3:5] = d_out
        if d_geo is not None:
            d_geo = _f32c(d_geo, "d_geo")
        grads = {}
        for i, n in enumerate(PARAM_NAMES):
            grads[n] = torch.zeros_like(params[n]) if ctx.needs_input_grad[7 + i] else None
        gs = NarutoGrads()
        for n in PARAM_NAMES:
            setattr(gs, n, _p(grads[n]))
        ps = _params_struct(params)
        pts, _ = _points_struct(x, rays_o, rays_d, z_vals)
        with _on_device(dev):
            ws_bytes = lib.naruto_query_bwd_workspace(ctx.handle.ptr, M)
            ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
            check(lib.naruto_query_bwd(ctx.handle.ptr, C.byref(ps), M, C.byref(pts), _p(feat), _p(d_raw), _p(d_geo), None, None,
                                       None, 0, C.byref(gs), _p(ws), _stream()), "naruto_query_bwd")
        return (None, None, None, None, None, None, None) + tuple(grads[n] for n in PARAM_NAMES)


def field_query(handle: FieldHandle, params: Dict[str, torch.Tensor], *, x=None, rays_o=None, rays_d=None, z_vals=None,
                color: bool = True, want_geo: bool = False):
    return _FieldQuery.apply(handle, color, want_geo, x, rays_o, rays_d, z_vals, *(params[n] for n in PARAM_NAMES))


# ---------------------------------------------------------------------------------------------------
# A6 + A7
# ---------------------------------------------------------------------------------------------------
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, raw, z_vals):
        lib = _lib.load()
        raw, z_vals = _f32c(raw, "raw"), _f32c(z_vals, "z_vals")
        N, S = z_vals.shape
        dev = raw.device
        rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        disp, acc, depth, depth_var, um = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(5))
        weights = torch.empty(N, S, dtype=torch.float32, device=dev)
        with _on_device(dev):
            check(lib.naruto_composite_fwd(handle.ptr, N, S, _p(raw), _p(z_vals), _p(rgb), _p(disp), _p(acc), _p(weights),
                                           _p(depth), _p(depth_var), _p(um), _stream()), "naruto_composite_fwd")
        ctx.handle = handle
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(raw, z_vals)
        return rgb, disp, acc, weights, depth, depth_var, um

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_acc, d_weights, d_depth, d_depth_var, d_um):
        lib = _lib.load()
        raw, z_vals = ctx.saved_tensors
        N, S = z_vals.shape
        cots = [None if g is None else _f32c(g, "cotangent") for g in (d_rgb, d_disp, d_acc, d_weights, d_depth, d_depth_var, d_um)]
        d_raw = torch.empty_like(raw)
        with _on_device(raw.device):
            check(lib.naruto_composite_bwd(ctx.handle.ptr, N, S, _p(raw), _p(z_vals), *(_p(c) for c in cots), _p(d_raw), 0,
                                           _stream()), "naruto_composite_bwd")
        return None, d_raw, None


def composite(handle: FieldHandle, raw: torch.Tensor, z_vals: torch.Tensor):
    """-> rgb_map, disp_map, acc_map, weights, depth_map, depth_var, uncert_map (raw2outputs order)."""
    return _Composite.apply(handle, raw, z_vals)


# ---------------------------------------------------------------------------------------------------
# A8: composite + losses in one autograd node
# ---------------------------------------------------------------------------------------------------
class _RenderLoss(torch.autograd.Function):
    """(raw, z_vals, targets) -> render outputs + losses[8].

    losses = [rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, uncert_loss, min(uncert_map), n_valid_depth].
    ``group``: optional torch.distributed process group; the loss sums are all-reduced over it so every
    rank normalises by the GLOBAL ray / sample counts (data-parallel ray sharding);
    ``n_rays_total``: rays over all ranks (0 = this rank's count x world size)."""

    @staticmethod
    def forward(ctx, handle, raw, z_vals, target_rgb, target_d, depth_trunc, rgb_missing, group, n_rays_total):
        lib = _lib.load()
        raw, z_vals = _f32c(raw, "raw"), _f32c(z_vals, "z_vals")
        target_rgb, target_d = _f32c(target_rgb, "target_rgb"), _f32c(target_d, "target_d").reshape(-1)
        N, S = z_vals.shape
        dev = raw.device
        rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        disp, acc, depth, depth_var, um = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(5))
        sums = torch.empty(_lib.LOSS_NSUMS, dtype=torch.float64, device=dev)
        losses = torch.empty(8, dtype=torch.float32, device=dev)
        n_total = N
        with _on_device(dev):
            st = _stream()
            check(lib.naruto_composite_fwd(handle.ptr, N, S, _p(raw), _p(z_vals), _p(rgb), _p(disp), _p(acc), None, _p(depth),
                                           _p(depth_var), _p(um), st), "naruto_composite_fwd")
            ws = torch.empty(lib.naruto_loss_workspace(N) // 4, dtype=torch.float32, device=dev)
            check(lib.naruto_loss_sums(handle.ptr, N, S, _p(raw), _p(z_vals), _p(rgb), _p(depth), _p(um), _p(target_rgb),
                                       _p(target_d), depth_trunc, rgb_missing, _p(sums), None if group is not None else _p(losses),
                                       _p(ws), st), "naruto_loss_sums")
            if group is not None:
                from . import parallel
                parallel.allreduce_loss_sums(sums, group)
                n_total = int(n_rays_total) if n_rays_total else N * parallel.world_size(group)
                check(lib.naruto_loss_finalize(_p(sums), n_total, S, _p(losses), st), "naruto_loss_finalize")
        ctx.handle, ctx.depth_trunc, ctx.rgb_missing, ctx.n_total = handle, depth_trunc, rgb_missing, n_total
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(raw, z_vals, target_rgb, target_d, sums)
        ctx.mark_non_differentiable(disp, acc, depth_var, um)
        return rgb, depth, disp, acc, depth_var, um, losses

    @staticmethod
    def backward(ctx, d_rgb, d_depth, _d_disp, _d_acc, _d_var, _d_um, d_losses):
        lib = _lib.load()
        raw, z_vals, target_rgb, target_d, sums = ctx.saved_tensors
        N, S = z_vals.shape
        dev = raw.device
        d_raw = torch.empty_like(raw)
        if d_losses is None:
            d_losses = torch.zeros(8, dtype=torch.float32, device=dev)
        d_losses = _f32c(d_losses, "d_losses")
        with _on_device(dev):
            st = _stream()
            check(lib.naruto_loss_bwd(ctx.handle.ptr, N, S, _p(raw), _p(z_vals), _p(target_rgb), _p(target_d), ctx.depth_trunc,
                                      ctx.rgb_missing, _p(sums), ctx.n_total, _p(d_losses), _p(d_raw), None, st), "naruto_loss_bwd")
            if d_rgb is not None or d_depth is not None:       # someone differentiated the rendered rgb / depth too
                d_rgb = None if d_rgb is None else _f32c(d_rgb, "d_rgb")
                d_depth = None if d_depth is None else _f32c(d_depth, "d_depth")
                check(lib.naruto_composite_bwd(ctx.handle.ptr, N, S, _p(raw), _p(z_vals), _p(d_rgb), None, None, None, _p(d_depth),
                                               None, None, _p(d_raw), 1, st), "naruto_composite_bwd")
        return None, d_raw, None, None, None, None, None, None, None


def render_loss(handle: FieldHandle, raw, z_vals, target_rgb, target_d, depth_trunc: float, rgb_missing: float, group=None,
                n_rays_total: int = 0):
    return _RenderLoss.apply(handle, raw, z_vals, target_rgb, target_d, float(depth_trunc), float(rgb_missing), group,
                             int(n_rays_total))


# ---------------------------------------------------------------------------------------------------
# A1..A8 as ONE autograd node: what JointEncodingNaruto.forward does in training mode
# ---------------------------------------------------------------------------------------------------
class _RenderTrain(torch.autograd.Function):
    """(rays, z_vals, targets, parameters) -> rgb, depth, losses[10] (+ non-differentiable render outputs).
    losses = [rgb, depth, sdf, fs, psnr, uncert, min(uncert_map), n_valid_depth, smoothness term (0 if off), 0].

    Same kernels as field_query + render_loss, but because the whole chain lives in one node the backward can
    use the structure of the mapping losses: every sample behind the surface band of its ray has an all-zero
    cotangent, so only the active prefix of each ray (typically 35-55 % of the samples) goes through the MLP
    backward and the table scatter (naruto_compact_active)."""

    @staticmethod
    def forward(ctx, handle, rays_o, rays_d, z_vals, target_rgb, target_d, depth_trunc, rgb_missing, group, n_rays_total, smooth,
                table, uncert_grid, sdf_w0, sdf_w1, col_w0, col_w1):
        lib = _lib.load()
        params = {"table": table, "uncert_grid": uncert_grid, "sdf_w0": sdf_w0, "sdf_w1": sdf_w1, "col_w0": col_w0, "col_w1": col_w1}
        params = {k: _f32c(v, k) for k, v in params.items()}
        rays_o, rays_d, z_vals = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d"), _f32c(z_vals, "z_vals")
        target_rgb, target_d = _f32c(target_rgb, "target_rgb"), _f32c(target_d, "target_d").reshape(-1)
        N, S = z_vals.shape
        M = N * S
        dev = z_vals.device
        need_grad = any(ctx.needs_input_grad[11:])
        raw = torch.empty(N, S, 5, dtype=torch.float32, device=dev)
        losses = torch.zeros(10, dtype=torch.float32, device=dev)
        sm_loss = sm_x = sm_d = None
        if smooth is not None:                      # (sample_points, voxel_size, margin, rand6): Co-SLAM smoothness term
            sp, vox, mar, rand6 = smooth
            n3 = (sp - 1) ** 3
            sm_x = torch.empty(n3, 3, dtype=torch.float32, device=dev)
            sm_d = torch.empty(n3, 32, dtype=torch.float32, device=dev)
            sm_loss = losses[8:10]                      # the term lands in slot 8 of the loss vector (slot 9: spare)
        feat = torch.empty(16, M, 2, dtype=torch.float32, device=dev) if need_grad else None
        rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        disp, acc, depth, depth_var, um = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(5))
        sums = torch.empty(_lib.LOSS_NSUMS, dtype=torch.float64, device=dev)
        ps = _params_struct(params)
        pts, _ = _points_struct(None, rays_o, rays_d, z_vals)
        n_total = N
        with _on_device(dev):
            st = _stream()
            check(lib.naruto_query_fwd(handle.ptr, C.byref(ps), M, C.byref(pts), _p(raw), None, None, _p(feat), st), "naruto_query_fwd")
            check(lib.naruto_composite_fwd(handle.ptr, N, S, _p(raw), _p(z_vals), _p(rgb), _p(disp), _p(acc), None, _p(depth),
                                           _p(depth_var), _p(um), st), "naruto_composite_fwd")
            ws = torch.empty(lib.naruto_loss_workspace(N) // 4, dtype=torch.float32, device=dev)
            check(lib.naruto_loss_sums(handle.ptr, N, S, _p(raw), _p(z_vals), _p(rgb), _p(depth), _p(um), _p(target_rgb),
                                       _p(target_d), depth_trunc, rgb_missing, _p(sums), None if group is not None else _p(losses),
                                       _p(ws), st), "naruto_loss_sums")
            if group is not None:
                from . import parallel
                parallel.allreduce_loss_sums(sums, group)
                n_total = int(n_rays_total) if n_rays_total else N * parallel.world_size(group)
                check(lib.naruto_loss_finalize(_p(sums), n_total, S, _p(losses), st), "naruto_loss_finalize")
            if smooth is not None:
                ws2 = torch.empty((lib.naruto_smoothness_workspace(sp) + 3) // 4, dtype=torch.float32, device=dev)
                check(lib.naruto_smoothness_fwd(handle.ptr, _p(params["table"]), sp, vox, mar, _p(_f32c(rand6, "rand6")), _p(sm_x), _p(sm_d),
                                                sm_loss.data_ptr(), _p(ws2), st), "naruto_smoothness_fwd")
        ctx.handle, ctx.depth_trunc, ctx.rgb_missing, ctx.n_total = handle, depth_trunc, rgb_missing, n_total
        ctx.has_smooth = smooth is not None
        ctx.set_materialize_grads(False)
        if need_grad:
            extra = (sm_x, sm_d) if smooth is not None else ()
            ctx.save_for_backward(raw, feat, rays_o, rays_d, z_vals, target_rgb, target_d, sums, *(params[k] for k in PARAM_NAMES), *extra)
        ctx.mark_non_differentiable(disp, acc, depth_var, um, raw)
        return rgb, depth, disp, acc, depth_var, um, raw, losses

    @staticmethod
    def backward(ctx, d_rgb, d_depth, _d_disp, _d_acc, _d_var, _d_um, _d_raw, d_losses):
        lib = _lib.load()
        raw, feat, rays_o, rays_d, z_vals, target_rgb, target_d, sums = ctx.saved_tensors[:8]
        params = dict(zip(PARAM_NAMES, ctx.saved_tensors[8:14]))
        sm_x, sm_d = (ctx.saved_tensors[14], ctx.saved_tensors[15]) if ctx.has_smooth else (None, None)
        N, S = z_vals.shape
        M = N * S
        dev = raw.device
        if d_losses is None:
            d_losses = torch.zeros(10, dtype=torch.float32, device=dev)
        d_losses = _f32c(d_losses, "d_losses")
        d_smooth = d_losses[8:9]
        d_raw = torch.empty_like(raw)
        extra = d_rgb is not None or d_depth is not None          # someone differentiated the rendered rgb / depth as well
        # weight / table gradients are WRITTEN by the reductions (no zero fill); the uncertainty grid is scattered into
        overwrite_table = handle_supports_overwrite(ctx.handle)
        grads = {}
        # table + MLP weight gradients are carved out of ONE flat buffer, so that data-parallel ranks can all-reduce
        # them with a single collective and no staging copies (naruto_amd.parallel.allreduce_grads)
        flat_names = [n for i, n in enumerate(PARAM_NAMES) if ctx.needs_input_grad[11 + i] and n != "uncert_grid"]
        flat = torch.empty(sum(params[n].numel() for n in flat_names), dtype=torch.float32, device=dev) if flat_names else None
        off = 0
        for n in flat_names:
            k = params[n].numel()
            grads[n] = flat[off:off + k].view_as(params[n])
            off += k
        if "table" in flat_names and not overwrite_table:
            grads["table"].zero_()
        for i, n in enumerate(PARAM_NAMES):
            if not ctx.needs_input_grad[11 + i]:
                grads[n] = None
            elif n == "uncert_grid":
                grads[n] = torch.zeros_like(params[n])
        flags = _lib.BWD_OVERWRITE_WEIGHT_GRADS | (_lib.BWD_OVERWRITE_TABLE_GRAD if overwrite_table else 0)
        gs = NarutoGrads()
        for n in PARAM_NAMES:
            setattr(gs, n, _p(grads[n]))
        ps = _params_struct(params)
        pts, _ = _points_struct(None, rays_o, rays_d, z_vals)
        with _on_device(dev):
            st = _stream()
            count = None if extra else torch.empty(N, dtype=torch.int32, device=dev)
            check(lib.naruto_loss_bwd(ctx.handle.ptr, N, S, _p(raw), _p(z_vals), _p(target_rgb), _p(target_d), ctx.depth_trunc,
                                      ctx.rgb_missing, _p(sums), ctx.n_total, _p(d_losses), _p(d_raw), _p(count), st), "naruto_loss_bwd")
            active = n_active = None
            if extra:
                d_rgb = None if d_rgb is None else _f32c(d_rgb, "d_rgb")
                d_depth = None if d_depth is None else _f32c(d_depth, "d_depth")
                check(lib.naruto_composite_bwd(ctx.handle.ptr, N, S, _p(raw), _p(z_vals), _p(d_rgb), None, None, None, _p(d_depth),
                                               None, None, _p(d_raw), 1, st), "naruto_composite_bwd")
            else:
                off = torch.empty(N, dtype=torch.int32, device=dev)
                active = torch.empty(M, dtype=torch.int32, device=dev)
                n_active = torch.empty(1, dtype=torch.int32, device=dev)
                check(lib.naruto_compact_active(N, S, _p(count), _p(off), _p(active), _p(n_active), st), "naruto_compact_active")
            ex = None
            n_extra = 0
            if ctx.has_smooth and grads["table"] is not None:
                ex = _lib.NarutoExtraPoints()
                ex.x, ex.d_feat, ex.scale, ex.n = _p(sm_x), _p(sm_d), d_smooth.data_ptr(), sm_x.shape[0]
                n_extra = sm_x.shape[0]
            ws = torch.empty((lib.naruto_query_bwd_workspace(ctx.handle.ptr, M + n_extra) + 3) // 4, dtype=torch.float32, device=dev)
            check(lib.naruto_query_bwd(ctx.handle.ptr, C.byref(ps), M, C.byref(pts), _p(feat), _p(d_raw), None, _p(active), _p(n_active),
                                       None if ex is None else C.byref(ex), flags, C.byref(gs), _p(ws), st), "naruto_query_bwd")
        return (None,) * 11 + tuple(grads[n] for n in PARAM_NAMES)


def handle_supports_overwrite(handle: FieldHandle) -> bool:
    """Table gradients can be written (not accumulated) and the optimiser fused into the backward for every table size; only
    the debug switch NARUTO_DEBUG_SCATTER_ATOMIC (large levels through global float atomics) turns that off."""
    return bool(_lib.load().naruto_field_scatter_overwrites(handle.ptr))


def render_train(handle: FieldHandle, params: Dict[str, torch.Tensor], rays_o, rays_d, z_vals, target_rgb, target_d,
                 depth_trunc: float, rgb_missing: float, group=None, n_rays_total: int = 0, smooth=None):
    """-> rgb, depth, disp, acc, depth_var, uncert_map, raw, losses[10].

    ``smooth`` = (sample_points, voxel_size, margin, rand6) adds Co-SLAM's feature-grid smoothness term as
    losses[8]; its table gradient is produced by the same scatter pass as the rendering losses'."""
    if smooth is not None:
        smooth = (int(smooth[0]), float(smooth[1]), float(smooth[2]), smooth[3])
    return _RenderTrain.apply(handle, rays_o, rays_d, z_vals, target_rgb, target_d, float(depth_trunc), float(rgb_missing), group,
                              int(n_rays_total), smooth, *(params[n] for n in PARAM_NAMES))


def adam_multi_(entries, *, betas, step: int = 0, step_dev: Optional[torch.Tensor] = None, advance: bool = False,
                zero_grad: bool = False) -> None:
    """entries: list of (param, grad, exp_avg, exp_avg_sq, lr, eps, weight_decay); one launch for all of them.
    ``advance``: step_dev is int32[2] = {completed steps, 0}; the launch is step step_dev[0]+1 and stores it back.
    ``zero_grad``: the gradients are zeroed once consumed."""
    lib = _lib.load()
    flags = (_lib.ADAM_ADVANCE if advance else 0) | (_lib.ADAM_ZERO_GRAD if zero_grad else 0)
    if advance:
        assert step_dev is not None and step_dev.numel() >= 2 and step_dev.dtype == torch.int32
    segs = (_lib.NarutoAdamSeg * len(entries))()
    for k, (p, g, m, v, lr, eps, wd) in enumerate(entries):
        for t in (p, g, m, v):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        segs[k].param, segs[k].grad, segs[k].exp_avg, segs[k].exp_avg_sq = _p(p), _p(g), _p(m), _p(v)
        segs[k].n, segs[k].lr, segs[k].eps, segs[k].weight_decay = p.numel(), lr, eps, wd
    with torch.cuda.device(entries[0][0].device):
        check(lib.naruto_adam_multi(segs, len(entries), betas[0], betas[1], step, _p(step_dev), flags, _stream()), "naruto_adam_multi")


def adam_multi_segs(segs, n: int, dev, *, betas, step: int = 0, step_dev: Optional[torch.Tensor] = None, advance: bool = False,
                    zero_grad: bool = False) -> None:
    """adam_multi_ over a prebuilt NarutoAdamSeg array (FusedAdam keeps one and refreshes the gradient pointers per step)."""
    lib = _lib.load()
    flags = (_lib.ADAM_ADVANCE if advance else 0) | (_lib.ADAM_ZERO_GRAD if zero_grad else 0)
    with _on_device(dev):
        check(lib.naruto_adam_multi(segs, n, betas[0], betas[1], step, _p(step_dev), flags, _stream()), "naruto_adam_multi")


class TrainStep:
    """The mapping iteration's forward + backward as two C calls on persistent buffers (naruto_train_forward /
    naruto_train_backward): no autograd graph, no per-iteration allocations or fills, a dozen launches.

    ``params``: the six parameter tensors (PARAM_NAMES).  Gradients: table + MLP weights are WRITTEN into views of
    ``self.flat_grad`` (one buffer, so data-parallel ranks reduce it with one collective); the uncertainty grid's
    gradient is ACCUMULATED into ``uncert_grad`` (it is stepped every 5th iteration, coslam.py:397-399)."""

    FLAT_NAMES = ("table", "sdf_w0", "sdf_w1", "col_w0", "col_w1")

    def __init__(self, handle: FieldHandle, params: Dict[str, torch.Tensor], uncert_grad: torch.Tensor, n_rays: int, *, n_samples_d: int,
                 n_range_d: int, near: float, far: float, range_d: float, depth_trunc: float, rgb_missing: float, perturb: bool,
                 loss_weights: torch.Tensor, smooth: Optional[Tuple[int, float, float]] = None, group=None, n_rays_total: int = 0,
                 device_rng: bool = True, seed: Optional[int] = None, rng_state: Optional[torch.Tensor] = None,
                 min_uncert_running: Optional[torch.Tensor] = None, own_grads: bool = True, table_grad_pad: int = 0):
        lib = _lib.load()
        self.handle, self.group = handle, group
        self.fuse_tail = os.environ.get("NARUTO_DEBUG_NO_FUSED_TAIL") is None     # run(): loss tail + compaction inside the backward's first launch
        self.params = {k: _f32c(params[k].detach(), k) for k in PARAM_NAMES}
        for k in PARAM_NAMES:
            assert self.params[k].data_ptr() == params[k].data_ptr(), f"{k}: parameters must be contiguous fp32 on the GPU"
        dev = self.params["table"].device
        self.device = dev
        N, S = int(n_rays), int(n_samples_d) + int(n_range_d)
        M = N * S
        self.N, self.S, self.perturb = N, S, bool(perturb)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.loss_weights = _f32c(loss_weights, "loss_weights")
        assert self.loss_weights.numel() == 10
        # device_rng: the kernels draw the depth jitter / lattice placement themselves from (seed, iteration counter) --
        # no RNG launch, and no generator state for hipGraph replay to refresh.  Otherwise: one torch RNG launch per
        # iteration into self.rand, or the caller's own numbers (run(rand=...)).
        self.device_rng = bool(device_rng)
        if rng_state is not None:          # int64 {seed, iteration counter} shared with the caller (the counter advances once per run)
            assert rng_state.is_cuda and rng_state.dtype == torch.int64 and rng_state.numel() == 2
            self.rng_state = rng_state
        else:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # follows torch.manual_seed
            self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
        self.rand = torch.empty(M + 6, **f32)                  # [N*S] depth jitter | 6 numbers placing the smoothness lattice
        self.z_vals = torch.empty(N, S, **f32)
        self.raw = torch.empty(N, S, 5, **f32)
        self.feat = torch.empty(16, M, 2, **f32)
        self.rgb, self.depth, self.uncert_map = torch.empty(N, 3, **f32), torch.empty(N, **f32), torch.empty(N, **f32)
        self.sums = torch.zeros(_lib.LOSS_NSUMS, dtype=torch.float64, device=dev)
        self.losses = torch.zeros(10, **f32)
        self.d_raw = torch.empty(N, S, 5, **f32)
        self.ray_count, self.ray_offset = torch.empty(N, **i32), torch.empty(N, **i32)
        self.active_idx, self.n_active = torch.empty(M, **i32), torch.zeros(1, **i32)
        self.grads = {n: None for n in PARAM_NAMES}
        if own_grads:
            # table_grad_pad: the table's bucket padded to this many floats (zeros behind the gradient) so that it splits evenly over the
            # ranks of a sharded optimiser (reduce-scatter; MappingTrainer(shard_table_optimizer=True))
            n_table = self.params["table"].numel()
            n_bucket = max(n_table, int(table_grad_pad))
            self.flat_grad = torch.zeros(n_bucket + sum(self.params[n].numel() for n in self.FLAT_NAMES[1:]), **f32)
            self.grads["table"] = self.flat_grad[:n_table].view_as(self.params["table"])
            off = n_bucket
            for n in self.FLAT_NAMES[1:]:
                k = self.params[n].numel()
                self.grads[n] = self.flat_grad[off:off + k].view_as(self.params[n])
                off += k
            self.grad_bucket_table, self.grad_bucket_mlp = self.flat_grad[:n_bucket], self.flat_grad[n_bucket:]      # the two collective buckets
            assert uncert_grad.is_cuda and uncert_grad.dtype == torch.float32 and uncert_grad.is_contiguous()
            self.grads["uncert_grid"] = uncert_grad
        world = 1
        if group is not None:
            from . import parallel
            world = parallel.world_size(group)
        t = _lib.NarutoTrainStep()
        self.t = t
        t.n_rays, t.n_samples_d, t.n_range_d = N, int(n_samples_d), int(n_range_d)
        t.near_, t.far_, t.range_d, t.depth_trunc, t.rgb_missing = float(near), float(far), float(range_d), float(depth_trunc), float(rgb_missing)
        if smooth is not None:
            t.smooth_points, t.smooth_voxel, t.smooth_margin = int(smooth[0]), float(smooth[1]), float(smooth[2])
            t.smooth_grad_scale = 1.0 / world            # every rank adds the same lattice's gradient; the sum over ranks is one term
        t.n_rays_total = int(n_rays_total) if n_rays_total else N * world
        self._set_rng_mode(self.device_rng)
        t.loss_weights = _p(self.loss_weights)
        t.z_vals, t.raw, t.feat_save = _p(self.z_vals), _p(self.raw), _p(self.feat)
        t.rgb, t.depth, t.uncert_map = _p(self.rgb), _p(self.depth), _p(self.uncert_map)
        t.sums, t.losses, t.d_raw = _p(self.sums), _p(self.losses), _p(self.d_raw)
        t.ray_count, t.ray_offset, t.active_idx, t.n_active = _p(self.ray_count), _p(self.ray_offset), _p(self.active_idx), _p(self.n_active)
        self.ws = torch.empty((lib.naruto_train_workspace(handle.ptr, C.byref(t)) + 3) // 4, **f32)
        t.workspace = _p(self.ws)
        if min_uncert_running is not None:      # float32[1] device word, +inf initially: every iteration's min(uncert_map) is folded into it
            assert min_uncert_running.is_cuda and min_uncert_running.dtype == torch.float32 and min_uncert_running.numel() == 1
            self.min_uncert_running = min_uncert_running
            t.min_uncert_running = _p(min_uncert_running)
        self.t = t
        self.ps = _params_struct(self.params)
        self.gs = NarutoGrads()
        for n in PARAM_NAMES:
            setattr(self.gs, n, _p(self.grads[n]))
        self.flags = _lib.BWD_OVERWRITE_WEIGHT_GRADS | (_lib.BWD_OVERWRITE_TABLE_GRAD if handle_supports_overwrite(handle) else 0)
        self._zero_table = not handle_supports_overwrite(handle)
        self.opt = None                      # NarutoFusedAdam: set by fuse_adam()
        self._gs_nograd = None

    def fuse_adam(self, entries: Dict[str, tuple], betas, step_dev: torch.Tensor, write_grads: bool = False):
        """Optimiser in the backward (single process): ``entries[name] = (exp_avg, exp_avg_sq, lr, eps, weight_decay)`` for the
        five FLAT_NAMES tensors; ``step_dev``: int32 device word holding the current step's 1-based number when the backward
        runs.  The launch that finishes the gradients then applies the Adam step in place; with ``write_grads=False`` the
        table / weight gradients are not materialised at all."""
        assert self.group is None, "the fused optimiser is single-process: data-parallel ranks must all-reduce gradients first"
        assert handle_supports_overwrite(self.handle), "the fused optimiser is not available with NARUTO_DEBUG_SCATTER_ATOMIC"
        o = _lib.NarutoFusedAdam()
        self._opt_keep = []
        for k, name in enumerate(self.FLAT_NAMES):
            m, v, lr, eps, wd = entries[name]
            for t_ in (m, v):
                assert t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == self.params[name].numel()
            o.param[k], o.exp_avg[k], o.exp_avg_sq[k] = self.params[name].data_ptr(), m.data_ptr(), v.data_ptr()
            o.lr[k], o.eps[k], o.weight_decay[k] = float(lr), float(eps), float(wd)
            self._opt_keep += [m, v]
        o.beta1, o.beta2 = float(betas[0]), float(betas[1])
        assert step_dev.dtype == torch.int32 and step_dev.is_cuda
        o.step_dev = step_dev.data_ptr()
        self._opt_keep.append(step_dev)
        self.opt = o
        gs = NarutoGrads()
        gs.uncert_grid = _p(self.grads["uncert_grid"])
        if write_grads:
            for n in self.FLAT_NAMES:
                setattr(gs, n, _p(self.grads[n]))
        self._gs_nograd = gs

    def _set_rng_mode(self, device_rng: bool):
        t, M = self.t, self.N * self.S
        t.perturb = 1 if self.perturb else 0
        if device_rng:
            t.rand, t.rand6, t.rng = None, None, self.rng_state.data_ptr()
        else:
            t.rand = self.rand.data_ptr() if self.perturb else None
            t.rand6 = self.rand[M:].data_ptr() if t.smooth_points else None
            t.rng = None

    def run(self, rays_o, rays_d, target_rgb, target_d, rand: Optional[torch.Tensor] = None):
        """One forward + backward.  Afterwards: self.losses[10], self.rgb / depth / uncert_map, gradients in self.grads.
        ``rand`` ([N,S], only with device_rng=False): the depth jitter to use instead of a fresh draw; the six lattice
        numbers in self.rand[N*S:] are then left as they are."""
        if self.group is None and self.fuse_tail:
            # back to back: the loss tail and the compaction ride in the backward's first launch (losses valid after the backward)
            self.run_forward(rays_o, rays_d, target_rgb, target_d, rand, _defer_tail=True)
            self.run_backward(_deferred_tail=True)
            return self.losses
        self.run_forward(rays_o, rays_d, target_rgb, target_d, rand)
        if self.group is not None:
            from . import parallel
            parallel.allreduce_loss_sums(self.sums, self.group)
        self.run_backward()
        return self.losses

    # The two halves of run(), split where data-parallel ranks exchange the loss sums: run_forward stops at this rank's
    # sums (a process group is set) or at the final losses (single process); run_backward finishes the losses from the
    # (all-reduced) sums if needed and runs the backward.  MappingTrainer captures them as separate hipGraph segments so
    # that the collectives stay ordinary eager RCCL calls between graph launches.
    def run_forward(self, rays_o, rays_d, target_rgb, target_d, rand: Optional[torch.Tensor] = None, _defer_tail: bool = False):
        lib = _lib.load()
        t = self.t
        for a, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (target_rgb, "target_rgb"), (target_d, "target_d")):
            if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
                raise RuntimeError(f"{n}: expected a contiguous fp32 tensor on the GPU")
        assert rays_o.shape[0] == self.N and target_d.numel() == self.N, "TrainStep was built for another ray count"
        t.rays_o, t.rays_d, t.target_rgb, t.target_d = _p(rays_o), _p(rays_d), _p(target_rgb), _p(target_d)
        if rand is not None:
            assert not self.device_rng, "an explicit jitter draw needs TrainStep(device_rng=False)"
            self.rand[:self.N * self.S].copy_(rand.reshape(-1))
        elif not self.device_rng and (self.perturb or t.smooth_points):
            self.rand.uniform_()                                # one RNG launch: jitter + lattice placement
        with _on_device(self.device):
            st = _stream()
            if self._zero_table:
                self.grads["table"].zero_()
            # data parallel: stop at this rank's sums; with the fused first launch of the backward the smoothness term is left to it
            # (five launches + the collectives instead of six: no k_sample_encode)
            fin = ((_lib.TRAIN_FWD_SUMS_TV_LATER if self.fuse_tail else 0) if self.group is not None
                   else (_lib.TRAIN_FWD_DEFER_TAIL if _defer_tail else 1))
            check(lib.naruto_train_forward(self.handle.ptr, C.byref(self.ps), C.byref(t), fin, st), "naruto_train_forward")

    def run_backward(self, phase: int = 0, _deferred_tail: bool = False):
        """phase 0: the whole backward.  Data parallel, for overlap: phase 1 = everything up to the MLP weight gradients (then
        complete in self.grads), phase 2 = the table scatter; the caller all-reduces the weight bucket in between."""
        lib = _lib.load()
        t = self.t
        with _on_device(self.device):
            st = _stream()
            # data parallel: self.sums holds the all-reduced sums; finalize + composite backward + compaction are one launch
            given = (_lib.TRAIN_BWD_SUMS_GIVEN | _lib.TRAIN_BWD_TV_MOVED) if (self.group is not None and self.fuse_tail) else 0
            if phase != 0:
                assert self.opt is None, "the fused optimiser runs the backward in one piece"
                if phase == 1 and self.group is not None and not given:
                    check(lib.naruto_train_finalize(self.handle.ptr, C.byref(t), st), "naruto_train_finalize")
                fl = self.flags | (_lib.TRAIN_BWD_MLP_ONLY | given if phase == 1 else _lib.TRAIN_BWD_TABLE_ONLY)
                check(lib.naruto_train_backward(self.handle.ptr, C.byref(self.ps), C.byref(t), C.byref(self.gs), fl, None, st), "naruto_train_backward")
                return
            if self.group is not None and not given:
                check(lib.naruto_train_finalize(self.handle.ptr, C.byref(t), st), "naruto_train_finalize")
            fl = self.flags | (_lib.TRAIN_BWD_DEFERRED_TAIL if _deferred_tail else 0) | given
            if self.opt is not None:
                check(lib.naruto_train_backward(self.handle.ptr, C.byref(self.ps), C.byref(t), C.byref(self._gs_nograd), fl, C.byref(self.opt), st),
                      "naruto_train_backward")
            else:
                check(lib.naruto_train_backward(self.handle.ptr, C.byref(self.ps), C.byref(t), C.byref(self.gs), fl, None, st),
                      "naruto_train_backward")



# ---------------------------------------------------------------------------------------------------
# JointEncodingNaruto.forward in training mode for an UNCHANGED caller (reference coslam.py:361-399: model.forward ->
# get_loss_from_ret -> loss.backward(retain_graph=True) -> Adam): one autograd node over naruto_train_forward / naruto_train_backward
# ---------------------------------------------------------------------------------------------------
class TrainNodeState(TrainStep):
    """Persistent buffers of the fused training kernels for one ray count (z_vals, raw, saved features, cotangents, lists, workspace);
    outputs and gradients are fresh tensors per call, so what the caller keeps (ret['rgb'], .grad) is never overwritten.  A second
    backward over the SAME graph is fine (coslam.py:368 keeps it); a backward over a graph whose forward is no longer the latest one
    for this state raises (its buffers have been reused)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, own_grads=False, **kw)
        self.version = 0
        dev = self.device
        self.w_unit = torch.zeros(10, dtype=torch.float32, device=dev)        # forward: the total in losses[9] is not meaningful yet
        self.zero = torch.zeros((), dtype=torch.float32, device=dev)
        self.losses_bwd = torch.zeros(10, dtype=torch.float32, device=dev)    # the backward's first launch rewrites losses[0..9]: keep the caller's copy out of it
        self.t.loss_weights = _p(self.w_unit)


class _TrainForward(torch.autograd.Function):
    """(rays, targets, parameters) -> rgb [N,3], depth [N], rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, uncert_loss, losses[10].
    The scalar losses are separate outputs so that the caller's weighted sum (get_loss_from_ret, coslam.py:154-174) hands their
    cotangents back as scalars: stacked, they ARE the loss-weight vector naruto_train_backward takes from device memory."""

    @staticmethod
    def forward(ctx, st: "TrainNodeState", rand, rays_o, rays_d, target_rgb, target_d, table, uncert_grid, sdf_w0, sdf_w1, col_w0, col_w1):
        lib = _lib.load()
        params = {"table": table, "uncert_grid": uncert_grid, "sdf_w0": sdf_w0, "sdf_w1": sdf_w1, "col_w0": col_w0, "col_w1": col_w1}
        params = {k: _f32c(v, k) for k, v in params.items()}
        rays_o, rays_d = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d")
        target_rgb, target_d = _f32c(target_rgb, "target_rgb"), _f32c(target_d, "target_d").reshape(-1)
        N = st.N
        assert rays_o.shape[0] == N and target_d.numel() == N
        dev = rays_o.device
        rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        losses = torch.empty(10, dtype=torch.float32, device=dev)
        t = st.t
        t.rays_o, t.rays_d, t.target_rgb, t.target_d = _p(rays_o), _p(rays_d), _p(target_rgb), _p(target_d)
        t.rgb, t.depth, t.losses, t.loss_weights = _p(rgb), _p(depth), _p(losses), _p(st.w_unit)
        if rand is not None:
            assert not st.device_rng, "an explicit jitter draw needs a state built with device_rng=False"
            st.rand[:N * st.S].copy_(_f32c(rand, "rand").reshape(-1))
        ps = _params_struct(params)
        with _on_device(dev):
            check(lib.naruto_train_forward(st.handle.ptr, C.byref(ps), C.byref(t), 1, _stream()), "naruto_train_forward")
        st.version += 1
        ctx.st, ctx.version = st, st.version
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(rays_o, rays_d, target_rgb, target_d, *(params[k] for k in PARAM_NAMES))
        return rgb, depth, losses[0], losses[1], losses[2], losses[3], losses[4], losses[5], losses

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d0, d1, d2, d3, _d4, d5, d_vec):
        lib = _lib.load()
        st = ctx.st
        if st.version != ctx.version:
            raise RuntimeError("backward over a training graph whose buffers a later forward of the same ray count has reused; call backward "
                               "before the next model.forward (the reference does, coslam.py:364-368), or set model.fused_train = False")
        if d_rgb is not None or d_depth is not None:
            raise NotImplementedError("the fused training node differentiates the losses only; to differentiate the rendered rgb / depth "
                                      "as well set model.fused_train = False (the modular operators)")
        rays_o, rays_d, target_rgb, target_d = ctx.saved_tensors[:4]
        params = dict(zip(PARAM_NAMES, ctx.saved_tensors[4:10]))
        dev = rays_o.device
        # the caller's loss weights: cotangents of the scalar losses (device scalars, gathered by the backward's own first launch) and /
        # or of the loss vector
        parts = (d0, d1, d2, d3, None, d5)
        if d_vec is None and all(p is None for p in parts):
            return (None,) * 12
        w = _f32c(d_vec, "d_losses") if d_vec is not None else None
        keep = []
        t = st.t
        for i in range(10):
            pi = parts[i] if i < 6 else None
            if pi is not None:
                pi = _f32c(pi, "loss cotangent")
                keep.append(pi)
            t.loss_weight_parts[i] = _p(pi)
        need = ctx.needs_input_grad[6:12]
        grads = {}
        flat_names = [n for i, n in enumerate(PARAM_NAMES) if need[i] and n != "uncert_grid"]
        flat = torch.empty(sum(params[n].numel() for n in flat_names), dtype=torch.float32, device=dev) if flat_names else None
        off = 0
        for n in flat_names:
            k = params[n].numel()
            grads[n] = flat[off:off + k].view_as(params[n])
            off += k
        overwrite = handle_supports_overwrite(st.handle)
        if "table" in flat_names and not overwrite:
            grads["table"].zero_()
        for i, n in enumerate(PARAM_NAMES):
            if not need[i]:
                grads[n] = None
            elif n == "uncert_grid":
                grads[n] = torch.zeros_like(params[n])       # accumulated into by the scatter's grid units
        gs = NarutoGrads()
        for n in PARAM_NAMES:
            setattr(gs, n, _p(grads[n]))
        t.rays_o, t.rays_d, t.target_rgb, t.target_d = _p(rays_o), _p(rays_d), _p(target_rgb), _p(target_d)
        t.loss_weights, t.losses = _p(w), _p(st.losses_bwd)
        ps = _params_struct(params)
        flags = _lib.BWD_OVERWRITE_WEIGHT_GRADS | (_lib.BWD_OVERWRITE_TABLE_GRAD if overwrite else 0) | _lib.TRAIN_BWD_SUMS_GIVEN
        with _on_device(dev):
            check(lib.naruto_train_backward(st.handle.ptr, C.byref(ps), C.byref(t), C.byref(gs), flags, None, _stream()), "naruto_train_backward")
        return (None,) * 6 + tuple(grads[n] for n in PARAM_NAMES)


def train_forward_node(st: "TrainNodeState", params: Dict[str, torch.Tensor], rays_o, rays_d, target_rgb, target_d, rand=None):
    return _TrainForward.apply(st, rand, rays_o, rays_d, target_rgb, target_d, *(params[n] for n in PARAM_NAMES))

# ---------------------------------------------------------------------------------------------------
# A10 helper
# ---------------------------------------------------------------------------------------------------
def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, step: int = 0,
               step_dev: Optional[torch.Tensor] = None) -> None:
    """In-place fused Adam.  ``step`` (host int, 1-based) or ``step_dev`` (int32 device tensor, graph-safe)."""
    lib = _lib.load()
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    if step_dev is not None:
        assert step_dev.dtype == torch.int32 and step_dev.is_cuda
    with _on_device(param.device):
        check(lib.naruto_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1], eps,
                                   weight_decay, step, _p(step_dev), _stream()), "naruto_adam_step")
