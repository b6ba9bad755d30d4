"""Shared helpers for the parity tests: identical parameters in the oracle (CPU) and the HIP module."""
import os

import numpy as np
import torch

from naruto_amd import config as C
from naruto_amd import synthetic as syn
from oracle import spec_torch as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def office_cfg(hash_size=16, perturb=0.0, n_samples_d=32, **kw):
    cfg = C.office0_config(perturb=perturb, n_samples_d=n_samples_d, **kw)
    cfg["grid"]["hash_size"] = hash_size
    return cfg


def make_oracle(cfg, table_amp, seed, weights=None, uncert_voxel=0.1):
    bbox = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32)
    ora = S.OracleField(cfg, bbox, uncert_voxel)
    w = weights if weights is not None else syn.mlp_weights(seed)
    dims = S.uncert_grid_dims(bbox, uncert_voxel)
    with torch.no_grad():
        ora.table.copy_(torch.from_numpy(syn.closed_form_table(ora.meta.n_params, table_amp)))
        ora.sdf_w0.copy_(torch.from_numpy(w["sdf_w0"]))
        ora.sdf_w1.copy_(torch.from_numpy(w["sdf_w1"]))
        ora.col_w0.copy_(torch.from_numpy(w["col_w0"]))
        ora.col_w1.copy_(torch.from_numpy(w["col_w1"]))
        ora.uncert_grid.copy_(torch.from_numpy(syn.closed_form_uncert_grid(dims)))
    return ora


def make_hip_from_oracle(cfg, ora, device, uncert_voxel=0.1):
    from naruto_amd.field import NarutoFieldHIP
    bbox = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=device)
    m = NarutoFieldHIP(cfg, bbox).to(device)
    m.get_uncert_grid(uncert_voxel)
    with torch.no_grad():
        assert m.embed_fn.params.numel() == ora.table.numel()
        m.embed_fn.params.copy_(ora.table)
        m.decoder.sdf_net.model[0].weight.copy_(ora.sdf_w0)
        m.decoder.sdf_net.model[2].weight.copy_(ora.sdf_w1)
        m.decoder.color_net.model[0].weight.copy_(ora.col_w0)
        m.decoder.color_net.model[2].weight.copy_(ora.col_w1)
        assert tuple(m.uncert_grid.shape) == tuple(ora.uncert_grid.shape)
        m.uncert_grid.copy_(ora.uncert_grid)
    return m


def hip_grads(m):
    return {"sdf_w0": m.decoder.sdf_net.model[0].weight.grad, "sdf_w1": m.decoder.sdf_net.model[2].weight.grad,
            "col_w0": m.decoder.color_net.model[0].weight.grad, "col_w1": m.decoder.color_net.model[2].weight.grad,
            "uncert_grid": m.uncert_grid.grad, "table": m.embed_fn.params.grad}


def ora_grads(o):
    return {"sdf_w0": o.sdf_w0.grad, "sdf_w1": o.sdf_w1.grad, "col_w0": o.col_w0.grad, "col_w1": o.col_w1.grad,
            "uncert_grid": o.uncert_grid.grad, "table": o.table.grad}


def assert_close(a, b, tol, what, rel=0.0):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: NaN pattern differs"
    a, b = torch.nan_to_num(a), torch.nan_to_num(b)
    err = (a - b).abs()
    bound = tol + rel * b.abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: max abs err {err.max().item():.3e} (tol {tol:g}, rel {rel:g}) at "
                           f"{int(bad.sum())}/{bad.numel()} elements; worst idx {int(err.argmax())}, "
                           f"got {a.reshape(-1)[int(err.argmax())].item():.6g} want {b.reshape(-1)[int(err.argmax())].item():.6g}")


# ---- the kernels' own uniform numbers (naruto_common.h: splitmix64 keyed by seed, iteration counter, index) ----
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def device_rng_uniform(seed: int, counter: int, idx):
    """float32 array of the values rng_uniform(rng_key({seed, counter}), i) for i in idx (python ints: exact)."""
    key = _splitmix64((seed & _M64) ^ _splitmix64(counter & _M64))
    return np.array([(_splitmix64((key + int(i)) & _M64) >> 40) * 2.0 ** -24 for i in idx], dtype=np.float32)
