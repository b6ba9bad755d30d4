#!/usr/bin/env python3
"""What the bf16 MLP mode (decoder.mlp_precision = 'bf16') costs in accuracy, measured against the exact fp32 mode of the same
library on the same inputs: raw outputs, rendered maps, the five losses, every gradient.  Run on the GPU box:

    python tools/bf16_error_study.py [> profiles/r02_bf16_error_study.txt]

Two cases: the golden rays of tests/golden/g1_render_train_t16 (64 rays x 43) and BASELINE configs[1]'s full batch
(2048 rays x 128 samples, random-init MLP + closed-form table of amplitude 0.05)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from naruto_amd import ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def run(cfg, ora, rays, rand, smooth):
    out = {}
    for mode in ("fp32", "bf16"):
        c = H.office_cfg(cfg["grid"]["hash_size"], perturb=cfg["training"]["perturb"], n_samples_d=cfg["training"]["n_samples_d"])
        c["decoder"]["mlp_precision"] = mode
        m = H.make_hip_from_oracle(c, ora, dev)
        tr, cam = c["training"], c["cam"]
        N = rays["rays_o"].shape[0]
        S = tr["n_samples_d"] + tr["n_range_d"]
        w = torch.tensor([tr["rgb_weight"], tr["depth_weight"], tr["sdf_weight"], tr["fs_weight"], 0.0, tr["uncert_weight"], 0.0, 0.0,
                          tr["smooth_weight"] if smooth else 0.0, 0.0], device=dev)
        ug = torch.zeros_like(m.uncert_grid)
        ts = ops.TrainStep(m._handle(), m._params(), ug, N, n_samples_d=tr["n_samples_d"], n_range_d=tr["n_range_d"], near=cam["near"], far=cam["far"],
                           range_d=tr["range_d"], depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"], perturb=rand is not None,
                           loss_weights=w, smooth=(tr["smooth_pts"], tr["smooth_vox"], tr["smooth_margin"]) if smooth else None, device_rng=False)
        args = [torch.from_numpy(rays[k]).to(dev).contiguous() for k in ("rays_o", "rays_d", "target_rgb")] + [torch.from_numpy(rays["target_d"]).to(dev).reshape(-1).contiguous()]
        if smooth:
            ts.rand[N * S:].copy_(torch.tensor([0.3, 0.6, 0.2, 0.1, 0.7, 0.4]))
        losses = ts.run(*args, rand=rand.to(dev) if rand is not None else None).clone()
        torch.cuda.synchronize()
        out[mode] = {"raw": ts.raw.clone(), "rgb": ts.rgb.clone(), "depth": ts.depth.clone(), "uncert": ts.uncert_map.clone(), "losses": losses,
                     "grads": {k: v.clone() for k, v in ts.grads.items()}, "ug": ug.clone()}
    return out


def report(name, o):
    a, b = o["fp32"], o["bf16"]
    print(f"== {name}")
    raw_a, raw_b = a["raw"].reshape(-1, 5), b["raw"].reshape(-1, 5)
    live = raw_a.abs().sum(1) > 0
    for ch, nm in enumerate(("r (pre-sigmoid)", "g", "b", "sdf", "uncert_raw")):
        d = (raw_a[live, ch] - raw_b[live, ch]).abs()
        print(f"  raw[{nm:16s}] max |d| {float(d.max()):.3e}  mean |d| {float(d.mean()):.3e}  scale max|x| {float(raw_a[live, ch].abs().max()):.3e}")
    for k in ("rgb", "depth", "uncert"):
        d = (a[k] - b[k]).abs()
        print(f"  rendered {k:7s} max |d| {float(d.max()):.3e}  mean |d| {float(d.mean()):.3e}  scale {float(a[k].abs().max()):.3e}")
    for i, nm in enumerate(("rgb_loss", "depth_loss", "sdf_loss", "fs_loss", "psnr", "uncert_loss")):
        la, lb = float(a["losses"][i]), float(b["losses"][i])
        print(f"  {nm:12s} fp32 {la:.6e}  bf16 {lb:.6e}  rel {abs(la - lb) / max(abs(la), 1e-30):.3e}")
    print(f"  total        fp32 {float(a['losses'][9]):.6e}  bf16 {float(b['losses'][9]):.6e}  rel {abs(float(a['losses'][9]) - float(b['losses'][9])) / abs(float(a['losses'][9])):.3e}")
    for k in list(a["grads"].keys()):
        ga, gb = a["grads"][k].double().reshape(-1), b["grads"][k].double().reshape(-1)
        if k == "uncert_grid":
            ga, gb = a["ug"].double().reshape(-1), b["ug"].double().reshape(-1)
        sc = float(ga.abs().max())
        d = (ga - gb).abs()
        cos = float((ga @ gb) / (ga.norm() * gb.norm() + 1e-300))
        print(f"  grad {k:11s} max |d| / max|g| {float(d.max()) / sc:.3e}   ||d||2 / ||g||2 {float(d.norm() / ga.norm()):.3e}   cosine {cos:.6f}")


g = H.load_golden("g1_render_train_t16")
cfg = H.office_cfg(int(g["hash_size"]), perturb=float(g["perturb"]), n_samples_d=int(g["n_samples_d"]))
ora = H.make_oracle(cfg, float(g["table_amp"]), int(g["seed"]), weights={k: g[k] for k in ("sdf_w0", "sdf_w1", "col_w0", "col_w1")})
rays = {k: g[k] for k in ("rays_o", "rays_d", "target_rgb", "target_d")}
report("golden g1_render_train_t16 (64 rays x 43 samples)", run(cfg, ora, rays, None, False))

cfg = H.office_cfg(16, perturb=1.0, n_samples_d=117)
ora = H.make_oracle(cfg, 0.05, 77)
rays = syn.random_rays(2048, cfg["mapping"]["bound"], seed=77, zero_depth_frac=0.05)
rand = torch.rand(2048, 128, generator=torch.Generator().manual_seed(11))
report("BASELINE configs[1]: 2048 rays x 128 samples, smoothness on", run(cfg, ora, rays, rand, True))
