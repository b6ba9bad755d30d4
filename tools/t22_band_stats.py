"""How much of a T = 2^22 batch does each training forward evaluate?  (samples with non-zero raw after a step; a-priori listed = z <= depth + k truncations)
    NARUTO_FWD_SORTED=0|1 python tools/t22_band_stats.py [n_rays] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from naruto_amd.trainer import MappingTrainer
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
cfg, _ = bench.workload("unit1024_T22_131072x43")
torch.manual_seed(0)
tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)
rays = {k: torch.from_numpy(v).to(dev) for k, v in bench.bench_rays(cfg, n_rays).items()}
for i in range(steps):
    tr.step(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"], smooth=True)
torch.cuda.synchronize()
ts = tr._train_step(n_rays, True)
S = ts.raw.numel() // (5 * n_rays)
raw = ts.raw.reshape(n_rays, S, 5)
z = ts.z_vals.reshape(n_rays, S)
td = rays["target_d"].reshape(-1, 1)
trunc = float(cfg["training"]["trunc"]) * float(cfg["training"].get("sc_factor", 1.0)) if "trunc" in cfg["training"] else None
print("cfg trunc / sc_factor:", cfg["training"].get("trunc"), cfg.get("data", {}).get("sc_factor"), "far", cfg["cam"]["far"], "S", S)
n_eval = int((raw.abs().sum(-1) > 0).sum())
print(f"after {steps} steps: evaluated (non-zero raw) {n_eval} of {n_rays * S} = {n_eval / (n_rays * S):.3f}; active (backward list) {int(ts.n_active.item())}")
for k in (1.0, 2.0):
    for tsc in (0.02, 0.05, 0.1):
        lst = (~(td > 0)) | ~(z > td + k * tsc)
        print(f"  a-priori listed with {k:.0f} x trunc_sc={tsc}: {int(lst.sum()) / (n_rays * S):.3f}")
