"""Per-phase timeline of the depth-ordered walk (k_query_fwd_loss, the headline's training forward): lane 0 of every ray WAVE stamps the 100 MHz
global counter; printed: when the phases end relative to the launch's first stamp, split by how many tiles the wave's ray evaluated.
    python tools/walk_timeline.py [workload] [mlp] [training steps before the stamped one: 600 = the state bench.py's timed region sees]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naruto_amd import _lib
from naruto_amd.trainer import MappingTrainer

wl = sys.argv[1] if len(sys.argv) > 1 else "office0_2048x128"
mlp = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
cfg, n_rays = bench.workload(wl)
cfg["decoder"]["mlp_precision"] = mlp
torch.manual_seed(0)
tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)
rays = {k: torch.from_numpy(v).to(dev) for k, v in bench.bench_rays(cfg, n_rays).items()}
step = lambda: tr.step(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"], smooth=True)
n_warm = int(sys.argv[3]) if len(sys.argv) > 3 else 600
for _ in range(n_warm):
    step()
n_wg = 4096
buf = torch.zeros((n_wg * 4 + 4 * n_wg * 4) * 8, dtype=torch.int64, device=dev)       # rows of 8: the walk's waves, then (from row 16384) the gather launch's
lib = _lib.load()
lib.naruto_debug_fwd_timeline(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.naruto_debug_fwd_timeline(None)
allrows = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = allrows[:n_wg * 4]
gw = allrows[n_wg * 4:]
gw = gw[gw[:, 6] > 0]
if len(gw):
    g0 = gw[:, 0].min()
    print(f"gather launch (k_gather_walk): {len(gw)} waves with a ray; microseconds after ITS first stamp")
    grp = gw[:, 6].astype(int) - 1
    for g in sorted(set(grp)):
        sel = grp == g
        line = f"  group {g} ({int(sel.sum())} waves, tiles gathered {gw[sel, 7].sum():.0f}):"
        for k, nm in {0: "start", 1: "depths", 2: "tile 0", 3: "end"}.items():
            v = (gw[sel, k] - g0) / 100.0
            v = v[gw[sel, k] > 0]
            line += f"  {nm} p50 {np.percentile(v, 50):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f} |"
        print(line)
    gw_end = (gw[:, 3].max() - g0) / 100.0
    print(f"  last gather wave ends at {gw_end:.2f}; the walk's first stamp {(t[t[:, 0] > 0][:, 0].min() - g0) / 100.0:.2f}")
t = t[t[:, 0] > 0]
t00 = t[:, 0].min()
t = t[(t[:, 0] - t00) / 100.0 < 50.0]           # (rows a later launch of the same kernel overwrote)
ts = tr._train_step(n_rays, True)
S = ts.raw.numel() // (5 * n_rays)
n_eval = int((ts.raw.reshape(n_rays, S, 5).abs().sum(dim=-1) > 0).sum().item())
print(f"after {n_warm} training steps: {n_eval} of {n_rays * S} samples evaluated (non-zero raw)")
tiles = t[:, 7].astype(int)
print(f"{wl}, {mlp}: {len(t)} ray waves; microseconds after the launch's first stamp; tiles evaluated: " + ", ".join(f"{k}: {int((tiles == k).sum())}" for k in sorted(set(tiles))))
names = {0: "start", 1: "weights staged", 2: "depths sampled", 3: "tile 0 gathers", 4: "tile 0 done", 5: "all tiles done", 6: "loss stage"}
for sel_name, sel in [("all waves", tiles >= 0)] + [(f"waves with {k} tiles", tiles == k) for k in sorted(set(tiles))]:
    if not sel.any():
        continue
    print(f" {sel_name} ({int(sel.sum())})")
    for k, nm in names.items():
        v = (t[sel, k] - t00) / 100.0
        print(f"   {nm:16s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
