"""Per-phase timeline of the short-ray training forward (k_query_fwd_loss_short, round 5): thread 0 of every workgroup -- ray workgroups and the
lattice-encode tail workgroups -- stamps the 100 MHz global counter; printed: when the phases end relative to the launch's first stamp.
    python tools/short_timeline.py [n_rays] [mlp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naruto_amd import _lib
from naruto_amd.trainer import MappingTrainer

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2148
mlp = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
cfg, _ = bench.workload("office0_2048x43")
cfg["decoder"]["mlp_precision"] = mlp
torch.manual_seed(0)
tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)
rays = {k: torch.from_numpy(v).to(dev) for k, v in bench.bench_rays(cfg, n_rays).items()}
step = lambda: tr.step(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"], smooth=True)
for _ in range(5):
    step()
n_wg = 4096
buf = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.naruto_debug_fwd_timeline(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.naruto_debug_fwd_timeline(None)
t = buf.cpu().numpy().reshape(n_wg, 8).astype(np.float64)
used = t[:, 0] > 0
t = t[used]
t00 = t[:, 0].min()
ray = t[:, 1] > 0
print(f"{n_rays} rays x 43, {mlp}: {int(ray.sum())} ray workgroups, {int((~ray).sum())} tail workgroups; microseconds after the launch's first stamp")
names = {0: "start", 5: "weights staged", 1: "depths sampled", 2: "gathers (wave 0)", 3: "tiles done", 4: "loss stage", 7: "end"}
for k, nm in names.items():
    v = (t[ray, k] - t00) / 100.0
    print(f"  ray workgroups  {nm:18s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
if (~ray).any():
    s0, s1 = (t[~ray, 0] - t00) / 100.0, (t[~ray, 7] - t00) / 100.0
    print(f"  tail workgroups start: mean {s0.mean():7.2f} p10 {np.percentile(s0, 10):7.2f} p90 {np.percentile(s0, 90):7.2f} | end: mean {s1.mean():7.2f} max {s1.max():7.2f} | duration mean {np.mean(s1 - s0):6.2f}")
    h, e = np.histogram(s0, bins=12)
    print("  tail workgroup starts, histogram:", " ".join(f"{e[i]:.0f}:{h[i]}" for i in range(len(h))))
