"""Per-step timeline of the packed training forward (k_query_fwd_loss_packed): thread 0 of every ray workgroup stamps the shader clock
(the constant-rate global counter, wall_clock64: 100 MHz) at the start and behind each step of its first chunk; printed: mean / p10 / p90
over the workgroups of the time each step takes, in microseconds.   python tools/fwd_timeline.py [workload] [mlp]"""
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
for _ in range(5):
    step()
n_wg = 256
buf = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.naruto_debug_fwd_timeline(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
step()
torch.cuda.synchronize()
lib.naruto_debug_fwd_timeline(None)
t = buf.cpu().numpy().reshape(n_wg, 16).astype(np.float64)
names = ["weights staged", "depths + counts", "list 1", "points 1", "gathers 1", "matrix chains 1", "needs 1", "list 2", "points 2", "gathers 2", "matrix chains 2", "needs 2"]
t0 = t[:, 0]
end = t[:, 15]
tot = float(np.mean(end - t0))
print(f"{wl} {mlp}: {n_wg} ray workgroups; microseconds (100 MHz global counter); "
      f"first chunk end - start: mean {tot / 100:.1f}, max {np.max(end - t0) / 100:.1f}")
prev = t0
for k, nm in enumerate(names, start=1):
    cur = t[:, k]
    has = cur > 0
    if not has.any():
        continue
    d = (cur - prev)[has] / 100.0
    print(f"  {nm:18s} {has.sum():4d} workgroups: mean {d.mean():7.2f} = {100 * d.mean() * 100 / tot:5.1f} %   p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}   (ends at {np.mean((cur - t0)[has]) / 100:7.2f})")
    prev = np.where(has, cur, prev)
print(f"  {'tail (zero fill, loss stage, rows)':18s}: mean {np.mean(end - prev) / 100:7.2f} = {100 * np.mean(end - prev) / tot:5.1f} %")
