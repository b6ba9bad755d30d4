"""Per-workgroup timeline of k_hash_scatter_lds (round 5): thread 0 of every workgroup stamps the 100 MHz global counter at its start, behind the
image's zero fill, behind its points and at its end, and says which unit it is.  Printed per unit type and level: start / duration of the three parts.
    python tools/scatter_timeline.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naruto_amd import _lib
from naruto_amd.trainer import MappingTrainer

wl = sys.argv[1] if len(sys.argv) > 1 else "office0_2048x128"
dev = torch.device("cuda:0")
cfg, n_rays = bench.workload(wl)
torch.manual_seed(0)
tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)
rays = {k: torch.from_numpy(v).to(dev) for k, v in bench.bench_rays(cfg, n_rays).items()}
step = lambda: tr.step(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"], smooth=True)
for _ in range(5):
    step()
n_wg = 4096
buf = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
os.environ.setdefault("NARUTO_WALK_PARTIAL", "0")
lib.naruto_debug_fwd_timeline(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.naruto_debug_fwd_timeline(None)
t = buf.cpu().numpy().reshape(n_wg, 8)
t = t[(t[:, 4] > 0) & (t[:, 4] < (1 << 24))]       # only rows a scatter workgroup tagged (the forward's waves leave time stamps -- huge numbers -- in the same buffer)
t0 = t[:, 0].min()
f = lambda a: (a.astype(np.float64) - t0) / 100.0
st = tr._train_steps[next(iter(tr._train_steps))] if hasattr(tr, "_train_steps") else None
print(f"{wl}: {len(t)} scatter workgroups; microseconds after the first workgroup's start; kernel ends at {f(t[:, 3]).max():.1f}")
print(f"{'workgroup':28s} {'n':>4s} {'start':>7s} {'zeroed':>7s} {'points':>8s} {'(max)':>7s} {'write':>6s} {'end max':>8s}")
tag = t[:, 4]
unc = tag >= 0x10000
rows = []
units = np.where(unc, -1, (tag - 1) >> 8)
for u in sorted(set(units.tolist())):
    m = units == u
    s0, z, p, e = f(t[m, 0]), f(t[m, 1]), f(t[m, 2]), f(t[m, 3])
    name = "uncertainty grid" if u < 0 else f"unit {u}"
    print(f"{name:28s} {int(m.sum()):4d} {s0.mean():7.2f} {(z - s0).mean():7.2f} {(p - z).mean():8.2f} {(p - z).max():7.2f} {(e - p).mean():6.2f} {e.max():8.2f}")
