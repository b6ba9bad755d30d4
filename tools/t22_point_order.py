"""does the ORDER of a launch's points matter for the hash gather at T = 2^22 (281 MB table, every gather an HBM line)?
The flat field query (k_query_fwd over explicit points) of BASELINE configs[4]'s batch -- 131 072 rays x 43 samples in the unit cube -- with
the points (a) in ray order, as the training forward has them, (b) sorted by a 30-bit Morton key of the position at 1/64 resolution (the
review's proposal), (c) at 1/1024, (d) shuffled: time per launch (HIP events), and what the sort itself would cost (torch.sort + two gathers).
    python tools/t22_point_order.py [log2_T] [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naruto_amd import ops
from naruto_amd.field import NarutoFieldHIP

log2T = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
dev = torch.device("cuda:0")
cfg, _ = bench.workload("unit1024_T22_131072x43" if log2T == 22 else "unit1024_131072x43")
torch.manual_seed(0)
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32)).to(dev)
m.get_uncert_grid(0.1)
with torch.no_grad():
    m.embed_fn.params.uniform_(-1e-2, 1e-2)
rays = {k: torch.from_numpy(v).to(dev) for k, v in bench.bench_rays(cfg, n_rays).items()}
tr, cam = cfg["training"], cfg["cam"]
S = tr["n_samples_d"] + tr["n_range_d"]
z = ops.sample_z(n_rays, rays["target_d"], float(cam["near"]), float(cam["far"]), tr["n_samples_d"], tr["n_range_d"], float(tr["range_d"]),
                 rand=torch.rand(n_rays, S, device=dev))
pts = rays["rays_o"][:, None, :] + rays["rays_d"][:, None, :] * z.reshape(n_rays, S, 1)
bb = m.bounding_box.to(dev)
x = ((pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])).reshape(-1, 3).contiguous()
M = x.shape[0]


def morton(x, bits):
    q = (x.clamp(0, 1 - 1e-7) * (1 << bits)).to(torch.int64)
    key = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
    for b in range(bits):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return key


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"T = 2^{log2T}: table {m.embed_fn.params.numel() * 4 / 1e6:.0f} MB; {n_rays} rays x {S} samples = {M} points; inside the unit cube: {float(((x >= 0) & (x < 1)).all(dim=1).float().mean()):.3f}")
with torch.no_grad():
    orders = {"ray order (as launched)": None, "Morton key, 6 bits per axis (1/64)": torch.argsort(morton(x, 6)), "Morton key, 10 bits per axis (1/1024)": torch.argsort(morton(x, 10)),
              "shuffled": torch.randperm(M, device=dev)}
    base = None
    for name, perm in orders.items():
        xp = x if perm is None else x[perm].contiguous()
        ms = timed(lambda: ops.field_query(m._handle(), m._params(), x=xp, color=True))
        base = base or ms
        print(f"  k_query_fwd<color>, points in {name:40s} {ms:8.3f} ms  ({ms / base:5.2f} of ray order)")
    key = morton(x, 10)
    raw = ops.field_query(m._handle(), m._params(), x=x, color=True)
    t_key = timed(lambda: morton(x, 10), 3)
    t_sort = timed(lambda: torch.sort(key), 3)
    perm = torch.argsort(key)
    t_g = timed(lambda: x[perm].contiguous(), 3)
    t_s = timed(lambda: raw[perm], 3)
    print(f"  what the re-ordering would cost with torch ops: key {t_key:.3f} ms (30 elementwise launches; one kernel would do), sort {t_sort:.3f} ms, gather of the points {t_g:.3f} ms, "
          f"scatter of the raw rows back {t_s:.3f} ms")
