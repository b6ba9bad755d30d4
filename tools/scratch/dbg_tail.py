import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import helpers as H
from naruto_amd import synthetic as syn
from naruto_amd import ops
gpu = torch.device('cuda:0')
cfg = H.office_cfg(12, perturb=1.0)
tr, cam = cfg["training"], cfg["cam"]
ora = H.make_oracle(cfg, 0.25, 31)
m = H.make_hip_from_oracle(cfg, ora, gpu)
n_rays, s_d, s_r = 96, 32, 11
rays = syn.random_rays(n_rays, cfg["mapping"]["bound"], seed=131, zero_depth_frac=0.1)
t = {k: torch.from_numpy(v) for k, v in rays.items()}
w = torch.tensor([tr["rgb_weight"], tr["depth_weight"], tr["sdf_weight"], tr["fs_weight"], 0.0, tr["uncert_weight"], 0.0, 0.0, 0.1, 0.0])
args = [t[k].to(gpu).contiguous() for k in ("rays_o", "rays_d", "target_rgb")] + [t["target_d"].to(gpu).reshape(-1).contiguous()]
out = []
for fuse in (False, True):
    ts = ops.TrainStep(m._handle(), m._params(), torch.zeros_like(m.uncert_grid), n_rays, n_samples_d=s_d, n_range_d=s_r,
                       near=cam["near"], far=cam["far"], range_d=tr["range_d"], depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"],
                       perturb=True, loss_weights=w.to(gpu), smooth=(12, 0.1, 0.05), device_rng=True, seed=77)
    ts.fuse_tail = fuse
    ts.run(*args)
    torch.cuda.synchronize()
    out.append((ts.d_raw.cpu().clone(), ts.raw.cpu().clone(), ts.active_idx.cpu().clone(), int(ts.n_active.cpu()), {k: v.cpu().clone() for k, v in ts.grads.items()}))
a, b = out
print("raw equal", torch.equal(a[1], b[1]), "d_raw equal", torch.equal(a[0], b[0]), "n", a[3], b[3], "idx", torch.equal(a[2][:a[3]], b[2][:b[3]]))
d = (a[0] - b[0]).abs(); print("d_raw maxdiff", d.max().item(), "nz", int((d > 0).sum()), "scale", a[0].abs().max().item())
for k in a[4]:
    dd = (a[4][k] - b[4][k]).abs()
    print(k, torch.equal(a[4][k], b[4][k]), dd.max().item(), a[4][k].abs().max().item(), int((dd > 0).sum()))
