"""How do the table gradient's magnitudes sit relative to the resolution of a fixed-point scatter image?

One mapping step of the headline workload (office_0, 2048 rays x 128 samples, smoothness term on) through the unchanged caller's
route (model.forward -> weighted loss -> backward), table gradient materialised.  Run once per library build
(NARUTO_HIP_LIB=... python tools/scatter_precision_study.py out.npz [train_iters]); compare two outputs with --compare a.npz b.npz:
per level, the distribution of |g| over the entries with a non-zero gradient and the error of b against a, in units of |g| (what an
Adam step with eps = 1e-15 sees: its update depends on an entry's gradient only through ratios)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out, iters):
    import torch
    from naruto_amd import config as C, synthetic as syn, trainer
    from naruto_amd.field import NarutoFieldHIP
    dev = torch.device("cuda:0")
    cfg = C.office0_config(perturb=0.0, n_samples_d=117)
    torch.manual_seed(0)
    m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)).to(dev)
    m.get_uncert_grid(0.1)
    with torch.no_grad():
        m.embed_fn.params.copy_(torch.from_numpy(syn.closed_form_table(m.embed_fn.params.numel(), 0.05)).to(dev))
    rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(2048, cfg["mapping"]["bound"], seed=0).items()}
    m.train()
    opt = trainer.create_optimizer(m, cfg)
    r6 = torch.tensor([0.3, 0.6, 0.2, 0.5, 0.1, 0.9])
    for it in range(iters + 1):
        opt.zero_grad(set_to_none=True)
        ret = m.forward(rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"])
        loss = trainer.get_loss_from_ret(m, cfg, ret, smooth=False)
        tr = cfg["training"]
        loss = loss + tr["smooth_weight"] * trainer.smoothness(m, cfg, tr["smooth_pts"], tr["smooth_vox"], margin=tr["smooth_margin"], offset_rand=r6[:3], jitter_rand=r6[3:])
        loss.backward()
        if it < iters:
            opt.step()
    g = m.embed_fn.params.grad.detach().float().cpu().numpy()
    np.savez(out, grad=g, offsets=np.asarray(m._handle().levels()[3], dtype=np.int64))
    print("saved", out, "nonzero entries", int((g != 0).sum()), "max|g|", float(np.abs(g).max()))


def compare(a, b):
    A, B = np.load(a), np.load(b)
    ga, gb = A["grad"].astype(np.float64), B["grad"].astype(np.float64)
    off = A["offsets"]
    if off.size == 0:
        off = np.array([0, ga.size // 2], dtype=np.int64)
    print(f"max|g| {np.abs(ga).max():.3e}; max abs diff {np.abs(ga - gb).max():.3e} = {np.abs(ga - gb).max() / np.abs(ga).max():.2e} of max|g|")
    for l in range(len(off) - 1):
        sa, sb = ga[2 * off[l]:2 * off[l + 1]], gb[2 * off[l]:2 * off[l + 1]]
        nz = sa != 0
        if not nz.any():
            continue
        mag = np.abs(sa[nz])
        rel = np.abs(sb[nz] - sa[nz]) / mag
        q = np.quantile(mag, [0.01, 0.1, 0.5, 0.9])
        print(f"level {l:2d}: {int(nz.sum()):7d} non-zero; |g| 1% {q[0]:.1e} 10% {q[1]:.1e} 50% {q[2]:.1e} 90% {q[3]:.1e} | "
              f"rel err: median {np.median(rel):.1e}, >1e-3: {100 * (rel > 1e-3).mean():5.2f} %, >1e-2: {100 * (rel > 1e-2).mean():5.2f} %, >0.1: {100 * (rel > 0.1).mean():5.2f} %, "
              f"lost (b == 0): {100 * (sb[nz] == 0).mean():5.2f} %")


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
