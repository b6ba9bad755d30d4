"""Where the unchanged caller's iteration goes (tools/dropin_caller.py = coslam.py:361-399): wall time of each piece with a device sync
around it, and the same pieces without syncs (host-side issue time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naruto_amd import config as C, synthetic as syn
from naruto_amd.field import NarutoFieldHIP
from dropin_caller import DropInCaller

dev = torch.device("cuda:0")
cfg = C.office0_config(perturb=1.0, n_samples_d=117)
n = 2048
opt = sys.argv[1] if len(sys.argv) > 1 else "torch"
sm = sys.argv[2] if len(sys.argv) > 2 else "reference"
torch.manual_seed(0)
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)).to(dev).train()
c = DropInCaller(m, cfg, 0.1, optimizer=opt, smoothness=sm)
rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(n, cfg["mapping"]["bound"], seed=0).items()}
a = (rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"])
for i in range(10):
    c.ba_iteration(i, *a)
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
N = 20
tr = cfg["training"]
for i in range(N):
    t0 = time.perf_counter()
    ret = m.forward(*a); t0 = tick("model.forward", t0)
    loss = c.get_loss_from_ret(ret); t0 = tick("get_loss_from_ret (no smooth)", t0)
    s = c.smoothness(tr['smooth_pts'], tr['smooth_vox'], margin=tr['smooth_margin']); t0 = tick("smoothness fwd", t0)
    loss = loss + tr['smooth_weight'] * s; t0 = tick("loss += w*s", t0)
    loss.backward(retain_graph=True); t0 = tick("loss.backward", t0)
    c.map_optimizer.step(); t0 = tick("map_optimizer.step", t0)
    c.map_optimizer.zero_grad(); t0 = tick("map_optimizer.zero_grad", t0)
    if (i + 1) % 5 == 0:
        c.uncert_optim.step(); c.uncert_optim.zero_grad(); t0 = tick("uncert step", t0)
for k, v in acc.items():
    print(f"{k:36s} {v / N * 1e3:8.3f} ms")
print("sum", sum(acc.values()) / N * 1e3)
# finer: pieces of the smoothness forward
bb = m.bounding_box
acc.clear()
for i in range(N):
    t0 = time.perf_counter()
    co = torch.stack(torch.meshgrid(*(torch.arange(0, 31),) * 3, indexing="ij"), -1).float(); t0 = tick("coordinates cpu", t0)
    co = co.to(bb); t0 = tick("coords H2D", t0)
    r = torch.rand(3).to(bb); t0 = tick("rand3 to", t0)
    pts = (co + r.reshape(1, 1, 1, 3)) * 0.1 + bb[:, 0] + r
    pts = (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0]); t0 = tick("pts ops", t0)
    e = m.query_sdf(pts, embed=True); t0 = tick("query_sdf embed", t0)
    tv = torch.pow(e[1:, ...] - e[:-1, ...], 2).sum() + torch.pow(e[:, 1:, ...] - e[:, :-1, ...], 2).sum() + torch.pow(e[:, :, 1:, ...] - e[:, :, :-1, ...], 2).sum(); t0 = tick("tv ops", t0)
    m.embed_fn.params.grad = None
    tv.backward(); t0 = tick("tv backward", t0)
for k, v in acc.items():
    print(f"  {k:34s} {v / N * 1e3:8.3f} ms")
