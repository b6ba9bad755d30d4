import os, sys, subprocess, json
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) == 1:
    # driver: one subprocess per configuration (the knobs are read at field creation)
    cfgs = [("default", {})]
    for sh, sd in [(1, 3), (2, 6), (2, 12), (3, 9), (4, 12), (4, 24)]:
        cfgs.append((f"all sh{sh} sd{sd}", {"NARUTO_DEBUG_SCATTER_SPLITS_HASHED": str(sh), "NARUTO_DEBUG_SCATTER_SPLITS_DENSE": str(sd)}))
    for l in (0, 1, 2, 3, 4, 5, 10, 15):
        cfgs.append((f"L{l} sh2 sd6", {"NARUTO_DEBUG_SCATTER_LEVELS": str(1 << l), "NARUTO_DEBUG_SCATTER_SPLITS_HASHED": "2", "NARUTO_DEBUG_SCATTER_SPLITS_DENSE": "6"}))
    for name, env in cfgs:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    sys.exit(0)
sys.path.insert(0, here); sys.path.insert(0, os.path.join(here, "tests"))
import numpy as np, torch, ctypes as C
from naruto_amd import _lib, ops, synthetic as syn
import helpers as H
dev = torch.device("cuda:0")
cfg = H.office_cfg(16, perturb=1.0, n_samples_d=117)
from naruto_amd.field import NarutoFieldHIP
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], device=dev)).to(dev); m.get_uncert_grid(0.1)
h = m._handle(); lib = _lib.load()
rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(2048, cfg["mapping"]["bound"], seed=0).items()}
z = ops.sample_z(2048, rays["target_d"], 0.0, 5.0, 117, 11, 0.1, rand=torch.rand(2048, 128, device=dev))
pts = rays["rays_o"][:, None] + rays["rays_d"][:, None] * z[..., None]
bb = m.bounding_box
x = ((pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0])).reshape(-1, 3).contiguous()
M = x.shape[0]
dfeat = torch.randn(M, 32, device=dev)
if os.environ.get("DBG_SPARSE"): dfeat = dfeat * (z.reshape(-1, 1) < rays["target_d"].repeat_interleave(128, 0) + 0.1)
dtab = torch.zeros(h.n_params, device=dev)
ws = torch.empty(lib.naruto_scatter_workspace(h.ptr) // 4 + 4, device=dev)
def run():
    _lib.check(lib.naruto_hash_encode_bwd(h.ptr, M, x.data_ptr(), dfeat.data_ptr(), dtab.data_ptr(), ws.data_ptr(), None))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); e1.synchronize()
print(f"scatter {e0.elapsed_time(e1)/10*1e3:.1f} us")
