import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here); sys.path.insert(0, os.path.join(here, "tests"))
import numpy as np, torch, ctypes as CT
from naruto_amd import _lib, ops, synthetic as syn
import helpers as H
dev = torch.device("cuda:0")
cfg = H.office_cfg(16, perturb=1.0, n_samples_d=117)
from naruto_amd.field import NarutoFieldHIP
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], device=dev)).to(dev); m.get_uncert_grid(0.1)
h = m._handle(); lib = _lib.load()
N, S = 2048, 128; M = N * S
rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(N, cfg["mapping"]["bound"], seed=0).items()}
z = ops.sample_z(N, rays["target_d"], 0.0, 5.0, 117, 11, 0.1, rand=torch.rand(N, S, device=dev))
params = {k: v.detach() for k, v in m._params().items()}
ps = ops._params_struct(params); pts, _ = ops._points_struct(None, rays["rays_o"], rays["rays_d"], z)
feat = torch.randn(16, M, 2, device=dev); d_raw = torch.randn(M, 5, device=dev)
grads = {k: torch.zeros_like(v) for k, v in params.items()}
gs = _lib.NarutoGrads()
for k in ("uncert_grid", "sdf_w0", "sdf_w1", "col_w0", "col_w1"): setattr(gs, k, grads[k].data_ptr())
ws = torch.empty(lib.naruto_query_bwd_workspace(h.ptr, M) // 4 + 16, device=dev)
act = torch.arange(M, dtype=torch.int32, device=dev)
def timeit(n_act):
    nact = torch.tensor([n_act], dtype=torch.int32, device=dev)
    f = lambda: _lib.check(lib.naruto_query_bwd(h.ptr, CT.byref(ps), M, CT.byref(pts), feat.data_ptr(), d_raw.data_ptr(), None, act.data_ptr(), nact.data_ptr(), None, 0, CT.byref(gs), ws.data_ptr(), None))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for n in (0, 32, 1024 * 32, 2 * 1024 * 32, 3 * 1024 * 32, 94000, 4 * 1024 * 32, 8 * 1024 * 32):
    print(f"n_active {n:7d}: query_bwd+wgrad_reduce {timeit(n):7.1f} us")
