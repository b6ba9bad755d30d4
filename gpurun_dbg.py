import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here); sys.path.insert(0, os.path.join(here, "tests"))
import numpy as np, torch
from naruto_amd import config as C, synthetic as syn
from naruto_amd.trainer import MappingTrainer, pack_rays
from naruto_amd.field import get_map_volumes
dev = torch.device("cuda:0")
def run(name, cfg, n_rays, steps=20, graph=True):
    torch.manual_seed(0)
    tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)
    rays = syn.random_rays(n_rays, cfg["mapping"]["bound"], seed=0)
    t = pack_rays(*(torch.from_numpy(rays[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d")))
    if graph: tr.capture(n_rays, smooth=True)
    for _ in range(5): ret, loss = tr.step(*t, smooth=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): ret, loss = tr.step(*t, smooth=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    tr.model.check_asserts()
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    print(f"{name}: {n_rays}x{S}  {dt*1e3:.3f} ms/iter  {n_rays/dt/1e6:.2f} M rays/s  loss {float(loss):.4f}", flush=True)
    return tr
tr = run("office0 2048x43 (shipped)", C.office0_config(perturb=1.0), 2048)
run("office0 2048x128", C.office0_config(perturb=1.0, n_samples_d=117), 2048)
run("office0 8192x43", C.office0_config(perturb=1.0), 8192)
run("mp3d 2048x256", C.mp3d_large_config(perturb=1.0, n_samples_d=245), 2048)
run("unit1024 T16 16384x43", C.unit_cube_config(1024, 16, perturb=1.0), 16384)
run("unit1024 T22 16384x43", C.unit_cube_config(1024, 22, perturb=1.0), 16384)
# planner query path (config 3): eval render of 8192 rays + dense map query
m = tr.model.eval()
rays = syn.random_rays(8192, tr.config["mapping"]["bound"], seed=1)
t = [torch.from_numpy(rays[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d")]
with torch.no_grad():
    for _ in range(3): r = m.forward(*t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = m.forward(*t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"eval render 8192x43 incl. uncert_map: {dt*1e3:.3f} ms  {8192/dt/1e6:.2f} M rays/s")
    for _ in range(3): v = get_map_volumes(m.query_sdf, m.bounding_box, 0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): v = get_map_volumes(m.query_sdf, m.bounding_box, 0.1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"get_map_volumes {v[0].shape} (96 040 pts) incl. D2H: {dt*1e3:.3f} ms")
