import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests")
import torch, numpy as np
import helpers as H
from naruto_amd import synthetic as syn
gpu = torch.device("cuda:0")
cfg = H.office_cfg(16, perturb=1.0); cfg["cam"]["far"] = 3.0
ora = H.make_oracle(cfg, 0.2, 29).eval(); m = H.make_hip_from_oracle(cfg, ora, gpu).eval()
N, S_tot = 8192, 43
rays = syn.random_rays(N, cfg["mapping"]["bound"], seed=29, zero_depth_frac=0.05)
ro, rd, td = (torch.from_numpy(rays[k]).to(gpu) for k in ("rays_o", "rays_d", "target_d"))
rand = torch.rand(N, S_tot, generator=torch.Generator().manual_seed(7)).to(gpu)
with torch.no_grad():
    whole = m.render_rays(ro, rd, target_d=td, rand=rand)["raw"]
    parts = torch.cat([m.render_rays(ro[i:i + 2048], rd[i:i + 2048], target_d=td[i:i + 2048], rand=rand[i:i + 2048])["raw"] for i in range(0, N, 2048)], 0)
    want = ora.double().render_rays(ro.cpu().double(), rd.cpu().double(), target_d=td.cpu().double(), rand=rand.cpu().double())["raw"] if False else ora.render_rays(ro.cpu(), rd.cpu(), target_d=td.cpu(), rand=rand.cpu())["raw"]
for c, name in enumerate(("r", "g", "b", "sdf")):
    a, b, w = whole[..., c].double().cpu(), parts[..., c].double().cpu(), want[..., c].double()
    sc = float(b.abs().max())
    d = (a - b).abs()
    print(name, "scale", sc, "x3-vs-fp32 max", float(d.max()), "mean", float(d.mean()), "p99.9", float(d.flatten().kthvalue(int(d.numel()*0.999)).values), "| x3-vs-oracle", float((a-w).abs().max()), "fp32chain-vs-oracle", float((b-w).abs().max()))
