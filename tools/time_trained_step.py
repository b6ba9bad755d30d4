"""How long does one mapping iteration take on a TRAINED map?  bench.py times random-initialised weights (BASELINE: synthetic data, random
init), where most rays show no early sign change of the sdf and the early-termination logic of the training forward has little to cut.
Here the analytic room scene of tests/accuracy_study.py is mapped first (first_frame_mapping + one global_BA per keyframe), then the
iteration is captured and replayed on a batch of that scene.   NARUTO_FWD_PACKED=0|1 python tools/time_trained_step.py [n_rays] [n_samples_d]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import accuracy_study as A            # the schedule / scene helpers only (no oracle involved here)
from naruto_amd import config as C
from naruto_amd.trainer import MappingTrainer

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
cfg = C.office0_config(perturb=1.0, n_samples_d=nd)
scene, phases = A.make_schedule(cfg, 20, n_rays, 200, 10, seed=0)
torch.manual_seed(0)
tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), dev, 0.1, fused_adam=True)

def timed(batch, tag):
    tr.capture(n_rays, smooth=True)
    b = A._dev_batch(batch, dev)
    for _ in range(20):
        tr.step(*b, smooth=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        tr.step(*b, smooth=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 300 * 1e3
    tr._graphs, tr._static = None, None
    print(f"{tag}: {ms:.4f} ms / iteration  ({n_rays} rays x {nd + 11} samples, NARUTO_FWD_PACKED={os.environ.get('NARUTO_FWD_PACKED', '1')})", flush=True)

last = phases[-1][1][-1]
snap = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
timed(last, "untrained map (random initialisation)")
tr.model.load_state_dict(snap)
for kind, batches in phases:
    db = [A._dev_batch(b, dev) for b in batches]
    if kind == "first":
        tr.first_frame_mapping(db)
    else:
        tr.global_BA(db, smooth=True)
torch.cuda.synchronize()
timed(last, "trained map (390 iterations)")
