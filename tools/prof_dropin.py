"""torch.profiler view of the unchanged caller's iteration (host-side op times + kernel times)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from naruto_amd import config as C, synthetic as syn
from naruto_amd.field import NarutoFieldHIP
from dropin_caller import DropInCaller

dev = torch.device("cuda:0")
cfg = C.office0_config(perturb=1.0, n_samples_d=117)
n = 2048
opt = sys.argv[1] if len(sys.argv) > 1 else "torch"
sm = sys.argv[2] if len(sys.argv) > 2 else "reference"
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if threads:
    torch.set_num_threads(threads)
torch.manual_seed(0)
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)).to(dev).train()
c = DropInCaller(m, cfg, 0.1, optimizer=opt, smoothness=sm)
rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(n, cfg["mapping"]["bound"], seed=0).items()}
a = (rays["rays_o"], rays["rays_d"], rays["target_rgb"], rays["target_d"])
for i in range(10):
    c.ba_iteration(i, *a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    c.ba_iteration(10 + i, *a)
torch.cuda.synchronize()
print("ms/iter", (time.perf_counter() - t0) / 50 * 1e3, "threads", torch.get_num_threads())
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(10):
        c.ba_iteration(60 + i, *a)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
