"""k_query_fwd<color> duration against the number of 64-sample tiles (how the launch's workgroups fall on the 256 CUs)."""
import sys, os, ctypes as CT
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naruto_amd import config as C, synthetic as syn, _lib, ops
from naruto_amd.field import NarutoFieldHIP

dev = torch.device("cuda:0")
cfg = C.office0_config(perturb=1.0)
m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)).to(dev)
m.get_uncert_grid(0.1)
with torch.no_grad():
    m.embed_fn.params.copy_(torch.from_numpy(syn.closed_form_table(m.embed_fn.params.numel(), 0.05)).to(dev))
lib = _lib.load()
h = m._handle()
ps = ops._params_struct({k: v.detach() for k, v in m._params().items()})
for tiles in [int(a) for a in sys.argv[1:]] or [512, 768, 1024, 1100, 1376, 1536, 2048, 3072, 4096]:
    M = tiles * 64
    x = torch.rand(M, 3, device=dev)
    raw = torch.empty(M, 5, device=dev)
    feat = torch.empty(16, M, 2, device=dev)
    pts, _ = ops._points_struct(x, None, None, None)
    run = lambda: _lib.check(lib.naruto_query_fwd(h.ptr, CT.byref(ps), M, CT.byref(pts), raw.data_ptr(), None, None, feat.data_ptr(), ops._stream()))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"tiles {tiles:5d}  workgroups {(tiles + 3) // 4:5d}  {e0.elapsed_time(e1) / 50 * 1e3:7.2f} us")
