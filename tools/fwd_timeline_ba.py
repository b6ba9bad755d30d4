"""tools/fwd_timeline.py on the BA workload (bench.ba_scene: 2 148 rays x 43 from the keyframe store): per-step clock stamps of the packed
training forward, plus how many samples each phase evaluated.   NARUTO_FWD_PACKED=3 python tools/fwd_timeline_ba.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naruto_amd import _lib
from naruto_amd.ba_loop import FusedBA

dev = torch.device("cuda:0")
use_graph = os.environ.get("TIMELINE_GRAPH", "0") == "1"
cfg, tr, store, smp, current, poses, vol, _ = bench.ba_scene("fp32", False, dev)
n_wg = 256
buf = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
if use_graph:
    lib.naruto_debug_fwd_timeline(buf.data_ptr())          # the captured launch carries the pointer: every replay stamps
ba = FusedBA(tr, store, smp, max_poses=256, use_graph=use_graph)
n_cur, n_train = ba.prepare(current, poses, None)
for i in range(20):
    ba.iteration(i)
torch.cuda.synchronize()
print("graph replay" if use_graph else "eager launches")
names = ["weights staged", "depths + counts", "list 1", "points 1", "gathers 1", "matrix chains 1", "needs 1", "list 2", "points 2", "gathers 2", "matrix chains 2", "needs 2"]
for rep in range(3):
    if not use_graph:
        buf.zero_()
        lib.naruto_debug_fwd_timeline(buf.data_ptr())
    for _ in range(30 if use_graph else 1):
        ba.iteration(20 + rep)
    torch.cuda.synchronize()
    if not use_graph:
        lib.naruto_debug_fwd_timeline(None)
    t = buf.cpu().numpy().reshape(n_wg, 16).astype(np.float64)
    t0, end = t[:, 0], t[:, 15]
    tot = float(np.mean(end - t0))
    print(f"iteration {20 + rep}: {n_train} rays; chunk end - start: mean {tot / 100:.1f}, max {np.max(end - t0) / 100:.1f} us; "
          f"starts spread over {(t0.max() - t0.min()) / 100:.1f}, last end - first start {(end.max() - t0.min()) / 100:.1f}; "
          f"start by workgroup id (every 32nd, relative): {[round((x - t0.min()) / 100) for x in t0[::32]]}")
    prev = t0
    for k, nm in enumerate(names, start=1):
        cur = t[:, k]
        has = cur > 0
        if not has.any():
            continue
        d = (cur - prev)[has] / 100.0
        print(f"  {nm:18s} {has.sum():4d} wgs: mean {d.mean():7.2f} = {100 * d.mean() * 100 / tot:5.1f} %  p10 {np.percentile(d, 10):7.2f} p90 {np.percentile(d, 90):7.2f} max {d.max():7.2f}")
        prev = np.where(has, cur, prev)
    print(f"  tail: mean {np.mean(end - prev) / 100:7.2f} = {100 * np.mean(end - prev) / tot:5.1f} %  max {np.max(end - prev) / 100:7.2f}")
    # how much of the batch was evaluated: raw rows that are not all zero
    ts = tr._train_step(n_train, True)
    raw = ts.t_raw if hasattr(ts, "t_raw") else None
