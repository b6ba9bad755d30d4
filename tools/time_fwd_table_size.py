#!/usr/bin/env python3
"""How much of k_query_fwd is the table not fitting one XCD's L2?  Same points, same instruction stream, hash tables of
2^16 (6.5 MB), 2^14, 2^12 entries per level.  Run on the GPU box:  python tools/time_fwd_table_size.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naruto_amd import config as C, ops, synthetic as syn       # noqa: E402
from naruto_amd.field import NarutoFieldHIP                      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n_rays, S = 2048, 128
    for hs in (16, 15, 14, 12):
        cfg = C.office0_config()
        cfg["grid"]["hash_size"] = hs
        bbox = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)
        m = NarutoFieldHIP(cfg, bbox).to(dev)
        m.get_uncert_grid(0.1)
        rays = {k: torch.from_numpy(v).to(dev) for k, v in syn.random_rays(n_rays, cfg["mapping"]["bound"], seed=0).items()}
        z = torch.sort(torch.rand(n_rays, S, device=dev) * 4.0 + 0.1, dim=1).values.contiguous()
        for grad in (True, False):
            def f():
                with torch.set_grad_enabled(grad):
                    return ops.field_query(m._handle(), m._params(), rays_o=rays["rays_o"], rays_d=rays["rays_d"], z_vals=z, color=True)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            e1.synchronize()
            print(f"hash_size {hs}  table {m.embed_fn.params.numel() * 4 / 2**20:.2f} MB  feat_save {'on ' if grad else 'off'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us / call")


if __name__ == "__main__":
    main()
