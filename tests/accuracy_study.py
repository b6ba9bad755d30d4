"""TEST INFRASTRUCTURE: "at matched reconstruction accuracy" (BASELINE.json north_star) measured end to end.

A consistent analytic scene (naruto_amd.synthetic.AnalyticRoom: box room + sphere inside the office_0 volume, closed-form ray casting ->
RGB-D frames from a ring of poses) is mapped with the reference's schedule -- ``first_frame_mapping`` on frame 0, then one ``global_BA`` call
per new keyframe over batches drawn from all frames so far (coslam.py:176-219, 293-399; shipped sampling 32 + 11, jitter on, smoothness on) --
by

  * the CPU oracle (oracle/spec_torch.py, the checker: the reference's arithmetic in plain torch, torch.optim.Adam),
  * the HIP path (MappingTrainer) in the exact fp32 mode and in the bf16-MFMA mode,

from the SAME initial parameters over the SAME ray batches (the depth jitter and the smoothness lattice's placement are each side's own
draws, as they are the reference's own torch.rand).  Compared: the reference's map-accuracy metric -- mean |predicted sdf| at ground-truth
surface points (src/evaluation/eval_mad.py:84-90 through predict_sdf, coslam.py:519-535), in cm --, the depth L1 and colour PSNR of eval-mode
renders from held-out poses, the error of the predicted sdf against the TRUE distance field in the truncation band, and the rank correlation
of the planner's uncertainty volume (get_map_volumes) between the implementations.  Further HIP runs give the run-to-run band (other jitter
seeds) and exercise one documented behaviour change on purpose: NaN depths are treated as missing -- 1 % of the batch's
depths set to NaN must cost nothing.

    python tests/accuracy_study.py [--rays 2048] [--frames 20] [--first 200] [--ba 10] [--out profiles/r04_accuracy_study.json]

tests/test_gpu_parity.py::test_matched_reconstruction_accuracy runs a reduced schedule of the same code.  Only tests/ may use oracle/."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from naruto_amd import config as C  # noqa: E402
from naruto_amd import synthetic as syn  # noqa: E402

KEYS = ("rays_o", "rays_d", "target_rgb", "target_d")


def make_schedule(cfg, n_frames: int, n_rays: int, n_first: int, n_ba: int, seed: int = 0):
    """The batches of the whole run as numpy arrays: [("first", [batch...]), ("ba", [batch...]) per new keyframe]."""
    scene = syn.AnalyticRoom(cfg["mapping"]["bound"])
    frames = [scene.rays(k, n_frames) for k in range(n_frames)]
    rs = np.random.RandomState(seed)

    def take(fr, idx):
        return {k: fr[k][idx] for k in KEYS}

    def cat(parts):
        return {k: np.concatenate([p[k] for p in parts], 0) for k in KEYS}
    phases = [("first", [take(frames[0], rs.randint(0, len(frames[0]["target_d"]), n_rays)) for _ in range(n_first)])]
    n_cur = max(n_rays // 8, 1)
    for kf in range(1, n_frames):
        pool = cat([{k: f[k] for k in KEYS} for f in frames[:kf]])
        n_pool = len(pool["target_d"])
        batches = []
        for _ in range(n_ba):
            a = take(pool, rs.randint(0, n_pool, n_rays - n_cur))
            b = take(frames[kf], rs.randint(0, len(frames[kf]["target_d"]), n_cur))
            batches.append(cat([a, b]))
        phases.append(("ba", batches))
    return scene, phases


def eval_sets(scene, n_frames: int, n_surface: int, seed: int = 1):
    """Ground-truth surface points seen from the training poses (random sub-pixel positions: not the training rays), held-out frames
    between the training poses, and free-space points of the truncation band with their true distance."""
    rs = np.random.RandomState(seed)
    per = -(-n_surface // n_frames)
    surf = np.concatenate([scene.rays(k, n_frames, jitter=rs, count=per)["hit"] for k in range(n_frames)], 0)[:n_surface]
    held = [scene.rays(k + 0.5, n_frames, H=30, W=40, f=30.0) for k in range(0, n_frames, max(n_frames // 4, 1))]
    held = {k: np.concatenate([h[k] for h in held], 0) for k in KEYS}
    # points at a random signed offset within +-8 cm of the surface along the viewing ray (where the sdf losses supervise the field)
    fr = [scene.rays(k, n_frames, jitter=rs, count=4000) for k in range(n_frames)]
    o = np.concatenate([f["rays_o"] for f in fr]); d = np.concatenate([f["rays_d"] for f in fr]); t = np.concatenate([f["target_d"] for f in fr])[:, 0]
    ok = t > 0
    off = rs.uniform(-0.08, 0.08, ok.sum())
    band = (o[ok] + d[ok] * (t[ok] + off)[:, None]).astype(np.float32)
    return {"surface": surf.astype(np.float32), "held": held, "band": band, "band_sdf": scene.sdf(band.astype(np.float64)).astype(np.float32)}


def _norm(points, bound):
    b = np.asarray(bound, np.float32)
    return (points - b[:, 0]) / (b[:, 1] - b[:, 0])


def metrics(model, cfg, ev, device):
    """model: NarutoFieldHIP or OracleField (same operator surface).  Everything in eval mode, no gradients."""
    from scipy.stats import spearmanr  # noqa: F401  (imported here so that the module loads without scipy)
    bound = cfg["mapping"]["bound"]
    was_training = model.training
    model.eval()
    out = {}
    with torch.no_grad():
        q = torch.from_numpy(_norm(ev["surface"], bound)).to(device)
        sdf = torch.cat([model.query_sdf(q[i:i + 65536]).reshape(-1) for i in range(0, len(q), 65536)])
        # eval_mad.py:84-90: mean |predicted sdf| at GT surface points "* 10  # unit: cm" -- the network's sdf is in units of the truncation
        # distance (0.1 m = 10 cm; get_sdf_loss: z + sdf * trunc = depth)
        out["mad_cm"] = float(sdf.abs().mean()) * cfg["training"]["trunc"] * 100.0
        qb = torch.from_numpy(_norm(ev["band"], bound)).to(device)
        sb = torch.cat([model.query_sdf(qb[i:i + 65536]).reshape(-1) for i in range(0, len(qb), 65536)])
        # the network's sdf is in units of the truncation distance (get_sdf_loss: z + sdf * trunc = depth)
        out["band_sdf_err_cm"] = float((sb.cpu() * cfg["training"]["trunc"] - torch.from_numpy(ev["band_sdf"])).abs().mean()) * 100.0
        h = {k: torch.from_numpy(v).to(device) for k, v in ev["held"].items()}
        rend = model.forward(h["rays_o"], h["rays_d"], h["target_rgb"], h["target_d"])
        valid = h["target_d"][:, 0] > 0
        out["heldout_depth_l1_cm"] = float((rend["depth"].reshape(-1)[valid] - h["target_d"][:, 0][valid]).abs().mean()) * 100.0
        mse = float(((rend["rgb"] - h["target_rgb"]) ** 2).mean())
        out["heldout_psnr_db"] = float(-10.0 * np.log10(max(mse, 1e-12)))
    model.train(was_training)
    return out


def uncert_volume(model, cfg, device):
    bbox = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=device)
    if device.type == "cuda":
        from naruto_amd.field import get_map_volumes
        um, _ = get_map_volumes(model.query_sdf, bbox, 0.1)
    else:
        from oracle import spec_torch as S
        with torch.no_grad():
            um, _ = S.get_map_volumes(model.query_sdf, bbox, 0.1)
    return torch.as_tensor(um).detach().float().cpu().numpy()


def _dev_batch(b, device, nan_frac=0.0, rs=None):
    t = {k: torch.from_numpy(b[k]).to(device) for k in KEYS}
    if nan_frac > 0.0:
        m = torch.from_numpy(rs.uniform(size=len(b["target_d"])) < nan_frac).to(device)
        t["target_d"] = torch.where(m[:, None], torch.full_like(t["target_d"], float("nan")), t["target_d"])
    return tuple(t[k] for k in KEYS)


def run_hip(cfg, phases, device, mlp: str = "fp32", seed: int = 0, init=None, nan_frac: float = 0.0):
    """MappingTrainer over the schedule.  ``init``: state_dict to start from (default: a fresh seeded initialisation, returned)."""
    from naruto_amd.trainer import MappingTrainer
    cfg = json.loads(json.dumps(cfg))
    cfg["decoder"]["mlp_precision"] = mlp
    torch.manual_seed(seed)
    tr = MappingTrainer(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32), device, 0.1, fused_adam=True)
    if init is not None:
        tr.model.load_state_dict(init)
    start = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    rs = np.random.RandomState(1234 + seed)
    t0 = time.perf_counter()
    for kind, batches in phases:
        dev_batches = [_dev_batch(b, device, nan_frac, rs) for b in batches]
        if kind == "first":
            tr.first_frame_mapping(dev_batches)
        else:
            tr.global_BA(dev_batches, smooth=True)
    tr.model.check_asserts(block=True)
    torch.cuda.synchronize()
    return tr.model, start, time.perf_counter() - t0


def run_oracle(cfg, phases, init, seed: int = 0, threads: int = 16):
    """The CPU oracle over the same schedule with the reference's loop bodies: first_frame_mapping (coslam.py:197-219: the uncertainty
    grid's gradient accumulates over the loop, one step at the end, not zeroed) and global_BA (coslam.py:361-399)."""
    from oracle import spec_torch as S
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    bbox = torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32)
    ora = S.OracleField(cfg, bbox, 0.1)
    with torch.no_grad():
        ora.table.copy_(init["embed_fn.params"].cpu())
        ora.sdf_w0.copy_(init["decoder.sdf_net.model.0.weight"].cpu())
        ora.sdf_w1.copy_(init["decoder.sdf_net.model.2.weight"].cpu())
        ora.col_w0.copy_(init["decoder.color_net.model.0.weight"].cpu())
        ora.col_w1.copy_(init["decoder.color_net.model.2.weight"].cpu())
        ora.uncert_grid.copy_(init["uncert_grid"].cpu())
    g1, g2 = ora.param_groups()
    o_map = torch.optim.Adam(g1, betas=(0.9, 0.99))
    o_unc = torch.optim.Adam(g2, lr=1)
    tr = cfg["training"]
    ora.train()
    t0 = time.perf_counter()
    for kind, batches in phases:
        if kind == "first":
            o_unc.zero_grad()
            for b in batches:
                o_map.zero_grad()
                ret = ora.forward(*(torch.from_numpy(b[k]) for k in KEYS))
                S.total_loss(ret, tr).backward()
                o_map.step()
            o_unc.step()
        else:
            for i, b in enumerate(batches):
                ret = ora.forward(*(torch.from_numpy(b[k]) for k in KEYS))
                sm = S.smoothness(ora, tr["smooth_pts"], tr["smooth_vox"], tr["smooth_margin"], torch.rand(3), torch.rand(3))
                S.total_loss(ret, tr, smooth_term=sm).backward()
                o_map.step()
                o_map.zero_grad()
                if (i + 1) % 5 == 0:
                    o_unc.step()
                    o_unc.zero_grad()
    return ora, time.perf_counter() - t0


def study(n_rays=2048, n_frames=20, n_first=200, n_ba=10, n_surface=200000, extra_seeds=(1, 2), oracle_threads=16, with_nan=True, verbose=True):
    from scipy.stats import spearmanr
    dev = torch.device("cuda:0")
    cfg = C.office0_config(perturb=1.0)
    scene, phases = make_schedule(cfg, n_frames, n_rays, n_first, n_ba, seed=0)
    ev = eval_sets(scene, n_frames, n_surface)
    n_iter = sum(len(b) for _, b in phases)
    res = {"schedule": {"rays_per_batch": n_rays, "frames": n_frames, "first_frame_iters": n_first, "ba_iters_per_keyframe": n_ba, "iterations": n_iter,
                        "samples_per_ray": cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"], "surface_points": int(len(ev["surface"]))},
           "runs": {}}

    def log(name, m, secs):
        res["runs"][name] = dict(m, train_seconds=round(secs, 2))
        if verbose:
            print(f"{name:28s} MAD {m['mad_cm']:.3f} cm | band sdf err {m['band_sdf_err_cm']:.3f} cm | held-out depth L1 {m['heldout_depth_l1_cm']:.3f} cm | PSNR {m['heldout_psnr_db']:.2f} dB"
                  f" | {secs:.1f} s for {n_iter} iterations", flush=True)
    m32, init, secs = run_hip(cfg, phases, dev, "fp32", seed=0)
    untrained = metrics_of_init(cfg, init, ev, dev)
    res["untrained"] = untrained
    if verbose:
        print(f"{'untrained':28s} MAD {untrained['mad_cm']:.3f} cm | band sdf err {untrained['band_sdf_err_cm']:.3f} cm | held-out depth L1 {untrained['heldout_depth_l1_cm']:.3f} cm", flush=True)
    log("hip_fp32", metrics(m32, cfg, ev, dev), secs)
    vol32 = uncert_volume(m32, cfg, dev)
    ora, secs = run_oracle(cfg, phases, init, seed=0, threads=oracle_threads)
    log("oracle_cpu_fp32", metrics(ora, cfg, ev, torch.device("cpu")), secs)
    vol_o = uncert_volume(ora, cfg, torch.device("cpu"))
    mbf, _, secs = run_hip(cfg, phases, dev, "bf16", seed=0, init=init)
    log("hip_bf16", metrics(mbf, cfg, ev, dev), secs)
    volbf = uncert_volume(mbf, cfg, dev)
    vols = {}
    for s in extra_seeds:
        ms, _, secs = run_hip(cfg, phases, dev, "fp32", seed=s, init=init)
        log(f"hip_fp32_jitter_seed{s}", metrics(ms, cfg, ev, dev), secs)
        vols[s] = uncert_volume(ms, cfg, dev)
    if with_nan:
        mn, _, secs = run_hip(cfg, phases, dev, "fp32", seed=0, init=init, nan_frac=0.01)
        log("hip_fp32_1pct_nan_depth", metrics(mn, cfg, ev, dev), secs)

    def rho(a, b):
        return float(spearmanr(a.reshape(-1), b.reshape(-1)).correlation)
    res["uncert_volume_spearman"] = {"hip_fp32_vs_oracle": rho(vol32, vol_o), "hip_bf16_vs_oracle": rho(volbf, vol_o),
                                     **{f"hip_fp32_vs_hip_fp32_seed{s}": rho(vol32, v) for s, v in vols.items()}}
    r = res["runs"]
    mads = [r["hip_fp32"]["mad_cm"]] + [r[f"hip_fp32_jitter_seed{s}"]["mad_cm"] for s in extra_seeds]
    res["summary"] = {"mad_hip_fp32_over_oracle": r["hip_fp32"]["mad_cm"] / r["oracle_cpu_fp32"]["mad_cm"],
                      "mad_hip_bf16_over_oracle": r["hip_bf16"]["mad_cm"] / r["oracle_cpu_fp32"]["mad_cm"],
                      "mad_hip_fp32_seed_spread": (max(mads) - min(mads)) / float(np.mean(mads)) if len(mads) > 1 else None}
    if verbose:
        print(json.dumps(res["summary"]), json.dumps(res["uncert_volume_spearman"]), flush=True)
    return res


def metrics_of_init(cfg, init, ev, dev):
    from naruto_amd.field import NarutoFieldHIP
    m = NarutoFieldHIP(cfg, torch.tensor(cfg["mapping"]["bound"], dtype=torch.float32, device=dev)).to(dev)
    m.get_uncert_grid(0.1)
    m.load_state_dict(init)
    return metrics(m, cfg, ev, dev)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--first", type=int, default=200)
    ap.add_argument("--ba", type=int, default=10)
    ap.add_argument("--surface", type=int, default=200000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = study(a.rays, a.frames, a.first, a.ba, a.surface, oracle_threads=a.threads)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)
