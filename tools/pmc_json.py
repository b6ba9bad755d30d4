#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as summarised by prof_summary.py.

    python tools/pmc_json.py <workload> <fetch.txt> <write.txt> [<workload> <fetch.txt> <write.txt> ...] > profiles/r02_pmc.json
    (round 5: a <write.txt> given as "<write.txt>,<tcc_req.txt>" adds tcc_req = TCC_REQ_sum per launch -- the requests reaching the L2)

The output is keyed by workload (bench.py --workload), then by bench.py's kernel-table row.

Units and corrections (MI355X_MICROARCH.md, "HBM"): the counters report KiB per dispatch; on gfx950 FETCH_SIZE counts
128-byte requests of wide coalesced streams as 64 bytes (x2 for those), other access widths are uncalibrated.  The
kernels here read mostly through 8-byte gathers, so the raw value is kept and the x2 bound is given next to it.
"""
import json
import re
import sys

NAMES = {"k_query_fwd<color>": ["k_query_fwdILb1"], "k_query_bwd": ["k_query_bwd"],
         "k_query_fwd_loss": ["k_query_fwd_lossI"], "k_query_fwd_loss_short": ["k_query_fwd_loss_shortI"], "k_query_fwd_list": ["k_query_fwd_listI"],
         "k_hash_scatter+reduce+k_wgrad_reduce": ["k_hash_scatter_lds", "k_scatter_reduce", "k_hash_scatter_atomic", "k_bwd_post", "k_wgrad_reduce", "k_bin_count",
                                                  "k_bin_colscan", "k_bin_start", "k_bin_fill", "k_bin_apply"],
         "k_hash_scatter_lds": ["k_hash_scatter_lds"], "k_bin_fill": ["k_bin_fill"], "k_bin_apply": ["k_bin_apply"], "k_bin_count": ["k_bin_count"], "k_bwd_finish": ["k_bwd_finish"], "k_tv_encode": ["k_tv_encode"],
         "k_loss_stage": ["k_loss_stage"], "k_composite_bwd<loss>": ["k_composite_bwdILb1"], "k_adam_multi": ["k_adam_multi"]}


def parse(path):
    """kernel -> per-dispatch average.  A kernel launched in several grid sizes is taken at its LARGEST one (the "by launch
    shape" table of prof_summary.py): for k_query_fwd that is the launch over all samples, the one bench.py's roofline times."""
    out, shaped = {}, {}
    section = None
    for line in open(path):
        if line.startswith("PMC counters by launch shape"):
            section = "shape"
            continue
        if line.startswith("PMC counters"):
            section = "all"
            continue
        if section is None or line.startswith("kernel"):
            continue
        if section == "all":
            m = re.match(r"(\S+)\s+(\w+)\s+(\d+)\s+([0-9.]+)\s*$", line)
            if m:
                out[m.group(1)] = float(m.group(4))
        else:
            m = re.match(r"(\S+)\s+(\d+)\s+(\w+)\s+(\d+)\s+([0-9.]+)\s*$", line)
            if m and (m.group(1) not in shaped or int(m.group(2)) > shaped[m.group(1)][0]):
                shaped[m.group(1)] = (int(m.group(2)), float(m.group(5)))
    for k, (g, v) in shaped.items():
        out[k] = v
    return out


out = {"note": "per-dispatch averages, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over eager launches "
               "(bench.py --no-graph); bytes = KiB * 1024.  fetch_bytes_x2 = the gfx950 wide-read correction applied (upper bound).  "
               "A kernel launched in several grid sizes is taken at its largest one (bench.py's kernel-table launch over all samples)."}
args = sys.argv[1:]
for k in range(0, len(args) - 2, 3):
    wl = args[k]
    wpath, _, tpath = args[k + 2].partition(",")
    fetch, write = parse(args[k + 1]), parse(wpath)
    tcc = parse(tpath) if tpath else {}
    res = {}
    for label, keys in NAMES.items():
        f = sum(v for kk, v in fetch.items() if any(x in kk for x in keys))
        w = sum(v for kk, v in write.items() if any(x in kk for x in keys))
        if f == 0 and w == 0:
            continue
        res[label] = {"fetch_bytes": int(f * 1024), "fetch_bytes_x2": int(2 * f * 1024), "write_bytes": int(w * 1024),
                      "traffic_bytes": int((f + w) * 1024)}
        q = sum(v for kk, v in tcc.items() if any(x in kk for x in keys))
        if q:
            res[label]["tcc_req"] = int(q)
    out[wl] = res
print(json.dumps(out, indent=1))
