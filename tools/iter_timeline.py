#!/usr/bin/env python3
"""Per-dispatch timeline of the LAST iterations of a rocprofv3 --kernel-trace run (rocpd .db): start / end of every kernel
relative to the first kernel of the window, with the queue it ran on -- shows what overlaps under graph replay.

    python tools/iter_timeline.py run.db [n_dispatches = 40]
"""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
dcols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else "0")
rows = list(c.execute(f"""select s.{name_col}, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                          on d.kernel_id = s.id order by d.start desc limit {n}"""))[::-1]
t0 = rows[0][1]
prev_end = t0
print(f"{'kernel':48s} {'queue':>6s} {'start_us':>10s} {'end_us':>10s} {'dur_us':>8s} {'gap_us':>8s}")
for name, st, en, q in rows:
    short = name.split("(")[0].replace("naruto::", "").replace("void ", "")[:48]
    print(f"{short:48s} {str(q):>6s} {(st - t0) / 1e3:10.2f} {(en - t0) / 1e3:10.2f} {(en - st) / 1e3:8.2f} {(st - prev_end) / 1e3:8.2f}")
    prev_end = max(prev_end, en)
