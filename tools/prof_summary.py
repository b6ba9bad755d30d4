#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel count / total / average duration."""
import sqlite3, sys
db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
q = f"""select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.{name_col} order by 3 desc"""
rows = list(c.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, cnt, s, a, mn, mx in rows[:45]:
    print(f"{n[:90]:90s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
