#!/usr/bin/env python3
"""Durations of one kernel in dispatch order from a rocprofv3 rocpd database (is a kernel's time drifting over a run?).
    python tools/kernel_series.py run_results.db <substring of the kernel name> [every]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
rows = list(c.execute(f"""select d.start, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        where s.{name_col} like ? order by d.start""", ("%" + sys.argv[2] + "%",)))
every = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, len(rows) // 40)
print(f"{len(rows)} dispatches of *{sys.argv[2]}*; every {every}th (us):")
print(" ".join(f"{d / 1e3:.1f}" for _, d in rows[::every]))
