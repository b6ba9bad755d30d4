#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) run: per-kernel count / total / average duration, and -- when the run
collected PMC counters -- the per-dispatch average of each counter.

    python tools/prof_summary.py run_results.db [> profiles/xyz.txt]
"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
rows = list(c.execute(f"""select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.{name_col} order by 3 desc"""))
tot = sum(r[2] for r in rows)
print(f"{'kernel':96s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, cnt, s, a, mn, mx in rows[:48]:
    print(f"{n[:96]:96s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
# The same kernel is launched in more than one shape (e.g. k_query_fwd: 1024 workgroups over all samples in bench.py's kernel
# table, 512 = one wave per ray in the training forward): split the big ones by grid size.
dcols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
gcol = next((g for g in ("grid_size_x", "grid_x", "grid_size") if g in dcols), None)
wcol = next((w for w in ("workgroup_size_x", "workgroup_x", "workgroup_size") if w in dcols), None)
if gcol and wcol:
    q = f"""select s.{name_col}, d.{gcol} / max(d.{wcol}, 1), count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col}, d.{gcol} / max(d.{wcol}, 1) order by 1, 2"""
    by_shape = {}
    for n, g, cnt, a, mn, mx in c.execute(q):
        by_shape.setdefault(n, []).append((g, cnt, a, mn, mx))
    print("\nby launch shape (kernels launched in more than one grid size):")
    print(f"{'kernel':64s} {'workgroups':>10s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
    for n, cnt, s_, a, mn, mx in rows[:12]:
        if len(by_shape.get(n, [])) > 1:
            for g, k, av, mn2, mx2 in by_shape[n]:
                print(f"{n[:64]:64s} {g:10d} {k:7d} {av/1e3:10.2f} {mn2/1e3:9.2f} {mx2/1e3:9.2f}")
n_pmc = c.execute("select count(*) from rocpd_pmc_event").fetchone()[0]
if n_pmc:
    q = f"""select s.{name_col}, p.name, count(*), avg(e.value), sum(e.value)
            from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col}, p.name order by 5 desc"""
    print("\nPMC counters (per-dispatch average):")
    print(f"{'kernel':96s} {'counter':>16s} {'calls':>7s} {'avg':>16s}")
    for n, pn, cnt, av, sm in list(c.execute(q))[:40]:
        print(f"{n[:96]:96s} {pn:>16s} {cnt:7d} {av:16.1f}")
    if gcol and wcol:
        q = f"""select s.{name_col}, d.{gcol} / max(d.{wcol}, 1), p.name, count(*), avg(e.value)
                from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                join rocpd_kernel_dispatch d on d.event_id = e.event_id
                join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                group by s.{name_col}, d.{gcol} / max(d.{wcol}, 1), p.name order by 1, 2"""
        shaped = {}
        for n, g, pn, cnt, av in c.execute(q):
            shaped.setdefault(n, []).append((g, pn, cnt, av))
        print("\nPMC counters by launch shape (kernels launched in more than one grid size):")
        print(f"{'kernel':96s} {'workgroups':>10s} {'counter':>16s} {'calls':>7s} {'avg':>16s}")
        for n, rows_ in shaped.items():
            if len({g for g, *_ in rows_}) > 1 and "naruto" in n:
                for g, pn, cnt, av in rows_:
                    print(f"{n[:96]:96s} {g:10d} {pn:>16s} {cnt:7d} {av:16.1f}")

