"""stdin: tools/scatter_timeline.py's table -> one line per workgroup TYPE (uncertainty grid / dense levels / hashed levels): workgroups, mean and max end."""
import sys, re
rows = {"uncertainty grid": [], "dense level units": [], "hashed level units": []}
head = None
for line in sys.stdin:
    if head is None and ":" in line and "scatter workgroups" in line:
        head = line.strip(); continue
    m = re.match(r"(uncertainty grid|unit (\d+))\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)", line)
    if not m:
        continue
    n, points_mean, end_max = int(m.group(3)), float(m.group(6)), float(m.group(9))
    if m.group(2) is None: key = "uncertainty grid"
    else: key = "dense level units" if int(m.group(2)) < 16 else "hashed level units"      # units 0..15: levels 0..4 (one chunk, two features... see the table)
    rows[key].append((n, points_mean, end_max))
print(head)
for k, v in rows.items():
    if v:
        wg = sum(r[0] for r in v)
        print(f"  {k:20s} workgroups {wg:4d}   points phase ends (mean over units) {sum(r[1] for r in v) / len(v):8.2f} us   last workgroup ends {max(r[2] for r in v):8.2f} us")
