#!/usr/bin/env python
"""Per-kernel totals out of a rocprofv3 rocpd database (what `--stats` prints when the csv writer is not selected):
python tools/rocpd_stats.py gpurun_out/vae_prof/vae_results.db [csv_out]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, n, t, avg, mn, mx in rows:
    lines.append(f'"{name}",{n},{t},{avg:.1f},{100.0 * t / tot:.2f},{mn},{mx}')
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
for ln in lines[:20]:
    print(ln[:200])
