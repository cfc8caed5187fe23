#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, avg/min us, share.
usage: prof_summary.py <kernel_trace.csv> [top_n]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    grid = f'{r.get("Grid_Size_X", r.get("Grid_Size", "?"))}x{r.get("Grid_Size_Y", "")}x{r.get("Grid_Size_Z", "")}'
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    agg[(name[:60], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':60s} {'grid':>18s} {'wg':>5s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'share':>7s}")
for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{name:60s} {grid:>18s} {wg:>5s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {100*sum(v)/tot:6.2f}%")
