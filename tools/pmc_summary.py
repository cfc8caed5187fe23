#!/usr/bin/env python3
"""Summarise the three rocprofv3 passes of tools/pmc_kernels.sh per (kernel, grid).
usage: pmc_summary.py <out-name> <kernel-substring[,substring...]> <command description>"""
import collections
import csv
import glob
import re
import sys

out, filt, what = sys.argv[1], sys.argv[2].split(","), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))    # (kernel, grid) -> counter -> sum over dispatches
calls = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:72]


for p in "ABC":
    cc = glob.glob(f"gpurun_out/{out}.{p}/**/*counter_collection.csv", recursive=True)
    if not cc:
        continue
    for r in csv.DictReader(open(cc[0])):
        if not any(f in r["Kernel_Name"] for f in filt):
            continue
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[key][r["Counter_Name"]] += 1
    if p == "A":
        kt = glob.glob(f"gpurun_out/{out}.{p}/**/*kernel_trace.csv", recursive=True)
        for r in csv.DictReader(open(kt[0])) if kt else []:
            if any(f in r["Kernel_Name"] for f in filt):
                grid = r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
                dur[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

print(f"# {what}")
print("# per dispatch averages; SQ_* wait/active counters as a fraction of SQ_WAVE_CYCLES; FETCH_SIZE doubled (gfx950)")
for key in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    c, n = agg[key], calls[key]
    avg = {k: v / max(n[k], 1) for k, v in c.items()}
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    d = dur.get(key) or []
    print(f"\n{key[0]}  grid={key[1]}  dispatches={n.get('SQ_WAVE_CYCLES', 0)}"
          + (f"  avg_us(under pmc)={sum(d) / len(d) / 1e3:.2f} min_us={min(d) / 1e3:.2f}" if d else ""))
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS"):
            if k in avg:
                print(f"  {k:28s} {avg[k] / wc:8.3f} of wave cycles")
    if "SQ_LDS_IDX_ACTIVE" in avg:
        print(f"  {'SQ_LDS_BANK_CONFLICT':28s} {avg.get('SQ_LDS_BANK_CONFLICT', 0) / max(avg['SQ_LDS_IDX_ACTIVE'], 1):8.3f} of LDS-array cycles"
              f"   (LDS active {avg['SQ_LDS_IDX_ACTIVE']:.3e} cycles)")
    if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
        print(f"  {'MFMA utilisation':28s} {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cyc * 1024, 1):8.3f}   "
              f"(SQ_VALU_MFMA_BUSY_CYCLES {avg['SQ_VALU_MFMA_BUSY_CYCLES']:.3e} / (GRBM_GUI_ACTIVE/8 = {cyc:.3e} x 1024 SIMDs))")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if k in avg:
            print(f"  {k:28s} {avg[k]:12.4e}")
    if "FETCH_SIZE" in avg:
        print(f"  {'HBM/fabric read bytes':28s} {2 * avg['FETCH_SIZE'] * 1024:12.4e}   (2 x FETCH_SIZE KB)")
