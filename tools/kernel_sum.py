import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[re.sub(r"\(.*", "", r["Kernel_Name"])[:44]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
att = [k for k in agg if "decode_attn_kernel" in k][0]
steps = len(agg[att]) / 32
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= steps * 0.9:
        per = sum(v) / steps / 1e3
        tot += per
        if per > 20: print(f"{k:44s} calls/step {len(v)/steps:6.1f} us/step {per:9.1f}")
print("sum of in-step kernel time per step (us):", round(tot, 1), "steps", steps)
# wall span per step: from first to last kernel of the last 20 attention-delimited steps
ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
print("trace span / steps (us):", round((ts[-1][1] - ts[0][0]) / 1e3 / steps, 1))
