#!/usr/bin/env python3
"""Per-kernel summary of the passes written by tools/prof_cfg2.sh.
Usage: summarize_pmc2.py <dir> [out.json]"""
import csv, glob, json, os, sys, statistics
from collections import defaultdict
d = sys.argv[1]
def rows(sub, suffix):
  out = []
  for p in glob.glob(os.path.join(d, sub, "**", "*" + suffix), recursive=True):
    with open(p, newline="") as f:
      out.extend(csv.DictReader(f))
  return out
def short(n): return n.replace("void ", "").split("(")[0]
res = defaultdict(dict)
for r in rows("trace", "kernel_stats.csv"):
  k = short(r["Name"])
  res[k].update(calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
def pmc(sub):
  acc = defaultdict(lambda: defaultdict(list))
  for r in rows(sub, "counter_collection.csv"):
    k = short(r["Kernel_Name"])
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[k][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
  return acc
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_GRBM_GUI_ACTIVE_SQ_VALU_MFMA_BUSY_CYCLES", "pmc_TCC_HIT_sum_TCC_MISS_sum"):
  for k, cs in pmc(sub).items():
    for c, vals in cs.items():
      res[k][c + "_avg"] = sum(v for v, _ in vals) / len(vals)
      res[k][c + "_n"] = len(vals)
      if c == "GRBM_GUI_ACTIVE":
        res[k]["clock_GHz"] = statistics.median(v / 8 / dur for v, dur in vals)
for k, r in res.items():
  if "FETCH_SIZE_avg" in r:
    r["hbm_MB_per_launch_corrected"] = (2 * r["FETCH_SIZE_avg"] + r.get("WRITE_SIZE_avg", 0)) / 1024
  if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in r and "GRBM_GUI_ACTIVE_avg" in r:
    r["mfma_busy_frac"] = r["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (r["GRBM_GUI_ACTIVE_avg"] / 8 * 1024)
  if "TCC_HIT_sum_avg" in r:
    r["l2_hit_rate"] = r["TCC_HIT_sum_avg"] / max(r["TCC_HIT_sum_avg"] + r["TCC_MISS_sum_avg"], 1)
out = {k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("total_ms", 0)) if k.startswith("psk::")}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
  json.dump(out, open(sys.argv[2], "w"), indent=1)
