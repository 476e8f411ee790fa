#!/usr/bin/env python3
"""Aggregate three separate rocprofv3 --pmc passes into profiles/rNN_pmc_<workload>_summary.json.

Usage: summarize_pmc.py FETCH.csv WRITE.csv CLOCK_MFMA.csv OUT.json [code-description]

The three inputs are the *_counter_collection.csv files of
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ...
of `python3 bench.py --no-cpu-baseline --no-headline --no-extras --steps 2 --warmup 1`.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE (KB) reports half of wide coalesced reads, so
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import statistics
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def load(path):
    rows = defaultdict(lambda: defaultdict(list))    # kernel -> counter -> [(value, dur_ns, grid)]
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if not k.startswith("psk::"):
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            rows[k][r["Counter_Name"]].append((float(r["Counter_Value"]), dur, int(r["Grid_Size"])))
    return rows


def main():
    fetch, write, clk = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    out_path = sys.argv[4]
    code = sys.argv[5] if len(sys.argv) > 5 else ""
    kernels = []
    for k in sorted(fetch):
        fv = [v for v, _, _ in fetch[k]["FETCH_SIZE"]]
        wv = [v for v, _, _ in write.get(k, {}).get("WRITE_SIZE", [])] or [0.0]
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        kernels.append({"kernel": k, "launches": len(fv), "FETCH_SIZE_KB_avg": round(fa, 1),
                        "WRITE_SIZE_KB_avg": round(wa, 1),
                        "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)})
    summary = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc GRBM_GUI_ACTIVE "
                   "SQ_VALU_MFMA_BUSY_CYCLES (three separate passes) --output-format csv -- python3 bench.py "
                   "--no-cpu-baseline --no-headline --no-extras --steps 2 --warmup 1",
        "workload": "cfg2_256x512_p4",
        "code": code,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section) => "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
        "kernels": kernels,
    }
    stage = [k for k in fetch if k.startswith("psk::newton_stage_kernel")]
    if stage:
        s = stage[0]
        by_grid = defaultdict(lambda: {"f": [], "w": []})
        for v, _, g in fetch[s]["FETCH_SIZE"]:
            by_grid[g]["f"].append(v)
        for v, _, g in write[s]["WRITE_SIZE"]:
            by_grid[g]["w"].append(v)
        summary["newton_stage_kernel_by_launch_size"] = {
            "%d_tiles" % (g // 256): {"launches": len(d["f"]),
                                      "fetch_MB_x2": round(2 * statistics.mean(d["f"]) / 1024, 1),
                                      "write_MB": round(statistics.mean(d["w"]) / 1024, 1) if d["w"] else None}
            for g, d in sorted(by_grid.items(), reverse=True)}
        act = clk[s]["GRBM_GUI_ACTIVE"]
        busy = clk[s]["SQ_VALU_MFMA_BUSY_CYCLES"]
        big = max(g for _, _, g in act)
        ghz = [v / 8 / d for v, d, g in act if g == big]
        frac = [b[0] / (a[0] / 8 * 1024) for a, b in zip(act, busy) if a[2] == big]
        summary["newton_stage_kernel_clock_GHz_median"] = round(statistics.median(ghz), 3)
        summary["newton_stage_kernel_mfma_busy_fraction_median"] = round(statistics.median(frac), 3)
    summary["notes"] = ("MFMA-busy is SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clock = "
                        "GRBM_GUI_ACTIVE/8/duration, both over the largest stage launches (profiled runs clock lower "
                        "than unprofiled ones).")
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
