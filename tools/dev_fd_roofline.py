"""Dev: cfg5 (FD branch) per-factor-update time and the HBM roofline of its C @ Y product, with the
row-major (PS_FD_TILED=0) and the tile-blocked (default) bf16 covariance."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for mode in ("1", "0", "1", "0"):
  os.environ["PS_FD_TILED"] = mode
  r = bench.fd_cfg5(torch.device("cuda:0"))
  ro = r.get("roofline") or {}
  print("tiled", mode, r["ms_per_factor_update"], "product ms", ro.get("ms_per_product_all_factors"),
        "HBM frac", ro.get("frac"), "tail", r.get("tail_after_updates"), flush=True)
