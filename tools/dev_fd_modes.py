"""Dev: cfg5 ms per factor update over the FD switches."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for cfg in sys.argv[1:] or ["default"]:
  env = dict(kv.split("=") for kv in cfg.split(",") if "=" in kv)
  os.environ.update(env)
  r = bench.fd_cfg5(dev)
  print(cfg, r["ms_per_factor_update"], "tails", [round(t, 3) for t in r.get("tail", r.get("tails", []))][:3] if isinstance(r.get("tail", r.get("tails", [])), list) else "", flush=True)
  for k in env:
    os.environ.pop(k)
