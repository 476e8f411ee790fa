import sys, os, json, torch
sys.path.insert(0, "/root/repo")
import bench
r = bench.fd_cfg5(torch.device("cuda:0"))
print(json.dumps(r.get("roofline"), indent=0)[:900]); print(r["ms_per_factor_update"])
