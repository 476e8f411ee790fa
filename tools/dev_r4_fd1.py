import os; os.environ.setdefault("PS_DEV_ENV", "1")
import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
r = bench.fd_cfg5(torch.device("cuda:0"), factors=1, updates=4)
print(json.dumps(r["ms_per_factor_update"]))
r = bench.fd_cfg5(torch.device("cuda:0"), factors=8, updates=4)
print(json.dumps(r["ms_per_factor_update"]), json.dumps(r["roofline"]))
