import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
factors = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print(json.dumps(bench.fd_cfg5(torch.device("cuda:0"), factors=factors)["ms_per_factor_update"]))
