import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K
from oracle import shampoo_oracle as orc
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests/golden/eigh_root.npz")); idx = json.load(open(os.path.join(ROOT, "tests/golden/eigh_root_index.json")))
for c in idx:
  a = z[c["name"]+"__a"]; ref = z[c["name"]+"__root"]
  ps = None if c["padding_start"] is None else [c["padding_start"]]
  r, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [c["p"]], ps, eigh=True)
  torch.cuda.synchronize()
  h = r[0].cpu().numpy(); m = m.cpu().numpy()[0]
  with np.errstate(all="ignore"):
    print(c["name"], "rel", np.linalg.norm(h-ref)/np.linalg.norm(ref), "err", m[0], "gold err", float(z[c["name"]+"__err"]), "sweeps", m[5], "pi", m[6], "finite", np.isfinite(h).all())
# small stats like in e2e: eps*I + rank-few
rng = np.random.default_rng(0)
for n, k in ((32, 3), (24, 2), (6, 1), (1, 1), (28, 40)):
  g = rng.standard_normal((n, k)).astype(np.float32)
  a = (np.float32(1e-6)*np.eye(n, dtype=np.float32) + np.float32(0.001) * (g @ g.T)).astype(np.float32)
  r, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [2], eigh=True)
  h = r[0].cpu().numpy(); href, mref = orc.matrix_inverse_pth_root_eigh(a, 2)
  print("lowrank", n, k, "rel", np.linalg.norm(h-href)/np.linalg.norm(href), m.cpu().numpy()[0][[0,5]], mref["inverse_pth_root_errors"])
