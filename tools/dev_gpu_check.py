"""Dev-only quick GPU check (run through gpurun). Not part of the test suite."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K

dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests/golden/newton_root.npz"))
idx = json.load(open(os.path.join(ROOT, "tests/golden/newton_root_index.json")))

# 1. plain gemm
rng = np.random.default_rng(0)
for (m, n, k) in [(128, 128, 128), (200, 130, 77), (512, 512, 512), (1000, 1000, 1000)]:
  a = rng.standard_normal((m, k)).astype(np.float32); b = rng.standard_normal((k, n)).astype(np.float32)
  c = K.matmul(torch.tensor(a, device=dev), torch.tensor(b, device=dev)).cpu().numpy()
  ref = a.astype(np.float64) @ b.astype(np.float64)
  print("gemm", m, n, k, "relerr", np.abs(c - ref).max() / np.abs(ref).max())

# 2. newton vs golden
for c in idx:
  if not c["full"]:
    continue
  a = g[c["name"] + "__a"]; root = g[c["name"] + "__root"]; mv = g[c["name"] + "__metrics"]
  ps = None if c["padding_start"] is None else [c["padding_start"]]
  t0 = time.time()
  outs, met = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [c["p"]], ps,
      ridge_epsilon=c["ridge"], relative_matrix_epsilon=c["rel"])
  torch.cuda.synchronize()
  h = outs[0].cpu().numpy(); m = met.cpu().numpy()[0]
  with np.errstate(all="ignore"):
    rel = np.linalg.norm(h - root) / np.linalg.norm(root)
  print(f"{c['name']:38s} relfro={rel:.2e} gold[err,it,ratio,ev,tries]={mv} got={m[:7]} {time.time()-t0:.3f}s")
