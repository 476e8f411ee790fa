"""dev: persistent vs staged Newton execution vs the oracle on the mixed batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
from tests.test_gpu_round2 import _mixed_batch

dev = torch.device("cuda:0")
mats, ps, pads = _mixed_batch()
mats_d = [torch.tensor(m, device=dev) for m in mats]
res = {}
for rep in range(3):
  for mode in ("1", "0"):
    os.environ["PS_NEWTON_PERSISTENT"] = mode
    roots, met = K.matrix_inverse_pth_root_batched(mats_d, ps, pads)
    torch.cuda.synchronize()
    res[(mode, rep)] = ([r.cpu().numpy() for r in roots], met.cpu().numpy())
for i, (a, p) in enumerate(zip(mats, ps)):
  h_ref, m_ref = orc.matrix_inverse_pth_root(a, p, padding_start=pads[i])
  line = f"blk {i:2d} n={a.shape[0]:4d} pad={pads[i]:4d} p={p} ref it={m_ref['inverse_pth_root_iters']} tr={m_ref['total_retries']} |"
  for mode in ("1", "0"):
    for rep in range(3):
      h, m = res[(mode, rep)]
      nr = np.linalg.norm(h_ref)
      rel = np.linalg.norm(h[i] - h_ref) / nr if nr > 0 else float(np.abs(h[i]).max())
      line += f" m{mode}r{rep}: rel={rel:.1e} it={m[i,1]:.0f} tr={m[i,4]:.0f} err={m[i,0]:.1e}"
    line += " |"
  print(line)
