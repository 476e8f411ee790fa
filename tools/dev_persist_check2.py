import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
from tests.test_gpu_round2 import _mixed_batch
dev = torch.device("cuda:0")
mats, ps, pads = _mixed_batch()
def run(idx, mode):
  os.environ["PS_NEWTON_PERSISTENT"] = mode
  md = [torch.tensor(mats[i], device=dev) for i in idx]
  r, m = K.matrix_inverse_pth_root_batched(md, [ps[i] for i in idx], [pads[i] for i in idx])
  torch.cuda.synchronize()
  return [x.cpu().numpy() for x in r], m.cpu().numpy()
for idx in ([6], [6, 6], [5, 6], list(range(13))):
  a = run(idx, "1"); b = run(idx, "0")
  k = idx.index(6)
  print(idx, "equal H:", np.array_equal(a[0][k], b[0][k]), "maxdiff", np.abs(a[0][k]-b[0][k]).max(),
        "metrics", a[1][k].view(np.uint32), b[1][k].view(np.uint32))
# same-mode determinism of block 6 alone with different garbage in the allocator
x = torch.full((64 << 20,), float("nan"), device=dev); del x
a = run([6], "1"); x = torch.full((64 << 20,), 1e30, device=dev); del x; b = run([6], "1")
print("persistent, NaN vs 1e30 garbage:", np.array_equal(a[0][0], b[0][0]))
x = torch.full((64 << 20,), float("nan"), device=dev); del x
a = run([6], "0"); x = torch.full((64 << 20,), 1e30, device=dev); del x; b = run([6], "0")
print("staged, NaN vs 1e30 garbage:", np.array_equal(a[0][0], b[0][0]))
