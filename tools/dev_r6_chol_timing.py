"""Dev (round 6): chol_rinv_kernel by block size (is a column step's cost fixed or proportional to its work?) and a
checksum of its output bits on fixed inputs (to compare kernel variants: same fma sequence per element = same bits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(11)
for b in (32, 64, 96):
  for B in (1, 8):
    x = torch.randn((B, 4 * b, b), generator=gen)
    g = (x.transpose(1, 2) @ x).to(dev).contiguous()
    out = K.chol_rinv_batched(g)
    for _ in range(5): K.chol_rinv_batched(g, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): K.chol_rinv_batched(g, out=out)
    torch.cuda.synchronize()
    bits = out.cpu().numpy().view(np.uint32).astype(np.uint64)
    print("b %3d  B %d  %.1f us per call   checksum %d" % (b, B, (time.perf_counter() - t0) / 50 * 1e6, int(bits.sum())), flush=True)
