import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K, _lib
L = _lib.lib()
dev = torch.device("cuda:0")
mats = []
for i in range(160):
  g = torch.randn((512, 2048), device=dev); mats.append(g @ g.T)
torch.cuda.synchronize()
lam0, it0 = K.power_iteration_batched(mats); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
for nb in (4, 32, 64, 100, 160):
  for cus, res in ((127, "1"), (127, "0"), (16, "1"), (250, "1")):
    os.environ["PS_PI_RESIDENT"] = res
    torch.cuda.synchronize()
    L.ps_diag_spin(side.cuda_stream, cus, 1024, 150 * 1024, 300.0)
    time.sleep(0.01)
    t0 = time.perf_counter()
    lam, it = K.power_iteration_batched(mats[:nb])
    torch.cuda.current_stream().synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("blocks %3d filler on %3d CUs resident=%s: PI call %.1f ms" % (nb, cus, res, dt * 1e3), flush=True)
