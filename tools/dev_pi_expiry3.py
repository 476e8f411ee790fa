import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K, _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def health():
  e, c, r = C.c_uint(), C.c_int(), C.c_int()
  L.ps_power_iteration_health(C.addressof(e), C.addressof(c), C.addressof(r)); return e.value, c.value, r.value
mats = []
for i in range(160):
  g = torch.randn((512, 2048), device=dev); mats.append(g @ g.T)
torch.cuda.synchronize()
lam0, it0 = K.power_iteration_batched(mats); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
os.environ["PS_PI_TIMEOUT_MS"] = "20"
for cus in (250, 248, 255, 200):
  L.ps_power_iteration_reset_health()
  torch.cuda.synchronize()
  L.ps_diag_spin(side.cuda_stream, cus, 1024, 150 * 1024, 300.0)
  time.sleep(0.01)
  t0 = time.perf_counter()
  lam, it = K.power_iteration_batched(mats)
  torch.cuda.current_stream().synchronize()
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  print("standalone PI, filler %d CUs: %.1f ms, NaN %d, equal %s, iters min %d, health %s" % (
      cus, dt * 1e3, int(torch.isnan(lam).sum()), bool(torch.equal(lam, lam0)), int(it.min()), health()), flush=True)
L.ps_power_iteration_reset_health()
torch.cuda.synchronize()
L.ps_diag_spin(side.cuda_stream, 250, 1024, 150 * 1024, 300.0)
time.sleep(0.01)
t0 = time.perf_counter()
roots, met = K.matrix_inverse_pth_root_batched(mats, [4] * len(mats))
torch.cuda.current_stream().synchronize()
print("newton: %.1f ms health %s nan %d" % ((time.perf_counter() - t0) * 1e3, health(), int(torch.isnan(met[:, :5]).sum())))
