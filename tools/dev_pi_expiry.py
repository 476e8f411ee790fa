import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K, _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def health():
  e, c, r = C.c_uint(), C.c_int(), C.c_int()
  L.ps_power_iteration_health(C.addressof(e), C.addressof(c), C.addressof(r)); return e.value, c.value, r.value
rng = np.random.default_rng(0)
mats = []
for i in range(160):
  g = torch.randn((512, 2048), device=dev); mats.append(g @ g.T)
torch.cuda.synchronize()
lam0, it0 = K.power_iteration_batched(mats); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
for cus, fill_ms, to in ((127, 300, "20"), (128, 300, "20"), (200, 300, "20"), (255, 300, "20"), (127, 300, "5000")):
  L.ps_power_iteration_reset_health()
  os.environ["PS_PI_TIMEOUT_MS"] = to
  torch.cuda.synchronize()
  L.ps_diag_spin(side.cuda_stream, cus, 1024, 150 * 1024, float(fill_ms))
  time.sleep(0.01)
  t0 = time.perf_counter()
  lam, it = K.power_iteration_batched(mats)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print("filler on %d CUs %d ms, timeout %s ms: PI call %.1f ms, NaN lambdas %d, equal %s, health %s" % (
      cus, fill_ms, to, dt * 1e3, int(torch.isnan(lam).sum()), bool(torch.equal(lam, lam0)), health()), flush=True)
