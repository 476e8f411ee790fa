"""Dev yardstick: the vendor float32 GEMM (torch.bmm / torch.mm -> rocBLAS / hipBLASLt, float32
'highest' precision = the fp32 MFMA path) on the product shapes of the bench workloads, next to
this library's exact-f32 core (ps_gemm via K.gemm) on the same shapes."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cuda.matmul.allow_tf32 = False
torch.set_float32_matmul_precision("highest")
dev = torch.device("cuda:0")
def t(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for b, n in ((256, 512), (64, 1024), (16, 2048), (1, 8192)):
  a = torch.randn(b, n, n, device=dev); c = torch.randn(b, n, n, device=dev); o = torch.empty_like(a)
  dt = t(lambda: torch.bmm(a, c, out=o))
  print("torch.bmm  %4d x %5d^3: %.3f ms  %.1f TFLOP/s (%.2f of 157.3)" % (b, n, dt * 1e3, 2 * b * n ** 3 / dt / 1e12, 2 * b * n ** 3 / dt / 157.3e12), flush=True)
  del a, c, o
from precondition_amd import kernels as K
for b, n in ((256, 512), (64, 1024), (16, 2048)):
  a = [torch.randn(n, n, device=dev) for _ in range(b)]; c = [torch.randn(n, n, device=dev) for _ in range(b)]
  o = [torch.empty(n, n, device=dev) for _ in range(b)]
  for tb in (False, True):
    items = [(x, y, z, False, tb) for x, y, z in zip(a, c, o)]
    dt = t(lambda: K.gemm_grouped(items))
    print("ps_gemm_grouped_f32 (full products, transb=%d) %4d x %5d^3: %.3f ms  %.1f TFLOP/s (%.2f of 157.3)" % (
        tb, b, n, dt * 1e3, 2 * b * n ** 3 / dt / 1e12, 2 * b * n ** 3 / dt / 157.3e12), flush=True)
  del a, c, o
