"""dev (GPU): the C @ Y product of the FD filter (8 x 4096^2 covariances, hi/lo bf16, b = 96) alone,
for a library variant given by PS_AB_LIB (tools/ab_build.sh): milliseconds per product and TB/s on
the algorithmic bytes.  B operands are per-factor [128, d] buffers so that timing-only variants that
read them with another layout stay inside their allocation.
Usage: PS_AB_LIB=.ab/x/libprecondition_amd.so python tools/dev_r4_cy.py [factors] [d] [b]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_DEV_ENV"] = "1"
import torch  # noqa: E402

from precondition_amd import _lib  # noqa: E402

if os.environ.get("PS_AB_LIB"):
  _lib.LIB_PATH = os.path.abspath(os.environ["PS_AB_LIB"])
from precondition_amd import kernels as K  # noqa: E402

factors = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(5)
cs = [torch.randn((d, d), generator=gen, device=dev) for _ in range(factors)]
c16 = [K.to_bf16(c, split=True, tiled=True) for c in cs]
z = torch.empty((factors, d, b), device=dev)
items = []
keep = []
for j in range(factors):
  y = torch.randn((d, b), generator=gen, device=dev)
  hi = torch.zeros((128, d), dtype=torch.bfloat16, device=dev)
  lo = torch.zeros((128, d), dtype=torch.bfloat16, device=dev)
  h, l = K.to_bf16(y, split=True, transpose=True)
  hi[:b].copy_(h); lo[:b].copy_(l)
  keep.append((hi, lo))
  items.append((c16[j], (hi[:b], lo[:b]), z[j]))
from precondition_amd._lib import GemmBf16Desc  # noqa: E402
L = _lib.lib()
descs = (GemmBf16Desc * factors)()
for dsc, (a, (b_hi, b_lo), c) in zip(descs, items):
  dsc.a_hi, dsc.a_lo, dsc.b_hi, dsc.b_lo = a.hi.data_ptr(), a.lo.data_ptr(), b_hi.data_ptr(), b_lo.data_ptr()
  dsc.c, dsc.m, dsc.n, dsc.k = c.data_ptr(), d, b, d
  dsc.lda, dsc.ldb, dsc.ldc, dsc.a_tiled = d, d, b, 1
ws = torch.empty((L.ps_gemm_bf16_grouped_workspace_bytes(descs, factors),), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def product():
  # the raw entry point with prebuilt descriptors: ~30 us of host time per call (the Python wrapper
  # takes ~130 us, which hid every variant faster than that)
  rc = L.ps_gemm_bf16_grouped(stream, descs, factors, ws.data_ptr(), ws.numel())
  assert rc == 0, rc


for _ in range(5):
  product()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
best = 1e9
t0 = time.perf_counter()
for _ in range(20):
  product()
host_us = (time.perf_counter() - t0) / 20 * 1e6
torch.cuda.synchronize()
for rep in range(3):
  e0.record()
  for _ in range(20):
    product()
  e1.record(); torch.cuda.synchronize()
  best = min(best, e0.elapsed_time(e1) / 20)
nbytes = factors * (4.0 * d * d + 8.0 * d * b)
ref = (cs[0] @ (keep[0][0][:b].float() + keep[0][1][:b].float()).T)
err = float((z[0] - ref).norm() / ref.norm())
print(f"{os.environ.get('PS_AB_LIB', 'in-tree'):44s} {best * 1e3:8.1f} us  {nbytes / best / 1e9:7.2f} TB/s  rel err vs torch {err:.2e}  host {host_us:.0f} us/call", flush=True)
