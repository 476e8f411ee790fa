"""dev (GPU): the fused filter step (ps_fd_cy_step_f32) alone on the cfg5 shape, for a library variant
given by PS_AB_LIB (tools/ab_build.sh).  Event-timed here; run under rocprofv3 for kernel-only times
(tools/dev_r4_cy_prof.sh).  Usage: [PS_AB_LIB=...] python tools/dev_r4_cystep.py [factors] [d] [b]"""
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
c16 = []
for _ in range(factors):
  c = torch.randn((d, d), generator=gen, device=dev)
  c16.append(K.to_bf16(c, split=True, tiled="frag"))
  del c
y = torch.randn((factors, d, b), generator=gen, device=dev)
y_prev = torch.randn((factors, d, b), generator=gen, device=dev)
z = torch.randn((factors, d, b), generator=gen, device=dev)
params = torch.tensor([[0.4, 0.5, 0.3, 12.0]] * factors, device=dev)
y1 = torch.empty_like(y)
yt = K.fd_filter_step(z, y, None, y1, params, 1, frag=True)
nt = (torch.empty_like(yt[0]), torch.empty_like(yt[1]))
out = torch.empty_like(y)
for _ in range(5):
  K.fd_cy_step(c16, yt, y, y_prev, out, nt, params, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
  e0.record()
  for _ in range(20):
    K.fd_cy_step(c16, yt, y, y_prev, out, nt, params, 3)
  e1.record(); torch.cuda.synchronize()
  best = min(best, e0.elapsed_time(e1) / 20)
nbytes = factors * (4.0 * d * d + 4.0 * d * b * 4 + 4.0 * d * b * 2)   # C hi+lo; y, y_prev, y_next, planes in + out
print(f"{os.environ.get('PS_AB_LIB', 'in-tree'):44s} {best * 1e3:8.1f} us  {nbytes / best / 1e9:7.2f} TB/s", flush=True)
