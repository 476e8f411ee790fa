"""dev (GPU): can an LDS / VALU-bound 1024-thread solver (eigh_small_kernel = the pivot solver of the
eigh path: 134 KB of LDS, one workgroup per CU) share CUs with an HBM-bound bf16-MFMA stream kernel that
needs little LDS (the fused FD filter step at b = 32: 17 KB)?  Times N launches of each alone and of
both on two streams.  If together ~ max(alone), a low-LDS update kernel could hide under the pivots."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_DEV_ENV"] = "1"
import torch  # noqa: E402

from precondition_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
# solver load: 1024 symmetric 128 x 128 matrices
g = torch.randn((1024, 128, 256), generator=gen, device=dev)
mats = list(torch.bmm(g, g.transpose(1, 2)).unbind(0))
# stream load: 8 x 4096^2 covariances, b = 32
factors, d, b = 8, 4096, 32
c16 = []
for _ in range(factors):
  c = torch.randn((d, d), generator=gen, device=dev)
  c16.append(K.to_bf16(c, split=True, tiled="frag"))
  del c
y = torch.randn((factors, d, b), generator=gen, device=dev)
y_prev = torch.randn((factors, d, b), generator=gen, device=dev)
z = torch.randn((factors, d, b), generator=gen, device=dev)
params = torch.tensor([[0.4, 0.5, 0.3, 12.0]] * factors, device=dev)
y1, out = torch.empty_like(y), torch.empty_like(y)
yt = K.fd_filter_step(z, y, None, y1, params, 1, frag=True)
nt = (torch.empty_like(yt[0]), torch.empty_like(yt[1]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def solver(n):
  for _ in range(n):
    K.eigh_batched(mats)


def stream(n):
  for _ in range(n):
    K.fd_cy_step(c16, yt, y, y_prev, out, nt, params, 3)


def timed(fn):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) * 1e3


solver(1); stream(5)
ta = timed(lambda: solver(3))
nstream = 60
tb = timed(lambda: stream(nstream))
print(f"solver alone: {ta:.2f} ms for 3 x 1024 problems;  stream alone: {tb:.2f} ms for {nstream} steps ({tb / nstream * 1e3:.1f} us each)", flush=True)
# size the stream work to the solver's duration
n2 = max(1, int(nstream * ta / tb))


def both():
  with torch.cuda.stream(s1):
    solver(3)
  with torch.cuda.stream(s2):
    stream(n2)


both()
tc = timed(both)
tb2 = timed(lambda: stream(n2))
print(f"together: {tc:.2f} ms  (solver alone {ta:.2f}, {n2} stream steps alone {tb2:.2f}; serial sum {ta + tb2:.2f})", flush=True)
