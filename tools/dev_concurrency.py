import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K, _lib
L = _lib.lib()
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
x = torch.randn(1 << 20, device=dev); torch.cuda.synchronize()
def trial(name, fn, cus=127, threads=1024, lds=150 * 1024):
  torch.cuda.synchronize()
  L.ps_diag_spin(side.cuda_stream, cus, threads, lds, 200.0)
  time.sleep(0.01)
  t0 = time.perf_counter(); fn(); torch.cuda.synchronize(torch.cuda.current_stream()) if False else torch.cuda.current_stream().synchronize()
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  print("%-50s %.2f ms" % (name, dt * 1e3), flush=True)
trial("tiny torch kernel", lambda: x.add_(1.0))
trial("torch H2D pageable", lambda: torch.tensor([1, 2, 3], device=dev))
trial("torch.empty + fill", lambda: torch.empty(1 << 20, device=dev).zero_())
main = torch.cuda.Stream(device=dev)
def on_main(fn):
  def f():
    with torch.cuda.stream(main):
      fn()
    main.synchronize()
  return f
trial("tiny torch kernel on a non-default stream", on_main(lambda: x.add_(1.0)))
trial("tiny kernel, filler 127 x 256 thr x 150KB", lambda: x.add_(1.0), threads=256)
trial("tiny kernel, filler 127 x 1024 thr x 64KB", lambda: x.add_(1.0), lds=64 * 1024)
trial("tiny kernel, filler 16 x 1024 thr x 150KB", lambda: x.add_(1.0), cus=16)
g = torch.randn((512, 2048), device=dev); a = g @ g.T; torch.cuda.synchronize()
trial("power iteration of one 512 block (default stream)", lambda: K.power_iteration_batched([a]))
trial("power iteration of one 512 block (other stream)", on_main(lambda: K.power_iteration_batched([a])))
