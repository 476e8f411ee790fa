"""dev (GPU): what the first FD update of a process pays for: kernel loading (small dummy update first)
vs allocation of the big buffers (full-size dummy first)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from precondition_amd import low_rank
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
def upd(d, rank, factors, seed):
  gen = torch.Generator(device=dev).manual_seed(seed)
  prevs = [torch.zeros((d, rank + 2), dtype=torch.float32, device=dev) for _ in range(factors)]
  grads = [torch.randn((d, d), generator=gen, device=dev) for _ in range(factors)]
  torch.cuda.synchronize(); t0 = time.perf_counter()
  calls = [dict(new_grad=low_rank.gram_of_block(grads[f], 0), p=4, rank=rank, ridge_epsilon=1e-6, decay=0.999,
                padding_start=d, prev=prevs[f], new_grad_is_gram=True) for f in range(factors)]
  out = low_rank._fd_update_root_batched(calls)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) * 1e3
torch.zeros(1, device=dev); torch.cuda.synchronize()
if mode == "small":
  print("small dummy (d=1024, rank 8, 1 factor):", round(upd(1024, 8, 1, 1), 1), "ms")
elif mode == "full":
  print("full-size dummy:", round(upd(4096, 64, 8, 1), 1), "ms")
  torch.cuda.synchronize()
print(mode, "-> first timed update (8 x 4096, rank 64):", round(upd(4096, 64, 8, 2), 1), "ms; second:", round(upd(4096, 64, 8, 3), 1))
