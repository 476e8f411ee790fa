"""Dev (round 5): leading dimensions.  Inputs and outputs of the batched roots as VIEWS into larger buffers (lda, ldo > n)
must give the bits of the contiguous call, and must not touch the buffer outside the view."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for eigh in (False, True):
  for rnd in range(6):
    mats, views, outs_v, holders, ps, pads = [], [], [], [], [], []
    for _ in range(6):
      n = int(rng.choice([1, 5, 64, 100, 129, 200, 257, 300]))
      g = rng.standard_normal((n, 2 * n + 3)); a = torch.tensor((g @ g.T).astype(np.float32), device=dev)
      ld = n + int(rng.choice([0, 1, 3, 64]))
      big = torch.full((n + 2, ld + 5), 7.0, device=dev); big[1:n + 1, 2:n + 2] = a
      ob = torch.full((n + 3, ld + 4), -3.0, device=dev)
      mats.append(a); views.append(big[1:n + 1, 2:n + 2]); outs_v.append(ob[2:n + 2, 1:n + 1]); holders.append((big, ob))
      ps.append(int(rng.choice([2, 4]))); pads.append(n if rng.random() < 0.7 else int(rng.integers(0, n + 1)))
    r0, m0 = K.matrix_inverse_pth_root_batched(mats, ps, pads, eigh=eigh)
    try:
      r1, m1 = K.matrix_inverse_pth_root_batched(views, ps, pads, eigh=eigh, out=outs_v)
    except Exception as ex:
      print(f"eigh={eigh} round {rnd}: views refused: {type(ex).__name__}: {str(ex)[:120]}"); continue
    for i, (a, b) in enumerate(zip(r0, outs_v)):
      n = a.shape[0]
      big, ob = holders[i]
      same = torch.equal(a, b)
      frame = ob.clone(); frame[2:n + 2, 1:n + 1] = -3.0
      untouched = bool((frame == -3.0).all()) and bool((big[0] == 7.0).all()) and bool((big[:, :2] == 7.0).all())
      if not (same and untouched):
        bad += 1
        print(f"eigh={eigh} round {rnd} block {i}: n={n} pad={pads[i]} same_bits={same} frame_untouched={untouched}", flush=True)
    if not torch.equal(m0, m1):
      d = (m0 != m1) & ~(torch.isnan(m0) & torch.isnan(m1))
      if bool(d.any()):
        bad += 1; print(f"eigh={eigh} round {rnd}: metrics differ", flush=True)
print("stride fuzz mismatches", bad)
