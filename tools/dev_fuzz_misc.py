"""Dev: fuzz the eigh root, the N-d statistics update and quantize against the oracles."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
from precondition_amd.blocking import BlockPartitioner
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
bad = 0
# ---- eigh root
for rnd in range(6):
  mats, ps, pads = [], [], []
  for _ in range(16):
    n = int(rng.choice([1, 2, 5, 31, 64, 65, 100, 128, 129, 200, 260]))
    p = int(rng.choice([1, 2, 4, 6, 8]))
    kind = rng.integers(0, 3)
    if kind == 0:
      g = rng.standard_normal((n, 2 * n + 1)); a = g @ g.T
    elif kind == 1:
      g = rng.standard_normal((n, max(1, n // 4))); a = g @ g.T
    else:
      q, _ = np.linalg.qr(rng.standard_normal((n, n))); a = (q * (10.0 ** rng.uniform(-3, 1, n))) @ q.T
    a = ((a + a.T) / 2 * 10.0 ** rng.uniform(-2, 2)).astype(np.float32)
    full = n + int(rng.choice([0, 0, 3, 40]))
    m = np.zeros((full, full), np.float32); m[:n, :n] = a
    mats.append(m); ps.append(p); pads.append(n)
  roots, met = K.matrix_inverse_pth_root_batched([torch.tensor(m, device=dev) for m in mats], ps, pads, eigh=True)
  met = met.cpu().numpy()
  for i, (m, p, pad) in enumerate(zip(mats, ps, pads)):
    h, mm = orc.matrix_inverse_pth_root_eigh(m, p, padding_start=pad)
    got = roots[i].cpu().numpy()
    a64 = m[:pad, :pad].astype(np.float64)
    w, v = np.linalg.eigh(a64)
    mx = max(w.max(), 0)
    # closed form of DS:1005-1014 in float64
    ridge = 1e-6 * max(mx, 1e-6)
    w2, v2 = np.linalg.eigh(a64 + ridge * np.eye(pad))
    inv = np.where(w2 == 0, 0.0, np.maximum(w2, ridge) ** (-1.0 / p))
    truth = (v2 * inv) @ v2.T
    tn = max(np.linalg.norm(truth), 1e-30)
    e_ref = np.linalg.norm(h[:pad, :pad] - truth) / tn
    e_got = np.linalg.norm(got[:pad, :pad] - truth) / tn
    ok = e_got <= 5 * e_ref + 5e-5 and np.all(got[pad:] == 0) and np.all(got[:, pad:] == 0)
    if not ok:
      bad += 1
      print("EIGH MISMATCH", m.shape[0], pad, p, f"hip {e_got:.2e} oracle {e_ref:.2e} err metric {met[i,0]:.2e} vs {mm['inverse_pth_root_errors']:.2e}")
print("eigh fuzz done, mismatches", bad)
# ---- statistics on N-d blocks
bad2 = 0
for shape, bs in (((70, 33), 32), ((5, 6, 7), 4), ((3, 130, 9), 64), ((2, 3, 4, 5), 3), ((260,), 100), ((17, 1, 19), 8), ((1, 50), 16), ((4, 4, 4, 4, 4), 4)):
  x = rng.standard_normal(shape).astype(np.float32)
  t = torch.tensor(x, device=dev)
  parts = BlockPartitioner(t, bs).partition(t)
  parts_np = BlockPartitioner(torch.from_numpy(x), bs).partition(torch.from_numpy(x))
  items, refs = [], []
  for blk, blk_np in zip(parts, parts_np):
    for axis in range(blk.dim()):
      d = blk.shape[axis]
      g0 = rng.standard_normal((d, d)).astype(np.float32); old_np = (g0 @ g0.T).astype(np.float32)
      old = torch.tensor(old_np, device=dev); new = torch.empty_like(old)
      items.append((blk, axis, old, new))
      refs.append(orc.gram_weighted_update(old_np, blk_np.numpy(), axis, 0.9, 0.1))
  K.stats_update_grouped(items, 0.9, 0.1)
  for (_, axis, _, new), ref in zip(items, refs):
    got = new.cpu().numpy()
    if not np.allclose(got, ref, rtol=2e-5, atol=1e-4 * max(np.abs(ref).max(), 1)):
      bad2 += 1; print("STATS MISMATCH", shape, bs, axis, np.abs(got - ref).max())
print("stats fuzz done, mismatches", bad2)
