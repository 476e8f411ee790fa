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

# ---- quantization (QU:45-113): every kernel family of csrc/quant.hip in mixed grouped calls ----------------------
from oracle import quantization_oracle as qorc
rng = np.random.default_rng(int(os.environ.get("SEED", "0")) + 3)
bad3 = 0
cases3 = 0
row_choices = [1, 2, 3, 17, 63, 64, 65, 100, 197, 512, 768, 1000, 1024, 1025, 1500, 2048, 3072, 4096, 4097, 5000]
col_choices = [1, 2, 3, 4, 7, 64, 66, 100, 128, 197, 256, 260, 768, 1000, 1024, 2048, 2050, 3072, 9000]
for rnd in range(int(os.environ.get("QROUNDS", "16"))):
  bits = int(rng.choice([8, 16]))
  extract = bool(rng.uniform() < 0.4)
  xs = []
  for _ in range(int(rng.integers(3, 12))):
    r = int(rng.choice(row_choices)); c = r if extract else int(rng.choice(col_choices))
    if r * c > 6_000_000:
      c = max(1, 6_000_000 // r)
      if extract: r = c = min(r, 2048)
    x = (rng.standard_normal((r, c)) * np.exp(rng.uniform(-10, 10, size=c))).astype(np.float32)
    if extract: x = (x + x.T).astype(np.float32)
    x[rng.uniform(size=x.shape) < 0.03] = 0.0
    if c > 2 and rng.uniform() < 0.5: x[:, int(rng.integers(0, c))] = 0.0
    xs.append(np.ascontiguousarray(x))
  tq = torch.int8 if bits == 8 else torch.int16
  npdt = np.int8 if bits == 8 else np.int16
  ts = [torch.tensor(x, device=dev) for x in xs]
  out = K.quantize_grouped(ts, tq, extract)
  fl = K.dequantize_grouped(out)
  for x, (q, d, b), f in zip(xs, out, fl):
    cases3 += 1
    oq, od, ob = qorc.quantize(x, npdt, extract)
    of = qorc.to_float(oq, od, ob, npdt, extract)
    ok = (np.array_equal(q.cpu().numpy(), oq) and
          np.array_equal(b.cpu().numpy().view(np.uint32), np.asarray(ob, np.float32).view(np.uint32)) and
          (not extract or np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))) and
          np.array_equal(f.cpu().numpy().view(np.uint32), of.view(np.uint32)))
    if not ok:
      bad3 += 1; print("QUANT MISMATCH", x.shape, bits, extract)
print("quant fuzz done, cases", cases3, "mismatches", bad3)
