"""dev (GPU, under rocprofv3 --kernel-trace): update() steps on the ViT-B tree, then the SAME application
launches (real gradients, real preconditioners) repeated back to back: does the application kernel take
longer inside a step than alone?"""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import precondition_amd as pa
from precondition_amd import plan as P
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
cap = {}
orig = P.TreePlan.apply_preconditioners
def spy(self, g, p, o, **k):
  cap["args"] = (self, g, p, o, k); return orig(self, g, p, o, **k)
P.TreePlan.apply_preconditioners = spy
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=50, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED)
st = opt.init(params)
for t in range(12):
  upd, st = opt.update(grads, st, params)
torch.cuda.synchronize()
marker = torch.zeros(7, device=dev); marker.fill_(1.0); torch.cuda.synchronize()   # a recognisable fill kernel
self_, g, p, o, k = cap["args"]
for _ in range(12):
  orig(self_, g, p, o, **k)
torch.cuda.synchronize()
# which operand makes the difference?  (marker kernels separate the groups in the trace)
def group(gg, pp, oo, tag):
  marker.fill_(float(tag)); torch.cuda.synchronize()
  for _ in range(6):
    orig(self_, gg, pp, oo, **k)
  torch.cuda.synchronize()
gen = torch.Generator(device=dev).manual_seed(1)
p_syn = []
for t in p:
  a = torch.randn(t.shape, generator=gen, device=dev); p_syn.append((a + a.T).contiguous())
g_syn = [torch.randn(t.shape, generator=gen, device=dev) * 0.02 for t in g]
p_clone = [t.clone() for t in p]          # real values, separate allocations
group(g, p_syn, o, 2)
group(g_syn, p, o, 3)
group(g, p_clone, o, 4)
print("P stats: mean abs", float(torch.stack([t.abs().mean() for t in p]).mean()), "frac zeros", float(torch.stack([(t == 0).float().mean() for t in p]).mean()),
      "denormal frac", float(torch.stack([((t.abs() < 1.2e-38) & (t != 0)).float().mean() for t in p]).mean()))
# layout rule: one flat buffer, every matrix at an offset rounded up to `align` bytes
def flat_views(align):
  offs, tot = [], 0
  for t in p:
    tot = (tot + align - 1) // align * align
    offs.append(tot // 4); tot += t.numel() * 4
  buf = torch.empty(tot // 4 + 1024, dtype=torch.float32, device=dev)
  vs = []
  for t, o_ in zip(p, offs):
    v = buf[o_:o_ + t.numel()].view(t.shape); v.copy_(t); vs.append(v)
  return vs
for tag, align in ((5, 4), (6, 4096), (7, 65536), (8, 1 << 21), (9, 1 << 22)):
  group(g, flat_views(align), o, tag)
print("first P data_ptr % 2MB:", p[0].data_ptr() % (1 << 21), "storage offset", p[0].storage_offset(), "p[1] offset", p[1].storage_offset())
