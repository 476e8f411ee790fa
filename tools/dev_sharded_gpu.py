import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
from tests.test_optimizer_host_logic import _sharded_index
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(gold, "e2e_sharded.npz"))
dev = torch.device("cuda:0")
for c in _sharded_index(gold, 1):
  name, n = c["name"], c["n_params"]
  kw = dict(c["kwargs"])
  if "graft_type" in kw:
    kw["graft_type"] = pa.GraftingType(kw["graft_type"])
  bs = kw.pop("block_size")
  opt = pa.distributed_shampoo(c["lr"], bs, batch_axis_name=None, shard_optimizer_states=True,
                               num_devices_for_pjit=1, **kw)
  params = tuple(torch.tensor(z[f"{name}__param{i}"], device=dev) for i in range(n))
  st = opt.init(params).init_fn(params)
  for t in range(c["steps"]):
    grads = tuple(torch.tensor(z[f"{name}__grad{i}_t{t}"], device=dev) for i in range(n))
    upd, st = opt.update(grads, st, params)
    errs = []
    for i in range(n):
      ref = z[f"{name}__upd{i}_t{t}"]; got = upd[i].cpu().numpy()
      errs.append(float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)))
    print(name, "step", t, ["%.1e" % e for e in errs], flush=True)
  gs = st.stats.global_stats
  ref_p = z[f"{name}__global_preconditioners"]; mine = gs.preconditioners.cpu().numpy()
  per = np.linalg.norm((mine - ref_p).reshape(len(ref_p), -1), axis=1) / np.maximum(np.linalg.norm(ref_p.reshape(len(ref_p), -1), axis=1), 1e-30)
  print(name, "precond rel err: max %.2e median %.2e argmax %d" % (per.max(), np.median(per), per.argmax()))
