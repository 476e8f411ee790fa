"""Dev: resident vs streaming power iteration (bit-identity + time)."""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
  from precondition_amd import kernels as K
  dev = torch.device("cuda:0")
  out = {}
  gen = torch.Generator(device=dev).manual_seed(3)
  for name, nb, n in (("c512", 256, 512), ("c1024", 64, 1024), ("mixed", 0, 0)):
    if name == "mixed":
      sizes = [1, 5, 127, 128, 129, 197, 300, 768, 1000, 1024, 0, 260] * 3
    else:
      sizes = [n] * nb
    mats = []
    for s in sizes:
      g = torch.randn((max(s, 1), 2 * max(s, 1)), generator=gen, device=dev)
      a = (g @ g.T)[:s, :s].contiguous() if s > 0 else torch.zeros((1, 1), device=dev)
      mats.append(a)
    pad = [s for s in sizes]
    for rep in range(2):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      lam, its = K.power_iteration_batched(mats, padding_starts=pad)
      torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name + "_lam"] = lam.cpu().numpy(); out[name + "_its"] = its.cpu().numpy()
    print(name, "ms", round(dt * 1e3, 3), "iters", int(its.min()), int(its.max()))
    # asymmetric inputs
    if name == "mixed":
      am = [m + 0.01 * torch.randn(m.shape, generator=gen, device=dev) for m in mats]
      lam, its = K.power_iteration_batched(am, padding_starts=pad)
      out["asym_lam"] = lam.cpu().numpy(); out["asym_its"] = its.cpu().numpy()
  np.savez(sys.argv[2], **out)
else:
  res = {}
  for mode in ("0", "1"):
    env = dict(os.environ, PS_PI_RESIDENT=mode)
    f = f"/tmp/pi_{mode}.npz"
    print("PS_PI_RESIDENT=" + mode)
    subprocess.check_call([sys.executable, __file__, "child", f], env=env)
    res[mode] = np.load(f)
  ok = True
  for k in res["0"].files:
    same = np.array_equal(res["0"][k], res["1"][k], equal_nan=True)
    ok &= same
    if not same:
      d = np.nanmax(np.abs(res["0"][k].astype(np.float64) - res["1"][k].astype(np.float64)))
      print("MISMATCH", k, d)
  print("bit-identical:", ok)
