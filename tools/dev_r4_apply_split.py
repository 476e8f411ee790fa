"""dev (GPU): the grouped application launches of the ViT-B tree, whole vs matrix blocks only vs the
one-row products of the vector blocks only (where do the 1.74 ms of a launch go?)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from precondition_amd import kernels as K, plan as P, _lib
from precondition_amd.blocking import Preconditioner
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
pcs = [Preconditioner(g, 1024, 4096, True) for g in grads]
merged = [g.reshape(pc._transformed_shape).contiguous() for g, pc in zip(grads, pcs)]
pl = P.TreePlan.build([tuple(g.shape) for g in merged], pcs, [False] * len(grads))
precs = []
for d in pl.stat_dims:
  a = torch.randn((int(d), int(d)), device=dev); precs.append((a + a.T).contiguous())
outs = [torch.empty_like(g) for g in merged]
x = torch.empty(max(pl.x_elems, 1), dtype=torch.float32, device=dev)
def table(which, sym):
  gp = pl._ptrs(merged); pp = pl._ptrs(precs); rp = pl._ptrs(outs); xp = np.uint64(x.data_ptr())
  ta = pl.a_tbl.copy()
  ta["a"] = gp[pl.a_param] + pl.a_goff; ta["b"] = pp[pl.a_stat]
  ta["c"] = np.where(pl.a_c_is_res, rp[pl.a_param], xp) + pl.a_coff
  if which == "A_all": return ta
  if which == "A_matrix": return ta[~pl.a_c_is_res]
  if which == "A_vector": return ta[pl.a_c_is_res]
  if which == "A_sym":
    sa = pl.sa_tbl.copy(); sa["a"] = pp[pl.sa_stat]; sa["b"] = gp[pl.sa_param] + pl.sa_goff; sa["c"] = xp + pl.sa_xoff
    return sa
  tb = (pl.sb_tbl if which == "B_sym" else pl.b_tbl).copy()
  tb["a"] = xp + pl.b_xoff; tb["b"] = pp[pl.b_stat]; tb["c"] = rp[pl.b_param] + pl.b_coff
  return tb
L = _lib.lib()
for which in ("A_all", "A_matrix", "A_vector", "A_sym", "B", "B_sym"):
  tbl = np.ascontiguousarray(table(which, False))
  n = len(tbl)
  descs = C.cast(tbl.ctypes.data, C.POINTER(_lib.GemmDesc))
  ws = K._workspace(L.ps_gemm_grouped_workspace_bytes(descs, n), dev)
  h = C.c_void_p()
  assert L.ps_gemm_grouped_plan_create(K._stream(), descs, n, ws.data_ptr(), ws.numel(), C.byref(h)) == 0
  L.ps_gemm_grouped_plan_launch(K._stream(), h); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): L.ps_gemm_grouped_plan_launch(K._stream(), h)
  e1.record(); torch.cuda.synchronize()
  fl = float((2.0 * tbl["m"].astype(np.float64) * tbl["n"] * tbl["k"]).sum())
  ms = e0.elapsed_time(e1) / 10
  print(f"{which:9s} tasks {n:4d}  {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s = {fl / ms / 1e9 / 157.3:.2f} of peak", flush=True)

# the same launches with the caches flushed in between (a 2 GB fill), as inside an update() step
flush = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)
def timed_cold(which):
  tbl = np.ascontiguousarray(table(which, False)); n = len(tbl)
  descs = C.cast(tbl.ctypes.data, C.POINTER(_lib.GemmDesc))
  ws = K._workspace(L.ps_gemm_grouped_workspace_bytes(descs, n), dev)
  h = C.c_void_p()
  assert L.ps_gemm_grouped_plan_create(K._stream(), descs, n, ws.data_ptr(), ws.numel(), C.byref(h)) == 0
  tot = 0.0
  for _ in range(6):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.ps_gemm_grouped_plan_launch(K._stream(), h); e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
  return tot / 6
for which in ("A_all", "A_matrix", "A_sym", "B", "B_sym"):
  print(f"cold {which:9s} {timed_cold(which):.3f} ms", flush=True)

# sustained: stats-like + A + B alternating for ~1 s (does the launch time drift with the power state?)
def mk(which):
  tbl = np.ascontiguousarray(table(which, False)); n = len(tbl)
  descs = C.cast(tbl.ctypes.data, C.POINTER(_lib.GemmDesc))
  ws = K._workspace(L.ps_gemm_grouped_workspace_bytes(descs, n), dev)
  h = C.c_void_p()
  assert L.ps_gemm_grouped_plan_create(K._stream(), descs, n, ws.data_ptr(), ws.numel(), C.byref(h)) == 0
  return h, ws
hA, wsA = mk("A_all"); hB, wsB = mk("B")
for rep in range(4):
  evs = []
  for i in range(60):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); L.ps_gemm_grouped_plan_launch(K._stream(), hA); e[1].record()
    L.ps_gemm_grouped_plan_launch(K._stream(), hB); e[2].record(); evs.append(e)
  torch.cuda.synchronize()
  a = [e[0].elapsed_time(e[1]) for e in evs]; b = [e[1].elapsed_time(e[2]) for e in evs]
  print(f"sustained rep {rep}: A first {a[0]:.3f} mid {np.mean(a[25:35]):.3f} last {np.mean(a[-5:]):.3f} | B first {b[0]:.3f} mid {np.mean(b[25:35]):.3f} last {np.mean(b[-5:]):.3f}", flush=True)
