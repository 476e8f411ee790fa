#!/usr/bin/env python3
"""bench.py — preconditioner-recompute benchmark (BASELINE.json metric).

A "step" is one preconditioner recompute over one batch of synthetic
statistics blocks: power iteration + coupled-Newton inverse p-th root for every
block (libprecondition_amd.so), plus, for N > 1, the RCCL all-gather that leaves
every rank holding every root (the reference's DS:2876).  Inputs are resident
in HBM before the timed region.  Default workload = BASELINE.json configs[1]:
256 independent 512x512 blocks per GPU, p = 4 (weak scaling: each rank owns its
own 256 blocks).  The north_star's 64 x 1024^2 set is measured beside it and
reported under "headline_1024".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak

WORKLOADS = {
    # name: (blocks per GPU, n, k of the Wishart factor, p, seed)
    "cfg2_256x512_p4": (256, 512, 2048, 4, 1234),
    "headline_64x1024_p4": (64, 1024, 4096, 4, 1024),
}


def c_of_p(p):
  """Algorithmic products per Newton step: c(2)=3, c(4)=4, ..."""
  return int(np.floor(np.log2(p))) + bin(p).count("1") - 1 + 2


def make_blocks(name, rank, dev):
  """A_b = G_b G_b^T, G_b ~ N(0,1) [n, k] from numpy default_rng(seed + rank)."""
  from precondition_amd import kernels as K
  nb, n, k, p, seed = WORKLOADS[name]
  rng = np.random.default_rng(seed + 1000003 * rank)
  stats = torch.zeros((nb, n, n), dtype=torch.float32, device=dev)
  chunk = 16
  for b0 in range(0, nb, chunk):
    g = rng.standard_normal((min(chunk, nb - b0), n, k), dtype=np.float32)
    g_d = torch.from_numpy(g).to(dev)
    items = [(g_d[i], 0, stats[b0 + i], stats[b0 + i]) for i in range(g_d.shape[0])]
    K.stats_update_grouped(items, 0.0, 1.0)  # S <- 0*S + 1*G G^T on the MFMA path
    torch.cuda.synchronize()
  return stats, p


class Workload:

  def __init__(self, name, rank, world, dev):
    self.name = name
    self.world = world
    self.stats, self.p = make_blocks(name, rank, dev)
    nb, n = self.stats.shape[0], self.stats.shape[1]
    self.nb, self.n = nb, n
    self.roots = torch.empty_like(self.stats)
    self.gathered = (torch.empty((world * nb, n, n), dtype=torch.float32, device=dev)
                     if world > 1 else None)
    self.metrics = None

  def step(self):
    from precondition_amd import kernels as K
    _, self.metrics = K.matrix_inverse_pth_root_batched(
        list(self.stats.unbind(0)), [self.p] * self.nb,
        padding_starts=[self.n] * self.nb, out=list(self.roots.unbind(0)))
    if self.world > 1:
      import torch.distributed as dist
      dist.all_gather_into_tensor(self.gathered.view(-1), self.roots.view(-1))

  def flops(self):
    """Algorithmic FLOPs of the last step on this rank."""
    it = self.metrics[:, 5].double().sum().item()  # PS_M_TOTAL_ITERS
    return it * c_of_p(self.p) * 2.0 * float(self.n) ** 3


def timed(work, steps, warmup, world):
  import torch.distributed as dist
  for _ in range(warmup):
    work.step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    work.step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  dt = time.perf_counter() - t0
  fl = work.flops()
  if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    f = torch.tensor([fl], dtype=torch.float64, device="cuda")
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    fl = f.item()
  return dt / steps, fl


def profile_stage_kernel(work):
  """One extra step with the library's per-launch HIP-event timing on."""
  from precondition_amd import _lib
  L = _lib.lib()
  L.ps_profile_reset()
  L.ps_profile_enable(1)
  try:
    work.step()
    torch.cuda.synchronize()
  finally:
    L.ps_profile_enable(0)
  stage_ms, pi_ms, other_ms = C.c_double(), C.c_double(), C.c_double()
  launches = C.c_int64()
  L.ps_profile_get(C.addressof(stage_ms), C.addressof(launches), C.addressof(pi_ms),
                   C.addressof(other_ms))
  return stage_ms.value, launches.value, pi_ms.value, other_ms.value


def cpu_baseline(name, budget_s=12.0):
  """The oracle executing the reference's op sequence (its 6 products per step
  at p=4) on the host cores, on a bounded sample of the same workload."""
  from oracle import shampoo_oracle as orc
  nb, n, k, p, seed = WORKLOADS[name]
  rng = np.random.default_rng(seed)
  threads = None
  try:
    from threadpoolctl import threadpool_info
    info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
    if info:
      threads = int(info[0]["num_threads"])
  except Exception:  # pylint: disable=broad-except
    pass
  done, flops, t_total = 0, 0.0, 0.0
  while done < nb and (t_total < budget_s or done < 2):
    g = rng.standard_normal((n, k), dtype=np.float32)
    a = g @ g.T
    t0 = time.perf_counter()
    _, m = orc.newton_root_reference_opcount(a, p, padding_start=n)
    t_total += time.perf_counter() - t0
    flops += m["inverse_pth_root_iters"] * c_of_p(p) * 2.0 * float(n) ** 3
    done += 1
  return dict(value=round(flops / t_total / 1e9, 2), unit="GFLOP/s",
              cores=threads or (os.cpu_count() or 1), kind="port",
              sample=f"first {done} of {nb} blocks of {name} (numpy/OpenBLAS float32, "
                     f"reference op sequence incl. its redundant products), "
                     f"{t_total:.1f} s, {t_total / done * 1e3:.1f} ms/block",
              host_cpus=os.cpu_count())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--workload", default="cfg2_256x512_p4", choices=sorted(WORKLOADS))
  ap.add_argument("--no-headline", action="store_true")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus and world > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X (no CPU path)")
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=dev)

  work = Workload(args.workload, rank, world, dev)
  sec, flops = timed(work, args.steps, args.warmup, world)
  nb, n, _, p, _ = WORKLOADS[args.workload]
  iters = work.metrics[:, 1].cpu().numpy()
  errs = work.metrics[:, 0].cpu().numpy()

  line = {
      "metric": "preconditioner-recompute throughput, batched n x n inverse p-th root "
                "(algorithmic GFLOP/s; step time in ms_per_step)",
      "value": round(flops / sec / 1e9, 1),
      "unit": "GFLOP/s",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": round(sec * 1e3, 3),
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f32",
      "data": "synthetic",
      "config": {
          "workload": f"{args.workload}: {nb} blocks/GPU of {n}x{n} fp32, p={p}, "
                      f"A=GG^T with G~N(0,1) [{n}x{WORKLOADS[args.workload][2]}], "
                      "ridge 1e-6 relative, Newton",
          "blocks_per_gpu": nb, "n": n, "p": p,
          "parallelism": f"blocks partitioned over {world} GPU(s)" +
                         (", RCCL all-gather of roots in the timed region" if world > 1 else ""),
          "newton_iters_per_block": {"min": float(iters.min()), "max": float(iters.max())},
          "max_newton_error": float(np.nanmax(errs)),
          "frac_of_f32_mfma_peak": round(flops / sec / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
      },
  }

  if rank == 0:
    stage_ms, launches, pi_ms, other_ms = profile_stage_kernel(work)
    fl1 = work.flops()
    ach = fl1 / (stage_ms * 1e-3) / 1e12 if stage_ms > 0 else 0.0
    line["roofline"] = {
        "kernel": "newton_stage_kernel",
        "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
        "traffic": None,
        "launches": int(launches),
        "avg_launch_ms": round(stage_ms / max(launches, 1), 4),
        "algorithmic_gflop_per_launch": round(fl1 / max(launches, 1) / 1e9, 3),
        "step_breakdown_ms": {"product_stages": round(stage_ms, 3),
                              "power_iteration": round(pi_ms, 3),
                              "init_control_copyout": round(other_ms, 3)},
    }
  elif world > 1:
    pass
  if world > 1:
    import torch.distributed as dist
    dist.barrier()

  if not args.no_headline and args.workload != "headline_64x1024_p4":
    del work
    torch.cuda.empty_cache()
    hw = Workload("headline_64x1024_p4", rank, world, dev)
    hsec, hflops = timed(hw, max(2, args.steps // 2), 1, world)
    head = {"workload": "64 blocks/GPU of 1024x1024 fp32, p=4",
            "value": round(hflops / hsec / 1e9, 1), "unit": "GFLOP/s",
            "ms_per_step": round(hsec * 1e3, 3),
            "frac_of_f32_mfma_peak": round(hflops / hsec / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4)}
    if rank == 0:
      sm, ln, pm, om = profile_stage_kernel(hw)
      f1 = hw.flops()
      head["roofline_stage_kernel_tflops"] = round(f1 / (sm * 1e-3) / 1e12, 2) if sm > 0 else None
      head["step_breakdown_ms"] = {"product_stages": round(sm, 3),
                                   "power_iteration": round(pm, 3),
                                   "init_control_copyout": round(om, 3)}
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
    line["headline_1024"] = head
    del hw

  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    line["cpu_baseline"] = cpu_baseline(args.workload)
  if rank == 0:
    print(json.dumps(line), flush=True)
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
