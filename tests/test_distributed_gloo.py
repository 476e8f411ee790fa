"""N > 1 path on CPU: two processes over gloo shard the statistics, root their
own chunk (CPU stand-in for the HIP root) and all-gather; every rank must end
with the single-process result in list order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _make_stats():
  rng = np.random.default_rng(11)
  sizes = [8, 12, 8, 5, 16, 8, 12]  # 7 statistics: odd count => a padding slot on 2 ranks
  exps = [4, 2, 4, 4, 2, 4, 2]
  stats = []
  for n in sizes:
    g = rng.standard_normal((n, 4 * n)).astype(np.float32)
    stats.append(torch.from_numpy((g @ g.T).astype(np.float32)))
  return stats, exps


def _worker(rank, world, port, ownership, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from precondition_amd import comm
    from tests import cpu_backend
    stats, exps = _make_stats()
    calls = []

    def root_fn(mats, ps, pads, **kw):
      calls.append(len(mats))
      return cpu_backend.matrix_inverse_pth_root_batched(mats, ps, pads, **kw)

    roots, metrics = comm.sharded_inverse_pth_roots(
        stats, exps, group=dist.group.WORLD, ownership=ownership, root_fn=root_fn)
    ret[rank] = ([r.clone().numpy() for r in roots], metrics.numpy().copy(), calls)
    # two-phase (compute/all-gather overlap) layout: same results, two root calls
    calls2 = []

    def root_fn2(mats, ps, pads, **kw):
      calls2.append(len(mats))
      return cpu_backend.matrix_inverse_pth_root_batched(mats, ps, pads, **kw)

    roots2, metrics2 = comm.sharded_inverse_pth_roots(
        stats, exps, group=dist.group.WORLD, ownership=ownership, root_fn=root_fn2,
        overlap_min_bytes=0)
    ret[rank + 200] = ([r.clone().numpy() for r in roots2], metrics2.numpy().copy(), calls2)

    # the optimizer surface over the same group: every rank must produce the
    # same update as a single process.
    import precondition_amd as pa
    params = (torch.ones(20, 12), torch.ones(7, 9))
    g = np.random.default_rng(3)
    grads = tuple(torch.from_numpy(g.standard_normal(p.shape).astype(np.float32)) for p in params)
    opt = pa.distributed_shampoo(0.1, 8, batch_axis_name=dist.group.WORLD,
                                 start_preconditioning_step=1, block_ownership=ownership,
                                 _backend_for_testing=cpu_backend)
    st = opt.init(params)
    for _ in range(3):
      upd, st = opt.update(grads, st, params)
    ret[rank + 100] = [u.numpy().copy() for u in upd]
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("ownership", ["reference", "lpt"])
def test_two_rank_sharding_matches_single_process(ownership):
  from precondition_amd import comm
  from tests import cpu_backend
  import precondition_amd as pa
  world = 2
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_worker, args=(world, _free_port(), ownership, ret), nprocs=world, join=True)
  stats, exps = _make_stats()
  base_roots, base_metrics = comm.sharded_inverse_pth_roots(
      stats, exps, group=None, root_fn=cpu_backend.matrix_inverse_pth_root_batched)
  for rank in range(world):
    roots, metrics, calls = ret[rank]
    assert len(roots) == len(stats)
    for a, b in zip(roots, base_roots):
      assert np.array_equal(a, b.numpy())  # same CPU arithmetic => identical
    assert np.array_equal(metrics, base_metrics.numpy())
    assert sum(calls) in (3, 4)  # 7 statistics over 2 ranks
    roots2, metrics2, calls2 = ret[rank + 200]
    for a, b in zip(roots2, base_roots):
      assert np.array_equal(a, b.numpy())
    assert np.array_equal(metrics2, base_metrics.numpy())
    assert len(calls2) in (1, 2) and sum(calls2) == sum(calls)
  if ownership == "reference":
    assert ret[0][2] == [4] and ret[1][2] == [3]  # chunks [0,4) and [4,7) + 1 padding slot
  # optimizer surface
  params = (torch.ones(20, 12), torch.ones(7, 9))
  g = np.random.default_rng(3)
  grads = tuple(torch.from_numpy(g.standard_normal(p.shape).astype(np.float32)) for p in params)
  opt = pa.distributed_shampoo(0.1, 8, batch_axis_name=None, start_preconditioning_step=1,
                               _backend_for_testing=cpu_backend)
  st = opt.init(params)
  for _ in range(3):
    upd, st = opt.update(grads, st, params)
  for rank in range(world):
    for a, b in zip(ret[rank + 100], upd):
      assert np.array_equal(a, b.numpy())


def _quant_problem():
  params = (torch.ones(20, 12), torch.ones(7, 9), torch.ones(3, 4, 5))
  g = np.random.default_rng(31)
  grads = [tuple(torch.from_numpy(g.standard_normal(p.shape).astype(np.float32)) for p in params)
           for _ in range(4)]
  kw = dict(start_preconditioning_step=1, preconditioning_compute_steps=2, matrix_epsilon=1e-3,
            best_effort_memory_usage_reduction=True)
  return params, grads, kw


def _run_quant(group):
  import precondition_amd as pa
  from tests import cpu_backend
  params, grads, kw = _quant_problem()
  opt = pa.distributed_shampoo(0.1, 8, batch_axis_name=group, _backend_for_testing=cpu_backend,
                               **kw)
  st = opt.init(params)
  for g in grads:
    upd, st = opt.update(g, st, params)
  s0 = st.stats[0]
  assert s0.statistics[0].quantized.dtype == torch.int16
  assert s0.preconditioners[0].quantized.dtype == torch.int16
  assert s0.momentum.quantized.dtype == torch.int8
  return ([u.numpy().copy() for u in upd],
          [p.quantized.numpy().copy() for s in st.stats for p in s.preconditioners],
          [p.bucket_size.numpy().copy() for s in st.stats for p in s.preconditioners])


def _quant_worker(rank, world, port, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    ret[rank] = _run_quant(dist.group.WORLD)
  finally:
    dist.destroy_process_group()


def test_two_rank_quantized_second_moment_matches_one_rank():
  """int16 preconditioners travel through the all-gather as codes + diagonal + bucket
  sizes (DS:3102-3127); two ranks must end bit-identical to a one-rank group."""
  from tests.conftest import single_rank_group
  world = 2
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_quant_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
  base = _run_quant(single_rank_group("gloo"))
  for rank in range(world):
    for got, ref in zip(ret[rank], base):
      assert len(got) == len(ref)
      for a, b in zip(got, ref):
        assert a.dtype == b.dtype and np.array_equal(a, b)


def _run_sharded_stats(group, shard, quant, ownership):
  import precondition_amd as pa
  from tests import cpu_backend
  params, grads, kw = _quant_problem()
  kw = dict(kw)
  kw["best_effort_memory_usage_reduction"] = quant
  opt = pa.distributed_shampoo(0.1, 8, batch_axis_name=group, _backend_for_testing=cpu_backend,
                               shard_statistics=shard, block_ownership=ownership, **kw)
  st = opt.init(params)
  for g in grads:
    upd, st = opt.update(g, st, params)
  stats = []
  for s in st.stats:
    for x in s.statistics:
      q = x.quantized if hasattr(x, "quantized") else x
      stats.append(q.numpy().copy())
  precs = [(p.quantized if hasattr(p, "quantized") else p).numpy().copy()
           for s in st.stats for p in s.preconditioners]
  return [u.numpy().copy() for u in upd], stats, precs


def _sharded_worker(rank, world, port, quant, ownership, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    ret[rank] = _run_sharded_stats(dist.group.WORLD, True, quant, ownership)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("quant,ownership", [(False, "reference"), (True, "lpt")])
def test_owner_only_statistics_two_ranks(quant, ownership):
  """shard_statistics: a rank keeps and updates only the statistics it
  roots.  Updates and (replicated) preconditioners must equal the replicated mode
  bit for bit; every statistic lives on exactly one rank and equals the replicated one."""
  from tests.conftest import single_rank_group
  world = 2
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_sharded_worker, args=(world, _free_port(), quant, ownership, ret), nprocs=world,
           join=True)
  base_upd, base_stats, base_precs = _run_sharded_stats(single_rank_group("gloo"), False, quant,
                                                        ownership)
  for rank in range(world):
    upd, stats, precs = ret[rank]
    for a, b in zip(upd, base_upd):
      assert np.array_equal(a, b)
    for a, b in zip(precs, base_precs):
      assert np.array_equal(a, b)
  n_owned = [0, 0]
  for j, ref in enumerate(base_stats):
    holders = [r for r in range(world) if ret[r][1][j].size > 0]
    assert len(holders) == 1, (j, holders)       # exactly one owner
    assert np.array_equal(ret[holders[0]][1][j], ref)
    n_owned[holders[0]] += 1
  assert min(n_owned) > 0


# ---------------------------------------------------------------------------
# shard_optimizer_states: the reference's pjit-mode API (DS:2162-2583, DS:3661-3673)
def _run_pjit_api(group, sharded):
  """Four steps; returns (updates of the last step, true-size statistics by global index
  or None where this rank does not hold them, true-size preconditioners)."""
  import precondition_amd as pa
  from tests import cpu_backend
  params, grads, kw = _quant_problem()
  kw = dict(kw)
  kw["best_effort_memory_usage_reduction"] = False
  opt = pa.distributed_shampoo(0.1, 8, batch_axis_name=group, _backend_for_testing=cpu_backend,
                               shard_optimizer_states=sharded, **kw)
  if not sharded:
    st = opt.init(params)
    for g in grads:
      upd, st = opt.update(g, st, params)
    stats = [x.numpy().copy() for s in st.stats for x in s.statistics]
    precs = [p.numpy().copy() for s in st.stats for p in s.preconditioners]
    return [u.numpy().copy() for u in upd], stats, precs, None
  fns = opt.init(params)
  assert isinstance(fns, pa.state.InitFnState)
  st = fns.init_fn(params)
  shapes = fns.shape_and_dtype_fn(params)
  g0 = st.stats.global_stats
  n_pad, max_size = shapes.stats.global_stats.statistics[0][0], shapes.stats.global_stats.statistics[0][1]
  assert list(g0.preconditioners.shape) == shapes.stats.global_stats.preconditioners[0]
  assert g0.statistics.shape[1:] == (max_size, max_size)
  for g in grads:
    upd, st = opt.update(g, st, params)
  gs = st.stats.global_stats
  import torch.distributed as dist
  world = dist.get_world_size(group) if group is not None else 1
  rank = dist.get_rank(group) if group is not None else 0
  b = gs.statistics.shape[0]
  assert b * world == n_pad
  from precondition_amd import pytree
  locs = [l for l in pytree.tree_flatten(
      st.stats.local_stats, is_leaf=lambda x: isinstance(x, pa.state.LocalShardedParameterStats))[0]]
  stats, precs = [], []
  for loc in locs:
    for j, n in enumerate(loc.sizes):
      i = loc.index_start + j
      stats.append(gs.statistics[i - rank * b, :n, :n].numpy().copy()
                   if rank * b <= i < (rank + 1) * b else None)
      precs.append(gs.preconditioners[i, :n, :n].numpy().copy())
      # padding of a slot: identity for statistics, zero for roots (DS:1324-1350, DST:367-398)
      if rank * b <= i < (rank + 1) * b and n < max_size:
        assert np.array_equal(gs.statistics[i - rank * b, n:, n:].numpy(),
                              np.eye(max_size - n, dtype=np.float32))
        assert not gs.statistics[i - rank * b, :n, n:].any()
      assert not gs.preconditioners[i, n:, :].any() and not gs.preconditioners[i, :, n:].any()
  return [u.numpy().copy() for u in upd], stats, precs, int(st.count)


def _pjit_worker(rank, world, port, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    ret[rank] = _run_pjit_api(dist.group.WORLD, True)
  finally:
    dist.destroy_process_group()


def test_sharded_optimizer_state_api_statistics_and_roots_match_plain_optimizer():
  """One process: the pjit-style state (stacked padded arrays + local stats, init-function
  triple) accumulates the plain optimizer's statistics and roots bit for bit.  Its UPDATES
  differ by design: sharded_update_fn transforms the gradient with the preconditioners the
  state came in with (DS:2443-2452), the pmap path roots first (DS:3648-3650); the updates are
  pinned by the reference's own pjit-mode goldens (test_e2e_sharded_state_*)."""
  base_upd, base_stats, base_precs, _ = _run_pjit_api(None, False)
  upd, stats, precs, count = _run_pjit_api(None, True)
  assert count == 4
  for a, b in zip(stats, base_stats):
    assert np.array_equal(a, b)
  for a, b in zip(precs, base_precs):
    assert np.array_equal(a, b)
  assert any(not np.array_equal(a, b) for a, b in zip(upd, base_upd))


def test_sharded_optimizer_state_api_two_ranks():
  """Two ranks: every rank holds one chunk of the stacked statistics (batch() order,
  DS:1827), all preconditioners; updates equal the single-process pjit-mode ones."""
  world = 2
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_pjit_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
  _, base_stats, base_precs, _ = _run_pjit_api(None, False)
  one_upd, _, _, _ = _run_pjit_api(None, True)
  for rank in range(world):
    upd, stats, precs, count = ret[rank]
    assert count == 4
    for a, b in zip(upd, one_upd):
      assert np.array_equal(a, b)
    for a, b in zip(precs, base_precs):
      assert np.array_equal(a, b)
  for j, ref in enumerate(base_stats):
    holders = [r for r in range(world) if ret[r][1][j] is not None]
    assert len(holders) == 1, (j, holders)
    assert np.array_equal(ret[holders[0]][1][j], ref)


def _sharded_golden_worker(rank, world, port, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from tests import cpu_backend
    from tests.test_optimizer_host_logic import (_sharded_index, check_sharded_final_state,
                                                 run_sharded_case)
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "e2e_sharded.npz"))
    worst = 0.0
    for case in _sharded_index(gold, world):
      st, w = run_sharded_case(case, z, torch.device("cpu"), cpu_backend, group=dist.group.WORLD)
      check_sharded_final_state(case, z, st, rank=rank, world=world)
      worst = max(worst, float(w))
    ret[rank] = worst
  finally:
    dist.destroy_process_group()


def test_e2e_sharded_state_two_ranks_vs_reference_golden():
  """The reference's pjit mode run with num_devices_for_pjit = 2 (tests/golden/e2e_sharded.npz:
  119 / 61 statistics padded to 120 / 62 with identity rows of exponent 1, DS:2476-2486) against
  two gloo ranks: every update of every step, and rank r's rows [r b, (r + 1) b) of the stacked
  statistics, all rows of the preconditioners, exponents, index_start / sizes."""
  world = 2
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_sharded_golden_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
  for rank in range(world):
    assert ret[rank] < 1e-3, ret[rank]


# ---------------------------------------------------------------------------
def test_bench_multi_rank_control_flow_over_gloo(tmp_path):
  """bench.py's N > 1 path end to end, launched exactly as the driver launches it
  (python -m torch.distributed.run --nproc-per-node 2 ... --gpus 2) through
  tests/bench_selftest_launcher.py, which swaps the kernels for the oracle and shrinks the sizes
  from the TEST side (bench.py has no such hook) and runs over gloo: rank set-up, the two-phase
  all-gather step, barriers, max-over-ranks timing, the ViT-B strong-scaling leg with LPT
  ownership, ONE JSON line from rank 0 with the multi-GPU self-diagnosis."""
  import json
  import subprocess
  env = dict(os.environ)
  env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(ROOT, "tests", "bench_selftest_launcher.py"), "--gpus", "2", "--steps", "2",
         "--warmup", "1"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  line = json.loads(lines[0])
  assert line["data"] == "selftest-cpu" and line["n_gpus"] == 2 and line["scaling"] == "weak"
  assert line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
  mg = line["multi_gpu"]
  assert mg["rccl_ranks_seen"] == 2 and mg["world_size"] == 2
  assert mg["gathered_list_matches_rank_order"] is True
  assert line["headline_1024"]["value"] > 0
  vb = line["vit_b_cfg4"]
  assert vb["failed_blocks"] == 0 and vb["ms_per_step"] > 0
  assert "cpu_baseline" not in line   # rank 0 at N = 1 only


def test_bench_plain_launch_starts_its_own_ranks(tmp_path):
  """`python bench.py --gpus 2` WITHOUT torchrun (how the driver launches --gpus 1): bench.py must
  start the two ranks itself, relay exactly one line with n_gpus == 2 and two ranks seen in the
  top-level config, and exit with the launcher's status."""
  import json
  import subprocess
  env = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
  cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_selftest_launcher.py"), "--gpus", "2",
         "--steps", "2", "--warmup", "1", "--no-extras"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  line = json.loads(lines[0])
  assert line["n_gpus"] == 2 and line["config"]["rccl_ranks_seen"] == 2
  assert line["multi_gpu"]["world_size"] == 2
  # flat scalars: the driver's record keeps the scalars of `config` and drops nested objects
  cfg = line["config"]
  assert cfg["headline_1024_ms"] > 0 and cfg["headline_1024_ms_no_hint"] > 0 and cfg["ms_per_step_no_hint"] > 0
  assert all(not isinstance(v, (dict, list)) for k, v in cfg.items() if k.startswith(("headline_", "eigh_", "fd_", "vit_")))


def test_bench_refuses_rank_counts_it_cannot_run(tmp_path):
  """A plain `bench.py --gpus 2` on a node with fewer than 2 GPUs exits non-zero with a message
  and prints no JSON line; so does a launcher/--gpus mismatch."""
  import subprocess
  import torch
  env = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  if torch.cuda.device_count() < 2:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "{" not in r.stdout
  env["WORLD_SIZE"] = "2"
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env,
                     capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
  assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout


@pytest.mark.parametrize("world", [2, 4, 8])
def test_vit_b_ownership_tables_and_gather_layout(world):
  """The 395 statistics of the ViT-B tree (cfg4) over 2 / 4 / 8 ranks, no communication:
  every rank derives the same ownership, phase and offset tables from shapes alone; the
  reference ownership is batch()'s contiguous chunks (DS:1827-1831) and the gathered list
  order is the statistics list order (DS:1834-1846); send buffers have the same size on
  every rank (all_gather_into_tensor requires it) with both ownership modes."""
  from precondition_amd import comm
  sizes = [768] * 172 + [768] * 112 + [1024] * 72 + [1024] * 36 + [1000, 1000, 197]
  exps = [4] * 172 + [2] * 112 + [4] * 72 + [2] * 36 + [4, 2, 4]
  assert len(sizes) == 395
  ref = comm.reference_ownership(395, world)
  b = (395 + (-395 % world)) // world
  assert ref == [i // b for i in range(395)] and max(ref) == world - 1
  for mode in ("reference", "lpt"):
    owner = comm.ownership_table(sizes, exps, world, mode)
    assert len(owner) == 395 and set(owner) == set(range(world))
    load = [0.0] * world
    for o, n, p in zip(owner, sizes, exps):
      load[o] += (4 if p == 4 else 3) * float(n) ** 3
    if mode == "lpt":
      assert max(load) / (sum(load) / world) < 1.02   # cost-balanced to 2 %
  # the in-process emulation of the gather: each "rank" fills its flat buffer with the index
  # of the statistic; after concatenation every statistic must be found at its own index
  import torch
  for mode in ("reference", "lpt"):
    owner = comm.ownership_table(sizes, exps, world, mode)
    seen = []
    small = [max(1, n // 64) for n in sizes]

    class FakeGroup:  # world/rank without a process group: run every rank in turn
      pass

    results = []
    for rank in range(world):
      calls = []

      def root_fn(mats, ps, pads, out=None, **kw):
        for m, o in zip(mats, out):
          o.fill_(float(m[0, 0]))
        calls.append(len(mats))
        return out, torch.zeros((len(mats), 8))

      stats = [torch.full((n, n), float(i)) for i, n in enumerate(small)]
      # one rank's view: group=None roots everything; emulate ownership by masking
      mine = [i for i in range(395) if owner[i] == rank]
      roots, _ = comm.sharded_inverse_pth_roots([stats[i] for i in mine], [exps[i] for i in mine],
                                                group=None, root_fn=root_fn)
      results.append({i: float(r[0, 0]) for i, r in zip(mine, roots)})
    merged = {}
    for r in results:
      merged.update(r)
    assert [merged[i] for i in range(395)] == [float(i) for i in range(395)]


def _vitb_worker(rank, world, port, ownership, ret):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from precondition_amd import comm
    sizes = [768] * 172 + [768] * 112 + [1024] * 72 + [1024] * 36 + [1000, 1000, 197]
    exps = [4] * 172 + [2] * 112 + [4] * 72 + [2] * 36 + [4, 2, 4]
    small = [max(2, n // 128) for n in sizes]
    stats = [torch.full((n, n), float(i)) for i, n in enumerate(small)]
    sent = []

    def root_fn(mats, ps, pads, out=None, **kw):
      for m, o in zip(mats, out):
        o.fill_(float(m[0, 0]) + 0.5)
      sent.append(len(mats))
      rows = torch.zeros((len(mats), 8))
      rows[:, 1] = torch.tensor([float(m[0, 0]) for m in mats])
      return out, rows

    roots, metrics = comm.sharded_inverse_pth_roots(
        stats, exps, group=dist.group.WORLD, ownership=ownership, root_fn=root_fn,
        overlap_min_bytes=0)   # force the two-phase (overlapped) layout
    ok = all(float(r[0, 0]) == i + 0.5 and tuple(r.shape) == (small[i], small[i])
             for i, r in enumerate(roots))
    ok &= bool(torch.equal(metrics[:, 1], torch.arange(395, dtype=torch.float32)))
    ret[rank] = (ok, sum(sent))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ownership", [(4, "reference"), (4, "lpt")])
def test_vit_b_statistics_gather_order_four_ranks(world, ownership):
  """395 statistics over 4 gloo ranks, two-phase layout: every rank ends with every root and
  metrics row at its list index (DS:1834-1846); each statistic is rooted exactly once."""
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_vitb_worker, args=(world, _free_port(), ownership, ret), nprocs=world, join=True)
  assert all(ret[r][0] for r in range(world))
  assert sum(ret[r][1] for r in range(world)) == 395


def test_cost_model_uses_iteration_counts():
  """LPT ownership weighs a block by last recompute's Newton iterations (ViT-B: 7-17)."""
  from precondition_amd import comm
  sizes = [1024] * 8
  exps = [4] * 8
  hint = [16, 16, 8, 8, 8, 8, 8, 8]
  owner = comm.ownership_table(sizes, exps, 2, "lpt", hint)
  load = [sum(h for h, o in zip(hint, owner) if o == r) for r in (0, 1)]
  assert load == [40, 40], (owner, load)
  flat = comm.ownership_table(sizes, exps, 2, "lpt")
  assert sorted(flat.count(r) for r in (0, 1)) == [4, 4]


def test_gathered_results_start_on_aligned_boundaries():
  """The per-statistic results are views into the gathered buffer and feed 16-byte vector loads of
  every later step (preconditioner application): each must start on a 4 KB boundary whatever the
  sizes before it (a 197 x 197 statistic once left the rest of a ViT-B tree 4 bytes off and its
  application on the scalar-load path)."""
  import torch
  from precondition_amd import comm
  sizes = [197, 768, 5, 1000, 33, 1024]
  stats = [torch.full((n, n), float(i)) for i, n in enumerate(sizes)]

  def root_fn(mats, ps, pads, out=None, **kw):
    for m, o in zip(mats, out):
      o.copy_(m + 0.5)
    return out, torch.zeros((len(mats), 8))

  roots, _ = comm.sharded_inverse_pth_roots(stats, [4] * len(sizes), group=None, root_fn=root_fn)
  for i, (r, n) in enumerate(zip(roots, sizes)):
    assert tuple(r.shape) == (n, n) and float(r[0, 0]) == i + 0.5
    assert r.storage_offset() % 1024 == 0 and r.data_ptr() % 16 == 0, (i, r.storage_offset())
    assert r.is_contiguous()
