"""Block/shape bookkeeping must be BIT-EXACT with the reference (goldens were
produced by the reference's own merge_small_dims / BlockPartitioner /
Preconditioner / pad_square_matrix / batch / unbatch)."""
import json
import os

import numpy as np
import pytest
import torch

from precondition_amd import blocking
from precondition_amd.state import PreconditionerType


@pytest.fixture(scope="module")
def book(golden_dir):
  with open(os.path.join(golden_dir, "bookkeeping.json")) as f:
    return json.load(f)


def test_merge_small_dims(book):
  assert len(book["merge_small_dims"]) > 100
  for c in book["merge_small_dims"]:
    assert blocking.merge_small_dims(c["shape"], c["max_dim"]) == c["out"], c
  # docstring examples of DS:1297-1298
  assert blocking.merge_small_dims([1, 2, 512, 1, 2048, 1, 3, 4], 1024) == [1024, 2048, 12]
  assert blocking.merge_small_dims([1, 2, 768, 1, 2048], 1024) == [2, 768, 2048]


def test_block_partitioner(book):
  for c in book["block_partitioner"]:
    x = np.random.default_rng(c["seed"]).standard_normal(c["shape"]).astype(np.float32)
    t = torch.from_numpy(x)
    bp = blocking.BlockPartitioner(t, c["block_size"])
    assert [[int(v) for v in s] for s in bp.split_sizes()] == c["split_sizes"]
    for s in bp.split_sizes():
      assert s.dtype == np.int32
    parts = bp.partition(t)
    assert [list(p.shape) for p in parts] == c["part_shapes"]
    assert [float(p.reshape(-1)[0]) for p in parts] == c["part_first"]
    assert [float(p.reshape(-1)[-1]) for p in parts] == c["part_last"]
    for p, s in zip(parts, c["part_sum"]):
      assert np.isclose(float(p.double().sum()), s, rtol=1e-12, atol=1e-9)
      assert p.data_ptr() >= t.data_ptr()  # a view, not a copy
    assert torch.equal(bp.merge_partitions(parts), t)


def test_preconditioner_shapes_exponents(book):
  assert len(book["preconditioner"]) > 500
  for c in book["preconditioner"]:
    param = torch.empty(c["shape"], dtype=torch.float32, device="meta")
    pc = blocking.Preconditioner(param, c["block_size"], c["merge_block"],
                                 c["best_effort"], PreconditionerType(c["ptype"]),
                                 c["rank"])
    assert list(pc._transformed_shape) == c["transformed"], c
    assert pc.shapes_for_preconditioners() == c["shapes"], c
    assert pc.exponent_for_preconditioner() == c["exponent"], c
    assert pc.should_precondition_dims() == c["should"], c


def test_vit_b_statistics_census():
  """ViT-B/16 tree at block_size=1024 has 395 statistics."""
  shapes = ([[16, 16, 3, 768], [768], [1, 1, 768], [1, 197, 768]] +
            12 * [[768], [768], [768, 12, 64], [12, 64], [768, 12, 64], [12, 64],
                  [768, 12, 64], [12, 64], [12, 64, 768], [768], [768], [768],
                  [768, 3072], [3072], [3072, 768], [768]] +
            [[768], [768], [768, 1000], [1000]])
  assert len(shapes) == 200
  census = {}
  for s in shapes:
    pc = blocking.Preconditioner(torch.empty(s, device="meta"), 1024, 4096, True)
    for shp in pc.shapes_for_preconditioners():
      key = (shp[0], pc.exponent_for_preconditioner())
      census[key] = census.get(key, 0) + 1
  assert census == {(768, 4): 172, (768, 2): 112, (1024, 4): 72, (1024, 2): 36,
                    (1000, 4): 1, (1000, 2): 1, (197, 4): 1}
  assert sum(census.values()) == 395


def test_pad_square_matrix(book):
  for c in book["pad_square_matrix"]:
    k = c["k"]
    m = torch.arange(k * k, dtype=torch.float32).reshape(k, k) + 1
    out = blocking.pad_square_matrix(m, c["max_size"])
    assert out.tolist() == c["out"]
  # DST:35-72 error cases
  with pytest.raises(ValueError, match="Must have rows == cols"):
    blocking.pad_square_matrix(torch.zeros(2, 3), 5)
  with pytest.raises(ValueError, match="Must have cols <= max_size"):
    blocking.pad_square_matrix(torch.zeros(6, 6), 5)
  assert blocking.pad_vector(torch.ones(3), 5).tolist() == [1, 1, 1, 0, 0]


def test_batch_unbatch_order(book):
  from precondition_amd import comm
  for c in book["batch_unbatch"]:
    n, d = c["n"], c["num_devices"]
    x = [torch.full((2, 2), float(i)) for i in range(n)]
    b = blocking.batch(x, d)
    assert list(b.shape) == c["batched_shape"]
    assert [int(b[r, j, 0, 0]) for r in range(d) for j in range(b.shape[1])] == c["owner_of"]
    assert [int(v.reshape(-1)[0]) for v in blocking.unbatch(b)] == c["unbatched_order"]
    # the ownership table the sharded root uses is the same chunking
    owner = comm.reference_ownership(n, d)
    assert owner == [i // (n // d) for i in range(n)]


def test_reference_ownership_with_padding():
  from precondition_amd import comm
  # 395 statistics on 8 ranks: padded to 400, b = 50
  owner = comm.reference_ownership(395, 8)
  assert owner[0] == 0 and owner[49] == 0 and owner[50] == 1 and owner[394] == 7
  assert [owner.count(r) for r in range(8)] == [50] * 7 + [45]
  lpt = comm.cost_balanced_ownership([768] * 6 + [1024] * 2, [4] * 8, 2)
  assert sorted(set(lpt)) == [0, 1]


def test_fd_pack_unpack_roundtrip_bit_exact():
  """DST:753-768: pack(unpack(x)) == x, bit for bit, and the exact cell layout of
  DS:555-592."""
  from precondition_amd import low_rank
  rng = np.random.default_rng(0)
  for d, r in ((12, 4), (40, 3), (9, 1)):
    vecs = torch.from_numpy(rng.standard_normal((d, r)).astype(np.float32))
    defl = torch.from_numpy(rng.random(r).astype(np.float32))
    inv = torch.from_numpy(rng.random(r).astype(np.float32))
    packed = low_rank._fd_low_rank_pack(vecs, defl, inv, 0.25, 0.75, True, r)
    assert tuple(packed.shape) == (d, r + 2)
    assert torch.equal(packed[:, :r], vecs) and torch.equal(packed[:r, -2], inv)
    assert float(packed[0, -1]) == 0.25 and float(packed[1, -1]) == 0.75
    assert torch.equal(packed[-r:, -1], defl) and float(packed[-1, -2]) == 1.0
    un = low_rank._fd_low_rank_unpack(packed, r)
    again = low_rank._fd_low_rank_pack(un[0], un[1], un[2], un[3], un[4], un[5], r)
    assert torch.equal(again, packed)
    ev, inv2, const, hz = low_rank._low_rank_unpack(packed, -r)
    assert torch.equal(inv2, inv) and float(const) == 0.25 and bool(hz)


def test_contraction_lengths_follow_the_statistics_order():
  """Preconditioner.contraction_lengths(): per statistic the columns of the block matricised along that
  axis (n * k = block size), in the order of shapes_for_preconditioners(), for every preconditioner type."""
  import itertools
  import torch
  from precondition_amd.blocking import Preconditioner
  from precondition_amd.state import PreconditionerType
  for shape in [(768, 3072), (3072,), (768, 12, 64), (5,), (130, 200, 3)]:
    for ptype in (PreconditionerType.ALL, PreconditionerType.INPUT, PreconditionerType.OUTPUT):
      pc = Preconditioner(torch.empty(shape), 128, 4096, True, preconditioner_type=ptype)
      shapes, ks = pc.shapes_for_preconditioners(), pc.contraction_lengths()
      assert len(shapes) == len(ks)
      blocks = list(itertools.product(*pc._partitioner.split_sizes()))
      per_block = len(shapes) // len(blocks)
      for b, blk in enumerate(blocks):
        total = 1
        for d in blk:
          total *= int(d)
        for s_, k in zip(shapes[b * per_block:(b + 1) * per_block], ks[b * per_block:(b + 1) * per_block]):
          assert s_[0] * k == total
