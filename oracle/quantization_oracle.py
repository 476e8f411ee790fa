"""CPU oracle for the quantized optimizer-state path.  TEST INFRASTRUCTURE ONLY.

NumPy float32 restatement of ``precondition/quantization_utils.py`` (cited as
``QU:<line>``) of google-research/precondition: ``QuantizedValue.quantize``
(QU:45-95) and ``QuantizedValue.to_float`` (QU:97-113).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package never does.

Parity pin: ``tools/gen_golden.py`` (section ``quant``) runs the reference's own
``QuantizedValue`` (imported from /root/reference over the NumPy stand-in for
jax) on seeded inputs, stores inputs + codes + diagonals + bucket sizes under
``tests/golden/quantization.npz`` and asserts this oracle agrees bit for bit;
``tests/test_oracle_golden.py`` re-checks it against the committed vectors
(bit-exact everywhere: the path is elementwise IEEE float32 arithmetic + a
round-half-to-even + an integer cast, no BLAS).

Not pinned (the reference leaves it to XLA's float->int cast): NaN/Inf inputs.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# QU:56-63: the most negative code is never used
NUM_BUCKETS = {np.dtype(np.int8): 127.0, np.dtype(np.int16): 32767.0}


def quantize(fvalue, quantized_dtype, extract_diagonal=False):
  """QU:45-95.  Returns (codes, diagonal, bucket_size); [] for absent parts."""
  fvalue = np.asarray(fvalue, dtype=F32)
  qd = np.dtype(quantized_dtype)
  if qd == np.dtype(np.float32):  # QU:48-49
    return fvalue, [], []
  if qd not in NUM_BUCKETS:
    raise ValueError(f"Quantized dtype {quantized_dtype} not supported.")  # QU:64
  num_buckets = F32(NUM_BUCKETS[qd])
  if extract_diagonal and fvalue.ndim != 2:  # QU:67-69
    raise ValueError("Input array must be 2D to work with extract_diagonal.")
  diagonal = []
  if extract_diagonal:  # QU:71-75
    diagonal = np.array(np.diag(fvalue), dtype=F32)
    fvalue = (fvalue - np.diag(diagonal)).astype(F32)
  if fvalue.ndim < 1:  # QU:80-83
    raise ValueError("Input array must have a strictly positive number of dimensions.")
  max_abs = np.max(np.abs(fvalue), axis=0)  # QU:85
  bucket_size = (max_abs / num_buckets).astype(F32)  # QU:86
  bs_expanded = bucket_size[np.newaxis, ...]
  bs_nonzero = np.where(bs_expanded > 0.0, bs_expanded, np.ones_like(bs_expanded))  # QU:89-90
  ratio = (fvalue / bs_nonzero).astype(F32)  # QU:91
  quantized = np.round(ratio)  # QU:93 (half to even, like jnp.round)
  return quantized.astype(qd), diagonal, bucket_size


def to_float(quantized, diagonal, bucket_size, quantized_dtype, extract_diagonal=False):
  """QU:97-113."""
  qd = np.dtype(quantized_dtype)
  if qd == np.dtype(np.float32):
    return quantized
  bs = np.asarray(bucket_size, dtype=F32)[np.newaxis, ...]
  val = (np.asarray(quantized).astype(F32) * bs).astype(F32)  # QU:109
  if extract_diagonal:
    val = (val + np.diag(np.asarray(diagonal, dtype=F32))).astype(F32)  # QU:111
  return val
