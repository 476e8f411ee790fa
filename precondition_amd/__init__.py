"""precondition_amd — MI355X-native preconditioner-compute path of Distributed Shampoo.

Drop-in for the hot path of google-research/precondition's
``precondition/distributed_shampoo.py``: statistics accumulation and the batched
matrix inverse p-th root, behind the reference's GradientTransformation /
ShampooState surface.  All arithmetic runs in hand-written HIP kernels for
gfx950 (libprecondition_amd.so, C-ABI in include/ps_api.h).
"""
__version__ = "0.1.0"
