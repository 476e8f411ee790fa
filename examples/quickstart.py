"""Quick start: train a small MLP with precondition_amd.distributed_shampoo on one MI355X.

    python examples/quickstart.py

Same call pattern as the reference's optax GradientTransformation:
    optim = distributed_shampoo(lr, block_size, **kwargs)
    state = optim.init(params)
    updates, state = optim.update(grads, state, params);  params <- params + updates
Multi-GPU: start one process per GPU (torchrun), init_process_group("nccl"), pass
batch_axis_name=<process group>; the statistics blocks are partitioned over the ranks and the
preconditioners all-gathered over RCCL.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import precondition_amd as pa


def main():
  dev = torch.device("cuda:0")
  torch.manual_seed(0)
  x = torch.randn(4096, 256, device=dev)
  w_true = torch.randn(256, 10, device=dev) / 16
  y = torch.tanh(x @ w_true).detach()
  params = {"w1": (torch.randn(256, 384, device=dev) / 16).requires_grad_(),
            "b1": torch.zeros(384, device=dev, requires_grad=True),
            "w2": (torch.randn(384, 10, device=dev) / 20).requires_grad_()}
  optim = pa.distributed_shampoo(
      2e-3, block_size=128, beta1=0.9, beta2=0.99, graft_type=pa.GraftingType.RMSPROP,
      preconditioning_compute_steps=5, start_preconditioning_step=5, matrix_epsilon=1e-6)
  state = optim.init({k: v.detach() for k, v in params.items()})
  first = last = None
  for step in range(200):
    h = torch.relu(x @ params["w1"] + params["b1"])
    loss = torch.mean((h @ params["w2"] - y) ** 2)
    grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
    updates, state = optim.update(grads, state, {k: v.detach() for k, v in params.items()})
    with torch.no_grad():
      for k in params:
        params[k] += updates[k]
    if step % 25 == 0 or step == 199:
      print(f"step {step:3d}  loss {loss.item():.5f}")
    first = loss.item() if first is None else first
    last = loss.item()
  tm = state.stats["w1"].training_metrics
  print("inverse-root errors of w1's blocks:", [f"{e:.1e}" for e in tm.inverse_pth_root_errors.tolist()])
  assert last < 0.2 * first, (first, last)
  print("ok")


if __name__ == "__main__":
  main()
