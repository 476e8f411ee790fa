"""Dev: one cfg2 / headline root call for per-launch kernel timing under rocprofv3."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_256x512_p4"
st, p = bench.make_blocks(name, 0, dev)
mats = list(st.unbind(0))
for _ in range(2):
  r, m = K.matrix_inverse_pth_root_batched(mats, [p] * len(mats))
torch.cuda.synchronize()
print(name, "iters", m[:, 1].min().item(), m[:, 1].max().item())
