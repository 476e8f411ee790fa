"""Dev (round 6): a few ViT-B tree recomputes (Newton roots) alone, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, bench
dev = torch.device("cuda:0")
w = bench.VitBWorkload(0, 1, dev, None)
for _ in range(2):
  w.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
  w.step()
torch.cuda.synchronize()
print("vit_b step %.1f ms" % ((time.perf_counter() - t0) / 4 * 1e3))
