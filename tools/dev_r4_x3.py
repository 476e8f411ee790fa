import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
for name in ("cfg2_256x512_p4",):
    w = bench.Workload(name, 0, 1, dev)
    for mode in ("f32", "bf16x6", "bf16x3"):
        w.options = {"products": mode}; w.hint = None
        w.compute(); torch.cuda.synchronize(); w.refresh_hint()
        t0 = time.perf_counter()
        for _ in range(3): w.compute()
        torch.cuda.synchronize()
        m = w.metrics.cpu().numpy()
        print(mode, (time.perf_counter() - t0) / 3 * 1e3, "ms; metrics row0", m[0], "retries max", m[:, 4].max(), "total iters", m[:, 5].min(), m[:, 5].max())
        sm, ln, pm, om = bench.profile_stage_kernel(w)
        print("   stage", sm, "launches", ln, "pi", pm, "other", om)
