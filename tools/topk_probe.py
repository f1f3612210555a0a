#!/usr/bin/env python
"""Where the selection kernel's time goes (round 6): k_topk_metrics on a [500, 20108] score matrix with RTX_TOPK_STOP = 1..5 (the kernel
returns after: 1 row load + per-thread maxima, 2 the bound L, 3 candidates placed, 4 candidates ranked, 5 relevance; 0 = whole kernel)."""
import os, sys, time
import numpy as np, torch
from scipy.sparse import csr_matrix
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectorch_amd.engine import CsrMatrix, topk_metrics
B, I = int(sys.argv[1]) if len(sys.argv) > 1 else 500, 20108
rng = np.random.RandomState(0)
torch.manual_seed(0)
sc = torch.randn(B, I, device="cuda")
if len(sys.argv) > 2 and sys.argv[2] == "ties":      # two equal scores among every row's top ten: the exact-rank pass runs in every workgroup
    top = sc.topk(10, dim=1).indices
    sc.scatter_(1, top[:, 3:4], sc.gather(1, top[:, 2:3]))
held = np.zeros((B, I)); 
for b in range(B): held[b, rng.choice(I, 20, replace=False)] = 1
hm = CsrMatrix(csr_matrix(held))
rows = torch.arange(B, dtype=torch.int32, device="cuda")
big = torch.empty(64 << 20, device="cuda")
for stop in [0, 1, 2, 3, 4, 5, 0]:
    os.environ["RTX_TOPK_STOP"] = str(stop)
    for cold in (0, 1):
        ts = []
        for it in range(12):
            if cold: big.fill_(1.0)          # 256 MB through the caches: the score rows come from HBM
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); res = topk_metrics(sc, hm, rows, [100, 50]); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("stop %d %s: median %.1f us  min %.1f" % (stop, "cold" if cold else "warm", sorted(ts)[len(ts) // 2], min(ts)))
