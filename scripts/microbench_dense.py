#!/usr/bin/env python3
"""Quick on-box timing of K1 (dense scan + top-k) with device-resident operands.
usage: python scripts/microbench_dense.py [n_rows] [dim] [k] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag.index import Index
from anrag import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
E = torch.randn((n, d), device=dev, dtype=torch.float32, generator=g)
E /= E.norm(dim=1, keepdim=True)
Q = E[:64] + 0.05 * torch.randn((64, d), device=dev, generator=g)
Q /= Q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
idx = Index(0)
torch.cuda.synchronize()
idx.dense_load((E.data_ptr(), n, d))
out = torch.empty((64 * k, 2), device=dev, dtype=torch.float64)
# correctness spot check against torch
idx.dense_search_device(Q.data_ptr(), 4, k, 0, out.data_ptr()); idx.sync()
cand = out.cpu().numpy().view(nat.CANDIDATE_DTYPE).reshape(64, k)
ref = (Q[:4] @ E.T).topk(k, dim=1)
print("ids match torch:", np.array_equal(cand["doc"][:4], ref.indices.cpu().numpy()),
      " max|dscore|:", float(np.abs(cand["score"][:4] - ref.values.cpu().numpy()).max()))
idx.profile(True)
for _ in range(5):
    idx.dense_search_device(Q.data_ptr(), 1, k, 0, out.data_ptr())
idx.sync(); idx.profile_reset()
t0 = time.perf_counter()
for i in range(iters):
    idx.dense_search_device(Q.data_ptr() + (i % 64) * d * 4, 1, k, 0, out.data_ptr())
idx.sync()
wall = (time.perf_counter() - t0) / iters
ms, launches = idx.profile_read(nat.KERNEL_DENSE_SCAN)
ms2, l2 = idx.profile_read(nat.KERNEL_SELECT)
gb = n * d * 4 / 1e9
print(f"n={n} d={d} k={k}: wall/query {wall*1e3:.3f} ms  scan kernel {ms/launches:.4f} ms "
      f"({gb/(ms/launches)*1e3:.0f} GB/s, {gb/(ms/launches)*1e3/8000*100:.1f}% of 8 TB/s)  merge {ms2/l2*1e3:.1f} us")
