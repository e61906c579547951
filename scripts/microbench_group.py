#!/usr/bin/env python3
"""Dense-only queries submitted n at a time to anrag_dense_search_device: the library scans them in groups of 4
per launch (api.hip kScanGroup), each query its own pass over the matrix.  n_per_call = 1 shows the one-launch-per-
query rate.  usage: python scripts/microbench_group.py rows [n_per_call]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anrag import _native as nat, synth
from anrag.index import Index
n = int(sys.argv[1]); per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, k = 768, 10
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 1234, dev)
Q, _ = synth.dense_queries(E, 64, 4321)
torch.cuda.synchronize()
idx = Index(0); idx.dense_load((E.data_ptr(), n, d))
out = torch.zeros((64, k, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ref = (Q @ E.T).topk(k, dim=1).indices.cpu().numpy()
lib = nat.load_library()
def run(calls):
    for c in range(calls):
        q0 = (c * per) % 64
        m = min(per, 64 - q0)
        nat.check(lib.anrag_dense_search_device(idx.handle, Q[q0].data_ptr(), m, k, None, out[q0].data_ptr()))
    idx.sync()
run(16)
got = out.cpu().numpy().view(np.float64).reshape(64, k, 2)
ids = out.cpu().numpy()[:, :, 1]
print("ids match torch:", bool((ids == ref).all()))
calls = 2000 // per
t0 = time.perf_counter(); run(calls); dt = time.perf_counter() - t0
print(f"rows={n} queries_per_call={per}: {calls*per/dt:.0f} q/s ({dt/(calls*per)*1e6:.1f} us/query)")
