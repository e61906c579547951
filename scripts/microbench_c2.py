#!/usr/bin/env python3
"""C2 as BASELINE states it: dense top-10 at ONE query per call (no grouping by the caller), operands in HBM.
Throughput of a stream of single-query submissions and the latency of one query alone.
usage: python scripts/microbench_c2.py [rows] [dim] [queries]   (env ANRAG_SCAN_LANES=1..4: scan streams in rotation)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 1234, dev)
Q, rows = synth.dense_queries(E, 64, 4321)
torch.cuda.synchronize()
idx = Index(0)
idx.dense_load((E.data_ptr(), n, d))
out = torch.zeros((64, 10, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()


def run(m):
    for i in range(m):
        q = i % 64
        idx.dense_search_device(Q[q].data_ptr(), 1, 10, 0, out[q].data_ptr())
    idx.sync()


run(256)
ok = bool(torch.equal(out[:, 0, 1].cpu(), rows.cpu()))
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    run(iters)
    best = min(best, (time.perf_counter() - t0) / iters)
lat = []
for i in range(200):
    t0 = time.perf_counter()
    idx.dense_search_device(Q[i % 64].data_ptr(), 1, 10, 0, out[i % 64].data_ptr())
    idx.sync()
    lat.append(time.perf_counter() - t0)
byts = n * d * 4
print(f"{n} x {d}, one query per call, ANRAG_SCAN_LANES={os.environ.get('ANRAG_SCAN_LANES', 'default')}: "
      f"{best*1e6:.2f} us/query = {byts/best/1e12:.3f} TB/s = {byts/best/8e12*100:.1f} % of 8 TB/s; "
      f"one query alone p50 {np.median(lat)*1e6:.1f} us p99 {np.percentile(lat, 99)*1e6:.1f} us; top-1 ok {ok}")
