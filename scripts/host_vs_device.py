#!/usr/bin/env python3
"""Is the per-query floor host enqueue time or device chain time?  python scripts/host_vs_device.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anrag import synth, _native as nat
from anrag.index import Index
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
E = synth.dense_corpus(rows, 768, 1234, dev)
torch.cuda.synchronize()
idx = Index(0); idx.dense_load((E.data_ptr(), rows, 768))
post = synth.bm25_postings(rows, 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), rows)
torch.cuda.synchronize()
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], post["total_len"] / rows, 1.7, 0.83)
Q, _ = synth.dense_queries(E, 64, 4321)
terms = synth.bm25_queries(post, 64, 99)
T = torch.full((64, 16), -1, dtype=torch.int32, device=dev)
for i, t in enumerate(terms): T[i, :len(t)] = torch.from_numpy(t).to(dev)
nt = [len(t) for t in terms]
out = torch.zeros((64, 10, 2), dtype=torch.int64, device=dev); cnt = torch.zeros(64, dtype=torch.int32, device=dev)
lib = nat.load_library()
def step(i):
    qi = i % 64
    nat.check(lib.anrag_hybrid_search_device(idx.handle, Q[qi].data_ptr(), T[qi].data_ptr(), nt[qi], 25, 5.0, 1.0, 40.0, 10,
                                             None, None, out[qi].data_ptr(), cnt[qi:].data_ptr()))
for i in range(200): step(i)
idx.sync()
n = 3000
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter()
idx.sync()
t2 = time.perf_counter()
print(f"rows={rows}: host enqueue {(t1-t0)/n*1e6:.1f} us/query; total {(t2-t0)/n*1e6:.1f} us/query (device drains {(t2-t1)*1e3:.2f} ms after the loop)")
