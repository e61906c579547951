#!/usr/bin/env python3
"""PCIe-inclusive rate: the host-pointer entry point `anrag_hybrid_search` (query + term ids H2D, results D2H,
one host sync per query) on the same 1M x 768 hybrid workload as bench.py.  Not the headline number."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anrag import synth
from anrag.index import Index

rows, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768
dev = torch.device("cuda", 0)
E = synth.dense_corpus(rows, dim, 1234, dev)
torch.cuda.synchronize()
idx = Index(0); idx.dense_load((E.data_ptr(), rows, dim))
post = synth.bm25_postings(rows, 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), rows)
torch.cuda.synchronize()
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], post["total_len"] / rows, 1.7, 0.83)
Q, _ = synth.dense_queries(E, 64, 4321); Qh = Q.cpu().numpy()
terms = synth.bm25_queries(post, 64, 99)
for i in range(20):
    idx.hybrid_search(Qh[i % 64], terms[i % 64], 25, 5.0, 1.0, 40, 10)
n = 300
t0 = time.perf_counter()
for i in range(n):
    idx.hybrid_search(Qh[i % 64], terms[i % 64], 25, 5.0, 1.0, 40, 10)
dt = (time.perf_counter() - t0) / n
print(f"host-pointer hybrid query (H2D + kernels + D2H + sync): {dt*1e3:.3f} ms/query = {1/dt:.0f} q/s")
