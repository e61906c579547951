#!/usr/bin/env python3
"""`anrag_hybrid_search_batch` (k <= 64 hybrid query lists through the device pipeline, host operands in and out) and
the host-synchronous single call, on a corpus of the given shape.
usage: python scripts/microbench_hybrid_batch.py [rows] [dim] [queries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9609
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 1234, dev)
Q, _ = synth.dense_queries(E, nq, 4321)
post = synth.bm25_postings(n, 50_000 if n < 100_000 else 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
terms = [np.asarray(t, np.int32) for t in synth.bm25_queries(post, min(nq, 128), 99)]
terms = [terms[i % len(terms)] for i in range(nq)]
torch.cuda.synchronize()
idx = Index(0)
idx.dense_load((E.data_ptr(), n, d))
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], float(post["total_len"]) / n,
              synth.BM25_K1, synth.BM25_B)
q = Q.cpu().numpy()
idx.hybrid_search_batch(q, terms, 25, 5.0, 1.0, 40.0, 15)
t0 = time.perf_counter()
ids, sc, cnt = idx.hybrid_search_batch(q, terms, 25, 5.0, 1.0, 40.0, 15)
tb = (time.perf_counter() - t0) / nq
lat = []
for i in range(200):
    t0 = time.perf_counter()
    a, b = idx.hybrid_search(q[i], terms[i], 25, 5.0, 1.0, 40.0, 15)
    lat.append(time.perf_counter() - t0)
    assert a.tolist() == ids[i, :cnt[i]].tolist()
print(f"{n} x {d} hybrid k=25 top-15: list of {nq}: {tb*1e6:.1f} us/query ({1/tb:.0f} q/s); one call at a time: p50 "
      f"{np.median(lat)*1e6:.1f} us, p99 {np.percentile(lat, 99)*1e6:.1f} us")
