#!/usr/bin/env python3
"""`anrag_hybrid_search` (host operands, host-synchronous: what the Python SearchEngine calls per query) on the reference's
corpus shape, one caller thread and four.  usage: python scripts/measure_host_hybrid_small.py [rows] [dim]
(ANRAG_HYBRID_LANES_MAX_MB=0: the three-stream pipeline instead of the lanes)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9609
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 7, dev)
Q, _ = synth.dense_queries(E, 64, 8)
post = synth.bm25_postings(n, 50_000, 9, dev)
df = post["df"].cpu().numpy()
idf = synth.bm25_idf(df, n)
terms = [np.asarray(t, np.int32) for t in synth.bm25_queries(post, 64, 10)]
q = Q.cpu().numpy()
torch.cuda.synchronize()
with Index(0) as idx:
    idx.dense_load((E.data_ptr(), n, d))
    idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                  (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], float(post["total_len"]) / n,
                  synth.BM25_K1, synth.BM25_B)
    ask = lambda i: idx.hybrid_search(q[i % 64], terms[i % 64], 25, 5.0, 1.0, 40, 10)
    for i in range(200):
        ask(i)
    lat = []
    for i in range(2000):
        t0 = time.perf_counter()
        ask(i)
        lat.append(time.perf_counter() - t0)
    print(f"{n} x {d}: one caller p50 {np.median(lat)*1e6:.1f} us, p99 {np.percentile(lat, 99)*1e6:.1f} us, "
          f"{1/np.mean(lat):.0f} q/s", flush=True)
    with ThreadPoolExecutor(4) as pool:
        list(pool.map(ask, range(400)))
        t0 = time.perf_counter()
        list(pool.map(ask, range(8000)))
        dt = time.perf_counter() - t0
    print(f"{n} x {d}: four caller threads {8000/dt:.0f} q/s", flush=True)
