#!/usr/bin/env python3
"""Full-ranking mode (retrieval_eval.py evaluates 7 of 9 configurations with
similarity_k = common_sections_n = 12000): host-synchronous dense / BM25 / WRRF calls with k > 64, i.e. the
score-array + rocPRIM radix-sort path.
usage: python scripts/microbench_largek.py [rows] [dim] [k] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9609
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 1234, dev)
Q, _ = synth.dense_queries(E, 16, 4321)
post = synth.bm25_postings(n, 50_000 if n < 100_000 else 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
terms = synth.bm25_queries(post, 16, 99)
torch.cuda.synchronize()
idx = Index(0)
idx.dense_load((E.data_ptr(), n, d))
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], float(post["total_len"]) / n,
              synth.BM25_K1, synth.BM25_B)
q = Q.cpu().numpy()
kk = min(k, n + 5)


def timed(fn):
    for i in range(3):
        fn(i)
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    return (time.perf_counter() - t0) / iters * 1e3


dense = lambda i: idx.dense_search(q[i % 16], kk)
bm25 = lambda i: idx.bm25_search(np.asarray(terms[i % 16], np.int32), kk)
t_dense, t_bm25 = timed(dense), timed(bm25)
dd, _, dc = dense(0)
bd, _, bc = bm25(0)
lists = [dd[0, : int(dc[0])].tolist(), bd[: bc].tolist()]
t_wrrf = timed(lambda i: idx.wrrf(lists, [5.0, 1.0], 40.0, kk))
print(f"{n} x {d}, k={kk}: dense full ranking {t_dense:.3f} ms, BM25 full ranking {t_bm25:.3f} ms, "
      f"WRRF over {len(lists[0])}+{len(lists[1])} ids {t_wrrf:.3f} ms -> {1e3/(t_dense+t_bm25+t_wrrf):.0f} hybrid q/s")
