#!/usr/bin/env python3
"""Full-ranking mode for query LISTS (`anrag_rank_batch`, rank_batch.hip) against the per-query entry points it
replaces for retrieval_eval's k = 12,000 configurations: wall-clock per query through the C ABI (host operands in,
host results out) for dense-only, BM25-only and dense + BM25 fused.
usage: python scripts/microbench_rank.py [rows] [dim] [k] [queries] [per-query sample]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import synth
from anrag.index import Index, rank_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9609
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
sample = int(sys.argv[5]) if len(sys.argv) > 5 else 32
dev = torch.device("cuda:0")
E = synth.dense_corpus(n, d, 1234, dev)
Q, _ = synth.dense_queries(E, nq, 4321)
post = synth.bm25_postings(n, 50_000 if n < 100_000 else 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
terms = [np.asarray(t, np.int32) for t in synth.bm25_queries(post, min(nq, 256), 99)]
terms = [terms[i % len(terms)] for i in range(nq)]
torch.cuda.synchronize()
idx = Index(0)
idx.dense_load((E.data_ptr(), n, d))
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], float(post["total_len"]) / n,
              synth.BM25_K1, synth.BM25_B)
q = Q.cpu().numpy()
kk = min(k, n)
dense_leg = dict(index=idx, weight=5.0, queries=q)
bm25_leg = dict(index=idx, weight=1.0, term_lists=terms)


def batch(legs, fuse, **kw):
    rank_batch(legs, nq, k, 40, k, id_space=n if fuse else 0, **kw)  # warm: scratch pool, LDS attribute
    t0 = time.perf_counter()
    out = rank_batch(legs, nq, k, 40, k, id_space=n if fuse else 0, **kw)
    return (time.perf_counter() - t0) / nq * 1e6, out


def per_query(fn):
    for i in range(2):
        fn(i)
    t0 = time.perf_counter()
    for i in range(sample):
        fn(i)
    return (time.perf_counter() - t0) / sample * 1e6


res = {"rows": n, "dim": d, "k": k, "queries": nq}
res["batch_dense_us"], (ids_d, _, cnt_d) = batch([dense_leg], False)
res["batch_bm25_us"], (ids_b, _, cnt_b) = batch([bm25_leg], False)
res["batch_hybrid_us"], (ids_f, _, cnt_f) = batch([dense_leg, bm25_leg], True)
expect = ids_f[:, 3].copy()
res["batch_hybrid_rank_only_us"], (_, _, _, ranks) = batch([dense_leg, bm25_leg], True, expect=expect, want_ids=False)
res["rank_only_ok"] = bool(np.all(ranks == 4))
res["per_query_dense_us"] = per_query(lambda i: idx.dense_search(q[i], kk))
res["per_query_bm25_us"] = per_query(lambda i: idx.bm25_search(terms[i], kk))
dd, _, dc = idx.dense_search(q[0], kk)
bd, _, bc = idx.bm25_search(terms[0], kk)
lists = [dd[0, :int(dc[0])], bd[:bc]]
res["per_query_wrrf_us"] = per_query(lambda i: idx.wrrf(lists, [5.0, 1.0], 40.0, kk))
res["per_query_hybrid_us"] = res["per_query_dense_us"] + res["per_query_bm25_us"] + res["per_query_wrrf_us"]
fid, _ = idx.wrrf(lists, [5.0, 1.0], 40.0, kk)
res["ids_equal_per_query"] = bool(ids_d[0, :cnt_d[0]].tolist() == dd[0, :int(dc[0])].tolist()
                                  and ids_b[0, :cnt_b[0]].tolist() == bd[:bc].tolist()
                                  and ids_f[0, :cnt_f[0]].tolist() == fid.tolist())
res["speedup_hybrid"] = res["per_query_hybrid_us"] / res["batch_hybrid_us"]
print(json.dumps({a: (round(b, 2) if isinstance(b, float) else b) for a, b in res.items()}))
