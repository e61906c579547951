#!/usr/bin/env python3
"""Stress of the query pipeline where every kernel lasts microseconds (tiny corpora): thousands of hybrid batch calls of
1..40 queries, mixed with single calls, candidates-mode calls and BM25-only / dense-only queries, each checked against
the single-query answers.  usage: python scripts/stress_small_batches.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anrag.bm25_index import Bm25Index
from anrag.index import Index
import faulthandler; faulthandler.enable()

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
calls = rounds = 0
while time.time() < t_end:
    rounds += 1
    n = int(rng.choice([1, 3, 17, 65, 300, 1000, 5000]))
    d = int(rng.choice([8, 24, 64, 384, 768]))
    vocab = int(rng.choice([5, 200, 2000]))
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
    corpus = [[str(t) for t in rng.zipf(1.3, size=int(rng.integers(0, 30))) % vocab] for _ in range(n)]
    if not any(corpus):
        corpus[0] = ["0"]
    bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
    with Index(0) as idx:
        idx.dense_load(e)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        for _ in range(40):
            nb = int(rng.integers(1, 41))
            qs = e[rng.integers(0, n, size=nb)] + 0.1 * rng.standard_normal((nb, d)).astype(np.float32)
            tls = [bi.term_ids([str(t) for t in rng.integers(0, vocab + 2, size=int(rng.integers(0, 6)))]) for _ in range(nb)]
            kb, tb = int(rng.choice([1, 10, 25, 64])), int(rng.choice([1, 10, 30]))
            ids, scores, counts = idx.hybrid_search_batch(qs, tls, kb, 5.0, 1.0, 40.0, tb)
            calls += 1
            for i in rng.choice(nb, size=min(nb, 3), replace=False):
                wid, ws = idx.hybrid_search(qs[i], tls[i], kb, 5.0, 1.0, 40.0, tb)
                c = int(counts[i])
                assert c == len(wid) and ids[i, :c].tolist() == wid.tolist() and scores[i, :c].tolist() == ws.tolist(), (n, d, nb, i)
print(f"stress ok: {rounds} corpora, {calls} batch calls in {budget:.0f} s")
