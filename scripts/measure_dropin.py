#!/usr/bin/env python3
"""Throughput of the DROP-IN surface: `RetrievalEvaluationSystem.retrieve_documents(...)` (the reference's
query_rag_retrieval.py:149 signature) over a synthetic corpus of bench.py's shape -- numpy query in, chunk-id
strings out, one host-synchronous call per query.  Compared with the raw C call underneath
(`anrag_hybrid_search`) so that the Python glue's share is visible.
usage: python scripts/measure_dropin.py [rows] [dim] [queries]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
from anrag import synth
from anrag.bm25_index import Bm25Index
from anrag.config import Config, InfoSource, LOCAL_ENCODER_KEY
from anrag.database_manager import Bm25Proxy, DenseHandle
from anrag.index_io import Section
from anrag.query_rag_retrieval import RetrievalEvaluationSystem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
t0 = time.time()
E = synth.dense_corpus(n, d, 1234, dev)
Q, _ = synth.dense_queries(E, 64, 4321)
post = synth.bm25_postings(n, 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
terms = synth.bm25_queries(post, 64, 99)
torch.cuda.synchronize()
e_host = E.cpu().numpy()
q_host = Q.cpu().numpy()
del E
ids = ["CG%d_%d" % (i % 300, i) for i in range(n)]
sources = ["CG%d" % (i % 300) if i % 300 >= 45 else "QS%d" % (i % 300) for i in range(n)]
df = pd.DataFrame({"id": ids, "document": [""] * n, "source": sources, "embedding": list(e_host)})
DenseHandle(e_host, sources).bind(df)
st = Bm25Index.__new__(Bm25Index)
st.k1, st.b, st.epsilon = synth.BM25_K1, synth.BM25_B, 0.05
st.vocab = {"t%d" % t: t for t in range(200_000)}
st.n_docs = n
st.doc_len = np.asarray(post["doc_len"], dtype=np.int32)
st.avgdl = float(post["total_len"]) / n
st.post_doc = post["post_doc"].cpu().numpy()
st.post_tf = post["post_tf"].cpu().numpy()
st.indptr = np.asarray(post["indptr"], dtype=np.int64)
st.idf = idf
st.average_idf = float(np.mean(idf))
proxy = Bm25Proxy(st, sources)
sections = [Section("", {"id": i, "source": s}) for i, s in zip(ids, sources)]
system = RetrievalEvaluationSystem.__new__(RetrievalEvaluationSystem)
system.config = Config()
from anrag.search_engine import SearchEngine
system.search_engine = SearchEngine(None, None)
system.voyage_client = None
system.fused = True
system.embeddings_data = {InfoSource.NICE: {LOCAL_ENCODER_KEY: df}}
system.bm25_data = {InfoSource.NICE: (proxy, sections, ids)}
tokens = [["t%d" % t for t in tl.tolist()] for tl in terms]
weights = {LOCAL_ENCODER_KEY: 5.0, "BM25": 1.0}
print("built %d x %d in %.0f s" % (n, d, time.time() - t0), flush=True)


def ask(i, flt=None):
    return system.retrieve_documents(query_embeddings={LOCAL_ENCODER_KEY: q_host[i % 64]}, query_tokens=tokens[i % 64],
                                     similarity_k=25, common_sections_n=10, model_weights=weights,
                                     filename_type_filter=flt, use_hybrid_search=True, wrrf_k=40, use_reranker=False)


for flt in (None, "CG,NG"):
    for i in range(20):
        r = ask(i, flt)
    assert len(r) == 10, r
    t1 = time.perf_counter()
    for i in range(nq):
        ask(i, flt)
    dt = (time.perf_counter() - t1) / nq
    print("retrieve_documents filter=%r: %.0f q/s (%.1f us/query)" % (flt, 1 / dt, dt * 1e6), flush=True)

# the list form: one library call for all queries (device pipeline, one host sync)
shared = dict(similarity_k=25, common_sections_n=10, model_weights=weights, use_hybrid_search=True, wrrf_k=40,
              use_reranker=False)
for flt in (None, "CG,NG"):
    batch = [{"query_embeddings": {LOCAL_ENCODER_KEY: q_host[i % 64]}, "query_tokens": tokens[i % 64]} for i in range(nq)]
    system.retrieve_documents_batch(batch[:16], filename_type_filter=flt, **shared)
    t1 = time.perf_counter()
    got = system.retrieve_documents_batch(batch, filename_type_filter=flt, **shared)
    dt = (time.perf_counter() - t1) / nq
    assert got[:8] == [ask(i, flt) for i in range(8)]
    print("retrieve_documents_batch filter=%r: %.0f q/s (%.1f us/query)" % (flt, 1 / dt, dt * 1e6), flush=True)

# several caller threads on ONE system (the reference's Streamlit sessions): enqueues overlap
from concurrent.futures import ThreadPoolExecutor
for workers in (2, 4, 8):
    with ThreadPoolExecutor(workers) as pool:
        list(pool.map(ask, range(64)))
        t1 = time.perf_counter()
        list(pool.map(ask, range(nq)))
        dt = (time.perf_counter() - t1) / nq
    print("retrieve_documents from %d threads: %.0f q/s (%.1f us/query)" % (workers, 1 / dt, dt * 1e6), flush=True)

# the raw ABI call underneath, same operands
from anrag.database_manager import FusedPair
pair = FusedPair.of(df, proxy, ids)
tids = [proxy.term_ids(t) for t in tokens]
t1 = time.perf_counter()
for i in range(nq):
    pair.dense.index.hybrid_search(q_host[i % 64], tids[i % 64], 25, 5.0, 1.0, 40.0, 10, None, None)
dt = (time.perf_counter() - t1) / nq
print("Index.hybrid_search (ctypes -> anrag_hybrid_search): %.0f q/s (%.1f us/query)" % (1 / dt, dt * 1e6), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    ask(i, "CG,NG")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
print(s.getvalue()[:2500])
