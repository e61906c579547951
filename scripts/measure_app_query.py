#!/usr/bin/env python3
"""The reference APP's query shape (src/app.py:109, src/query_rag.py:269: one hybrid query at a time, similarity_k = 25,
common_sections_n = 15, wrrf_k = 40, filter, ~9.6k chunks) through `retrieve_documents` on the stand-in corpus: latency
per call and where the Python side spends it.
usage: python scripts/measure_app_query.py [calls]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
from anrag import niceqa
from anrag.bm25_index import Bm25Index
from anrag.config import Config, InfoSource, LOCAL_ENCODER_KEY
from anrag.database_manager import Bm25Proxy, DenseHandle
from anrag.index_io import Section
from anrag.query_rag_retrieval import RetrievalEvaluationSystem
from anrag.search_engine import SearchEngine

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
data = niceqa.load_standin(os.path.join(G, "suggested_queries_bm25_preprocessed.json.gz"), os.path.join(G, "NICEQA.csv"))
ids, sources, emb = data["ids"], data["sources"], data["embeddings"].astype(np.float32)
df = pd.DataFrame({"id": ids, "document": [" ".join(t) for t in data["tokens"]], "source": sources, "embedding": list(emb)})
DenseHandle(emb, sources).bind(df)
proxy = Bm25Proxy(Bm25Index(data["tokens"], k1=1.7, b=0.83, epsilon=0.05), sources)
sections = [Section(d, {"id": i, "source": s}) for i, s, d in zip(ids, sources, df["document"])]
system = RetrievalEvaluationSystem.__new__(RetrievalEvaluationSystem)
system.config = Config(); system.search_engine = SearchEngine(None, None); system.voyage_client = None; system.fused = True
system.embeddings_data = {InfoSource.NICE: {LOCAL_ENCODER_KEY: df}}
system.bm25_data = {InfoSource.NICE: (proxy, sections, ids)}
qv, qt = niceqa.encode_questions(data)
W = {LOCAL_ENCODER_KEY: 5.0, "BM25": 1.0}


def ask(i, **kw):
    j = i % len(qv)
    return system.retrieve_documents(query_embeddings={LOCAL_ENCODER_KEY: qv[j]}, query_tokens=qt[j] or ["asthma"],
                                     similarity_k=25, common_sections_n=15, filename_type_filter="CG,NG", wrrf_k=40,
                                     use_reranker=False, use_hybrid_search=True, model_weights=W, **kw)


for i in range(50):
    ask(i)
lat = []
for i in range(calls):
    t0 = time.perf_counter()
    r = ask(i)
    lat.append(time.perf_counter() - t0)
print(f"retrieve_documents (ids out, fused route): p50 {np.median(lat)*1e6:.1f} us, p99 {np.percentile(lat, 99)*1e6:.1f} us, "
      f"{len(r)} ids", flush=True)
lat = []
for i in range(min(calls, 500)):
    t0 = time.perf_counter()
    r = ask(i, return_docs=True)
    lat.append(time.perf_counter() - t0)
print(f"retrieve_documents (return_docs=True, method-by-method route): p50 {np.median(lat)*1e6:.1f} us, p99 "
      f"{np.percentile(lat, 99)*1e6:.1f} us, {len(r)} documents", flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(300):
    ask(i)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:2600])
