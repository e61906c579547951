#!/usr/bin/env python3
"""retrieval_eval's full-ranking configurations (similarity_k = common_sections_n = 12000, filter "CG,NG") through
`retrieve_documents` on the stand-in corpus (9,609 shipped chunk ids x 384-d): time per query and where it goes.
usage: python scripts/measure_eval_route.py [queries]"""
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
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 70
data = niceqa.load_standin(os.path.join(G, "suggested_queries_bm25_preprocessed.json.gz"), os.path.join(G, "NICEQA.csv"))
ids, sources, emb = data["ids"], data["sources"], data["embeddings"].astype(np.float32)
df = pd.DataFrame({"id": ids, "document": [" ".join(t) for t in data["tokens"]], "source": sources, "embedding": list(emb)})
DenseHandle(emb, sources).bind(df)
bi = Bm25Index(data["tokens"], k1=1.7, b=0.83, epsilon=0.05)
proxy = Bm25Proxy(bi, sources)
sections = [Section(d, {"id": i, "source": s}) for i, s, d in zip(ids, sources, df["document"])]
system = RetrievalEvaluationSystem.__new__(RetrievalEvaluationSystem)
system.config = Config(); system.search_engine = SearchEngine(None, None); system.voyage_client = None; system.fused = True
system.embeddings_data = {InfoSource.NICE: {LOCAL_ENCODER_KEY: df}}
system.bm25_data = {InfoSource.NICE: (proxy, sections, ids)}
qv, qt = niceqa.encode_questions(data)
configs = {
    "dense full ranking": dict(model_weights={LOCAL_ENCODER_KEY: 1.0, "BM25": 0.0}, use_hybrid_search=False),
    "BM25 full ranking": dict(model_weights={LOCAL_ENCODER_KEY: 0.0, "BM25": 1.0}, use_hybrid_search=True),
    "hybrid full ranking": dict(model_weights={LOCAL_ENCODER_KEY: 5.0, "BM25": 1.0}, use_hybrid_search=True),
}


def ask(i, cfg):
    j = i % len(qv)
    return system.retrieve_documents(query_embeddings={LOCAL_ENCODER_KEY: qv[j]}, query_tokens=qt[j] or ["asthma"],
                                     similarity_k=12000, common_sections_n=12000, filename_type_filter="CG,NG",
                                     wrrf_k=40, use_reranker=False, **cfg)


from anrag.retrieval_eval import RetrievalEvaluator

ev = RetrievalEvaluator(retrieval_system=system)
big = max(nq, 2048)
for name, cfg in configs.items():
    for i in range(3):
        r = ask(i, cfg)
    t0 = time.perf_counter()
    for i in range(nq):
        r = ask(i, cfg)
    dt = (time.perf_counter() - t0) / nq
    # the same questions as LISTS: retrieve_documents_batch (id strings out) and evaluate_queries (rank of the expected
    # id only) -- one anrag_rank_batch per list (rank_batch.hip)
    asks = [{"query_embeddings": {LOCAL_ENCODER_KEY: qv[i % len(qv)]}, "query_tokens": qt[i % len(qv)] or ["asthma"]}
            for i in range(big)]
    shared = dict(similarity_k=12000, common_sections_n=12000, filename_type_filter="CG,NG", wrrf_k=40, use_reranker=False,
                  **cfg)
    system.retrieve_documents_batch(asks, **shared)  # warm-up at full size: the scratch pool grows once
    t0 = time.perf_counter()
    lists = system.retrieve_documents_batch(asks, **shared)
    dl = (time.perf_counter() - t0) / big
    same = all(lists[i] == ask(i, cfg) for i in range(4))
    items = [{"query": "", "expected_id": lists[i][min(7, len(lists[i]) - 1)], "query_embeddings": a["query_embeddings"],
              "query_tokens": a["query_tokens"]} for i, a in enumerate(asks)]
    params = dict(shared)
    ev.evaluate_queries(items, params)
    t0 = time.perf_counter()
    res = ev.evaluate_queries(items, params)
    de = (time.perf_counter() - t0) / big
    ok = all(r_["rank"] == min(7, len(lists[i]) - 1) + 1 for i, r_ in enumerate(res))
    print(f"{name}: one by one {dt*1e3:.2f} ms/query ({1/dt:.0f} q/s), {len(r)} ids returned; as a list of {big}: "
          f"{dl*1e6:.0f} us/query with the id strings ({1/dl:.0f} q/s; equal to one-by-one: {same}), "
          f"{de*1e6:.1f} us/query for the evaluation harness's rank-of-expected ({1/de:.0f} q/s; ranks right: {ok})", flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    ask(i, configs["hybrid full ranking"])
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(16); print(s.getvalue()[:2800])
