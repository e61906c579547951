"""Oracle (test infrastructure): the CPU reference path over the NICEQA stand-in corpus
(anrag.niceqa.load_standin): dense numpy scan, restated BM25, Python WRRF -- src/query_rag_retrieval.py:197-378
for one dense model + BM25, weights 5:1, wrrf_k 40 (retrieval_eval.py:279)."""
from . import ref_search
from .ref_bm25 import CsrBM25


def cpu_ranked_ids(data, query_vectors, query_tokens, k=25, top_n=10):
    bm = CsrBM25(data["tokens"], k1=1.7, b=0.83, epsilon=0.05)
    out = []
    for v, t in zip(query_vectors, query_tokens):
        rows, _ = ref_search.similarity_search_with_embedding(v, data["embeddings"], None, k, None, canonical=True)
        lists = [([data["ids"][r] for r in rows], "dense")]
        if t:
            brow = ref_search.canonical_topk(bm.get_scores(t), k)
            lists.append(([data["ids"][r] for r in brow], "BM25"))
        fused = ref_search.weighted_reciprocal_rank_fusion(lists, {"dense": 5.0, "BM25": 1.0}, 40)[:top_n]
        out.append([i for i, _ in fused])
    return out
