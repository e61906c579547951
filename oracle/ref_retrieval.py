"""Oracle (test infrastructure): restatement of the retrieval glue and the metrics.

Follows src/query_rag_retrieval.py:149-411 (`retrieve_documents`) and
src/retrieval_eval.py:51-116 (`evaluate_query`, `calculate_metrics`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import ref_search

# The reference walks its dense models in this fixed order
# (query_rag_retrieval.py:197, :222, :253, :282).
MODEL_ORDER = ("voyage-3-large", "voyage-3.5", "text-embedding-3-large", "Qwen3")

# src/config.py:30-36
DEFAULT_MODEL_WEIGHTS = {
    "voyage-3-large": 5.0,
    "text-embedding-3-large": 0.0,
    "voyage-3.5": 0.0,
    "Qwen3": 0.0,
    "BM25": 1.0,
}


@dataclass
class DenseCorpus:
    ids: Sequence[str]
    sources: Sequence[str]
    embeddings: np.ndarray  # [n, d] float32
    documents: Optional[Sequence[str]] = None  # the `document` column (only `retrieve_docs` reads it)


@dataclass
class Bm25Corpus:
    bm25: object  # anything with get_scores(tokens) -> float64[n]
    section_ids: Sequence[str]
    section_sources: Sequence[str]
    section_contents: Optional[Sequence[str]] = None  # `page_content` of each section


def retrieve_ids(
    dense: Dict[str, DenseCorpus],
    bm25: Optional[Bm25Corpus],
    query_embeddings: Dict[str, np.ndarray],
    query_tokens: Optional[List[str]] = None,
    similarity_k: int = 25,
    common_sections_n: int = 15,
    model_weights: Optional[Dict[str, float]] = None,
    filename_type_filter: Optional[str] = None,
    use_hybrid_search: bool = False,
    wrrf_k=60,
    canonical: bool = False,
    model_order: Sequence[str] = MODEL_ORDER,
) -> List[str]:
    """query_rag_retrieval.py:149-411 with `use_reranker=False`, ids out (:403-407)."""
    if model_weights is None:
        model_weights = dict(DEFAULT_MODEL_WEIGHTS)
    ranked_lists = []
    seen = {}  # insertion-ordered: id -> True ("results_dict", :373)
    for name in model_order:
        corpus = dense.get(name)
        # gating: :199-204
        if corpus is None or len(corpus.ids) == 0:
            continue
        if not (model_weights.get(name, 0) > 0 and name in query_embeddings):
            continue
        rows, _ = ref_search.similarity_search_with_embedding(
            query_embeddings[name], corpus.embeddings, corpus.sources,
            similarity_k, filename_type_filter, canonical=canonical,
        )
        if len(rows) == 0:  # :213 `if not results.empty`
            continue
        ranked = [corpus.ids[r] for r in rows]
        ranked_lists.append((ranked, name))
        for i in ranked:
            seen.setdefault(i, True)
    # BM25 gating: :304
    if use_hybrid_search and bm25 is not None and model_weights.get("BM25", 0) > 0:
        ranked = None
        if query_tokens:  # :307
            scores = bm25.bm25.get_scores(query_tokens)
            rows = ref_search.core_bm25_search(
                scores, bm25.section_sources, similarity_k, filename_type_filter, canonical=canonical
            )
            ranked = [bm25.section_ids[r] for r in rows]
        if ranked:  # :333
            ranked_lists.append((ranked, "BM25"))
            for i in ranked:
                seen.setdefault(i, True)
    # fusion: :356-370
    if len(ranked_lists) > 1:
        fused = ref_search.weighted_reciprocal_rank_fusion(ranked_lists, model_weights, wrrf_k)
        most_common = [i for i, _ in fused[:common_sections_n]]
    elif len(ranked_lists) == 1:
        most_common = list(ranked_lists[0][0][:common_sections_n])
    else:
        most_common = []
    # :372-378
    return [i for i in most_common if i in seen][:common_sections_n]


def rerank_documents(client, query_text: str, documents: List[dict], reranker_model: str = "rerank-2",
                     reranker_top_k: Optional[int] = None) -> List[dict]:
    """src/search_engine.py:161-203: the documents' texts go to `vo.rerank`; the answer's (index, relevance_score)
    pairs re-order the records (`rerank_score` added); any failure returns the original order (:201-203)."""
    try:
        if not documents:
            return documents
        texts = [doc.get("document", "") for doc in documents]
        result = client.rerank(query=query_text, documents=texts, model=reranker_model,
                               top_k=reranker_top_k or len(texts), truncation=True)
        return [{**documents[r.index], "rerank_score": r.relevance_score} for r in result.results
                if r.index < len(documents)]
    except Exception:
        return documents


def retrieve_docs(
    dense: Dict[str, DenseCorpus],
    bm25: Optional[Bm25Corpus],
    query_embeddings: Dict[str, np.ndarray],
    query_text: Optional[str] = None,
    query_tokens: Optional[List[str]] = None,
    similarity_k: int = 25,
    common_sections_n: int = 15,
    model_weights: Optional[Dict[str, float]] = None,
    filename_type_filter: Optional[str] = None,
    use_hybrid_search: bool = False,
    wrrf_k=60,
    use_reranker: bool = True,
    reranker_model: str = "rerank-2-lite",
    reranker_top_k: Optional[int] = 5,
    return_docs: bool = False,
    rerank_client=None,
    canonical: bool = False,
    model_order: Sequence[str] = MODEL_ORDER,
):
    """query_rag_retrieval.py:149-411 including what `retrieve_ids` leaves out: the per-document records
    (:216-217 `to_dict("records")`; a later model's rows are dropped when an earlier model already returned their id,
    :242-245, :272-275, :293-294 -- the FIRST model's record stands; BM25-only sections :336-350), `results_dict`
    (:373-374), the reranker hand-off (:383-387) and the two exits (`return_docs` :400-401, ids :403-407).  Records carry id, document,
    source, similarity and, for dense rows, `embedding0` (first value of the row: which model's record it is)."""
    if model_weights is None:
        model_weights = dict(DEFAULT_MODEL_WEIGHTS)
    ranked_lists = []
    all_results = []
    for name in model_order:
        corpus = dense.get(name)
        if corpus is None or len(corpus.ids) == 0:
            continue
        if not (model_weights.get(name, 0) > 0 and name in query_embeddings):
            continue
        rows, sims = ref_search.similarity_search_with_embedding(
            query_embeddings[name], corpus.embeddings, corpus.sources, similarity_k, filename_type_filter,
            canonical=canonical)
        if len(rows) == 0:
            continue
        ranked_lists.append(([corpus.ids[r] for r in rows], name))
        existing = {i for i, _ in all_results}
        for r, sim in zip(rows, sims):
            if corpus.ids[r] in existing:
                continue
            all_results.append((corpus.ids[r], dict(
                id=corpus.ids[r], document=corpus.documents[r] if corpus.documents is not None else None,
                source=corpus.sources[r], similarity=sim, embedding0=float(corpus.embeddings[r, 0]))))
    if use_hybrid_search and bm25 is not None and model_weights.get("BM25", 0) > 0:
        ranked = None
        if query_tokens:
            scores = bm25.bm25.get_scores(query_tokens)
            rows = ref_search.core_bm25_search(scores, bm25.section_sources, similarity_k, filename_type_filter,
                                               canonical=canonical)
            ranked = [bm25.section_ids[r] for r in rows]
        if ranked:
            ranked_lists.append((ranked, "BM25"))
            existing = {i for i, _ in all_results}
            pos = {sid: j for j, sid in enumerate(bm25.section_ids)}
            for sid in ranked:
                if sid not in existing:
                    j = pos[sid]
                    all_results.append((sid, dict(
                        id=sid, document=bm25.section_contents[j] if bm25.section_contents is not None else None,
                        source=bm25.section_sources[j], similarity=0.0)))
    if len(ranked_lists) > 1:
        fused = ref_search.weighted_reciprocal_rank_fusion(ranked_lists, model_weights, wrrf_k)
        most_common = [i for i, _ in fused[:common_sections_n]]
    elif len(ranked_lists) == 1:
        most_common = list(ranked_lists[0][0][:common_sections_n])
    else:
        most_common = []
    results_dict = {doc_id: doc for doc_id, doc in all_results}
    common_docs = [results_dict[i] for i in most_common if i in results_dict][:common_sections_n]
    if use_reranker and common_docs and len(common_docs) > 1 and query_text:
        common_docs = rerank_documents(rerank_client, query_text, common_docs, reranker_model, reranker_top_k)
    if return_docs:
        return common_docs
    return [doc.get("id", "Unknown section") for doc in common_docs]


# --------------------------------------------------------------------------- metrics
def rank_of(expected_id: str, retrieved: Sequence[str]) -> int:
    """retrieval_eval.py:75-82 -- 1-based rank, -1 if absent."""
    for i, doc_id in enumerate(retrieved):
        if doc_id == expected_id:
            return i + 1
    return -1


def calculate_metrics(results: List[Dict]) -> Dict:
    """retrieval_eval.py:90-116."""
    found = [r for r in results if r.get("found")]
    found_ranks = [r["rank"] for r in found]
    all_ranks = [r["rank"] if r.get("found") else 100000 for r in results]
    n = len(results)
    mrr = sum(1.0 / r["rank"] for r in found) / n if results else 0.0

    def recall_at(k):
        return sum(1 for r in found if r["rank"] <= k) / n if results else 0.0

    return {
        "total": n,
        "found": len(found),
        "success_rate": len(found) / n if results else 0.0,
        "mean_rank": np.mean(found_ranks) if found_ranks else None,
        "median_rank": np.median(found_ranks) if found_ranks else None,
        "max_rank": np.max(all_ranks) if all_ranks else None,
        "mrr": mrr,
        "recall@1": recall_at(1),
        "recall@5": recall_at(5),
        "recall@10": recall_at(10),
        "recall@15": recall_at(15),
    }
