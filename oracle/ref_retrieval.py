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


@dataclass
class Bm25Corpus:
    bm25: object  # anything with get_scores(tokens) -> float64[n]
    section_ids: Sequence[str]
    section_sources: Sequence[str]


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
