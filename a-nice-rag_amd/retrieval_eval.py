"""Retrieval benchmark harness: the reference's src/retrieval_eval.py (query loading :17-40,
`RetrievalEvaluator.evaluate_query` :51-84, `calculate_metrics` :90-116, CSV schema :292-294)
over `anrag.query_rag_retrieval.RetrievalEvaluationSystem`."""
from __future__ import annotations

import ast
import csv
import os
import sqlite3
from typing import Dict, List, Optional

import numpy as np
import pandas as pd

from .query_rag_retrieval import RetrievalEvaluationSystem

CSV_HEADER = ["Model", "MRR", "Recall@1", "Recall@5", "Recall@10", "Recall@15", "Median_Rank", "Mean_Rank", "Max_Rank"]


def load_queries_from_db(db_path: str) -> pd.DataFrame:
    """retrieval_eval.py:17-25: `queries(id, query, query_embedding BLOB float32)`."""
    conn = sqlite3.connect(db_path)
    df = pd.read_sql_query("SELECT * FROM queries WHERE query_embedding IS NOT NULL;", conn)
    conn.close()
    df["query_embedding"] = df["query_embedding"].apply(lambda x: np.frombuffer(x, dtype=np.float32))
    return df.reset_index(drop=True)


def load_bm25_preprocessed_queries(csv_path: str) -> pd.DataFrame:
    """retrieval_eval.py:28-40."""
    df = pd.read_csv(csv_path, encoding="utf-8")

    def safe_eval(x):
        try:
            return ast.literal_eval(x) if isinstance(x, str) else x
        except (ValueError, SyntaxError):
            return []

    df["tokens_lemmatized"] = df["tokens_lemmatized"].apply(safe_eval)
    return df


def rank_of(expected_id: str, docs) -> int:
    """retrieval_eval.py:75-82: 1-based rank of the expected chunk id, -1 if absent."""
    for i, doc in enumerate(docs):
        doc_id = doc.get("section_id") if isinstance(doc, dict) else doc
        if doc_id == expected_id:
            return i + 1
    return -1


class RetrievalEvaluator:
    def __init__(self, db_paths: Optional[Dict[str, str]] = None, retrieval_system=None):
        self.db_paths = db_paths or {}
        self.retrieval_system = retrieval_system or RetrievalEvaluationSystem()

    def evaluate_query(self, query: str, expected_id: str, query_embeddings: Dict[str, np.ndarray], params: Dict,
                       query_tokens: Optional[List[str]] = None) -> Dict:
        """retrieval_eval.py:51-84."""
        try:
            results = self.retrieval_system.retrieve_documents(
                query_text=query, query_embeddings=query_embeddings, query_tokens=query_tokens,
                similarity_k=params["similarity_k"], common_sections_n=params["common_sections_n"],
                info_source=params.get("info_source", "NICE"), model_weights=params["model_weights"],
                filename_type_filter=params.get("filename_type_filter"),
                use_hybrid_search=params["use_hybrid_search"], use_reranker=params.get("use_reranker", False),
                reranker_model=params.get("reranker_model", "rerank-2"), reranker_top_k=params.get("reranker_top_k"),
                wrrf_k=params["wrrf_k"])
            docs = results[0] if isinstance(results, tuple) else results
            rank = rank_of(expected_id, docs)
            return {"rank": rank, "found": rank > 0, "total_retrieved": len(docs)}
        except Exception as e:
            return {"rank": -1, "found": False, "total_retrieved": 0, "error": str(e)}


    def evaluate_queries(self, items: List[Dict], params: Dict) -> List[Dict]:
        """`evaluate_query` over a list of {"query", "expected_id", "query_embeddings", "query_tokens"?} items with
        ONE retrieval call (`retrieve_documents_batch`); element i equals `evaluate_query(**items[i], params=params)`."""
        try:
            shared = dict(
                similarity_k=params["similarity_k"], common_sections_n=params["common_sections_n"],
                info_source=params.get("info_source", "NICE"), model_weights=params["model_weights"],
                filename_type_filter=params.get("filename_type_filter"), use_hybrid_search=params["use_hybrid_search"],
                use_reranker=params.get("use_reranker", False), reranker_model=params.get("reranker_model", "rerank-2"),
                reranker_top_k=params.get("reranker_top_k"), wrrf_k=params["wrrf_k"])
            lists = self.retrieval_system.retrieve_documents_batch(
                [{"query_text": it["query"], "query_embeddings": it["query_embeddings"],
                  "query_tokens": it.get("query_tokens")} for it in items], **shared)
        except Exception:
            return [self.evaluate_query(it["query"], it["expected_id"], it["query_embeddings"], params,
                                        it.get("query_tokens")) for it in items]
        out = []
        for it, docs in zip(items, lists):
            rank = rank_of(it["expected_id"], docs)
            out.append({"rank": rank, "found": rank > 0, "total_retrieved": len(docs)})
        return out


def calculate_metrics(results: List[Dict]) -> Dict:
    """retrieval_eval.py:90-116 (a miss counts as rank 100000 for max_rank only)."""
    found = [r for r in results if r.get("found")]
    found_ranks = [r["rank"] for r in found]
    all_ranks = [r["rank"] if r.get("found") else 100000 for r in results]
    n = len(results)
    mrr = sum(1.0 / r["rank"] for r in found) / n if results else 0.0
    recall_at = lambda k: sum(1 for r in found if r["rank"] <= k) / n if results else 0.0
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


def append_results_csv(path: str, model_label: str, metrics: Dict) -> None:
    """One row per configuration in the reference's schema (retrieval_eval.py:292-294, :401-419)."""
    new = not os.path.exists(path)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "a", newline="") as f:
        w = csv.writer(f)
        if new:
            w.writerow(CSV_HEADER)
        w.writerow([model_label, metrics["mrr"], metrics["recall@1"], metrics["recall@5"], metrics["recall@10"],
                    metrics["recall@15"], metrics["median_rank"], metrics["mean_rank"], metrics["max_rank"]])
