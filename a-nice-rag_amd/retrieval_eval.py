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
            asks = [{"query_text": it["query"], "query_embeddings": it["query_embeddings"],
                     "query_tokens": it.get("query_tokens")} for it in items]
            # the harness only looks the expected id up in each list (:75-82): ask the device for that position
            ranked = getattr(self.retrieval_system, "rank_of_expected_batch", None)
            got = ranked(asks, [it["expected_id"] for it in items], **shared) if ranked else None
            if got is not None:
                ranks, totals = got
                return [{"rank": int(r), "found": bool(r > 0), "total_retrieved": int(t)}
                        for r, t in zip(ranks.tolist(), totals.tolist())]
            lists = self.retrieval_system.retrieve_documents_batch(asks, **shared)
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


# ---------------------------------------------------------------------------------------------------
# The evaluation run (reference: retrieval_eval.py:121-420, a script with its paths and nine configurations
# written into `main`).  Same selection rules, same split, same CSV rows; the queries of a configuration go to
# the retrieval system as lists (`evaluate_queries`), not one call each.
DENSE_KEYS = ("voyage-3-large", "voyage-3.5", "text-embedding-3-large", "Qwen3")


def _configuration(name, weights, hybrid, k=12000, n=12000, **extra):
    w = {key: 0.0 for key in DENSE_KEYS}
    w["BM25"] = 0.0
    w.update(weights)
    return dict(name=name, model_weights=w, use_hybrid_search=hybrid, similarity_k=k, common_sections_n=n, **extra)


# retrieval_eval.py:130-268 as a table: (label, non-zero weights, hybrid?, k, n, reranker)
REFERENCE_CONFIGURATIONS = [
    _configuration("Voyage-3-Large", {"voyage-3-large": 1.0}, False),
    _configuration("Voyage-3.5", {"voyage-3.5": 1.0}, False),
    _configuration("Text-Embedding-3-Large", {"text-embedding-3-large": 1.0}, False),
    _configuration("Qwen3-Embedding-0.6B", {"Qwen3": 1.0}, False),
    _configuration("BM25", {"BM25": 1.0}, True),
    _configuration("Voyage-3-Large + BM25", {"voyage-3-large": 5.0, "BM25": 1.0}, True),
    _configuration("Voyage-3-Large + Text-Embedding-3-Large", {"voyage-3-large": 2.0, "text-embedding-3-large": 1.0}, False),
    _configuration("Voyage-3-Large + BM25 (Reranker 2 Lite)", {"voyage-3-large": 5.0, "BM25": 1.0}, True, 25, 15,
                   use_reranker=True, reranker_model="rerank-2-lite", reranker_top_k=10),
    _configuration("Voyage-3-Large + BM25 (Reranker 2)", {"voyage-3-large": 5.0, "BM25": 1.0}, True, 25, 15,
                   use_reranker=True, reranker_model="rerank-2", reranker_top_k=10),
]
BASE_PARAMS = {"wrrf_k": 40, "filename_type_filter": "CG,NG"}  # retrieval_eval.py:278-281


def select_queries(cached_dbs: Dict[str, pd.DataFrame], model_weights: Dict[str, float],
                   reference_model: str = "voyage-3-large"):
    """retrieval_eval.py:309-336: the query table of the heaviest active dense model (the reference model's for a
    BM25-only run), inner-joined on query id with every other active model's embeddings."""
    active = [k for k in model_weights if k != "BM25" and model_weights.get(k, 0) > 0]
    main = max(active, key=lambda k: model_weights[k]) if active else reference_model
    queries = cached_dbs[main].copy()
    embeddings = {main: queries["query_embedding"].values}
    for model in active:
        if model == main:
            continue
        merged = queries.merge(cached_dbs[model][["id", "query_embedding"]], on="id", how="left",
                               suffixes=("", f"_{model}"))
        col = f"query_embedding_{model}"
        embeddings[model] = merged[col].values
        queries = merged[merged[col].notna()].copy()
    return queries, embeddings


def format_csv_row(name: str, m: Dict) -> str:
    """retrieval_eval.py:401-417: three decimals, 'N/A' for the ranks of an all-miss run."""
    mean = f"{m['mean_rank']:.3f}" if m["mean_rank"] is not None else "N/A"
    median = str(m["median_rank"]) if m["median_rank"] is not None else "N/A"
    return (f"{name},{m['mrr']:.3f},{m['recall@1']:.3f},{m['recall@5']:.3f},{m['recall@10']:.3f},"
            f"{m['recall@15']:.3f},{median},{mean},{m['max_rank']}\n")


def run_evaluation(evaluator: RetrievalEvaluator, cached_dbs: Dict[str, pd.DataFrame], preprocessed: pd.DataFrame,
                   output_file: str, configurations: Optional[List[Dict]] = None, base_params: Optional[Dict] = None,
                   reference_model: str = "voyage-3-large", chunk: int = 2048) -> List[Dict]:
    """retrieval_eval.py:270-420.  -> one metrics dict per configuration, each also appended to `output_file`."""
    from sklearn.model_selection import train_test_split

    configurations = REFERENCE_CONFIGURATIONS if configurations is None else configurations
    base_params = BASE_PARAMS if base_params is None else base_params
    # :273-276 -- the split is drawn once over the reference model's table; the 85 % side is what gets evaluated
    train_idx, _ = train_test_split(range(len(cached_dbs[reference_model])), test_size=0.15, random_state=42,
                                    shuffle=True)
    train_idx = np.asarray(train_idx)
    os.makedirs(os.path.dirname(output_file) or ".", exist_ok=True)
    if not os.path.exists(output_file):
        with open(output_file, "w") as f:
            f.write(",".join(CSV_HEADER) + "\n")
    tokens_of = dict(zip(preprocessed["id"], preprocessed["tokens_lemmatized"]))
    all_metrics = []
    for config in configurations:
        weights = config["model_weights"]
        queries, embeddings = select_queries(cached_dbs, weights, reference_model)
        test_queries = queries.iloc[train_idx].reset_index(drop=True)
        test_embeddings = {k: v[train_idx] for k, v in embeddings.items()}
        need_tokens = weights.get("BM25", 0) > 0 and config["use_hybrid_search"]
        params = dict(base_params)
        params.update({k: v for k, v in config.items() if k != "name"})
        items = []
        for i, row in enumerate(test_queries.itertuples(index=False)):
            qe = {k: v[i] for k, v in test_embeddings.items() if i < len(v) and v[i] is not None}
            items.append({"query": row.query, "expected_id": row.id, "query_embeddings": qe,
                          "query_tokens": tokens_of.get(row.id) if need_tokens else None})
        results: List[Dict] = []
        for lo in range(0, len(items), chunk):
            results.extend(evaluator.evaluate_queries(items[lo: lo + chunk], params))
        metrics = calculate_metrics(results)
        with open(output_file, "a") as f:
            f.write(format_csv_row(config["name"], metrics))
        all_metrics.append(metrics)
    return all_metrics
