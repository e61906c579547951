"""NICEQA Recall@10 on a stand-in corpus.

The reference ships `data/NICEQA.csv` (70 rows: Guideline ID, Section, Question -- no answers, no chunk ids)
and no corpus (its *.db / *.pkl are git-ignored, the NICE text is not redistributable), and publishes no
Recall@10 for NICEQA.  What can be measured offline is EQUALITY of Recall@10 between the GPU path and the CPU
reference path on identical inputs.  Stand-in, built only from shipped data files (tests/golden/):
  * corpus      the 9,609 chunk ids of `suggested_queries_bm25_preprocessed.csv`; a chunk's text is its id's
                title part + the LLM-generated query the reference stores for it; BM25 tokens = tokenised title
                + the shipped `tokens_lemmatized`
  * embeddings  no encoder weights exist offline: a deterministic hashed bag-of-words -> 384-d unit vector
                (`hashed_bow`), used identically for chunks and questions, by both paths
  * gold        a NICEQA row (guideline g, section s) is a hit@10 if any of the top-10 chunk ids has source g
                and a title containing s as a whole dotted number
"""
from __future__ import annotations

import csv
import gzip
import json
import re
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .preprocess_bm25 import preprocess_text

DIM = 384


def hashed_bow(tokens: Sequence[str], dim: int = DIM) -> np.ndarray:
    v = np.zeros(dim, dtype=np.float32)
    for t in tokens:
        h = zlib.crc32(t.encode("utf-8"))
        v[h % dim] += 1.0 if (h >> 16) & 1 else -1.0
    n = float(np.linalg.norm(v))
    if n == 0.0:
        v[0] = 1.0
        n = 1.0
    return v / np.float32(n)


def load_standin(chunks_json_gz: str, niceqa_csv: str) -> Dict[str, object]:
    with gzip.open(chunks_json_gz, "rt", encoding="utf-8") as f:
        rows = json.load(f)
    ids, sources, titles, tokens = [], [], [], []
    for r in rows:
        cid = r["id"]
        source, _, title = cid.partition("_")
        ids.append(cid)
        sources.append(source)
        titles.append(title)
        tokens.append(preprocess_text(title, use_lemmatization=True) + list(r["tokens_lemmatized"]))
    emb = np.stack([hashed_bow(t) for t in tokens])
    questions = []
    with open(niceqa_csv, encoding="utf-8-sig") as f:
        for r in csv.DictReader(f):
            g = (r.get("Guideline ID") or "").strip()
            s = (r.get("Section") or "").strip()
            q = (r.get("Question") or "").strip()
            if g and q:
                questions.append((g, s, q))
    return dict(ids=ids, sources=sources, titles=titles, tokens=tokens, embeddings=emb, questions=questions)


def gold_chunks(data: Dict[str, object], guideline: str, section: str) -> List[int]:
    rx = re.compile(r"(^|[^\d.])" + re.escape(section) + r"(?![\d.]*\d)") if section else None
    return [i for i, (s, t) in enumerate(zip(data["sources"], data["titles"]))
            if s == guideline and (rx is None or rx.search(t))]


def encode_questions(data: Dict[str, object]) -> Tuple[np.ndarray, List[List[str]]]:
    toks = [preprocess_text(q, use_lemmatization=True) for _, _, q in data["questions"]]
    return np.stack([hashed_bow(t) for t in toks]), toks


def recall_at_10(data: Dict[str, object], ranked_ids_per_question: Sequence[Sequence[str]]) -> Dict[str, float]:
    pos = {cid: i for i, cid in enumerate(data["ids"])}
    hits = answerable = 0
    for (g, s, _), ranked in zip(data["questions"], ranked_ids_per_question):
        gold = set(gold_chunks(data, g, s))
        if not gold:
            continue  # reported, not dropped from the denominator choice below
        answerable += 1
        if any(pos[c] in gold for c in list(ranked)[:10]):
            hits += 1
    n = len(data["questions"])
    return {"questions": n, "with_gold_chunk": answerable, "hits_at_10": hits,
            "recall_at_10": hits / n if n else 0.0,
            "recall_at_10_answerable": hits / answerable if answerable else 0.0}


def gpu_ranked_ids(data: Dict[str, object], similarity_k: int = 25, top_n: int = 10, w_dense: float = 5.0,
                   w_bm25: float = 1.0, wrrf_k: float = 40.0, device: int = 0) -> List[List[str]]:
    """The product path: one fused `anrag_hybrid_search` per question."""
    from .bm25_index import Bm25Index
    from .index import Index

    bi = Bm25Index(data["tokens"], k1=1.7, b=0.83, epsilon=0.05)
    qv, qt = encode_questions(data)
    out = []
    with Index(device) as idx:
        idx.dense_load(data["embeddings"])
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        for v, t in zip(qv, qt):
            ids, _ = idx.hybrid_search(v, bi.term_ids(t), similarity_k, w_dense, w_bm25, wrrf_k, top_n)
            out.append([data["ids"][i] for i in ids.tolist()])
    return out
