"""Oracle (test infrastructure): numpy restatement of `SearchEngine`'s arithmetic.

Every function cites the reference lines it follows (paths relative to
`/root/reference/`).  Row indices are positions in the *unfiltered* corpus so
that results can be compared with the device path, which never materialises a
filtered copy.
"""
from __future__ import annotations

import re
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- filter
def filter_prefixes(filename_type_filter: str) -> Tuple[str, ...]:
    """src/search_engine.py:39 and :222 -- split on ',', strip, upper-case."""
    return tuple(p.strip().upper() for p in filename_type_filter.split(","))


def dense_filter_mask(sources: Sequence[Optional[str]], filename_type_filter: str) -> np.ndarray:
    """src/search_engine.py:36-48 (`_filter_by_filename_type`).

    One prefix -> `str.startswith`; several -> an UN-ESCAPED regex
    `^(?:A|B)` evaluated with `str.contains` (pandas uses `re.search`).
    A missing source (None/NaN) is dropped (`na=False`).
    """
    prefixes = filter_prefixes(filename_type_filter)
    up = [s.upper() if isinstance(s, str) else None for s in sources]
    if len(prefixes) == 1:
        return np.array([u is not None and u.startswith(prefixes[0]) for u in up], dtype=bool)
    pattern = re.compile("^(?:" + "|".join(prefixes) + ")")
    return np.array([u is not None and pattern.search(u) is not None for u in up], dtype=bool)


def bm25_filter_mask(sources: Sequence[str], filename_type_filter: str) -> np.ndarray:
    """src/search_engine.py:221-231 -- `any(source.upper().startswith(p))`."""
    prefixes = filter_prefixes(filename_type_filter)
    return np.array([any(s.upper().startswith(p) for p in prefixes) for s in sources], dtype=bool)


# --------------------------------------------------------------------------- top-k idioms
def numpy_topk_idiom(scores: np.ndarray, k: int) -> np.ndarray:
    """The reference's selection idiom, verbatim in behaviour
    (src/search_engine.py:83-87 dense, :236-241 BM25 no-filter path).

    `argpartition(-k)` + `argsort()[::-1]`: order among EQUAL scores is whatever
    numpy's introselect/quicksort leave behind (not stable, not specified).
    """
    if len(scores) > k:
        top = np.argpartition(scores, -k)[-k:]
        return top[scores[top].argsort()[::-1]]
    return scores.argsort()[::-1]


def canonical_topk(scores: np.ndarray, k: int, allowed: Optional[np.ndarray] = None) -> np.ndarray:
    """The build's deterministic rule: score descending, then row ascending.

    Identical to the reference wherever scores are distinct; inside a group of
    equal scores the reference's order is unspecified (dense, BM25 no-filter) or
    low-index-first (BM25 filter path, a stable `sorted`, :233) -- the latter is
    exactly this rule.
    """
    idx = np.arange(len(scores)) if allowed is None else np.nonzero(allowed)[0]
    s = scores[idx]
    order = np.lexsort((idx, -s))  # primary: -score ascending == score descending
    return idx[order[:k]]


# --------------------------------------------------------------------------- dense
def dense_scores(query_embedding: np.ndarray, embeddings: np.ndarray) -> np.ndarray:
    """src/search_engine.py:77-81 -- raw dot product, no normalisation.

    dtype follows numpy promotion: fp32 query x fp32 matrix -> fp32 (eval path,
    retrieval_eval.py:22-24); an fp64 query (text path, :157) up-casts to fp64.
    """
    q = query_embedding.reshape(1, -1) if query_embedding.ndim == 1 else query_embedding
    return np.dot(q, embeddings.T).flatten()


def similarity_search_with_embedding(
    query_embedding: np.ndarray,
    embeddings: np.ndarray,
    sources: Optional[Sequence[str]] = None,
    similarity_k: int = 25,
    filename_type_filter: Optional[str] = None,
    canonical: bool = False,
) -> Tuple[np.ndarray, np.ndarray]:
    """src/search_engine.py:57-98.  Returns (rows into the unfiltered corpus,
    similarities) in rank order.  Empty arrays if the filter leaves nothing
    (:71-75 returns the empty frame)."""
    n = len(embeddings)
    keep = np.arange(n)
    if filename_type_filter:
        keep = np.nonzero(dense_filter_mask(sources, filename_type_filter))[0]
    if len(keep) == 0:
        return np.empty(0, np.int64), np.empty(0, np.float32)
    sims = dense_scores(query_embedding, embeddings[keep])
    top = canonical_topk(sims, similarity_k) if canonical else numpy_topk_idiom(sims, similarity_k)
    return keep[top].astype(np.int64), sims[top]


# --------------------------------------------------------------------------- BM25 selection
def core_bm25_search(
    bm25_scores: np.ndarray,
    section_sources: Sequence[str],
    similarity_k: int,
    filename_type_filter: Optional[str],
    canonical: bool = False,
) -> np.ndarray:
    """src/search_engine.py:219-243, from the score vector on.  Returns rows.

    Filter path (:221-234): Python `sorted(..., reverse=True)[:k]` -- stable, so
    equal scores keep ascending row order.  No-filter path (:236-243): the numpy
    idiom.  Zero-score documents are NOT dropped on either path.
    """
    if filename_type_filter:
        mask = bm25_filter_mask(section_sources, filename_type_filter)
        pairs = [(i, bm25_scores[i]) for i in range(len(bm25_scores)) if mask[i]]
        top = sorted(pairs, key=lambda x: x[1], reverse=True)[:similarity_k]
        return np.array([i for i, _ in top], dtype=np.int64)
    scores = np.array(bm25_scores)
    if canonical:
        return canonical_topk(scores, similarity_k).astype(np.int64)
    return numpy_topk_idiom(scores, similarity_k).astype(np.int64)


# --------------------------------------------------------------------------- fusion
def weighted_reciprocal_rank_fusion(
    ranked_lists: List[Tuple[Sequence, str]], model_weights: Dict[str, float], k=50
) -> List[Tuple[object, float]]:
    """src/search_engine.py:21-34.

    `score[id] += w * (1 / (k + rank))`, rank from 1, in list order; missing
    weight -> 1.0; the final `sorted(..., reverse=True)` is stable, so equal
    fused scores keep first-insertion order.
    """
    rrf = defaultdict(float)
    for ranked, name in ranked_lists:
        w = model_weights.get(name, 1.0)
        for rank, doc_id in enumerate(ranked, start=1):
            rrf[doc_id] = rrf[doc_id] + w * (1 / (k + rank))
    return sorted(rrf.items(), key=lambda x: x[1], reverse=True)
