"""Synthetic corpora of the shapes BASELINE.json names (no corpus ships with the reference:
its *.db / *.pkl are git-ignored and the NICE text is not redistributable).

Generated on the GPU with torch (PyTorch is plumbing here: device memory + RNG), per SURVEY.md
section 8(d): unit-norm N(0,1) rows; queries = a corpus row + 0.05 noise, re-normalised; BM25:
200k-term vocabulary, Zipf(1.07) term draw, log-normal document length (median 120, mean ~150),
9-term queries drawn from a document's own terms, 6.5 % with one duplicated term;
k1=1.7, b=0.83, epsilon=0.05 (src/processing/bm25_search.py:134-139).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

BM25_K1, BM25_B, BM25_EPSILON = 1.7, 0.83, 0.05


def dense_corpus(n_rows: int, dim: int, seed: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    e = torch.empty((n_rows, dim), device=device, dtype=torch.float32)
    step = max(1, (1 << 28) // max(dim, 1))  # ~1 GiB of fp32 per chunk
    for lo in range(0, n_rows, step):
        hi = min(n_rows, lo + step)
        blk = torch.randn((hi - lo, dim), device=device, dtype=torch.float32, generator=g)
        blk /= blk.norm(dim=1, keepdim=True)
        e[lo:hi] = blk
    return e


def dense_queries(e: torch.Tensor, n_queries: int, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (queries [nq, d], planted rows [nq]): query i sits next to corpus row planted[i]."""
    g = torch.Generator(device=e.device)
    g.manual_seed(seed)
    rows = torch.randint(0, e.shape[0], (n_queries,), device=e.device, generator=g)
    q = e[rows] + 0.05 * torch.randn((n_queries, e.shape[1]), device=e.device, dtype=torch.float32, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    return q.contiguous(), rows


def bm25_postings(n_docs: int, vocab: int, seed: int, device, median_len: float = 120.0, sigma: float = 0.668,
                  zipf_s: float = 1.07) -> Dict[str, object]:
    """Term-major CSR postings of a synthetic shard.

    -> dict(indptr int64[vocab+1] (host numpy), post_doc / post_tf int32 (torch, on `device`),
            doc_len int32 (host numpy), df int64 (torch), total_len int)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.exp(torch.randn(n_docs, device=device, generator=g) * sigma + math.log(median_len))
    lens = lens.clamp_(1, 2000).to(torch.int64)
    total = int(lens.sum().item())
    p = 1.0 / torch.arange(1, vocab + 1, device=device, dtype=torch.float64) ** zipf_s
    cdf = torch.cumsum(p / p.sum(), 0).to(torch.float32)
    doc_ids = torch.repeat_interleave(torch.arange(n_docs, device=device, dtype=torch.int64), lens)
    keys = torch.empty(total, device=device, dtype=torch.int64)
    step = 1 << 26
    for lo in range(0, total, step):
        hi = min(total, lo + step)
        u = torch.rand(hi - lo, device=device, generator=g)
        term = torch.searchsorted(cdf, u).clamp_(max=vocab - 1)
        keys[lo:hi] = term * n_docs + doc_ids[lo:hi]
    del doc_ids
    keys, _ = torch.sort(keys)
    uniq, counts = torch.unique_consecutive(keys, return_counts=True)
    del keys
    term = torch.div(uniq, n_docs, rounding_mode="floor")
    post_doc = (uniq - term * n_docs).to(torch.int32)
    post_tf = counts.to(torch.int32)
    df = torch.bincount(term, minlength=vocab)
    indptr = torch.zeros(vocab + 1, dtype=torch.int64, device=device)
    torch.cumsum(df, 0, out=indptr[1:])
    return dict(indptr=indptr.cpu().numpy(), post_doc=post_doc.contiguous(), post_tf=post_tf.contiguous(),
                doc_len=lens.to(torch.int32).cpu().numpy(), df=df, total_len=total)


def bm25_idf(df: np.ndarray, n_docs: int, epsilon: float = BM25_EPSILON) -> np.ndarray:
    """rank_bm25's `_calc_idf` over global document frequencies (Python floats, vocabulary order).
    Terms with df == 0 (possible in a synthetic vocabulary) get the idf of an unseen term and are never
    queried; they are left out of the average like terms that do not exist."""
    idf = np.zeros(len(df), dtype=np.float64)
    idf_sum = 0
    n_seen = 0
    negative: List[int] = []
    for t, f in enumerate(df.tolist()):
        if f == 0:
            continue
        v = math.log(n_docs - f + 0.5) - math.log(f + 0.5)
        idf[t] = v
        idf_sum += v
        n_seen += 1
        if v < 0:
            negative.append(t)
    eps = epsilon * (idf_sum / max(n_seen, 1))
    for t in negative:
        idf[t] = eps
    return idf


def bm25_queries(post: Dict[str, object], n_queries: int, seed: int, n_terms: int = 9, dup_rate: float = 0.065):
    """Term-id queries drawn from the terms of random documents of this shard (host numpy lists)."""
    rng = np.random.default_rng(seed)
    post_doc = post["post_doc"]
    indptr = post["indptr"]
    device = post_doc.device
    n_docs = len(post["doc_len"])
    term_of_posting = torch.repeat_interleave(
        torch.arange(len(indptr) - 1, device=device, dtype=torch.int32),
        torch.from_numpy(np.diff(indptr)).to(device))
    out = []
    for _ in range(n_queries):
        terms = np.empty(0, dtype=np.int32)
        while len(terms) < 2:
            d = int(rng.integers(n_docs))
            terms = term_of_posting[post_doc == d].cpu().numpy()
        q = rng.choice(terms, size=min(n_terms, len(terms)), replace=False).astype(np.int32)
        if rng.random() < dup_rate:
            q[-1] = q[0]
        out.append(q)
    return out
