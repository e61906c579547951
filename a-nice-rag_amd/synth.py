"""Synthetic corpora of the shapes BASELINE.json names (no corpus ships with the reference:
its *.db / *.pkl are git-ignored and the NICE text is not redistributable).

Generated with torch (PyTorch is plumbing here: device memory + RNG), per SURVEY.md
section 8(d): unit-norm N(0,1) rows; queries = a corpus row + 0.05 noise, re-normalised; BM25:
200k-term vocabulary, Zipf(1.07) term draw, log-normal document length (median 120, mean ~150),
9-term queries drawn from a document's own terms, 6.5 % with one duplicated term;
k1=1.7, b=0.83, epsilon=0.05 (src/processing/bm25_search.py:134-139).

The corpus is a function of (seed, GLOBAL row number) only: rows and documents are drawn in fixed-size
blocks, block b from a generator seeded with (seed, b), so any row range [lo, hi) can be produced on its
own -- the union of the N ranks' shards is bit-identical to the single-GPU corpus (bench.py checks the
sharded answers against a single index built from the same blocks), and queries are planted next to
GLOBAL rows, identical on every rank without a broadcast.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

BM25_K1, BM25_B, BM25_EPSILON = 1.7, 0.83, 0.05
DENSE_BLOCK = 16384   # rows per seeded block of the dense corpus
DOC_BLOCK = 65536     # documents per seeded block of the BM25 collection


def _block_generator(device, seed: int, block: int) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 1_000_003 + int(block))
    return g


def _dense_block(block: int, dim: int, seed: int, device) -> torch.Tensor:
    """Rows [block * DENSE_BLOCK, (block + 1) * DENSE_BLOCK) of the global corpus (always a whole block, so a
    row does not depend on where the corpus or a shard ends)."""
    blk = torch.randn((DENSE_BLOCK, dim), device=device, dtype=torch.float32,
                      generator=_block_generator(device, seed, block))
    blk /= blk.norm(dim=1, keepdim=True)
    return blk


def dense_corpus(n_rows: int, dim: int, seed: int, device, row_lo: int = 0) -> torch.Tensor:
    """Rows [row_lo, row_lo + n_rows) of the global corpus `seed`."""
    e = torch.empty((n_rows, dim), device=device, dtype=torch.float32)
    row_hi = row_lo + n_rows
    for b in range(row_lo // DENSE_BLOCK, (row_hi + DENSE_BLOCK - 1) // DENSE_BLOCK):
        b_lo = b * DENSE_BLOCK
        lo, hi = max(row_lo, b_lo), min(row_hi, b_lo + DENSE_BLOCK)
        e[lo - row_lo:hi - row_lo] = _dense_block(b, dim, seed, device)[lo - b_lo:hi - b_lo]
    return e


def dense_queries(e: torch.Tensor, n_queries: int, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (queries [nq, d], planted rows [nq]): query i sits next to row planted[i] of the LOCAL matrix `e`."""
    g = torch.Generator(device=e.device)
    g.manual_seed(seed)
    rows = torch.randint(0, e.shape[0], (n_queries,), device=e.device, generator=g)
    q = e[rows] + 0.05 * torch.randn((n_queries, e.shape[1]), device=e.device, dtype=torch.float32, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    return q.contiguous(), rows


def dense_queries_global(n_total_rows: int, dim: int, n_queries: int, seed: int, corpus_seed: int,
                         device) -> Tuple[torch.Tensor, np.ndarray]:
    """Queries planted next to GLOBAL rows of corpus `corpus_seed` (regenerating the rows' blocks), so that every
    rank of a sharded run -- and the single-GPU run -- asks the same questions.  -> (queries, planted rows)."""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, n_total_rows, size=n_queries)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base = torch.empty((n_queries, dim), device=device, dtype=torch.float32)
    cache: Dict[int, torch.Tensor] = {}
    for i, r in enumerate(rows.tolist()):
        b = r // DENSE_BLOCK
        if b not in cache:
            cache.clear()  # one block (50 MB at 768-d) at a time
            cache[b] = _dense_block(b, dim, corpus_seed, device)
        base[i] = cache[b][r - b * DENSE_BLOCK]
    q = base + 0.05 * torch.randn((n_queries, dim), device=device, dtype=torch.float32, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    return q.contiguous(), rows


def _zipf_cdf(vocab: int, zipf_s: float, device) -> torch.Tensor:
    p = 1.0 / torch.arange(1, vocab + 1, device=device, dtype=torch.float64) ** zipf_s
    return torch.cumsum(p / p.sum(), 0).to(torch.float32)


def _doc_block(block: int, vocab: int, seed: int, device, cdf: torch.Tensor, median_len: float, sigma: float):
    """Documents [block * DOC_BLOCK, (block + 1) * DOC_BLOCK) of the global collection:
    -> (lens int64 [DOC_BLOCK], doc int64 [tokens] (global doc number of each token), term int64 [tokens])."""
    g = _block_generator(device, seed, block)
    lens = torch.exp(torch.randn(DOC_BLOCK, device=device, generator=g) * sigma + math.log(median_len))
    lens = lens.clamp_(1, 2000).to(torch.int64)
    total = int(lens.sum().item())
    doc = torch.repeat_interleave(torch.arange(DOC_BLOCK, device=device, dtype=torch.int64), lens) + block * DOC_BLOCK
    u = torch.rand(total, device=device, generator=g)
    term = torch.searchsorted(cdf, u).clamp_(max=vocab - 1)
    return lens, doc, term


def bm25_postings(n_docs: int, vocab: int, seed: int, device, median_len: float = 120.0, sigma: float = 0.668,
                  zipf_s: float = 1.07, doc_lo: int = 0) -> Dict[str, object]:
    """Term-major CSR postings of documents [doc_lo, doc_lo + n_docs) of the global collection `seed`
    (document numbers in the postings are LOCAL: 0 .. n_docs-1).

    -> dict(indptr int64[vocab+1] (host numpy), post_doc / post_tf int32 (torch, on `device`),
            doc_len int32 (host numpy), df int64 (torch), total_len int)."""
    doc_hi = doc_lo + n_docs
    cdf = _zipf_cdf(vocab, zipf_s, device)
    all_lens, all_keys = [], []
    for b in range(doc_lo // DOC_BLOCK, (doc_hi + DOC_BLOCK - 1) // DOC_BLOCK):
        lens, doc, term = _doc_block(b, vocab, seed, device, cdf, median_len, sigma)
        b_lo = b * DOC_BLOCK
        lo, hi = max(doc_lo, b_lo), min(doc_hi, b_lo + DOC_BLOCK)
        all_lens.append(lens[lo - b_lo:hi - b_lo])
        if lo != b_lo or hi != b_lo + DOC_BLOCK:
            keep = (doc >= lo) & (doc < hi)
            doc, term = doc[keep], term[keep]
        all_keys.append(term * n_docs + (doc - doc_lo))
        del doc, term
    lens = torch.cat(all_lens)
    total = int(lens.sum().item())
    keys = torch.cat(all_keys)
    del all_keys
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


def _draw_query(rng, terms: np.ndarray, n_terms: int, dup_rate: float) -> np.ndarray:
    q = rng.choice(terms, size=min(n_terms, len(terms)), replace=False).astype(np.int32)
    if rng.random() < dup_rate:
        q[-1] = q[0]
    return q


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
        out.append(_draw_query(rng, terms, n_terms, dup_rate))
    return out


def bm25_queries_global(n_total_docs: int, vocab: int, n_queries: int, seed: int, corpus_seed: int, device,
                        n_terms: int = 9, dup_rate: float = 0.065, median_len: float = 120.0, sigma: float = 0.668,
                        zipf_s: float = 1.07):
    """Like `bm25_queries`, drawn from GLOBAL documents of collection `corpus_seed` (regenerating the documents'
    blocks): the same queries on every rank and in the single-GPU run."""
    rng = np.random.default_rng(seed)
    cdf = _zipf_cdf(vocab, zipf_s, device)
    out = []
    cache: Dict[int, tuple] = {}
    for _ in range(n_queries):
        terms = np.empty(0, dtype=np.int32)
        while len(terms) < 2:
            d = int(rng.integers(n_total_docs))
            b = d // DOC_BLOCK
            if b not in cache:
                cache.clear()
                _, doc, term = _doc_block(b, vocab, corpus_seed, device, cdf, median_len, sigma)
                cache[b] = (doc, term)
            doc, term = cache[b]
            terms = torch.unique(term[doc == d]).to(torch.int32).cpu().numpy()
        out.append(_draw_query(rng, terms, n_terms, dup_rate))
    return out
