"""Host-side BM25 index builder: tokenised corpus -> term-major CSR postings + statistics.

Takes the place of `BM25Okapi(corpus, k1=, b=, epsilon=)` at
src/processing/bm25_search.py:77 of the reference (rank_bm25's `_initialize` and
`_calc_idf`): same vocabulary order (first appearance walking the corpus), same
Python-float idf (`math.log`, summed in vocabulary order, epsilon floor for
negative idf), same avgdl.  What it produces is what `anrag_bm25_load` uploads;
scoring itself happens only on the GPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np


class Bm25Index:
    def __init__(self, corpus: Sequence[Sequence[str]], k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25):
        self.k1, self.b, self.epsilon = float(k1), float(b), float(epsilon)
        vocab: Dict[str, int] = {}
        term_of: List[int] = []   # one entry per (doc, distinct term), doc-major
        doc_of: List[int] = []
        tf_of: List[int] = []
        doc_len = np.empty(len(corpus), dtype=np.int32)
        total = 0
        for d, document in enumerate(corpus):
            doc_len[d] = len(document)
            total += len(document)
            counts: Dict[int, int] = {}
            for word in document:
                t = vocab.get(word)
                if t is None:
                    t = vocab[word] = len(vocab)
                counts[t] = counts.get(t, 0) + 1
            term_of.extend(counts.keys())
            tf_of.extend(counts.values())
            doc_of.extend([d] * len(counts))
        self.n_docs = len(corpus)
        if self.n_docs == 0:
            raise ValueError("BM25 index over an empty corpus")
        self.doc_len = doc_len
        self.avgdl = total / self.n_docs
        self.vocab = vocab
        n_terms = len(vocab)
        term_arr = np.asarray(term_of, dtype=np.int64)
        order = np.argsort(term_arr, kind="stable")  # stable: documents stay ascending inside a term
        self.post_doc = np.asarray(doc_of, dtype=np.int32)[order]
        self.post_tf = np.asarray(tf_of, dtype=np.int32)[order]
        df = np.bincount(term_arr, minlength=n_terms).astype(np.int64)
        self.indptr = np.zeros(n_terms + 1, dtype=np.int64)
        np.cumsum(df, out=self.indptr[1:])
        # idf exactly as rank_bm25._calc_idf: Python floats, vocabulary order
        idf = np.empty(n_terms, dtype=np.float64)
        idf_sum = 0
        negative = []
        n = self.n_docs
        for t, f in enumerate(df.tolist()):
            v = math.log(n - f + 0.5) - math.log(f + 0.5)
            idf[t] = v
            idf_sum += v
            if v < 0:
                negative.append(t)
        self.average_idf = idf_sum / n_terms if n_terms else 0.0
        eps = self.epsilon * self.average_idf
        for t in negative:
            idf[t] = eps
        self.idf = idf

    @property
    def n_terms(self) -> int:
        return len(self.vocab)

    @property
    def n_postings(self) -> int:
        return int(self.indptr[-1])

    def term_ids(self, tokens: Sequence[str]) -> np.ndarray:
        """Query tokens -> term ids in query order; -1 for tokens outside the vocabulary
        (`self.idf.get(q) or 0` in rank_bm25: they contribute nothing)."""
        get = self.vocab.get
        return np.fromiter((get(q, -1) for q in tokens), dtype=np.int32, count=len(tokens))
