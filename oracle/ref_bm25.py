"""Oracle (test infrastructure): restatement of `rank_bm25.BM25Okapi`.

PARITY UNPINNED for the score values: the reference calls the third-party
package `rank-bm25` (requirements.txt:5, no version pin; 0.2.2 is the only
release line since 2022) at src/search_engine.py:219 (`bm25.get_scores`) and
src/processing/bm25_search.py:77 (`BM25Okapi(corpus, k1=, b=, epsilon=)`).
Its source is not under /root/reference and the package is not installed in
this image, and the reference holds no test or golden vector at that call
boundary.  What follows restates the published 0.2.2 algorithm: same data
structures, same operator order, same Python-float / numpy-fp64 types.

Two independent forms are kept and tested bit-for-bit against each other:
  * `BM25Okapi`   -- dict-of-tf per document, `get_scores` as a numpy fp64
                     vector expression (the shape of the upstream class);
  * `CsrBM25`     -- term-major CSR postings + per-posting arithmetic, the
                     shape the HIP kernel uses.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np


class BM25Okapi:
    """rank_bm25.BM25Okapi (0.2.2): `__init__` -> `_initialize` -> `_calc_idf`,
    and `get_scores`.  Reference parameters: k1=1.7, b=0.83, epsilon=0.05
    (src/processing/bm25_search.py:48-50, :134-139); package defaults are
    k1=1.5, b=0.75, epsilon=0.25."""

    def __init__(self, corpus: Sequence[Sequence[str]], k1=1.5, b=0.75, epsilon=0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.avgdl = 0
        self.doc_freqs: List[Dict[str, int]] = []
        self.idf: Dict[str, float] = {}
        self.doc_len: List[int] = []
        nd = self._initialize(corpus)
        self._calc_idf(nd)

    def _initialize(self, corpus):
        nd: Dict[str, int] = {}  # word -> number of documents containing it
        num_doc = 0
        for document in corpus:
            self.doc_len.append(len(document))
            num_doc += len(document)
            frequencies: Dict[str, int] = {}
            for word in document:
                if word not in frequencies:
                    frequencies[word] = 0
                frequencies[word] += 1
            self.doc_freqs.append(frequencies)
            for word in frequencies:
                nd[word] = nd.get(word, 0) + 1
            self.corpus_size += 1
        self.avgdl = num_doc / self.corpus_size
        return nd

    def _calc_idf(self, nd):
        # idf = ln(N - df + 0.5) - ln(df + 0.5); terms with idf < 0 are floored
        # to epsilon * mean(idf over the vocabulary, BEFORE flooring).
        idf_sum = 0
        negative_idfs = []
        for word, freq in nd.items():
            idf = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
            self.idf[word] = idf
            idf_sum += idf
            if idf < 0:
                negative_idfs.append(word)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for word in negative_idfs:
            self.idf[word] = eps

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        # Term-at-a-time, in query order, duplicates counted again.
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(doc.get(q) or 0) for doc in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (
                q_freq * (self.k1 + 1) / (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl))
            )
        return score


class CsrBM25:
    """Independent restatement over term-major CSR postings.

    Layout (the one `anrag_bm25_load` takes, include/anrag.h):
      vocab[t]            term string of term id t (insertion order of first
                          appearance while walking the corpus, as `nd` above)
      indptr[t]..[t+1]    slice of `post_doc` / `post_tf` for term t, documents
                          ascending inside a term
      idf[t], doc_len[d], avgdl, k1, b
    """

    def __init__(self, corpus: Sequence[Sequence[str]], k1=1.5, b=0.75, epsilon=0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        term_id: Dict[str, int] = {}
        plists: List[List[int]] = []
        tfs: List[List[int]] = []
        doc_len = []
        total = 0
        for d, document in enumerate(corpus):
            doc_len.append(len(document))
            total += len(document)
            counts: Dict[int, int] = {}
            for word in document:
                t = term_id.get(word)
                if t is None:
                    t = term_id[word] = len(term_id)
                    plists.append([])
                    tfs.append([])
                counts[t] = counts.get(t, 0) + 1
            for t, c in counts.items():
                plists[t].append(d)
                tfs[t].append(c)
        self.n_docs = len(doc_len)
        self.doc_len = np.asarray(doc_len, dtype=np.int32)
        self.avgdl = total / self.n_docs
        self.term_id = term_id
        self.vocab = list(term_id)
        df = np.array([len(p) for p in plists], dtype=np.int64)
        self.indptr = np.zeros(len(plists) + 1, dtype=np.int64)
        np.cumsum(df, out=self.indptr[1:])
        self.post_doc = np.fromiter((d for p in plists for d in p), dtype=np.int32, count=int(df.sum()))
        self.post_tf = np.fromiter((c for p in tfs for c in p), dtype=np.int32, count=int(df.sum()))
        # idf with Python floats and math.log, summed in vocabulary order
        idf = []
        idf_sum = 0
        for f in df.tolist():
            v = math.log(self.n_docs - f + 0.5) - math.log(f + 0.5)
            idf.append(v)
            idf_sum += v
        self.average_idf = idf_sum / len(idf)
        eps = self.epsilon * self.average_idf
        self.idf = np.array([eps if v < 0 else v for v in idf], dtype=np.float64)

    def term_ids(self, query: Sequence[str]) -> List[int]:
        """Query tokens -> term ids in query order; unknown tokens map to -1
        (they contribute `(idf.get(q) or 0) == 0`, i.e. nothing)."""
        return [self.term_id.get(q, -1) for q in query]

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        return csr_get_scores(self.indptr, self.post_doc, self.post_tf, self.idf, self.doc_len, self.avgdl,
                              self.k1, self.b, self.term_ids(query))


def csr_get_scores(indptr, post_doc, post_tf, idf, doc_len, avgdl, k1, b, term_ids) -> np.ndarray:
    """`BM25Okapi.get_scores` over raw CSR arrays (term ids in query order, negative = unknown token).
    Same operator order as the dict form: only documents that hold the term receive a non-zero addend,
    and adding the reference's `idf * 0.0` to the others changes nothing."""
    score = np.zeros(len(doc_len))
    for t in term_ids:
        if t < 0:
            continue
        w = idf[t] or 0
        lo, hi = indptr[t], indptr[t + 1]
        docs = post_doc[lo:hi]
        f = post_tf[lo:hi].astype(np.int64)
        dl = doc_len[docs].astype(np.int64)
        score[docs] += w * (f * (k1 + 1) / (f + k1 * (1 - b + b * dl / avgdl)))
    return score
