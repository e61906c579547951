"""CPU: known-answer and cross-restatement checks for the BM25 oracle.

rank_bm25 is not installed and the reference pins nothing at that boundary
("parity unpinned", oracle/ref_bm25.py).  These tests anchor the restatement on
(i) values worked out by hand from the published formula and (ii) bit-for-bit
agreement between two independently written forms.
"""
import math

import numpy as np

from oracle.make_golden import synth_chunks
from oracle.ref_bm25 import BM25Okapi, CsrBM25

CORPUS = [
    ["asthma", "inhaler", "dose", "asthma"],          # dl 4, asthma tf 2
    ["asthma", "child"],                              # dl 2
    ["dose", "adult", "review", "dose", "dose"],      # dl 5, dose tf 3
    ["asthma", "review"],                             # dl 2
    ["stroke"],                                       # dl 1
]


def test_index_statistics_by_hand():
    bm = BM25Okapi(CORPUS, k1=1.7, b=0.83, epsilon=0.05)
    assert bm.corpus_size == 5 and bm.doc_len == [4, 2, 5, 2, 1]
    assert bm.avgdl == 14 / 5
    # df: asthma 3, inhaler 1, dose 2, child 1, adult 1, review 2, stroke 1
    raw = {"asthma": math.log(5 - 3 + 0.5) - math.log(3.5), "inhaler": math.log(4.5) - math.log(1.5),
           "dose": math.log(3.5) - math.log(2.5), "child": math.log(4.5) - math.log(1.5),
           "adult": math.log(4.5) - math.log(1.5), "review": math.log(3.5) - math.log(2.5),
           "stroke": math.log(4.5) - math.log(1.5)}
    assert list(bm.idf) == list(raw)  # vocabulary order = first appearance
    avg = sum(raw.values()) / 7
    assert bm.average_idf == avg
    assert raw["asthma"] < 0  # df > N/2  -> floored to epsilon * average idf
    assert bm.idf["asthma"] == 0.05 * avg
    for w in raw:
        if w != "asthma":
            assert bm.idf[w] == raw[w]


def _by_hand(bm, query):
    out = []
    for d, doc in enumerate(CORPUS):
        s = 0.0
        for q in query:
            f = doc.count(q)
            idf = bm.idf.get(q) or 0
            s += idf * (f * (bm.k1 + 1) / (f + bm.k1 * (1 - bm.b + bm.b * len(doc) / bm.avgdl)))
        out.append(s)
    return out


def test_scores_by_hand_including_duplicates_and_unknowns():
    bm = BM25Okapi(CORPUS, k1=1.7, b=0.83, epsilon=0.05)
    for query in (["dose"], ["asthma", "dose"], ["dose", "asthma"], ["dose", "dose"], ["nothere"],
                  ["review", "nothere", "asthma", "review"], []):
        got = bm.get_scores(query)
        assert got.dtype == np.float64
        assert got.tolist() == _by_hand(bm, query), query
    # a duplicated token counts twice; an unknown token changes nothing
    a = bm.get_scores(["dose"])
    assert np.array_equal(bm.get_scores(["dose", "dose"]), a + a)
    assert np.array_equal(bm.get_scores(["dose", "zzz"]), a)
    # documents without the term stay at exactly 0.0
    assert a[1] == 0.0 and a[3] == 0.0 and a[4] == 0.0


def test_idf_exactly_zero_is_dropped():
    # N=2, df=1: ln(1.5) - ln(1.5) == 0.0 -> `idf.get(q) or 0` -> 0
    bm = BM25Okapi([["a", "b"], ["a", "c"]], k1=1.7, b=0.83, epsilon=0.05)
    assert bm.idf["b"] == 0.0
    assert bm.get_scores(["b"]).tolist() == [0.0, 0.0]


def test_csr_restatement_is_bit_identical():
    chunks = [c["tokens"] for c in synth_chunks(400, 42) if c["tokens"]]
    a = BM25Okapi(chunks, k1=1.7, b=0.83, epsilon=0.05)
    b = CsrBM25(chunks, k1=1.7, b=0.83, epsilon=0.05)
    assert b.vocab == list(a.idf)
    assert b.idf.tolist() == [a.idf[w] for w in b.vocab]
    assert b.avgdl == a.avgdl and b.doc_len.tolist() == a.doc_len
    # postings ascending by document inside each term
    for t in range(len(b.vocab)):
        docs = b.post_doc[b.indptr[t]:b.indptr[t + 1]]
        assert np.all(np.diff(docs) > 0)
        assert [a.doc_freqs[d][b.vocab[t]] for d in docs] == b.post_tf[b.indptr[t]:b.indptr[t + 1]].tolist()
    rng = np.random.default_rng(7)
    for _ in range(50):
        doc = chunks[int(rng.integers(len(chunks)))]
        q = [str(x) for x in rng.choice(doc, size=int(rng.integers(1, 8)))]
        if rng.random() < 0.3:
            q.append(q[0])
        if rng.random() < 0.3:
            q.insert(0, "unknownterm")
        assert np.array_equal(a.get_scores(q), b.get_scores(q)), q
