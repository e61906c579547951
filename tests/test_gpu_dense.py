"""GPU parity: K1 dense scan + top-k through the C ABI vs the oracle and the reference's golden vectors.

Tolerance: |dscore| <= 1e-4 (BASELINE.json north_star, "within 1e-4 cosine for the dense side");
rows must match wherever the reference's scores are further apart than that."""
import numpy as np
import pytest

from helpers import assert_ranking_matches, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _intern(sources):
    table = {}
    ids = np.array([table.setdefault(s, len(table)) for s in sources], dtype=np.uint16)
    return ids, list(table)


def _allow(distinct, flt):
    from oracle import ref_search

    return ref_search.dense_filter_mask(distinct, flt).astype(np.uint8)


@pytest.fixture(scope="module")
def Index():
    from anrag.index import Index

    return Index


def test_golden_dense_vectors(Index):
    from oracle.make_golden import synth_dense, synth_query, synth_sources
    from oracle import ref_search

    cases = [c for c in load_golden("ref_dense.json") if "rows" in c]
    by_corpus = {}
    for c in cases:
        by_corpus.setdefault((c["n"], c["d"], c["corpus_seed"], tuple(c.get("dups", ()))), []).append(c)
    checked = 0
    for (n, d, seed, dups), group in by_corpus.items():
        e = synth_dense(n, d, seed)
        for j in dups:
            e[j] = e[group[0]["dup_of"]]
        sources = synth_sources(n, group[0]["source_seed"])
        sid, distinct = _intern(sources)
        with Index(0) as idx:
            idx.dense_load(e, source_id=sid)
            for c in group:
                q = e[c["dup_of"]].copy() if dups else synth_query(e, c["query_seed"], c["query_row"])
                allow = _allow(distinct, c["filter"]) if c["filter"] else None
                doc, score, count = idx.dense_search(q.astype(np.float32), c["k"], allow)
                m = int(count[0])
                assert m == len(c["rows"]), (c["k"], c["filter"], m)
                full = ref_search.dense_scores(q.astype(np.float32), e)
                assert_ranking_matches(c["rows"], c["sims"], doc[0, :m], score[0, :m], TOL, full,
                                       f"golden n={n} d={d} k={c['k']} f={c['filter']} {c['qdtype']}")
                assert np.all(doc[0, m:] == -1)
                checked += 1
    assert checked > 60


@pytest.mark.parametrize("n,d", [(5000, 768), (3001, 384), (1000, 1024), (777, 2048), (600, 3072), (300, 4096),
                                 (900, 1536), (2000, 192), (500, 100), (64, 8), (3, 768), (1, 384)])
def test_random_dense_vs_oracle(Index, n, d):
    from oracle import ref_search

    rng = np.random.default_rng(n * 31 + d)
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    sources = [("CG%d" % (i % 7)) if i % 3 else ("NG%d" % (i % 5)) for i in range(n)]
    sid, distinct = _intern(sources)
    with Index(0) as idx:
        idx.dense_load(e, source_id=sid, doc_id_base=1000)
        qs = rng.standard_normal((3, d), dtype=np.float32)
        qs /= np.linalg.norm(qs, axis=1, keepdims=True)
        # all scores of one query
        got = idx.dense_scores(qs[0])
        ref = ref_search.dense_scores(qs[0], e)
        assert np.max(np.abs(got - ref)) <= TOL
        for k in (1, 10, 25, 64):
            for flt in (None, "NG"):
                allow = _allow(distinct, flt) if flt else None
                doc, score, count = idx.dense_search(qs, k, allow)  # n_queries = 3 in one call
                for qi in range(3):
                    rows, sims = ref_search.similarity_search_with_embedding(qs[qi], e, sources, k, flt, canonical=True)
                    m = int(count[qi])
                    assert m == len(rows)
                    full = ref_search.dense_scores(qs[qi], e)
                    assert_ranking_matches(rows + 1000, sims, doc[qi, :m], score[qi, :m], TOL, None, f"n={n} d={d} k={k}")
                    # every returned score is that row's true score
                    assert np.max(np.abs(full[doc[qi, :m] - 1000] - score[qi, :m])) <= TOL


def test_exact_ties_are_row_ordered(Index):
    """Duplicate rows give bit-equal scores: the build's rule is row ascending."""
    rng = np.random.default_rng(5)
    e = rng.standard_normal((4096, 256), dtype=np.float32)
    e[100] = e[7]; e[3000] = e[7]; e[4095] = e[7]; e[2048] = e[7]
    with Index(0) as idx:
        idx.dense_load(e)
        doc, score, count = idx.dense_search(e[7], 5)
        assert doc[0].tolist() == [7, 100, 2048, 3000, 4095]
        assert len(set(score[0].tolist())) == 1


def test_explicit_doc_ids_and_errors(Index):
    from anrag._native import AnragError

    rng = np.random.default_rng(9)
    e = rng.standard_normal((300, 128), dtype=np.float32)
    ids = rng.permutation(10_000)[:300].astype(np.int64)
    with Index(0) as idx:
        with pytest.raises(AnragError):
            idx.dense_search(e[0], 5)  # search before load
        idx.dense_load(e, doc_id=ids)
        doc, score, count = idx.dense_search(e[17], 3)
        assert doc[0, 0] == ids[17]
        with pytest.raises(AnragError):
            idx.dense_search(e[0], 5, allow_source=np.ones(4, np.uint8))  # filter without source ids
