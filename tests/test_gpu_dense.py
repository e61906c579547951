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
    checked = checked64 = 0
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
                if c.get("qdtype") == "float64":
                    # the reference scored this case in fp64 (numpy promotes, search_engine.py:81): so does the fp64
                    # entry point -- its fp64 similarities at 1e-12, not 1e-4 (sums of <= 384 products of unit vectors)
                    q64 = q.astype(np.float64)
                    d64, s64, m64 = idx.dense_search_f64(q64, c["k"], allow)
                    assert m64 == len(c["rows"]) and c["sim_dtype"] in (None, "float64")
                    assert_ranking_matches(c["rows"], c["sims"], d64[:m64], s64[:m64], 1e-12,
                                           ref_search.dense_scores(q64, e), f"golden fp64 n={n} k={c['k']} f={c['filter']}")
                    assert np.all(d64[m64:] == -1)
                    checked64 += 1
                checked += 1
    assert checked > 60 and checked64 > 25


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


def test_fp64_query_scores_like_numpy(Index):
    """A float64 query (the reference's text path, search_engine.py:157): numpy promotes the fp32 matrix and scores in
    fp64; `anrag_dense_search_f64` accumulates in fp64 on the device.  768-d and an odd dimension, k below and above
    the fused limit, a filter; and the Python shim routes float64 queries there."""
    from oracle import ref_search

    rng = np.random.default_rng(77)
    for n, d in ((20000, 768), (501, 100)):
        e = rng.standard_normal((n, d), dtype=np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        sid = (np.arange(n) % 5).astype(np.uint16)
        allow = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
        with Index(0) as idx:
            idx.dense_load(e, source_id=sid)
            for t in range(3):
                q = rng.standard_normal(d)
                q /= np.linalg.norm(q)  # float64, NOT representable in fp32
                want = np.dot(q.reshape(1, -1), e.T).flatten()  # exactly what the reference computes (:81)
                assert want.dtype == np.float64
                for k, flt in ((10, None), (25, allow), (300, None), (n + 7, allow)):
                    doc, score, cnt = idx.dense_search_f64(q, k, flt)
                    order = ref_search.canonical_topk(want, k, None if flt is None else flt.astype(bool)[sid])
                    assert cnt == len(order)
                    assert_ranking_matches(order, want[order], doc[:cnt], score[:cnt], 1e-12, want, f"fp64 n={n} k={k}")
                    # the fp32 entry point rounds the query: inside 1e-4, but not these bits
                    if k <= 25 and flt is None:
                        d32, s32, _ = idx.dense_search(q.astype(np.float32), k)
                        assert np.max(np.abs(s32[0] - want[order])) < 1e-4 and np.max(np.abs(s32[0] - want[order])) > 1e-12


def test_nan_scores_rank_first_like_numpy(Index):
    """numpy ranks NaN above every number (argpartition / argsort, search_engine.py:83-87): a row whose dot product is
    NaN is returned FIRST by the reference.  The device carries such a score as +inf: same ranks, reported +inf.
    K1 (batch = 1), its score-array form (k > 64), K2 (batched) and the fp64 entry point."""
    rng = np.random.default_rng(5)
    n, d = 70000, 256
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    bad = [123, 40000, 69999]
    e[bad[0], 7] = np.nan
    e[bad[1], :] = np.nan
    e[bad[2], 100] = np.inf  # x * inf summed with finite terms = +-inf or NaN; here q[100] > 0 -> +inf
    q = np.abs(rng.standard_normal((20, d), dtype=np.float32))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    with np.errstate(invalid="ignore"):
        sims = np.dot(q[0].reshape(1, -1), e.T).flatten()
    k = 10
    ref_top = np.argpartition(sims, -k)[-k:]
    ref_top = ref_top[np.argsort(sims[ref_top])[::-1]]  # the reference's idiom: NaN rows first, then +inf, then numbers
    assert set(ref_top[:2].tolist()) == set(bad[:2]) and ref_top[2] == bad[2]
    with Index(0) as idx:
        idx.dense_load(e)
        doc, score, cnt = idx.dense_search(q[0], k)                      # K1
        assert set(doc[0, :3].tolist()) == set(bad) and doc[0, 3:].tolist() == ref_top[3:].tolist()
        assert np.all(np.isposinf(score[0, :3])) and np.all(np.isfinite(score[0, 3:]))
        doc_l, score_l, _ = idx.dense_search(q[0], 200)                  # score array + sort
        assert set(doc_l[0, :3].tolist()) == set(bad) and doc_l[0, 3:10].tolist() == ref_top[3:].tolist()
        doc_b, score_b, _ = idx.dense_search(q, k)                       # 20 queries: K2
        assert set(doc_b[0, :3].tolist()) == set(bad) and doc_b[0, 3:].tolist() == ref_top[3:].tolist()
        d64, s64, _ = idx.dense_search_f64(q[0].astype(np.float64), k)   # fp64 entry point
        assert set(d64[:3].tolist()) == set(bad) and d64[3:].tolist() == ref_top[3:].tolist()


def test_bm25_load_rejects_non_finite_statistics(Index):
    from anrag import _native as nat

    indptr = np.array([0, 1, 2], dtype=np.int64)
    post_doc = np.array([0, 1], dtype=np.int32)
    post_tf = np.array([1, 1], dtype=np.int32)
    doc_len = np.array([3, 4], dtype=np.int32)
    with Index(0) as idx:
        for idf, avgdl in (([1.0, np.nan], 3.5), ([np.inf, 1.0], 3.5), ([1.0, 1.0], np.inf)):
            with pytest.raises(nat.AnragError):
                idx.bm25_load(indptr, post_doc, post_tf, np.array(idf), doc_len, avgdl, 1.7, 0.83)
        idx.bm25_load(indptr, post_doc, post_tf, np.array([1.0, -0.5]), doc_len, 3.5, 1.7, 0.83)
