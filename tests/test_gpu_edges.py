"""GPU: edge cases the reference's behaviour defines (nulls, empties, ragged sizes, long queries)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def test_null_sources_and_tiny_corpora():
    from oracle import ref_search
    from anrag.search_engine import SearchEngine

    rng = np.random.default_rng(0)
    e = rng.standard_normal((5, 64), dtype=np.float32)
    df = pd.DataFrame({"id": list("abcde"), "source": ["CG1", None, "ng2", float("nan"), "QS3"], "embedding": list(e)})
    se = SearchEngine(None, None)
    q = e[2] + 0.01
    full = ref_search.dense_scores(q.astype(np.float32), e)
    r = se.similarity_search_with_embedding(q, df, "m", 25)               # k > n: everything, ranked
    assert r["id"].tolist() == [df["id"][i] for i in np.lexsort((np.arange(5), -full))]
    r = se.similarity_search_with_embedding(q, df, "m", 25, "CG,NG")      # null sources never match (na=False)
    assert sorted(r["id"].tolist()) == ["a", "c"]
    r = se.similarity_search_with_embedding(q, df, "m", 1, "qs")
    assert r["id"].tolist() == ["e"]
    assert se.similarity_search_with_embedding(q, df.iloc[0:0], "m", 3).empty


def test_bm25_ragged():
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    corpus = [["a"], ["a", "b", "a"], ["c"]]
    ref = BM25Okapi(corpus, 1.7, 0.83, 0.05)
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    with Index(0) as idx:
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        for q in (["a"], ["b", "c", "b"], ["zzz"], ["a"] * 300 + ["c"] * 77):   # 377 tokens: three term batches
            s = idx.bm25_scores(bi.term_ids(q))
            assert np.array_equal(s, ref.get_scores(q)), q
            doc, sc, cnt = idx.bm25_search(bi.term_ids(q), 25)                   # k > n_docs
            want = np.lexsort((np.arange(3), -s))
            assert cnt == 3 and doc[:3].tolist() == want.tolist() and np.all(doc[3:] == -1)
    one = Bm25Index([["solo", "solo"]], 1.7, 0.83, 0.05)
    with Index(0) as idx:
        idx.bm25_load(one.indptr, one.post_doc, one.post_tf, one.idf, one.doc_len, one.avgdl, one.k1, one.b)
        assert np.array_equal(idx.bm25_scores(one.term_ids(["solo"])), BM25Okapi([["solo", "solo"]], 1.7, 0.83, 0.05).get_scores(["solo"]))


def test_wrrf_degenerate_lists():
    from oracle import ref_search
    from anrag.search_engine import SearchEngine

    se = SearchEngine(None, None)
    w = {"m1": 5.0, "BM25": 1.0}
    for lists in ([(["a", "b"], "m1")], [([], "m1"), (["x"], "BM25")], [(["a", "b", "a"], "m1"), (["b"], "BM25")],
                  [([], "m1")]):
        assert se.weighted_reciprocal_rank_fusion(lists, w, 60) == ref_search.weighted_reciprocal_rank_fusion(lists, w, 60)
    assert se.weighted_reciprocal_rank_fusion([], w, 60) == []
