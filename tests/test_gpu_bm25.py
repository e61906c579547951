"""GPU parity: K3 BM25 scorer + selection, K4 large-k, through the C ABI vs the oracle.
Bar: bit-exact fp64 scores; ids/ranks identical under (score desc, row asc)."""
import numpy as np
import pytest

from helpers import assert_ranking_matches

pytestmark = pytest.mark.gpu


def zipf_corpus(n_docs, vocab, seed, mean_len=40):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1) ** 1.07
    p /= p.sum()
    cdf = np.cumsum(p)
    lens = np.maximum(1, rng.lognormal(np.log(mean_len), 0.5, n_docs).astype(int))
    flat = np.searchsorted(cdf, rng.random(int(lens.sum())))
    flat = np.minimum(flat, vocab - 1)
    out, pos = [], 0
    for ln in lens:
        out.append([f"t{j}" for j in flat[pos:pos + ln]])
        pos += ln
    return out


@pytest.fixture(scope="module")
def small():
    from oracle.make_golden import synth_chunks
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    chunks = [c for c in synth_chunks(400, 42) if c["tokens"]]
    corpus = [c["tokens"] for c in chunks]
    sources = [c["source"] for c in chunks]
    ref = BM25Okapi(corpus, k1=1.7, b=0.83, epsilon=0.05)
    bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
    table = {}
    sid = np.array([table.setdefault(s, len(table)) for s in sources], dtype=np.uint16)
    idx = Index(0)
    idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid)
    yield corpus, sources, list(table), ref, bi, idx
    idx.close()


@pytest.fixture(scope="module")
def big():
    from oracle.ref_bm25 import CsrBM25
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    corpus = zipf_corpus(30000, 3000, 7)
    ref = CsrBM25(corpus, k1=1.7, b=0.83, epsilon=0.05)
    bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
    idx = Index(0)
    idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, doc_id_base=500)
    yield corpus, ref, bi, idx
    idx.close()


def test_index_builder_matches_oracle_statistics(small, big):
    corpus, sources, distinct, ref, bi, idx = small
    assert list(bi.vocab) == list(ref.idf)
    assert bi.idf.tolist() == [ref.idf[w] for w in bi.vocab]
    assert bi.avgdl == ref.avgdl and bi.doc_len.tolist() == ref.doc_len
    _, cref, cbi, _ = big
    assert np.array_equal(cbi.indptr, cref.indptr) and np.array_equal(cbi.post_doc, cref.post_doc)
    assert np.array_equal(cbi.post_tf, cref.post_tf) and np.array_equal(cbi.idf, cref.idf)
    assert (np.diff(cbi.indptr) >= 2048).sum() > 5, "corpus must exercise the frequent-term partition tables"


def _queries(corpus, rng, n):
    qs = []
    for _ in range(n):
        doc = corpus[int(rng.integers(len(corpus)))]
        q = [str(x) for x in rng.choice(doc, size=int(rng.integers(1, 12)))]
        r = rng.random()
        if r < 0.25:
            q.append(q[0])                # duplicated token
        elif r < 0.5:
            q.insert(len(q) // 2, "zz-unknown")
        qs.append(q)
    qs.append([])                          # empty query: all zeros
    qs.append(["zz-unknown"])
    return qs


def test_scores_bit_exact_small(small):
    corpus, sources, distinct, ref, bi, idx = small
    rng = np.random.default_rng(3)
    for q in _queries(corpus, rng, 40):
        got = idx.bm25_scores(bi.term_ids(q))
        want = ref.get_scores(q)
        assert np.array_equal(got, want), q


def test_scores_bit_exact_big(big):
    corpus, ref, bi, idx = big
    rng = np.random.default_rng(4)
    for q in _queries(corpus, rng, 25) + [["t0", "t1", "t2", "t0", "t5"]]:  # the most frequent terms
        got = idx.bm25_scores(bi.term_ids(q))
        want = ref.get_scores(q)
        assert np.array_equal(got, want), q


@pytest.mark.parametrize("k", [1, 10, 25, 64])
def test_topk_matches_oracle(small, big, k):
    from oracle import ref_search

    corpus, sources, distinct, ref, bi, idx = small
    rng = np.random.default_rng(10 + k)
    for q in _queries(corpus, rng, 15):
        if not q:
            continue  # _core_bm25_search returns [] before scoring (search_engine.py:216-217); shim's job
        scores = ref.get_scores(q)
        for flt in (None, "CG,NG", "ZZ"):
            allow = None if flt is None else ref_search.bm25_filter_mask(distinct, flt).astype(np.uint8)
            doc, sc, cnt = idx.bm25_search(bi.term_ids(q), k, allow)
            want = ref_search.core_bm25_search(scores, sources, k, flt, canonical=True)
            assert cnt == len(want)
            assert doc[:cnt].tolist() == want.tolist(), (q, flt)
            assert np.array_equal(sc[:cnt], scores[want])
            assert np.all(doc[cnt:] == -1)
    corpus, cref, cbi, cidx = big
    for q in _queries(corpus, rng, 10):
        if not q:
            continue
        scores = cref.get_scores(q)
        doc, sc, cnt = cidx.bm25_search(cbi.term_ids(q), k)
        want = ref_search.canonical_topk(scores, k)
        assert (doc[:cnt] - 500).tolist() == want.tolist(), q
        assert np.array_equal(sc[:cnt], scores[want])


@pytest.mark.parametrize("k", [65, 300, 30005])
def test_large_k_full_ranking(small, big, k):
    from oracle import ref_search

    corpus, sources, distinct, ref, bi, idx = small
    rng = np.random.default_rng(20)
    q = _queries(corpus, rng, 1)[0]
    scores = ref.get_scores(q)
    for flt in (None, "CG,NG"):
        allow = None if flt is None else ref_search.bm25_filter_mask(distinct, flt).astype(np.uint8)
        doc, sc, cnt = idx.bm25_search(bi.term_ids(q), k, allow)
        want = ref_search.core_bm25_search(scores, sources, k, flt, canonical=True)
        assert cnt == len(want) and doc[:cnt].tolist() == want.tolist()
        assert np.array_equal(sc[:cnt], scores[want])
    corpus, cref, cbi, cidx = big
    q = _queries(corpus, rng, 1)[0]
    scores = cref.get_scores(q)
    doc, sc, cnt = cidx.bm25_search(cbi.term_ids(q), k)
    want = ref_search.canonical_topk(scores, k)
    assert cnt == len(want) and (doc[:cnt] - 500).tolist() == want.tolist()


def test_dense_large_k():
    from oracle import ref_search
    from anrag.index import Index

    rng = np.random.default_rng(2)
    e = rng.standard_normal((3000, 384), dtype=np.float32)
    e[11] = e[5]
    q = rng.standard_normal(384, dtype=np.float32)
    with Index(0) as idx:
        idx.dense_load(e)
        for k in (65, 500, 3005):
            doc, sc, cnt = idx.dense_search(q, k)
            full = idx.dense_scores(q)
            want = ref_search.canonical_topk(full, k)  # the device's own scores: exact order check
            assert cnt[0] == len(want) and doc[0, :cnt[0]].tolist() == want.tolist()
            rows, sims = ref_search.similarity_search_with_embedding(q, e, None, k, None, canonical=True)
            assert_ranking_matches(rows, sims, doc[0, :cnt[0]], sc[0, :cnt[0]], 1e-4, None, f"dense large k={k}")


def test_mid_size_partitions_vs_oracle():
    """600k documents = partitions of 2,560 documents: the 1,024-thread form of K3 with partly filled lanes and a short
    last partition, multi-term queries with a source filter, against the oracle's CSR scorer (bit-exact scores, exact
    ranks).  (The fixtures above are <= 30k documents: partitions of 256, the 256-thread form.)"""
    import torch
    from oracle import ref_bm25, ref_search
    from anrag import synth
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n = 600_123
    post = synth.bm25_postings(n, 50_000, 31, dev, median_len=30.0)
    idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
    avgdl = post["total_len"] / n
    sid = (np.arange(n) % 7).astype(np.uint16)
    allow = np.array([1, 1, 0, 1, 0, 1, 1], dtype=np.uint8)
    torch.cuda.synchronize()
    post_doc, post_tf = post["post_doc"].cpu().numpy(), post["post_tf"].cpu().numpy()
    with Index(0) as idx:
        idx.bm25_load(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, source_id=sid)
        terms = [list(map(int, t)) for t in synth.bm25_queries(post, 5, 3)] + [[], [-1]]
        for t in terms:
            want = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, t)
            assert np.array_equal(idx.bm25_scores(t), want), t
            for k, flt in ((25, None), (64, allow), (3, allow)):
                doc, sc, cnt = idx.bm25_search(t, k, flt)
                order = ref_search.canonical_topk(want, k, None if flt is None else flt.astype(bool)[sid])
                assert cnt == k and doc.tolist() == order.tolist() and np.array_equal(sc, want[order]), (t, k)
