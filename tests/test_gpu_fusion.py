"""GPU parity: K5 weighted RRF (bit-exact fp64, tie order = first insertion) and the fused hybrid query."""
import numpy as np
import pytest

from helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def idx():
    from anrag.index import Index

    i = Index(0)
    yield i
    i.close()


def test_wrrf_golden_bitwise(idx):
    for c in load_golden("ref_wrrf.json"):
        ids = {}
        lists, weights = [], []
        for l, name in c["lists"]:
            lists.append([ids.setdefault(x, len(ids) + 100) for x in l])
            weights.append(c["weights"].get(name, 1.0))
        back = {v: k for k, v in ids.items()}
        for top_n in (len(ids), 15, 1):
            out_id, out_score = idx.wrrf(lists, weights, c["k"], top_n)
            want = c["fused"][:top_n]
            assert [back[i] for i in out_id.tolist()] == [w[0] for w in want]
            assert out_score.tolist() == [w[1] for w in want]


def test_wrrf_long_lists(idx):
    """Lists beyond one workgroup: the all-pairs grid form (1,024 < M <= 4,096 entries) and the sort-based form
    (retrieval_eval's full-ranking mode: two 12,000-id lists), with repeated ids inside a list, ids shared by all
    lists, a zero weight and many equal scores -- bit-for-bit the dict + stable-sort result."""
    from oracle import ref_search

    rng = np.random.default_rng(1)
    cases = [
        ([rng.permutation(9000)[:n].tolist() for n in (1500, 900, 600)], {"a": 5.0, "b": 1.0, "c": 2.0}),     # grid
        ([rng.permutation(9000)[:n].tolist() for n in (3000, 2500, 1200)], {"a": 5.0, "b": 1.0, "c": 2.0}),   # sorted
        ([rng.permutation(12000).tolist(), rng.permutation(12000).tolist()], {"a": 5.0, "b": 1.0}),            # eval shape
        ([rng.integers(0, 500, 4000).tolist(), rng.integers(0, 500, 3000).tolist(), list(range(2000))],
         {"a": 1.0, "b": 1.0, "c": 0.0}),                                                                      # repeats, ties
    ]
    for lists, weights in cases:
        names = list(weights)[: len(lists)]
        ref = ref_search.weighted_reciprocal_rank_fusion(list(zip(lists, names)), weights, 40)
        for top_n in (len(ref), 30000, 15):
            out_id, out_score = idx.wrrf(lists, [weights[n] for n in names], 40, top_n)
            want = ref[:top_n]
            assert out_id.tolist() == [i for i, _ in want]
            assert out_score.tolist() == [s for _, s in want]


def test_hybrid_matches_oracle(idx):
    from oracle import ref_search
    from oracle.make_golden import synth_chunks, synth_dense, synth_query
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    chunks = synth_chunks(600, 77)
    n, d = len(chunks), 128
    e = synth_dense(n, d, 78)
    kept = [i for i, c in enumerate(chunks) if c["tokens"]]
    corpus = [chunks[i]["tokens"] for i in kept]
    ref = BM25Okapi(corpus, k1=1.7, b=0.83, epsilon=0.05)
    bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
    sources = [c["source"] for c in chunks]
    table = {}
    sid = np.array([table.setdefault(s, len(table)) for s in sources], dtype=np.uint16)
    distinct = list(table)
    rng = np.random.default_rng(5)
    with Index(0) as h:
        h.dense_load(e, source_id=sid)  # doc id = dense row
        h.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b,
                    source_id=sid[kept], doc_id=np.array(kept, dtype=np.int64))
        for trial in range(20):
            target = int(rng.integers(n))
            q = synth_query(e, 900 + trial, target)
            toks = [str(t) for t in rng.choice(chunks[target]["tokens"] or ["asthma"], size=int(rng.integers(1, 8)))]
            for sim_k, top_n, wrrf_k, flt in ((25, 15, 40, None), (25, 10, 40, "CG,NG"), (64, 15, 60, "NG"), (10, 10, 50, None)):
                ad = None if flt is None else ref_search.dense_filter_mask(distinct, flt).astype(np.uint8)
                ab = None if flt is None else ref_search.bm25_filter_mask(distinct, flt).astype(np.uint8)
                out_id, out_score = h.hybrid_search(q, bi.term_ids(toks), sim_k, 5.0, 1.0, wrrf_k, top_n, ad, ab)
                # oracle, built from the DEVICE's dense ranking (dense parity has its own test) and exact BM25
                ddoc, _, dcnt = h.dense_search(q, sim_k, ad)
                dense_list = ddoc[0, :dcnt[0]].tolist()
                scores = ref.get_scores(toks)
                rows = ref_search.core_bm25_search(scores, [sources[i] for i in kept], sim_k, flt, canonical=True)
                bm_list = [kept[r] for r in rows]
                fused = ref_search.weighted_reciprocal_rank_fusion(
                    [(dense_list, "d"), (bm_list, "BM25")], {"d": 5.0, "BM25": 1.0}, wrrf_k)[:top_n]
                assert out_id.tolist() == [i for i, _ in fused], (trial, sim_k, flt)
                assert out_score.tolist() == [s for _, s in fused]
            # dense-only and BM25-only degenerate to the single list (query_rag_retrieval.py:363-366)
            out_id, _ = h.hybrid_search(q, [], 25, 5.0, 1.0, 40, 15)
            ddoc, _, dcnt = h.dense_search(q, 25)
            assert out_id.tolist() == ddoc[0, :15].tolist()
            out_id, _ = h.hybrid_search(q, bi.term_ids(toks), 25, 0.0, 1.0, 40, 15)
            bdoc, _, bcnt = h.bm25_search(bi.term_ids(toks), 25)
            assert out_id.tolist() == bdoc[:15].tolist()


def test_hybrid_batch_equals_single_calls():
    """`anrag_hybrid_search_batch` (one call, device pipeline, > kPipeSlots queries in flight) row by row equals
    `anrag_hybrid_search`, with and without a source filter, including a query without terms (dense list only)."""
    from oracle import ref_search
    from oracle.make_golden import synth_chunks, synth_dense, synth_query
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    chunks = synth_chunks(900, 91)
    n, d = len(chunks), 256
    e = synth_dense(n, d, 92)
    kept = [i for i, c in enumerate(chunks) if c["tokens"]]
    bi = Bm25Index([chunks[i]["tokens"] for i in kept], k1=1.7, b=0.83, epsilon=0.05)
    table = {}
    sid = np.array([table.setdefault(c["source"], len(table)) for c in chunks], dtype=np.uint16)
    distinct = list(table)
    rng = np.random.default_rng(6)
    nq = 37
    targets = rng.integers(n, size=nq)
    qs = np.stack([synth_query(e, 500 + i, int(t)) for i, t in enumerate(targets)])
    terms = [bi.term_ids([str(t) for t in rng.choice(chunks[int(t)]["tokens"] or ["asthma"], size=int(rng.integers(1, 9)))])
             for t in targets]
    terms[5] = np.zeros(0, np.int32)          # no tokens: the dense list alone
    terms[11] = np.array([-1, -1], np.int32)  # tokens outside the vocabulary: every BM25 score is zero
    with Index(0) as h:
        h.dense_load(e, source_id=sid)
        h.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b,
                    source_id=sid[kept], doc_id=np.array(kept, dtype=np.int64))
        for sim_k, top_n, flt in ((25, 10, None), (25, 15, "CG,NG"), (64, 12, "NG")):
            ad = None if flt is None else ref_search.dense_filter_mask(distinct, flt).astype(np.uint8)
            ab = None if flt is None else ref_search.bm25_filter_mask(distinct, flt).astype(np.uint8)
            ids, scores, counts = h.hybrid_search_batch(qs, terms, sim_k, 5.0, 1.0, 40.0, top_n, ad, ab)
            assert ids.shape == (nq, top_n)
            for i in range(nq):
                want_id, want_score = h.hybrid_search(qs[i], terms[i], sim_k, 5.0, 1.0, 40.0, top_n, ad, ab)
                c = int(counts[i])
                assert c == len(want_id)
                assert ids[i, :c].tolist() == want_id.tolist(), (i, flt)
                assert scores[i, :c].tolist() == want_score.tolist()
                assert (ids[i, c:] == -1).all()
        # an empty batch is a no-op
        ids, _, counts = h.hybrid_search_batch(np.zeros((0, d), np.float32), [], 25, 5.0, 1.0, 40.0, 10)
        assert ids.shape == (0, 10) and counts.shape == (0,)


def test_hybrid_batch_ranking_route_equals_single_calls_and_the_pipeline():
    """Row numbers as document ids on both sides and 16 queries or more: `anrag_hybrid_search_batch` takes the ranking route
    (score tiles + per-list sorts + fusion in LDS, rank_batch.hip).  Row by row it equals `anrag_hybrid_search` -- ids, fused
    fp64 scores, counts, padding -- with and without a source filter, including a query without terms and one whose terms
    are outside the vocabulary; and it equals the device pipeline's list form (ANRAG_BATCH_PIPELINE=1) bit for bit."""
    import os

    from oracle import ref_search
    from oracle.make_golden import synth_chunks, synth_dense, synth_query
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    chunks = [c for c in synth_chunks(1300, 191) if c["tokens"]][:1100]
    n, d = len(chunks), 384
    e = synth_dense(n, d, 192)
    e[700] = e[3]  # equal rows: ties inside the dense list
    bi = Bm25Index([c["tokens"] for c in chunks], k1=1.7, b=0.83, epsilon=0.05)
    table = {}
    sid = np.array([table.setdefault(c["source"], len(table)) for c in chunks], dtype=np.uint16)
    distinct = list(table)
    rng = np.random.default_rng(16)
    nq = 41
    targets = rng.integers(n, size=nq)
    qs = np.stack([synth_query(e, 900 + i, int(t)) for i, t in enumerate(targets)])
    terms = [bi.term_ids([str(t) for t in rng.choice(chunks[int(t)]["tokens"], size=int(rng.integers(1, 9)))]) for t in targets]
    terms[5] = np.zeros(0, np.int32)
    terms[11] = np.array([-1, -1], np.int32)
    with Index(0) as h:
        h.dense_load(e, source_id=sid)
        h.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid)
        for sim_k, top_n, flt in ((25, 10, None), (25, 15, "CG,NG"), (64, 128, "NG"), (7, 3, None)):
            ad = None if flt is None else ref_search.dense_filter_mask(distinct, flt).astype(np.uint8)
            ab = None if flt is None else ref_search.bm25_filter_mask(distinct, flt).astype(np.uint8)
            ids, scores, counts = h.hybrid_search_batch(qs, terms, sim_k, 5.0, 1.0, 40.0, top_n, ad, ab)
            os.environ["ANRAG_BATCH_PIPELINE"] = "1"
            try:
                ids_p, scores_p, counts_p = h.hybrid_search_batch(qs, terms, sim_k, 5.0, 1.0, 40.0, top_n, ad, ab)
            finally:
                del os.environ["ANRAG_BATCH_PIPELINE"]
            assert np.array_equal(ids, ids_p) and np.array_equal(counts, counts_p), (sim_k, top_n, flt)
            assert np.array_equal(scores.view(np.int64), scores_p.view(np.int64)), (sim_k, top_n, flt)
            for i in range(nq):
                want_id, want_score = h.hybrid_search(qs[i], terms[i], sim_k, 5.0, 1.0, 40.0, top_n, ad, ab)
                c = int(counts[i])
                assert c == len(want_id) and ids[i, :c].tolist() == want_id.tolist(), (i, flt)
                assert scores[i, :c].tolist() == want_score.tolist()
                assert (ids[i, c:] == -1).all()
