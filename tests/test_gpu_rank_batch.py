"""GPU parity: `anrag_rank_batch` -- full ranking (similarity_k = 12,000, src/retrieval_eval.py:142-143) for lists of
queries on the device -- against the oracle (numpy restatement of src/search_engine.py), against the per-query entry
points (same ids, same score bits) and, for the fusion, bit for bit against the reference's dict + stable sort."""
import os

import numpy as np
import pytest

from helpers import assert_ranking_matches

pytestmark = pytest.mark.gpu

K_FULL = 12000  # retrieval_eval.py's similarity_k = common_sections_n


def _sources(n, rng):
    kinds = ["CG", "NG", "QS", "TA", "PH"]
    p = [0.13, 0.72, 0.07, 0.05, 0.03]
    codes = [f"{kinds[i]}{int(j)}" for i, j in zip(rng.choice(5, 300, p=p), rng.integers(1, 250, 300))]
    return [codes[i % 300] for i in range(n)]


@pytest.fixture(scope="module")
def c1():
    """The shape of the reference's own corpus: 9,609 chunks x 384-d, BM25 over a permuted subset of the chunks plus
    sections no dense row has (the two row spaces share ids, not positions: bm25_search.py:67-68)."""
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.database_manager import intern_sources
    from anrag.index import Index

    rng = np.random.default_rng(42)
    n, d, vocab = 9609, 384, 3000
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e[4001] = e[17]  # equal rows: ties in every query
    sources = _sources(n, rng)
    p = 1.0 / np.arange(1, vocab + 1) ** 1.07
    p /= p.sum()
    nb = 9000
    docs = [[f"t{j}" for j in rng.choice(vocab, size=int(rng.integers(3, 40)), p=p)] for _ in range(nb)]
    # BM25 section j <-> document id: a permutation of 8,800 dense rows + 200 ids past the dense rows
    ids_b = np.concatenate([rng.permutation(n)[:8800], np.arange(n, n + 200)])
    ids_b = ids_b[rng.permutation(nb)].astype(np.int64)
    src_b = [sources[i] if i < n else "NG999" for i in ids_b]
    ref = BM25Okapi(docs, k1=1.7, b=0.83, epsilon=0.05)
    bi = Bm25Index(docs, k1=1.7, b=0.83, epsilon=0.05)
    sid_d, distinct_d = intern_sources(sources)
    sid_b, distinct_b = intern_sources(src_b)
    di, bx = Index(0), Index(0)
    di.dense_load(e, source_id=sid_d)
    bx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid_b)
    nq = 24
    q = e[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    toks = [[str(t) for t in rng.choice(docs[int(rng.integers(nb))], size=int(rng.integers(2, 10)))] for _ in range(nq)]
    toks[3] = []                       # no tokens: the reference skips BM25 for this query
    toks[5] = ["nowhere", "unseen"]    # tokens outside the vocabulary: every score 0, all sections ranked
    toks[7] = toks[7] + toks[7][:1]    # a duplicated token counts again
    yield dict(e=e, sources=sources, docs=docs, ids_b=ids_b, src_b=src_b, ref=ref, bi=bi, di=di, bx=bx, q=q, toks=toks,
               distinct_d=distinct_d, distinct_b=distinct_b, n=n, nb=nb, id_space=n + 200)
    di.close()
    bx.close()


def _allows(w, flt):
    from anrag.search_engine import bm25_allow, dense_allow

    if not flt:
        return None, None
    return dense_allow(w["distinct_d"], flt), bm25_allow(w["distinct_b"], flt)


@pytest.mark.parametrize("flt", [None, "CG,NG", "ZZ"])
def test_single_legs_match_oracle_and_per_query_path(c1, flt):
    from oracle import ref_search
    from anrag.index import rank_batch

    w = c1
    ad, ab = _allows(w, flt)
    nq = len(w["q"])
    ids, sc, cnt = rank_batch([dict(index=w["di"], weight=1.0, allow=ad, queries=w["q"])], nq, K_FULL, 40, K_FULL,
                              want_scores=True)
    for i in range(nq):
        rows, sims = ref_search.similarity_search_with_embedding(w["q"][i], w["e"], w["sources"], K_FULL, flt,
                                                                 canonical=True)
        assert cnt[i] == len(rows)
        if flt == "ZZ":
            assert cnt[i] == 0 and np.all(ids[i] == -1)
            continue
        full = np.dot(w["e"], w["q"][i])
        assert_ranking_matches(rows, sims, ids[i, :cnt[i]], sc[i, :cnt[i]], 1e-4, full, f"dense q{i} {flt}")
        # the per-query entry point (K1 score dump + library sort): same ids, same score bits
        doc, s1, c1_ = w["di"].dense_search(w["q"][i], K_FULL, ad)
        assert c1_[0] == cnt[i] and doc[0, :cnt[i]].tolist() == ids[i, :cnt[i]].tolist()
        assert np.array_equal(s1[0, :cnt[i]].astype(np.float64), sc[i, :cnt[i]])
        assert np.all(ids[i, cnt[i]:] == -1)
    term_lists = [w["bi"].term_ids(t) for t in w["toks"]]
    ids, sc, cnt = rank_batch([dict(index=w["bx"], weight=1.0, allow=ab, term_lists=term_lists)], nq, K_FULL, 40,
                              K_FULL, want_scores=True)
    for i in range(nq):
        if not w["toks"][i]:
            assert cnt[i] == 0
            continue
        scores = w["ref"].get_scores(w["toks"][i])
        want = ref_search.core_bm25_search(scores, w["src_b"], K_FULL, flt, canonical=True)
        assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (i, flt)
        assert np.array_equal(sc[i, :cnt[i]], scores[want])  # bit-exact fp64


@pytest.mark.parametrize("flt,k,top_n", [(None, K_FULL, K_FULL), ("CG,NG", K_FULL, K_FULL), ("CG,NG", 300, 300),
                                         (None, 2500, 15), ("NG", 65, 12000)])
def test_fused_matches_reference_fusion_bitwise(c1, flt, k, top_n):
    """Two dense models + BM25 (three lists, weights 5 : 2 : 1, wrrf_k 40): the fused ids and fp64 scores equal the
    reference's dict-update + stable-sort over the device's own three lists, and the lists equal the per-query path's."""
    from oracle import ref_search
    from anrag.index import Index, rank_batch

    w = c1
    ad, ab = _allows(w, flt)
    nq = len(w["q"])
    rng = np.random.default_rng(9)
    e2 = np.ascontiguousarray(w["e"][::-1] * np.float32(0.5) + rng.standard_normal(w["e"].shape, dtype=np.float32) * 0.02)
    q2 = np.ascontiguousarray(w["q"][:, ::-1])
    term_lists = [w["bi"].term_ids(t) for t in w["toks"]]
    perm = rng.permutation(w["n"]).astype(np.int64)  # second model's rows are stored in another order
    with Index(0) as d2:
        from anrag.database_manager import intern_sources

        sid2, distinct2 = intern_sources([w["sources"][int(i)] for i in perm])
        d2.dense_load(e2[perm], source_id=sid2)
        from anrag.search_engine import dense_allow

        ad2 = dense_allow(distinct2, flt) if flt else None
        legs = [dict(index=w["di"], weight=5.0, allow=ad, queries=w["q"]),
                dict(index=d2, weight=2.0, allow=ad2, queries=q2, doc_of_row=perm),
                dict(index=w["bx"], weight=1.0, allow=ab, term_lists=term_lists, doc_of_row=w["ids_b"])]
        ids, sc, cnt = rank_batch(legs, nq, k, 40, top_n, id_space=w["id_space"], want_scores=True)
        # an id space of 9,609 is fused inside one workgroup's LDS (rank_fuse_sort_kernel); the route larger id spaces
        # take (sums in HBM: fill, accumulate per leg, sort, emit) must give the same bits
        os.environ["ANRAG_RANK_FUSE_IN_HBM"] = "1"
        try:
            ids_h, sc_h, cnt_h = rank_batch(legs, nq, k, 40, top_n, id_space=w["id_space"], want_scores=True)
        finally:
            del os.environ["ANRAG_RANK_FUSE_IN_HBM"]
        assert np.array_equal(ids, ids_h) and np.array_equal(cnt, cnt_h) and np.array_equal(sc.view(np.int64), sc_h.view(np.int64))
        for i in range(nq):
            l1, _, c1_ = w["di"].dense_search(w["q"][i], k, ad)
            l2, _, c2_ = d2.dense_search(q2[i], k, ad2)
            lists = [(l1[0, :c1_[0]].tolist(), "a"), (perm[l2[0, :c2_[0]]].tolist(), "b")]
            if w["toks"][i]:
                l3, _, c3_ = w["bx"].bm25_search(term_lists[i], k, ab)
                lists.append((w["ids_b"][l3[:c3_]].tolist(), "c"))
            lists = [l for l in lists if l[0]]
            fused = ref_search.weighted_reciprocal_rank_fusion(lists, {"a": 5.0, "b": 2.0, "c": 1.0}, 40)[:top_n]
            assert cnt[i] == len(fused), (i, cnt[i], len(fused))
            assert ids[i, :cnt[i]].tolist() == [d for d, _ in fused], (i, flt, k)
            assert sc[i, :cnt[i]].tolist() == [s for _, s in fused]
            assert np.all(ids[i, cnt[i]:] == -1)


def test_expected_rank_equals_position_in_the_list(c1):
    """retrieval_eval.py:75-82 looks the expected chunk id up in the returned list; `expect` asks the device for that
    position directly (with or without the lists themselves)."""
    from anrag.index import rank_batch

    w = c1
    nq = len(w["q"])
    ad, ab = _allows(w, "CG,NG")
    term_lists = [w["bi"].term_ids(t) for t in w["toks"]]
    legs = [dict(index=w["di"], weight=5.0, allow=ad, queries=w["q"]),
            dict(index=w["bx"], weight=1.0, allow=ab, term_lists=term_lists, doc_of_row=w["ids_b"])]
    rng = np.random.default_rng(77)
    for use, k, top_n in ((legs, K_FULL, K_FULL), (legs[:1], K_FULL, K_FULL), (legs, 300, 40), (legs[1:], K_FULL, 500)):
        ids, _, cnt = rank_batch(use, nq, k, 40, top_n, id_space=w["id_space"])
        expect = np.array([ids[i, rng.integers(0, cnt[i])] if cnt[i] and i % 5 else w["id_space"] - 1 - (i % 3)
                           for i in range(nq)], dtype=np.int64)
        ids2, _, cnt2, ranks = rank_batch(use, nq, k, 40, top_n, id_space=w["id_space"], expect=expect)
        none, _, cnt3, ranks3 = rank_batch(use, nq, k, 40, top_n, id_space=w["id_space"], expect=expect, want_ids=False)
        assert none is None and np.array_equal(ids, ids2) and np.array_equal(cnt, cnt2) and np.array_equal(cnt, cnt3)
        assert np.array_equal(ranks, ranks3)
        for i in range(nq):
            pos = np.nonzero(ids[i, :cnt[i]] == expect[i])[0]
            assert ranks[i] == (pos[0] + 1 if len(pos) else -1), (i, k, top_n)


@pytest.mark.parametrize("d", [64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1536, 2048, 3072, 8])
def test_score_tiles_equal_the_scan_scores_at_every_kernel_shape(d):
    """K1T (dense_tile.hip: a batch of rows in registers, all the queries of a launch over it) against K1's own score
    pass for one query (`dense_scores`), bit for bit, at every dimension with a shaped kernel (3,072: no tile kernel, a
    pass per query; 8: the generic kernel), corpora from one row to a few per wave to many, 19 queries (one launch takes
    16 at most, 6 at 2,048-d), with and without a source filter."""
    from oracle import ref_search
    from anrag.index import Index, rank_batch

    rng = np.random.default_rng(d)
    nq = 19
    for n in (1, 3, 63, 1000, 5003):
        e = rng.standard_normal((n, d), dtype=np.float32)
        if n > 10:
            e[n // 2] = e[1]
            e[7] = np.nan if d == 384 else e[7]  # a NaN row ranks first (numpy's order): carried as +inf
        q = rng.standard_normal((nq, d), dtype=np.float32)
        sid = (np.arange(n) % 5).astype(np.uint16)
        allow = np.array([1, 0, 1, 1, 0], np.uint8)
        with Index(0) as di:
            di.dense_load(e, source_id=sid)
            for al in (None, allow):
                ids, sc, cnt = rank_batch([dict(index=di, weight=1.0, allow=al, queries=q)], nq, n, 40, n, want_scores=True)
                ok = None if al is None else al.astype(bool)[sid]
                for i in range(nq):
                    full = di.dense_scores(q[i])
                    want = ref_search.canonical_topk(full, n, ok)
                    assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (d, n, i)
                    assert np.array_equal(sc[i, :cnt[i]], full[want].astype(np.float64)), (d, n, i)


@pytest.mark.parametrize("d", [512, 768, 256])
def test_matrix_core_score_tiles_equal_the_scan_scores(d):
    """K1T's matrix-core form (dense_tile_mfma.hip: one v_mfma_f32_16x16x4_f32 accumulation per lane slice = the scan's
    per-lane FMA chain, then the scan's lane tree as vector adds) takes corpora of at least 32 rows per CU at 512 / 768
    dimensions (256-d rows are scanned in another shape and stay on the VALU form: checked here too): every score bit for bit K1's (`dense_scores`, one query at a time) -- a row count that is no multiple
    of the 32-row blocks, 37 queries (launch groups of 32 and tiles of 16 end inside the list), equal rows, a NaN row
    (ranks first), a source filter; and the VALU form (ANRAG_TILE_MFMA=0 is read once per process: here through the
    small-corpus route, which always takes it) agrees by the other test."""
    from oracle import ref_search
    from anrag.index import Index, rank_batch

    rng = np.random.default_rng(1000 + d)
    n, nq = 8192 + 37, 37
    e = rng.standard_normal((n, d), dtype=np.float32)
    e *= np.exp(rng.uniform(-6, 6, size=(n, 1))).astype(np.float32)  # magnitudes over five decades
    e[n // 2] = e[1]
    e[n - 1] = e[2]
    e[77] = np.nan
    q = rng.standard_normal((nq, d), dtype=np.float32)
    q[5] *= np.float32(1e-12)
    sid = (np.arange(n) % 5).astype(np.uint16)
    allow = np.array([1, 0, 1, 1, 0], np.uint8)
    with Index(0) as di:
        di.dense_load(e, source_id=sid)
        for al in (None, allow):
            ids, sc, cnt = rank_batch([dict(index=di, weight=1.0, allow=al, queries=q)], nq, n, 40, n, want_scores=True)
            ok = None if al is None else al.astype(bool)[sid]
            for i in range(nq):
                full = di.dense_scores(q[i])
                want = ref_search.canonical_topk(full, n, ok)
                assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (d, i)
                assert np.array_equal(sc[i, :cnt[i]], full[want].astype(np.float64)), (d, i)


@pytest.mark.parametrize("nb", [12287, 12288, 12289, 64, 1])
def test_radix_sort_capacity_edges_bm25(nb):
    """fp64 segments at the LDS radix sort's capacity (12,288 scores), one past it (the network takes over) and tiny ones;
    heavy ties (most documents score 0: their order is the row order), k below / at / above the number of documents."""
    from oracle import ref_search
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index, rank_batch

    rng = np.random.default_rng(nb)
    docs = [[f"w{int(j)}" for j in rng.integers(0, 400, size=int(rng.integers(1, 5)))] for _ in range(nb)]
    bi = Bm25Index(docs, k1=1.7, b=0.83, epsilon=0.05)
    ref = BM25Okapi(docs, k1=1.7, b=0.83, epsilon=0.05)
    toks = [["w3", "w7", "w11"], ["absent"], ["w1"], [], ["w399", "w399", "w2"]]
    tl = [bi.term_ids(t) for t in toks]
    with Index(0) as bx:
        bx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        for k in (nb, max(1, nb // 3), nb + 5):
            ids, sc, cnt = rank_batch([dict(index=bx, weight=1.0, term_lists=tl)], len(toks), k, 40, k, want_scores=True)
            for i, t in enumerate(toks):
                if not t:
                    assert cnt[i] == 0
                    continue
                scores = ref.get_scores(t)
                want = ref_search.canonical_topk(scores, k)
                assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (nb, k, t)
                assert np.array_equal(sc[i, :cnt[i]], scores[want])


def test_select_path_heavy_ties_and_edges():
    """Segments longer than a workgroup's LDS (the radix select runs first): all-equal scores (every BM25 score 0: the
    k lowest rows win), duplicated rows, k = 1, a segment one element past the cap, a filter that keeps fewer than k."""
    from oracle import ref_search
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index, rank_batch, rank_caps

    cap32, cap64 = rank_caps()
    rng = np.random.default_rng(3)
    n, d = 50000, 64
    e = rng.standard_normal((n, d), dtype=np.float32)
    e[30000:30100] = e[5]            # a hundred equal rows
    e[40000:] = 0.0                  # ten thousand zero scores
    q = rng.standard_normal((5, d), dtype=np.float32)
    sid = (np.arange(n) % 11).astype(np.uint16)
    allow = np.zeros(11, np.uint8)
    allow[[2, 7]] = 1
    docs = [[f"w{int(j)}" for j in rng.integers(0, 50, size=int(rng.integers(1, 6)))] for _ in range(20000)]
    bi = Bm25Index(docs, k1=1.7, b=0.83, epsilon=0.05)
    from oracle.ref_bm25 import BM25Okapi

    ref = BM25Okapi(docs, k1=1.7, b=0.83, epsilon=0.05)
    with Index(0) as di, Index(0) as bx, Index(0) as edge:
        di.dense_load(e, source_id=sid)
        bx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        for k, al in ((K_FULL, None), (1, None), (cap32, None), (K_FULL, allow), (300, allow)):
            ids, sc, cnt = rank_batch([dict(index=di, weight=1.0, allow=al, queries=q)], 5, k, 40, k, want_scores=True)
            for i in range(5):
                full = di.dense_scores(q[i])  # the device's own scores: the order must be exact
                ok = None if al is None else al.astype(bool)[sid]
                want = ref_search.canonical_topk(full, k, ok)
                assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (k, i)
                assert np.array_equal(sc[i, :cnt[i]], full[want].astype(np.float64))
        toks = [["w3", "w7"], ["absent"], ["w1"], ["w49", "w49", "w2"]]
        tl = [bi.term_ids(t) for t in toks]
        for k in (K_FULL, 7, cap64):
            ids, sc, cnt = rank_batch([dict(index=bx, weight=1.0, term_lists=tl)], 4, k, 40, k, want_scores=True)
            for i, t in enumerate(toks):
                scores = ref.get_scores(t)
                want = ref_search.canonical_topk(scores, k)
                assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), (k, t)
                assert np.array_equal(sc[i, :cnt[i]], scores[want])
        # one element past the cap, and exactly the cap
        for m in (cap32 + 1, cap32, 65):
            em = np.ascontiguousarray(e[:m])
            edge.dense_load(em)
            ids, sc, cnt = rank_batch([dict(index=edge, weight=1.0, queries=q[:2])], 2, cap32, 40, cap32, want_scores=True)
            for i in range(2):
                want = ref_search.canonical_topk(edge.dense_scores(q[i]), cap32)
                assert cnt[i] == len(want) and ids[i, :cnt[i]].tolist() == want.tolist(), m
        # outside the envelope: refused, not truncated
        from anrag._native import AnragError

        with pytest.raises(AnragError):
            rank_batch([dict(index=di, weight=1.0, queries=q)], 5, cap32 + 1, 40, 10)
        with pytest.raises(AnragError):  # a leg naming a document twice cannot be fused in the batch
            rank_batch([dict(index=di, weight=1.0, queries=q, doc_of_row=np.zeros(n, np.int64)),
                        dict(index=bx, weight=1.0, term_lists=tl + [tl[0]])], 5, 100, 40, 10, id_space=n)


def test_full_size_1m_matches_per_query_path():
    """1M x 768 (BASELINE C3's corpus), k = 12,000: the radix select over a 4 MB score segment per query.  Dense and
    BM25 lists equal the per-query entry points' (score dump + library sort of all N), the fusion equals the
    reference's over those lists."""
    import torch
    from oracle import ref_search
    from anrag import synth
    from anrag.index import Index, rank_batch

    dev = torch.device("cuda", 0)
    n, d, nq = 1_000_000, 768, 12
    e = synth.dense_corpus(n, d, 1234, dev)
    q, _ = synth.dense_queries(e, nq, 4321)
    post = synth.bm25_postings(n, 200_000, 777, dev)
    idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
    avgdl = post["total_len"] / n
    terms = [list(map(int, t)) for t in synth.bm25_queries(post, nq, 5)]
    terms[2] = []
    torch.cuda.synchronize()
    qh = q.cpu().numpy()
    with Index(0) as idx:
        idx.dense_load((e.data_ptr(), n, d))
        idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                      (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], avgdl, 1.7, 0.83)
        del e
        legs = [dict(index=idx, weight=5.0, queries=qh), dict(index=idx, weight=1.0, term_lists=terms)]
        ids_d, sc_d, cnt_d = rank_batch(legs[:1], nq, K_FULL, 40, K_FULL, want_scores=True)
        ids_b, sc_b, cnt_b = rank_batch(legs[1:], nq, K_FULL, 40, K_FULL, want_scores=True)
        ids_f, sc_f, cnt_f = rank_batch(legs, nq, K_FULL, 40, K_FULL, id_space=n, want_scores=True)
        for i in range(nq):
            doc, s1, c1 = idx.dense_search(qh[i], K_FULL)
            assert cnt_d[i] == K_FULL and ids_d[i].tolist() == doc[0].tolist()
            assert np.array_equal(sc_d[i], s1[0].astype(np.float64))
            lists = [(doc[0].tolist(), "d")]
            if terms[i]:
                bdoc, bs, bc = idx.bm25_search(terms[i], K_FULL)
                assert cnt_b[i] == K_FULL and ids_b[i].tolist() == bdoc.tolist() and np.array_equal(sc_b[i], bs)
                lists.append((bdoc.tolist(), "b"))
            else:
                assert cnt_b[i] == 0
            if len(lists) == 2:
                fused = ref_search.weighted_reciprocal_rank_fusion(lists, {"d": 5.0, "b": 1.0}, 40)[:K_FULL]
                assert ids_f[i, :cnt_f[i]].tolist() == [x for x, _ in fused]
                assert sc_f[i, :cnt_f[i]].tolist() == [s for _, s in fused]
            else:
                assert ids_f[i].tolist() == doc[0].tolist()
