"""GPU, BASELINE.json's full sizes (1M x 768 dense, 1M-document postings): size-independent properties, since the
CPU oracle cannot afford many full-size queries (bench.py's cpu_baseline leg re-checks 3 of them every run).

  * two independent kernels agree: K1 (batch=1 scan) == K2 (MFMA batched) == torch.topk on the same device data
  * planted neighbours rank first; results are sorted (score desc, row asc); the same query twice = same bits
  * sharding: top-k of the whole == merge of the two halves' top-k (doc ids = global rows)
  * BM25: scores of a single-term query == idf * impact on exactly that term's postings, zero elsewhere; a
    duplicated term doubles them exactly; top-k by the fused kernel == top-k of the full score array
  * hybrid: fused ids/scores == oracle WRRF of the device's own per-modality lists
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D = 1_000_000, 768


@pytest.fixture(scope="module")
def world():
    import torch
    from anrag import synth
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    E = synth.dense_corpus(N, D, 1234, dev)
    Q, planted = synth.dense_queries(E, 256, 4321)
    idx = Index(0)
    torch.cuda.synchronize()  # device-pointer operands must be complete before the library copies them
    idx.dense_load((E.data_ptr(), N, D))
    post = synth.bm25_postings(N, 200_000, 777, dev)
    idf = synth.bm25_idf(post["df"].cpu().numpy(), N)
    avgdl = post["total_len"] / N
    torch.cuda.synchronize()
    idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                  (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], avgdl, 1.7, 0.83)
    yield dict(torch=torch, E=E, Q=Q, planted=planted, idx=idx, post=post, idf=idf, avgdl=avgdl)
    idx.close()


def test_dense_kernels_agree_at_full_size(world):
    torch, E, Q, idx = world["torch"], world["E"], world["Q"], world["idx"]
    qh = Q.cpu().numpy()
    k = 10
    d1, s1, c1 = idx.dense_search(qh[:8], k)          # 8 < 16 queries: K1, one scan each
    d2, s2, c2 = idx.dense_search(qh, k)              # BASELINE configs[3]: 256 queries, ONE full-width MFMA pass
    ref = (Q @ E.T).topk(k, dim=1)
    ri, rv = ref.indices.cpu().numpy(), ref.values.cpu().numpy()
    assert np.all(c1 == k) and np.all(c2 == k)
    assert np.array_equal(d1, ri[:8]) and np.array_equal(d2, ri)
    assert np.max(np.abs(s1 - rv[:8])) <= 1e-4 and np.max(np.abs(s2 - rv)) <= 1e-4
    assert np.array_equal(d2[:, 0], world["planted"].cpu().numpy())          # planted neighbour first
    assert np.all(np.diff(s1, axis=1) <= 0) and np.all(np.diff(s2, axis=1) <= 0)
    again = idx.dense_search(qh[:8], k)
    assert np.array_equal(again[0], d1) and np.array_equal(again[1], s1)     # bit-identical on repeat
    # the opt-in split-precision arithmetic (bf16 x 3 products): the same 256 x 1M pass inside the 1e-4 bar
    from helpers import assert_ranking_matches
    idx.set_batched_precision("bf16x3")
    try:
        d3, s3, c3 = idx.dense_search(qh, k)
    finally:
        idx.set_batched_precision("f32")
    assert np.all(c3 == k)
    for i in range(len(qh)):
        assert_ranking_matches(ri[i], rv[i], d3[i], s3[i], 1e-4, None, f"bf16x3 query {i}")
    assert np.mean(d3 == ri) > 0.999  # and in practice the same ids: the split products are ~1e-6 from f32


def test_c2_100k_x768_top10_vs_oracle(world):
    """BASELINE configs[1] at its exact size: 100,000 x 768, dense-only top-10 at batch = 1, against the CPU oracle
    (one np.dot over 0.3 GB per query) -- ids equal, scores within 1e-4."""
    from oracle import ref_search
    from helpers import assert_ranking_matches
    from anrag.index import Index

    E, Q = world["E"], world["Q"]
    n = 100_000
    e_host = E[:n].cpu().numpy()
    with Index(0) as sub:
        sub.dense_load((E.data_ptr(), n, D))  # the first 100k rows of the same corpus
        for qi in (0, 1, 2, 3, 100, 255):
            q = Q[qi].cpu().numpy()
            want = ref_search.dense_scores(q, e_host)
            order = ref_search.canonical_topk(want, 10)
            doc, score, cnt = sub.dense_search(q, 10)
            assert cnt[0] == 10
            assert_ranking_matches(order, want[order], doc[0], score[0], 1e-4, want, f"C2 query {qi}")


def test_c5_rank_shape_1m_x1024_hybrid():
    """BASELINE configs[4] as ONE rank sees it: a 1M x 1024 shard (4.1 GB) with its postings, hybrid.  Size-independent
    properties: K1 == torch.topk on the device's own data, the fused answer == oracle WRRF of the device's own
    per-modality lists, planted neighbours first, the candidate payload == the two lists."""
    import torch
    from oracle import ref_search
    from anrag import _native as nat, synth
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, d, k = 1_000_000, 1024, 25
    E = synth.dense_corpus(n, d, 99, dev, row_lo=3_000_000)        # rows [3M, 4M) of an 8M-row corpus
    Q, planted = synth.dense_queries(E, 8, 17)
    post = synth.bm25_postings(n, 200_000, 55, dev, doc_lo=3_000_000)
    idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
    terms = synth.bm25_queries(post, 8, 19)
    torch.cuda.synchronize()
    lib = nat.load_library()
    with Index(0) as idx:
        idx.dense_load((E.data_ptr(), n, d), doc_id_base=3_000_000)
        idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                      (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"],
                      post["total_len"] / n, 1.7, 0.83, doc_id_base=3_000_000)
        ref = (Q @ E.T).topk(k, dim=1)
        out = torch.zeros((2 * k, 2), dtype=torch.int64, device=dev)
        T = torch.full((16,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # torch fills / copies on ITS stream; the library reads and writes on its own streams
        for qi in range(8):
            q = Q[qi].cpu().numpy()
            doc, score, cnt = idx.dense_search(q, k)
            assert cnt[0] == k and (doc[0] - 3_000_000).tolist() == ref.indices[qi].cpu().tolist()
            assert np.max(np.abs(score[0] - ref.values[qi].cpu().numpy())) <= 1e-4
            assert doc[0, 0] - 3_000_000 == int(planted[qi])
            bl = idx.bm25_search(terms[qi], k)[0]
            ids, scores = idx.hybrid_search(q, terms[qi], k, 5.0, 1.0, 40, 10)
            want = ref_search.weighted_reciprocal_rank_fusion([(doc[0].tolist(), "d"), (bl.tolist(), "b")],
                                                              {"d": 5.0, "b": 1.0}, 40)[:10]
            assert ids.tolist() == [i for i, _ in want] and scores.tolist() == [s for _, s in want]
            T[: len(terms[qi])] = torch.from_numpy(np.asarray(terms[qi], np.int32)).to(dev)
            torch.cuda.synchronize()
            nat.check(lib.anrag_hybrid_candidates_device(idx.handle, Q[qi].data_ptr(), T.data_ptr(), len(terms[qi]), k,
                                                         None, None, out.data_ptr()))
            idx.sync()
            rec = out.cpu().numpy()
            assert rec[:k, 1].tolist() == doc[0].tolist() and rec[k:, 1].tolist() == bl.tolist()


def test_sharded_merge_equals_whole(world):
    torch, E, Q, idx = world["torch"], world["E"], world["Q"], world["idx"]
    from anrag import _native as nat
    from anrag.index import Index

    k, half = 25, N // 2
    lib = nat.load_library()
    with Index(0) as a, Index(0) as b, Index(0) as util:
        a.dense_load((E.data_ptr(), half, D), doc_id_base=0)
        b.dense_load((E.data_ptr() + half * D * 4, N - half, D), doc_id_base=half)
        lists = torch.zeros((2, k, 2), dtype=torch.int64, device=E.device)
        out = torch.zeros((k, 2), dtype=torch.int64, device=E.device)
        torch.cuda.synchronize()  # the zero fills run on torch's stream: over before the library writes the buffers
        for qi in range(4):
            a.dense_search_device(Q[qi].data_ptr(), 1, k, 0, lists[0].data_ptr())
            b.dense_search_device(Q[qi].data_ptr(), 1, k, 0, lists[1].data_ptr())
            a.sync(); b.sync()
            nat.check(lib.anrag_merge_candidates_device(util.handle, lists.data_ptr(), 2, k, k, out.data_ptr()))
            util.sync()
            whole_doc, whole_score, _ = idx.dense_search(Q[qi].cpu().numpy(), k)
            got = out.cpu().numpy()
            assert got[:, 1].tolist() == whole_doc[0].tolist()
            assert np.array_equal(got[:, 0].copy().view(np.float64).astype(np.float32), whole_score[0])


def test_bm25_properties_at_full_size(world):
    idx, post, idf = world["idx"], world["post"], world["idf"]
    indptr = post["indptr"]
    df = np.diff(indptr)
    for t in (int(np.argmax(df)), int(np.nonzero((df > 3000) & (df < 6000))[0][0]), int(np.nonzero((df > 0) & (df < 20))[0][0])):
        lo, hi = indptr[t], indptr[t + 1]
        docs = post["post_doc"][lo:hi].cpu().numpy()
        tf = post["post_tf"][lo:hi].cpu().numpy().astype(np.int64)
        dl = post["doc_len"][docs].astype(np.int64)
        want = idf[t] * (tf * (1.7 + 1) / (tf + 1.7 * (1 - 0.83 + 0.83 * dl / world["avgdl"])))
        s = idx.bm25_scores([t])
        assert np.array_equal(s[docs], want)                      # bit-exact on the term's postings
        assert np.count_nonzero(s) == np.count_nonzero(want)      # and exactly zero everywhere else
        s2 = idx.bm25_scores([t, t])
        assert np.array_equal(s2, s + s)                          # a duplicated query token counts twice
        doc, sc, cnt = idx.bm25_search([t], 25)
        order = np.lexsort((np.arange(N), -s))[:25]
        assert cnt == 25 and doc.tolist() == order.tolist() and np.array_equal(sc, s[order])


def test_bm25_multi_term_queries_vs_oracle_at_full_size(world):
    """9-term queries (bench.py's shape, one with a duplicated term, one with an unknown token) at 1M documents
    against the oracle's CSR restatement of BM25Okapi.get_scores: every one of the 1M fp64 scores bit for bit, and
    the fused top-25 == (score desc, row asc) over that array.  This is the 1,024-thread form of K3 (partitions of
    4,096 documents); the small corpora of test_gpu_bm25.py run its 256-thread form."""
    from oracle import ref_bm25, ref_search
    from anrag import synth

    idx, post, idf = world["idx"], world["post"], world["idf"]
    post_doc, post_tf = post["post_doc"].cpu().numpy(), post["post_tf"].cpu().numpy()
    terms = [list(map(int, t)) for t in synth.bm25_queries(post, 6, 17)]
    terms[1][-1] = terms[1][0]       # a duplicated query token counts again
    terms[2].insert(3, -1)           # a token outside the vocabulary contributes nothing
    terms.append([int(np.argmax(np.diff(post["indptr"])))] * 2 + terms[0][:3])  # the most frequent term, twice
    for t in terms:
        want = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], world["avgdl"],
                                       1.7, 0.83, t)
        got = idx.bm25_scores(t)
        assert np.array_equal(got, want), t
        for k in (1, 25, 64):
            doc, sc, cnt = idx.bm25_search(t, k)
            order = ref_search.canonical_topk(want, k)
            assert cnt == k and doc.tolist() == order.tolist() and np.array_equal(sc, want[order]), (t, k)


def test_hybrid_consistent_at_full_size(world):
    from oracle import ref_search
    from anrag import synth

    idx, Q = world["idx"], world["Q"]
    terms = synth.bm25_queries(world["post"], 4, 5)
    for qi in range(4):
        q = Q[qi].cpu().numpy()
        ids, scores = idx.hybrid_search(q, terms[qi], 25, 5.0, 1.0, 40, 10)
        dl = idx.dense_search(q, 25)[0][0].tolist()
        bl = idx.bm25_search(terms[qi], 25)[0].tolist()
        want = ref_search.weighted_reciprocal_rank_fusion([(dl, "d"), (bl, "b")], {"d": 5.0, "b": 1.0}, 40)[:10]
        assert ids.tolist() == [i for i, _ in want] and scores.tolist() == [s for _, s in want]


def test_c4_with_a_source_filter_at_full_size(world):
    """BASELINE configs[3] (256 queries x 1M x 768) under a source filter, both K2 arithmetic modes: the filtered
    epilogues of dense_batched_kernel<256> and dense_batched_split_dma_kernel at the full size, against torch.topk
    over the allowed rows."""
    from anrag.index import Index
    from helpers import assert_ranking_matches

    torch, E, Q = world["torch"], world["E"], world["Q"]
    k = 10
    src = (np.arange(N, dtype=np.int64) % 300).astype(np.uint16)
    allow = np.zeros(300, np.uint8)
    allow[45:] = 1                                            # ~15 % of the rows drop out
    keep = torch.from_numpy(allow[src].astype(bool)).to(E.device)
    scores = Q @ E.T
    scores[:, ~keep] = float("-inf")
    ref = scores.topk(k, dim=1)
    ri, rv = ref.indices.cpu().numpy(), ref.values.cpu().numpy()
    qh = Q.cpu().numpy()
    torch.cuda.synchronize()
    with Index(0) as idx:
        idx.dense_load((E.data_ptr(), N, D), source_id=src)
        for mode in ("f32", "bf16x3"):
            idx.set_batched_precision(mode)
            d, s, c = idx.dense_search(qh, k, allow)
            assert np.all(c == k)
            assert np.all(allow[src[d]] == 1), mode
            for i in range(len(qh)):
                assert_ranking_matches(ri[i], rv[i], d[i], s[i], 1e-4, None, f"{mode} query {i}")


def test_corpus_past_2_to_31_elements():
    """3M x 768 = 2.3e9 elements, 9.2 GB: every offset past 2^31 elements / 2^32 bytes (a C5 shard at 1M x 1024 stays just
    below both).  Queries planted next to the LAST rows; K1, both K2 modes, the score-array form, the fp64 entry point and
    anrag_dense_scores against torch."""
    import torch
    from anrag import synth
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, d, k = 3_000_000, 768, 10
    E = synth.dense_corpus(n, d, 5, dev)
    Q, _ = synth.dense_queries(E, 40, 6)
    hi_rows = torch.tensor([n - 1, n - 7, 2_900_000, 2_800_001], device=dev)
    Q[:4] = E[hi_rows] + 0.02 * torch.randn((4, d), device=dev)
    Q[:4] /= Q[:4].norm(dim=1, keepdim=True)
    tops = [(Q[i:i + 8] @ E.T).topk(k, dim=1) for i in range(0, 40, 8)]
    ri = torch.cat([t.indices for t in tops]).cpu().numpy()
    rv = torch.cat([t.values for t in tops]).cpu().numpy()
    full0 = (Q[0] @ E.T).cpu().numpy()
    qh = Q.cpu().numpy()
    torch.cuda.synchronize()
    with Index(0) as idx:
        idx.dense_load((E.data_ptr(), n, d))
        d1, s1, _ = idx.dense_search(qh[:6], k)                      # K1
        assert np.array_equal(d1, ri[:6]) and np.max(np.abs(s1 - rv[:6])) <= 1e-4
        assert d1[:4, 0].tolist() == hi_rows.cpu().tolist()
        for mode in ("f32", "bf16x3"):                                # K2
            idx.set_batched_precision(mode)
            d2, s2, _ = idx.dense_search(qh, k)
            assert np.mean(d2 == ri) > 0.995 and np.max(np.abs(s2 - rv)) <= 1e-4, mode
            assert d2[:4, 0].tolist() == hi_rows.cpu().tolist()
        idx.set_batched_precision("f32")
        d3, _, _ = idx.dense_search(qh[0], 200)                      # score array + sort
        assert d3[0, :k].tolist() == ri[0].tolist()
        d4, _, _ = idx.dense_search_f64(qh[0].astype(np.float64), k)  # fp64 entry point
        assert d4.tolist() == ri[0].tolist()
        assert np.max(np.abs(idx.dense_scores(qh[0]) - full0)) <= 1e-5


def test_bm25_at_4m_documents_vs_oracle():
    """4M documents, 408M postings, 977 partitions of 4,096 documents (a C5 rank holds 1M; the whole C5 corpus, 8M
    documents / 816M postings, was run the same way once by hand): ids, scores and the score vector bit for bit against
    the oracle's CSR scorer, with and without a source filter."""
    import torch
    from oracle import ref_bm25, ref_search
    from anrag import synth
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, vocab = 4_000_000, 200_000
    post = synth.bm25_postings(n, vocab, 31, dev)
    idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
    avgdl = post["total_len"] / n
    torch.cuda.synchronize()
    post_doc, post_tf = post["post_doc"].cpu().numpy(), post["post_tf"].cpu().numpy()
    queries = synth.bm25_queries(post, 4, 7)
    sid = np.random.default_rng(3).integers(0, 5, size=n).astype(np.uint16)
    with Index(0) as idx:
        idx.bm25_load(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, source_id=sid)
        for qi, terms in enumerate(queries):
            terms = [int(t) for t in terms]
            want = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, terms)
            for allow in (None, np.array([1, 0, 1, 1, 0], np.uint8)):
                mask = None if allow is None else allow[sid].astype(bool)
                doc, sc, cnt = idx.bm25_search(terms, 25, allow)
                rows = ref_search.canonical_topk(want, 25, mask)
                assert cnt == len(rows) and doc[:cnt].tolist() == rows.tolist(), qi
                assert np.array_equal(sc[:cnt], want[rows])
            if qi == 0:
                assert np.array_equal(idx.bm25_scores(terms), want)
