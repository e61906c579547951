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
    Q, planted = synth.dense_queries(E, 32, 4321)
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
    d2, s2, c2 = idx.dense_search(qh, k)              # 32 queries: K2, one MFMA pass
    ref = (Q @ E.T).topk(k, dim=1)
    ri, rv = ref.indices.cpu().numpy(), ref.values.cpu().numpy()
    assert np.all(c1 == k) and np.all(c2 == k)
    assert np.array_equal(d1, ri[:8]) and np.array_equal(d2, ri)
    assert np.max(np.abs(s1 - rv[:8])) <= 1e-4 and np.max(np.abs(s2 - rv)) <= 1e-4
    assert np.array_equal(d2[:, 0], world["planted"].cpu().numpy())          # planted neighbour first
    assert np.all(np.diff(s1, axis=1) <= 0) and np.all(np.diff(s2, axis=1) <= 0)
    again = idx.dense_search(qh[:8], k)
    assert np.array_equal(again[0], d1) and np.array_equal(again[1], s1)     # bit-identical on repeat


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
