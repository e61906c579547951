"""The synthetic corpus is a function of (seed, global row): the union of the ranks' shards must be bit-identical
to the single-GPU corpus, or a sharded answer could not be checked against a single index (bench.py does)."""
import numpy as np
import torch

from anrag import synth
from anrag.sharded import shard_bounds


def test_dense_shards_concatenate_to_the_whole():
    n, d = 40_000, 8  # 2.4 blocks of 16,384 rows
    whole = synth.dense_corpus(n, d, 1234, "cpu")
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            parts.append(synth.dense_corpus(hi - lo, d, 1234, "cpu", row_lo=lo))
        assert torch.equal(torch.cat(parts), whole)
    assert not torch.equal(synth.dense_corpus(n, d, 1235, "cpu"), whole)
    np.testing.assert_allclose(whole.norm(dim=1).numpy(), 1.0, rtol=1e-5)


def test_global_queries_sit_next_to_global_rows():
    n, d = 40_000, 16
    whole = synth.dense_corpus(n, d, 1234, "cpu")
    q, rows = synth.dense_queries_global(n, d, 12, 4321, 1234, "cpu")
    q2, rows2 = synth.dense_queries_global(n, d, 12, 4321, 1234, "cpu")
    assert torch.equal(q, q2) and rows.tolist() == rows2.tolist()
    sims = q @ whole.T
    assert sims.argmax(dim=1).tolist() == rows.tolist()


def _postings_to_global(post, doc_lo):
    indptr = post["indptr"]
    term = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    return term, post["post_doc"].numpy().astype(np.int64) + doc_lo, post["post_tf"].numpy()


def test_posting_shards_union_to_the_whole():
    n, vocab = 70_000, 500  # 1.07 blocks of 65,536 documents
    whole = synth.bm25_postings(n, vocab, 777, "cpu", median_len=12.0)
    wt, wd, wf = _postings_to_global(whole, 0)
    for world in (2, 3):
        terms, docs, tfs, lens = [], [], [], []
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            p = synth.bm25_postings(hi - lo, vocab, 777, "cpu", median_len=12.0, doc_lo=lo)
            t, dd, f = _postings_to_global(p, lo)
            terms.append(t), docs.append(dd), tfs.append(f), lens.append(p["doc_len"])
        t, dd, f = np.concatenate(terms), np.concatenate(docs), np.concatenate(tfs)
        order = np.lexsort((dd, t))  # term-major, documents ascending: the whole's order
        assert np.array_equal(t[order], wt) and np.array_equal(dd[order], wd) and np.array_equal(f[order], wf)
        assert np.array_equal(np.concatenate(lens), whole["doc_len"])
    # queries drawn from global documents name terms those documents hold
    qs = synth.bm25_queries_global(n, vocab, 5, 99, 777, "cpu", median_len=12.0)
    qs2 = synth.bm25_queries_global(n, vocab, 5, 99, 777, "cpu", median_len=12.0)
    assert all(np.array_equal(a, b) for a, b in zip(qs, qs2))
    for q in qs:
        assert 2 <= len(q) <= 9 and all(0 <= int(x) < vocab for x in q)
