"""GPU parity: K2 (batched queries on the fp32 matrix cores, sampled-threshold exact top-k) vs the oracle's loop
of the single-query path.  Tolerance 1e-4 on scores (north_star), rows exact outside near-ties."""
import numpy as np
import pytest

from helpers import assert_ranking_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(77)
    n, d = 70001, 768  # >= 65536 rows: the batched path engages; odd size exercises the tile tail
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    return e


@pytest.mark.parametrize("nq,k", [(256, 10), (40, 25), (300, 10), (17, 64)])
def test_batched_matches_oracle(corpus, nq, k):
    from oracle import ref_search
    from anrag.index import Index

    e = corpus
    n, d = e.shape
    rng = np.random.default_rng(nq * 100 + k)
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[1] = rng.standard_normal(d, dtype=np.float32)  # an un-planted query
    sources = np.array([i % 5 for i in range(n)], dtype=np.uint16)
    with Index(0) as idx:
        idx.dense_load(e, source_id=sources, doc_id_base=7)
        for allow in (None, np.array([1, 0, 1, 0, 0], dtype=np.uint8)):
            doc, score, count = idx.dense_search(q, k, allow)   # n_queries >= 16 -> K2 passes of <= 256
            keep = np.ones(n, bool) if allow is None else allow[sources].astype(bool)
            for qi in range(nq):
                full = ref_search.dense_scores(q[qi], e)
                want = ref_search.canonical_topk(full, k, keep)
                m = int(count[qi])
                assert m == len(want)
                assert_ranking_matches(want + 7, full[want], doc[qi, :m], score[qi, :m], 1e-4, None, f"q{qi}")
            # the batch = 1 kernel agrees row for row
            d1, s1, c1 = idx.dense_search(q[:3], k, allow)
            for qi in range(3):
                assert_ranking_matches(d1[qi, :c1[qi]], s1[qi, :c1[qi]], doc[qi, :count[qi]], score[qi, :count[qi]], 1e-5)


def test_overflowing_survivor_lists_fall_back():
    """Every row equal: all N scores tie with the sampled threshold, every list overflows its 8192 slots, and the
    call must still return the exact (row-ascending) answer through the batch = 1 redo."""
    from anrag.index import Index

    rng = np.random.default_rng(1)
    d, n = 256, 66000
    row = rng.standard_normal(d, dtype=np.float32)
    e = np.tile(row, (n, 1))
    q = rng.standard_normal((16, d), dtype=np.float32)
    with Index(0) as idx:
        idx.dense_load(e)
        doc, score, count = idx.dense_search(q, 5)
        assert np.all(count == 5)
        assert np.all(doc == np.arange(5)[None, :])


def test_split_precision_option(corpus):
    """bf16 x 3 split products (opt-in): every score within 1e-4 of the f32 oracle (bound ~3e-5 for unit-norm rows),
    rows identical outside near-ties, and the measured error is reported."""
    from oracle import ref_search
    from anrag.index import Index

    e = corpus
    n, d = e.shape
    rng = np.random.default_rng(5)
    nq, k = 200, 10
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    with Index(0) as idx:
        idx.dense_load(e)
        idx.set_batched_precision("bf16x3")
        doc, score, count = idx.dense_search(q, k)
        idx.set_batched_precision("f32")
        doc32, score32, _ = idx.dense_search(q, k)
    worst = 0.0
    for qi in range(nq):
        full = ref_search.dense_scores(q[qi], e)
        want = ref_search.canonical_topk(full, k)
        assert_ranking_matches(want, full[want], doc[qi], score[qi], 1e-4, full, f"bf16x3 q{qi}")
        worst = max(worst, float(np.max(np.abs(full[doc[qi]] - score[qi]))))
    assert worst <= 5e-5, worst
    assert (doc == doc32).mean() > 0.99
    print(f"bf16x3 max |score - f32 oracle| over {nq * k} results: {worst:.2e}")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_exactly_k_rows_stand_out(mode):
    """Adversarial for the sampled threshold: only k rows score high for a query and none of them need be in the
    sample.  The threshold is then the k-th best of ORDINARY rows -- still a lower bound -- and all k planted rows must
    come back, in both arithmetic modes (the sampled and the full pass of a mode score a row bit-identically, so a
    sampled row can never fall just below its own threshold)."""
    from oracle import ref_search
    from anrag.index import Index

    rng = np.random.default_rng(11)
    n, d, nq, k = 66_003, 128, 48, 10
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    planted = rng.choice(n, size=(nq, k), replace=False)
    for qi in range(nq):
        for j, r in enumerate(planted[qi]):
            v = q[qi] + (0.02 + 0.01 * j) * rng.standard_normal(d, dtype=np.float32)
            e[r] = v / np.linalg.norm(v)
    with Index(0) as idx:
        idx.dense_load(e)
        idx.set_batched_precision(mode)
        doc, score, count = idx.dense_search(q, k)
    for qi in range(nq):
        assert int(count[qi]) == k
        assert set(doc[qi].tolist()) == set(planted[qi].tolist()), qi
        full = ref_search.dense_scores(q[qi], e)
        want = ref_search.canonical_topk(full, k)
        assert_ranking_matches(want, full[want], doc[qi], score[qi], 1e-4, full, f"{mode} q{qi}")


def test_split_precision_filter_nan_reload():
    """The split-precision full pass (LDS-DMA images of the corpus): source filter, a NaN row (ranks first, like numpy's
    argsort puts it and like K1), a tail tile (rows not a multiple of 256), 17 queries (15/16 of the block is padding)
    and a second dense_load on the same index (the images must be rebuilt, not reused)."""
    from oracle import ref_search
    from anrag.index import Index

    rng = np.random.default_rng(12)
    n, d, nq, k = 65_536 + 77, 256, 17, 12
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e2 = rng.standard_normal((n + 300, d), dtype=np.float32)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    q = e[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sources = (np.arange(n) % 7).astype(np.uint16)
    allow = np.array([1, 1, 0, 1, 0, 0, 1], dtype=np.uint8)
    with Index(0) as idx:
        idx.dense_load(e, source_id=sources)
        idx.set_batched_precision("bf16x3")
        for al in (None, allow):
            doc, score, count = idx.dense_search(q, k, al)
            keep = np.ones(n, bool) if al is None else al[sources].astype(bool)
            for qi in range(nq):
                full = ref_search.dense_scores(q[qi], e)
                want = ref_search.canonical_topk(full, k, keep)
                assert int(count[qi]) == k
                assert_ranking_matches(want, full[want], doc[qi], score[qi], 1e-4, full, f"q{qi}")
        # a NaN row
        e_nan = e.copy()
        e_nan[40_000, 3] = np.nan
        idx.dense_load(e_nan)
        doc, score, count = idx.dense_search(q, k)
        assert np.all(doc[:, 0] == 40_000) and np.all(np.isinf(score[:, 0]))
        for qi in range(nq):
            full = ref_search.dense_scores(q[qi], e)
            full[40_000] = -np.inf
            want = ref_search.canonical_topk(full, k - 1)
            assert_ranking_matches(want, full[want], doc[qi, 1:], score[qi, 1:], 1e-4, full, f"nan q{qi}")
        # another corpus on the same index
        idx.dense_load(e2)
        doc, score, count = idx.dense_search(q, k)
        for qi in range(nq):
            full = ref_search.dense_scores(q[qi], e2)
            want = ref_search.canonical_topk(full, k)
            assert_ranking_matches(want, full[want], doc[qi], score[qi], 1e-4, full, f"reload q{qi}")
