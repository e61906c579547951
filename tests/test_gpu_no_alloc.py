"""GPU: (1) a steady-state query loop allocates nothing -- every hipMalloc / hipFree / hipHostMalloc of the library is
counted (`anrag_debug_alloc_calls`), and the list entry points keep their per-call buffers in grow-only pools;
(2) stream ordering without host syncs (`anrag_index_wait_stream` / `anrag_index_signal_stream`): the race of a
framework's zero fill against the library's answer (it bit bench.py once) cannot happen under the documented recipe;
(3) indexes on two devices of one process are independent (arms itself on the first multi-GPU box)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _alloc_calls():
    from anrag import _native as nat

    n = C.c_int64(0)
    nat.check(nat.load_library().anrag_debug_alloc_calls(C.byref(n)))
    return n.value


def _small_world(rng, n=20000, d=128, vocab=500):
    from anrag.bm25_index import Bm25Index

    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    docs = [[f"t{int(j)}" for j in rng.integers(0, vocab, size=int(rng.integers(3, 30)))] for _ in range(n)]
    return e, Bm25Index(docs, k1=1.7, b=0.83, epsilon=0.05)


def test_steady_state_loops_do_not_allocate():
    from anrag.index import Index, rank_batch

    rng = np.random.default_rng(1)
    e, bi = _small_world(rng)
    big = rng.standard_normal((70000, 128), dtype=np.float32)
    q = rng.standard_normal((64, 128), dtype=np.float32)
    terms = [bi.term_ids([f"t{int(j)}" for j in rng.integers(0, 500, size=5)]) for _ in range(64)]
    with Index(0) as idx, Index(0) as bidx:
        idx.dense_load(e)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        bidx.dense_load(big)

        def loop():
            out = []
            out.append(idx.hybrid_search_batch(q, terms, 25, 5.0, 1.0, 40.0, 10))           # anrag_hybrid_search_batch
            out.append(idx.hybrid_search_batch(q[:9], terms[:9], 25, 5.0, 1.0, 40.0, 10))   # a smaller list: same pool
            out.append(bidx.dense_search(q, 10))                                             # K2 passes from host memory
            out.append(idx.dense_search(q[:5], 10))
            out.append(idx.bm25_search(terms[0], 25))
            out.append(idx.hybrid_search(q[0], terms[0], 25, 5.0, 1.0, 40.0, 10))
            out.append(idx.dense_search(q[0], 3000))                                         # score array + library sort
            out.append(idx.wrrf([list(range(50)), list(range(25, 75))], [5.0, 1.0], 40.0, 10))
            legs = [dict(index=idx, weight=5.0, queries=q), dict(index=idx, weight=1.0, term_lists=terms)]
            out.append(rank_batch(legs, 64, 12000, 40, 12000, id_space=20000))               # anrag_rank_batch
            return out

        first = loop()   # warms every pool and scratch buffer
        loop()
        before = _alloc_calls()
        for _ in range(3):
            again = loop()
        assert _alloc_calls() == before, "a steady-state query loop allocated or freed device / pinned memory"
        for a, b in zip(first, again):  # and the pooled buffers did not change any answer
            for x, y in zip(a, b):
                if x is not None:
                    assert np.array_equal(np.asarray(x), np.asarray(y))


def test_stream_ordering_recipe_without_host_sync():
    """torch allocates and zero-fills the output on ITS stream; the library answers on its own streams.  With
    wait_stream / signal_stream around the call the answer survives and torch reads it on its stream, no host sync."""
    import torch
    from anrag import _native as nat
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    e = rng.standard_normal((50000, 256), dtype=np.float32)
    with Index(0) as idx:
        idx.dense_load(e)
        lib = nat.load_library()
        E = torch.from_numpy(e).to(dev)
        for trial in range(20):
            rows = torch.randint(0, 50000, (8,), device=dev)
            Q = E[rows].contiguous()                                   # produced on torch's stream
            out = torch.zeros((8, 10, 2), dtype=torch.int64, device=dev)  # zero fill on torch's stream
            junk = torch.zeros((1 << 22,), device=dev)                 # more queued work in front of the fill
            junk += 1
            out.zero_()
            st = torch.cuda.current_stream().cuda_stream
            idx.wait_stream(st)
            nat.check(lib.anrag_dense_search_device(idx.handle, Q.data_ptr(), 8, 10, None, out.data_ptr()))
            idx.signal_stream(st)
            top = out[:, 0, 1].clone()                                 # read on torch's stream, after the signal
            assert torch.equal(top.cpu(), rows.cpu()), trial           # (.cpu() syncs only torch's stream)


def test_indexes_on_two_devices_are_independent():
    from anrag import _native as nat
    from anrag.index import Index

    if nat.device_count() < 2:
        pytest.skip("one visible device: arms itself on the first multi-GPU box")
    from oracle import ref_search
    from oracle.ref_bm25 import BM25Okapi

    rng = np.random.default_rng(3)
    n, d = 70000, 256
    e = rng.standard_normal((n, d), dtype=np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    docs = [[f"t{int(j)}" for j in rng.integers(0, 300, size=int(rng.integers(3, 20)))] for _ in range(5000)]
    from anrag.bm25_index import Bm25Index

    bi = Bm25Index(docs, k1=1.7, b=0.83, epsilon=0.05)
    ref = BM25Okapi(docs, k1=1.7, b=0.83, epsilon=0.05)
    q = e[rng.integers(0, n, 32)] + 0.05 * rng.standard_normal((32, d), dtype=np.float32)
    toks = ["t1", "t7", "t7", "t250"]
    # device 1 FIRST: a process-wide "attribute already set" flag would leave device 0's kernels without their LDS
    for order in ((1, 0), (0, 1)):
        idxs = {dv: Index(dv) for dv in order}
        try:
            for dv in order:
                idxs[dv].dense_load(e)
                idxs[dv].bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
            for dv in order:
                ix = idxs[dv]
                doc, sc, cnt = ix.dense_search(q, 10)  # 32 queries on 70k rows: K2 (> 64 KB of dynamic LDS)
                one, s1, _ = ix.dense_search(q[0], 10)  # K1
                assert doc[0].tolist() == one[0].tolist()
                rows, sims = ref_search.similarity_search_with_embedding(q[0], e, None, 10, None, canonical=True)
                assert doc[0].tolist() == rows.tolist()
                assert np.array_equal(ix.bm25_scores(bi.term_ids(toks)), ref.get_scores(toks))  # K3
        finally:
            for ix in idxs.values():
                ix.close()
