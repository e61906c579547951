"""GPU: the three-stream hybrid pipeline under back-to-back, MIXED traffic (fused queries, candidate-only queries,
different k, no host sync inside a burst): every result must equal what the same query returns when run alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mixed_bursts_match_serial():
    import torch
    from oracle.make_golden import synth_chunks, synth_dense
    from anrag import _native as nat
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, d = 60000, 384
    chunks = [c for c in synth_chunks(n + 2500, 21) if c["tokens"]][:n]
    corpus = [c["tokens"] for c in chunks]
    e = synth_dense(n, d, 22)
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    rng = np.random.default_rng(23)
    nq = 48
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    terms = [bi.term_ids([str(t) for t in rng.choice(corpus[r], size=int(rng.integers(1, 9)))]) for r in rows]
    lib = nat.load_library()
    with Index(0) as idx:
        idx.dense_load(e)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
        # serial references through the synchronous host API
        fused_ref = [idx.hybrid_search(q[i], terms[i], 25, 5.0, 1.0, 40, 10) for i in range(nq)]
        dense_ref = [idx.dense_search(q[i], 10)[0][0] for i in range(nq)]
        bm25_ref = [idx.bm25_search(terms[i], 25)[0] for i in range(nq)]
        dense25_ref = [idx.dense_search(q[i], 25)[0][0] for i in range(nq)]

        Q = torch.from_numpy(q).to(dev)
        T = torch.full((nq, 16), -1, dtype=torch.int32, device=dev)
        for i, t in enumerate(terms):
            T[i, : len(t)] = torch.from_numpy(t).to(dev)
        out_f = torch.zeros((nq, 10, 2), dtype=torch.int64, device=dev)
        cnt_f = torch.zeros(nq, dtype=torch.int32, device=dev)
        out_c = torch.zeros((nq, 50, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for rounds in range(3):
            out_f.zero_(); out_c.zero_(); cnt_f.zero_()
            torch.cuda.synchronize()
            for i in range(nq):  # one burst: alternating entry points, nothing waits in between
                if i % 3 != 2:
                    nat.check(lib.anrag_hybrid_search_device(idx.handle, Q[i].data_ptr(), T[i].data_ptr(), len(terms[i]), 25,
                                                             5.0, 1.0, 40.0, 10, None, None, out_f[i].data_ptr(),
                                                             cnt_f[i:].data_ptr()))
                else:
                    nat.check(lib.anrag_hybrid_candidates_device(idx.handle, Q[i].data_ptr(), T[i].data_ptr(),
                                                                 len(terms[i]), 25, None, None, out_c[i].data_ptr()))
            idx.sync()
            of, oc, cf = out_f.cpu().numpy(), out_c.cpu().numpy(), cnt_f.cpu().numpy()
            for i in range(nq):
                if i % 3 != 2:
                    ids, scores = fused_ref[i]
                    assert cf[i] == len(ids) and of[i, :cf[i], 1].tolist() == ids.tolist(), (rounds, i)
                    assert of[i, :cf[i], 0].copy().view(np.float64).tolist() == scores.tolist()
                else:
                    assert oc[i, :25, 1].tolist() == dense25_ref[i].tolist(), (rounds, i)
                    assert oc[i, 25:, 1].tolist() == bm25_ref[i].tolist(), (rounds, i)
        # plain dense bursts after the pipeline has been busy
        out_d = torch.zeros((nq, 10, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        idx.dense_search_device(Q.data_ptr(), nq, 10, 0, out_d.data_ptr())
        idx.sync()
        od = out_d.cpu().numpy()
        for i in range(nq):
            assert od[i, :, 1].tolist() == dense_ref[i].tolist()


def test_concurrent_host_callers_match_serial():
    """One index, many host threads (the reference shares one SearchEngine across Streamlit session threads,
    src/app.py:17-27): `anrag_hybrid_search` callers overlap -- each holds the index lock only while it enqueues --
    and are mixed with other entry points that drain the pipeline (dense / BM25 search, a source filter).  Every
    answer must equal the serial one."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.make_golden import synth_chunks, synth_dense
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    n, d = 40000, 256
    chunks = [c for c in synth_chunks(n + 2000, 31) if c["tokens"]][:n]
    corpus = [c["tokens"] for c in chunks]
    e = synth_dense(n, d, 32)
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    table = {}
    sid = np.array([table.setdefault(c["source"], len(table)) for c in chunks], dtype=np.uint16)
    allow = (np.arange(len(table)) % 3 != 0).astype(np.uint8)
    rng = np.random.default_rng(33)
    nq = 96
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    terms = [bi.term_ids([str(t) for t in rng.choice(corpus[r], size=int(rng.integers(1, 9)))]) for r in rows]
    with Index(0) as idx:
        idx.dense_load(e, source_id=sid)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid)

        def ask(i):
            kind = i % 8
            if kind == 6:
                return ("dense", idx.dense_search(q[i], 10)[0][0].tolist())
            if kind == 7:
                return ("bm25", idx.bm25_search(terms[i], 25)[0].tolist())
            flt = allow if kind == 3 else None
            ids, scores = idx.hybrid_search(q[i], terms[i], 25, 5.0, 1.0, 40, 10, flt, flt)
            return ("hybrid", ids.tolist(), scores.tolist())

        serial = [ask(i) for i in range(nq)]
        for workers in (2, 8, 16):
            with ThreadPoolExecutor(workers) as pool:
                assert list(pool.map(ask, range(nq))) == serial, workers


def test_bm25_device_then_hybrid_on_a_fresh_index():
    """ADVICE r1 (api.hip:525): `anrag_bm25_search_device` used list set 0 on the primary stream without taking a
    pipeline slot; the very next hybrid query of a fresh index (slot 0, secondary stream) could overwrite the set
    under it.  It is a member of the pipeline now: both answers must equal the oracle's, call after call with no
    host sync in between, BM25-only queries interleaved with hybrid ones and a query without terms."""
    import torch
    from oracle import ref_search
    from oracle.make_golden import synth_chunks, synth_dense
    from oracle.ref_bm25 import BM25Okapi
    from anrag import _native as nat
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, d, k = 30000, 256, 25
    chunks = [c for c in synth_chunks(n + 1500, 41) if c["tokens"]][:n]
    corpus = [c["tokens"] for c in chunks]
    e = synth_dense(n, d, 42)
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    ref = BM25Okapi(corpus, 1.7, 0.83, 0.05)
    rng = np.random.default_rng(43)
    nq = 40
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    toks = [[str(t) for t in rng.choice(corpus[r], size=int(rng.integers(1, 9)))] for r in rows]
    toks[5] = []  # BM25-only with no terms: the all-zero score array is ranked (rows 0..k-1), like anrag_bm25_search
    terms = [bi.term_ids(t) for t in toks]
    lib = nat.load_library()
    Q = torch.from_numpy(q).to(dev)
    T = torch.full((nq, 16), -1, dtype=torch.int32, device=dev)
    for i, t in enumerate(terms):
        T[i, : len(t)] = torch.from_numpy(t).to(dev)
    for first in ("bm25", "hybrid"):
        with Index(0) as idx:  # FRESH index each time: sequence number 0 is the slot the old code shared
            idx.dense_load(e)
            idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b)
            out_b = torch.zeros((nq, k, 2), dtype=torch.int64, device=dev)
            out_f = torch.zeros((nq, 10, 2), dtype=torch.int64, device=dev)
            cnt_f = torch.zeros(nq, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for i in range(nq):
                bm_turn = (i % 2 == 0) == (first == "bm25")
                if bm_turn or i == 5:
                    nat.check(lib.anrag_bm25_search_device(idx.handle, T[i].data_ptr(), len(terms[i]), k, None,
                                                           out_b[i].data_ptr()))
                else:
                    nat.check(lib.anrag_hybrid_search_device(idx.handle, Q[i].data_ptr(), T[i].data_ptr(), len(terms[i]),
                                                             k, 5.0, 1.0, 40.0, 10, None, None, out_f[i].data_ptr(),
                                                             cnt_f[i:].data_ptr()))
            idx.sync()
            ob, of, cf = out_b.cpu().numpy(), out_f.cpu().numpy(), cnt_f.cpu().numpy()
            for i in range(nq):
                scores = ref.get_scores(toks[i])
                bl = ref_search.canonical_topk(scores, k)
                bm_turn = (i % 2 == 0) == (first == "bm25")
                if bm_turn or i == 5:
                    assert ob[i, :, 1].tolist() == bl.tolist(), (first, i)
                    assert np.array_equal(ob[i, :, 0].copy().view(np.float64), scores[bl]), (first, i)
                else:
                    dl = ref_search.canonical_topk(ref_search.dense_scores(q[i], e), k).tolist()
                    want = ref_search.weighted_reciprocal_rank_fusion([(dl, "d"), (bl.tolist(), "b")],
                                                                      {"d": 5.0, "b": 1.0}, 40)[:10]
                    assert cf[i] == len(want) and of[i, :cf[i], 1].tolist() == [j for j, _ in want], (first, i)
                    assert of[i, :cf[i], 0].copy().view(np.float64).tolist() == [s for _, s in want], (first, i)


def test_single_hybrid_queries_on_the_scan_lanes():
    """A corpus of at most 1 GiB: `anrag_hybrid_search_device` launches a query's scan, BM25 kernel and tail back to back
    on one of four lane streams (no events) and rotates over the lanes; single DENSE queries use the same lanes and leave
    their list merge to the lane's next scan launch.  Interleaved bursts of both (more than the 32 list sets, so sets are
    reused under back-pressure), a query without tokens, a source filter, then a group call (the pipeline: a change of
    mode) -- every answer equals the synchronous host API's, and the steady state allocates nothing."""
    import torch
    from oracle.make_golden import synth_chunks, synth_dense
    from anrag import _native as nat
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index

    dev = torch.device("cuda", 0)
    n, d = 9609, 384
    chunks = [c for c in synth_chunks(n + 800, 41) if c["tokens"]][:n]
    corpus = [c["tokens"] for c in chunks]
    e = synth_dense(n, d, 42)
    bi = Bm25Index(corpus, 1.7, 0.83, 0.05)
    table = {}
    sid = np.array([table.setdefault(c["source"], len(table)) for c in chunks], dtype=np.uint16)
    allow = (np.arange(len(table)) % 4 != 1).astype(np.uint8)
    rng = np.random.default_rng(43)
    nq = 150
    rows = rng.integers(0, n, nq)
    q = e[rows] + 0.05 * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    terms = [bi.term_ids([str(t) for t in rng.choice(corpus[r], size=int(rng.integers(1, 9)))]) for r in rows]
    terms[7] = np.zeros(0, np.int32)  # no tokens: the dense list is the answer (query_rag_retrieval.py:363-366)
    lib = nat.load_library()
    with Index(0) as idx:
        idx.dense_load(e, source_id=sid)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid)
        flt = lambda i: allow if i % 11 == 3 else None
        fused_ref = [idx.hybrid_search(q[i], terms[i], 25, 5.0, 1.0, 40, 10, flt(i), flt(i)) for i in range(nq)]
        dense_ref = [idx.dense_search(q[i], 10)[0][0] for i in range(nq)]
        Q = torch.from_numpy(q).to(dev)
        T = torch.full((nq, 16), -1, dtype=torch.int32, device=dev)
        for i, t in enumerate(terms):
            T[i, : len(t)] = torch.from_numpy(t).to(dev)
        words = np.zeros(2048, np.uint32)  # the device form of a source filter: bit s of 65,536 = source s allowed
        for s_, a in enumerate(allow):
            if a:
                words[s_ >> 5] |= np.uint32(1 << (s_ & 31))
        bits = torch.from_numpy(words.view(np.int32)).to(dev)
        out_f = torch.zeros((nq, 10, 2), dtype=torch.int64, device=dev)
        cnt_f = torch.zeros(nq, dtype=torch.int32, device=dev)
        out_d = torch.zeros((nq, 10, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        allocs = None
        for rounds in range(3):
            out_f.zero_(); out_d.zero_(); cnt_f.zero_()
            torch.cuda.synchronize()
            for i in range(nq):
                if i % 4 == 1:  # a single dense query: its merge waits for the lane's next scan launch
                    nat.check(lib.anrag_dense_search_device(idx.handle, Q[i].data_ptr(), 1, 10, None, out_d[i].data_ptr()))
                else:
                    b = bits.data_ptr() if flt(i) is not None else None
                    nat.check(lib.anrag_hybrid_search_device(idx.handle, Q[i].data_ptr(), T[i].data_ptr(), len(terms[i]), 25,
                                                             5.0, 1.0, 40.0, 10, b, b, out_f[i].data_ptr(), cnt_f[i:].data_ptr()))
            idx.sync()
            of, od, cf = out_f.cpu().numpy(), out_d.cpu().numpy(), cnt_f.cpu().numpy()
            for i in range(nq):
                if i % 4 == 1:
                    assert od[i, :, 1].tolist() == dense_ref[i].tolist(), (rounds, i)
                else:
                    ids, scores = fused_ref[i]
                    assert cf[i] == len(ids) and of[i, :cf[i], 1].tolist() == ids.tolist(), (rounds, i)
                    assert of[i, :cf[i], 0].copy().view(np.float64).tolist() == scores.tolist()
            import ctypes as C

            calls = C.c_int64(0)
            nat.check(lib.anrag_debug_alloc_calls(C.byref(calls)))
            now = calls.value
            assert allocs is None or now == allocs, "the steady state of the lane route allocates"
            allocs = now
        # a group call takes the pipeline: the lanes are drained first
        out_g = torch.zeros((16, 10, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        idx.dense_search_device(Q.data_ptr(), 16, 10, 0, out_g.data_ptr())
        idx.sync()
        og = out_g.cpu().numpy()
        for i in range(16):
            assert og[i, :, 1].tolist() == dense_ref[i].tolist()
