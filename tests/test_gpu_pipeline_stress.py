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
