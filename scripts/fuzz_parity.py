#!/usr/bin/env python3
"""Randomised differential check of the HIP path against the oracle (test infrastructure, like tests/): random
shapes, k, filters, duplicate / unknown / empty term lists, ties, the list form against single calls.  Dense: ids equal
up to 1e-4 score ties; BM25 and
fusion: bit-exact.  usage: python scripts/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import ref_search
from oracle.ref_bm25 import BM25Okapi
from anrag.bm25_index import Bm25Index
from anrag.index import Index
from helpers import assert_ranking_matches

import faulthandler
faulthandler.enable()
LOG = open(os.environ.get("FUZZ_LOG", "/tmp/fuzz.log"), "w")


def log(*a):
    print(*a, file=LOG, flush=True)


def run(budget: float = 60.0, seed: int = 0) -> int:
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = checks = 0
    DIMS = [8, 24, 64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1536, 2048, 3072, 4096]
    while time.time() < t_end:
        rounds += 1
        n = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 255, 257, 1000, 4095, 4096, 4097, 9000, 20000]))
        d = int(rng.choice(DIMS))
        n_src = int(rng.integers(1, 12))
        e = rng.standard_normal((n, d), dtype=np.float32)
        if rng.random() < 0.3:  # exact ties: duplicated rows
            e[rng.integers(0, n, size=max(1, n // 4))] = e[0]
        e /= np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
        sid = rng.integers(0, n_src, size=n).astype(np.uint16)
        vocab = int(rng.choice([5, 50, 2000]))
        lens = rng.integers(0, 40, size=n)
        zipf = rng.zipf(1.3, size=int(lens.sum())) % vocab
        corpus, at = [], 0
        for L in lens:
            corpus.append([str(t) for t in zipf[at: at + L]])
            at += L
        if not any(corpus):
            corpus[0] = ["0"]
        log("corpus", rounds, "n", n, "d", d, "n_src", n_src, "vocab", vocab, "postings", int(lens.sum()))
        ref = BM25Okapi(corpus, k1=1.7, b=0.83, epsilon=0.05)
        bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
        with Index(0) as idx:
            idx.dense_load(e, source_id=sid)
            idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid)
            for trial in range(6):
                k = int(rng.choice([1, 2, 10, 25, 63, 64, 65, 100, n + 3]))
                allow = None if rng.random() < 0.5 else (rng.random(n_src) < 0.6).astype(np.uint8)
                mask = None if allow is None else allow[sid].astype(bool)
                q = e[int(rng.integers(n))] + 0.1 * rng.standard_normal(d).astype(np.float32)
                log("  trial", trial, "k", k, "filter", allow is not None)
                # dense
                doc, score, cnt = idx.dense_search(q, k, allow)
                log("   dense ok")
                sims = ref_search.dense_scores(q, e)
                want_n = min(k, n if mask is None else int(mask.sum()))
                assert int(cnt[0]) == want_n, ("dense count", n, d, k, int(cnt[0]), want_n)
                if want_n:
                    rr = ref_search.canonical_topk(sims, k, mask)
                    assert_ranking_matches(rr, sims[rr], doc[0, :want_n], score[0, :want_n], 1e-4, sims,
                                           f"dense n={n} d={d} k={k}")
                # BM25: exact
                nt = int(rng.integers(0, 12))
                toks = [str(t) for t in rng.integers(0, vocab + 3, size=nt)]
                if nt and rng.random() < 0.3:
                    toks.append(toks[0])
                tid = bi.term_ids(toks)
                log("   bm25", toks)
                bdoc, bscore, bcnt = idx.bm25_search(tid, k, allow)
                log("   bm25 ok")
                sc = ref.get_scores(toks) if toks else np.zeros(n)
                rows = ref_search.canonical_topk(sc, k, mask)
                if toks:
                    assert bdoc[:bcnt].tolist() == rows.tolist(), ("bm25 ids", n, vocab, k, toks)
                    assert bscore[:bcnt].tolist() == sc[rows].tolist(), ("bm25 scores", n, vocab, k)
                    assert np.array_equal(idx.bm25_scores(tid), sc), ("bm25 score vector", n, vocab)
                # fused hybrid (k <= 64) from the device's dense ranking + exact BM25
                if k <= 64 and toks:
                    top_n = int(rng.choice([1, 10, 15, 2 * k]))
                    log("   hybrid top_n", top_n)
                    hid, hs = idx.hybrid_search(q, tid, k, 5.0, 1.0, 40.0, top_n, allow, allow)
                    log("   hybrid ok")
                    dl = doc[0, :want_n].tolist()
                    bl = rows.tolist()
                    want = ref_search.weighted_reciprocal_rank_fusion([(dl, "d"), (bl, "b")], {"d": 5.0, "b": 1.0}, 40)[:top_n]
                    assert hid.tolist() == [i for i, _ in want], ("hybrid ids", n, d, k, top_n)
                    assert hs.tolist() == [s for _, s in want], ("hybrid scores", n, d, k)
                checks += 1
            # the list form against the single calls (scan groups of 8, partial groups, a query without terms)
            nb = int(rng.integers(1, 21))
            qs = e[rng.integers(0, n, size=nb)] + 0.1 * rng.standard_normal((nb, d)).astype(np.float32)
            tls = [bi.term_ids([str(t) for t in rng.integers(0, vocab + 2, size=int(rng.integers(0, 6)))]) for _ in range(nb)]
            kb, tb = int(rng.choice([1, 10, 25, 64])), int(rng.choice([1, 10, 30]))
            log("  batch", nb, kb, tb)
            ids, scores, counts = idx.hybrid_search_batch(qs, tls, kb, 5.0, 1.0, 40.0, tb)
            for i in range(nb):
                wid, ws = idx.hybrid_search(qs[i], tls[i], kb, 5.0, 1.0, 40.0, tb)
                c = int(counts[i])
                assert c == len(wid) and ids[i, :c].tolist() == wid.tolist() and scores[i, :c].tolist() == ws.tolist(), \
                    ("batch row", n, d, nb, i, kb, tb)
    print(f"fuzz ok: {rounds} corpora, {checks} query checks in {budget:.0f} s (seed {seed})")
    return checks


def run_batched(budget: float = 60.0, seed: int = 0) -> int:
    """K2 (16+ queries per call on the matrix cores, corpora of 65,536+ rows), both arithmetic modes: random sizes with
    tail tiles, dims (multiples of 32), query counts incl. 256+ (two passes) and counts that leave most of the block
    padding, k up to 64, source filters, duplicated rows (exact ties), CLUSTERED corpora (near-duplicates of a few
    centres: the sampled threshold sits in a dense score region and lists run long or overflow into the K1 redo).
    Every score within 1e-4 of the oracle's, rows exact outside near-ties, counts exact."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = checks = 0
    while time.time() < t_end:
        rounds += 1
        n = int(rng.choice([65536, 65537, 70001, 100_000, 131_072 + 255, 200_003]))
        d = int(rng.choice([32, 64, 128, 256, 384, 768]))
        n_src = int(rng.integers(2, 9))
        kind = rng.choice(["normal", "ties", "clustered"])
        if kind == "clustered":
            centres = rng.standard_normal((int(rng.integers(3, 40)), d), dtype=np.float32)
            e = centres[rng.integers(0, len(centres), size=n)] + np.float32(rng.choice([1e-3, 0.05, 0.3])) * rng.standard_normal((n, d), dtype=np.float32)
        else:
            e = rng.standard_normal((n, d), dtype=np.float32)
            if kind == "ties":
                originals = e[rng.integers(0, n, size=4)].copy()
                e[rng.integers(0, n, size=n // 8)] = originals[rng.integers(0, 4, size=n // 8)]
        e /= np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
        sid = rng.integers(0, n_src, size=n).astype(np.uint16)
        log("K2 corpus", rounds, "n", n, "d", d, kind)
        with Index(0) as idx:
            idx.dense_load(e, source_id=sid)
            for trial in range(3):
                mode = "bf16x3" if rng.random() < 0.6 else "f32"
                nq = int(rng.choice([16, 17, 33, 100, 128, 129, 200, 256, 300]))
                k = int(rng.choice([1, 5, 10, 25, 64]))
                allow = None if rng.random() < 0.5 else (rng.random(n_src) < 0.6).astype(np.uint8)
                if allow is not None and not allow.any():
                    allow[0] = 1
                mask = None if allow is None else allow[sid].astype(bool)
                q = e[rng.integers(0, n, size=nq)] + 0.1 * rng.standard_normal((nq, d)).astype(np.float32)
                q[0] = rng.standard_normal(d).astype(np.float32)
                log("  trial", trial, mode, "nq", nq, "k", k, "filter", allow is not None)
                idx.set_batched_precision(mode)
                doc, score, count = idx.dense_search(q, k, allow)
                for qi in range(nq):
                    full = ref_search.dense_scores(q[qi], e)
                    want = ref_search.canonical_topk(full, k, mask)
                    m = int(count[qi])
                    assert m == len(want), (mode, qi, m, len(want))
                    assert_ranking_matches(want, full[want], doc[qi, :m], score[qi, :m], 1e-4, full, f"K2 {mode} r{rounds} q{qi}")
                    checks += 1
    log("K2 done", rounds, checks)
    return checks


def run_big(budget: float = 60.0, seed: int = 0) -> int:
    """BM25 on corpora of 262k .. 1.2M documents (partitions of 1,280 .. 4,096 documents: the 1,024-thread form of
    K3, which the corpora of run() never reach), built on the GPU (anrag.synth), against the oracle's CSR scorer:
    score vectors and top-k bit for bit; random vocabulary sizes and document lengths (all-zero partitions, heavy
    ties, terms that are in every document), term lists with duplicates / unknown ids / up to 40 terms, filters."""
    import torch
    from oracle import ref_bm25
    from anrag import synth

    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    t_end = time.time() + budget
    rounds = checks = 0
    while time.time() < t_end:
        rounds += 1
        n = int(rng.integers(262_145, 1_200_000))
        vocab = int(rng.choice([40, 3000, 120_000]))
        median_len = float(rng.choice([1.5, 12.0, 60.0]))
        post = synth.bm25_postings(n, vocab, int(rng.integers(1 << 30)), dev, median_len=median_len)
        idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
        if rng.random() < 0.3:
            idf[rng.integers(0, vocab, size=max(1, vocab // 10))] = 0.0  # `idf or 0`: such terms contribute nothing
        avgdl = post["total_len"] / n
        n_src = int(rng.integers(1, 9))
        sid = rng.integers(0, n_src, size=n).astype(np.uint16)
        torch.cuda.synchronize()
        post_doc, post_tf = post["post_doc"].cpu().numpy(), post["post_tf"].cpu().numpy()
        df = np.diff(post["indptr"])
        log("big corpus", rounds, "n", n, "vocab", vocab, "median_len", median_len, "postings", len(post_doc))
        with Index(0) as idx:
            idx.bm25_load(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, source_id=sid)
            for trial in range(5):
                nt = int(rng.choice([0, 1, 2, 9, 17, 40]))
                frequent = np.argsort(-df)[: max(1, min(vocab, 50))]
                terms = [int(rng.choice(frequent)) if rng.random() < 0.4 else int(rng.integers(-1, vocab)) for _ in range(nt)]
                if nt and rng.random() < 0.3:
                    terms.append(terms[0])
                k = int(rng.choice([1, 5, 25, 64]))
                allow = None if rng.random() < 0.5 else (rng.random(n_src) < 0.6).astype(np.uint8)
                mask = None if allow is None else allow[sid].astype(bool)
                log("  trial", trial, "terms", terms, "k", k, "filter", allow is not None)
                want = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl, 1.7, 0.83, terms)
                assert np.array_equal(idx.bm25_scores(terms), want), ("big bm25 score vector", n, vocab, terms)
                doc, sc, cnt = idx.bm25_search(terms, k, allow)
                rows = ref_search.canonical_topk(want, k, mask)
                assert cnt == len(rows) and doc[:cnt].tolist() == rows.tolist(), ("big bm25 ids", n, vocab, k, terms)
                assert np.array_equal(sc[:cnt], want[rows]), ("big bm25 scores", n, vocab, k, terms)
                checks += 1
    print(f"big-partition fuzz ok: {rounds} corpora, {checks} query checks in {budget:.0f} s (seed {seed})")
    return checks


def run_rank(budget: float = 60.0, seed: int = 0) -> int:
    """Full-ranking lists (`anrag_rank_batch`, rank_batch.hip) against the per-query entry points and the oracle: random
    corpora below / around / above the sort kernel's LDS capacity (13,312 / 16,384 composites), shaped and odd dimensions,
    duplicated rows and constant vectors (heavy ties at every cut), filters incl. one that keeps almost nothing, k from 1
    to past the capacity of the smaller leg, 1 - 3 legs with permuted row -> document maps, queries without BM25 terms,
    top_n below / above the fused length.  Single legs: ids and scores equal the per-query calls'; BM25 against the oracle
    bit for bit; fusions bit for bit against the reference's dict + stable sort over the per-query lists.  Interleaved
    with single dense queries on the scan lanes (their answers against `dense_search`)."""
    from anrag.index import rank_batch, rank_caps
    from anrag import _native as nat
    import ctypes as C

    cap32, cap64 = rank_caps()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = checks = 0
    while time.time() < t_end:
        rounds += 1
        n = int(rng.choice([3, 65, 1000, 4097, 9609, 12287, 12288, 12289, cap64 - 1, cap64, cap64 + 1, cap32, cap32 + 1, 20000, 40000]))  # 12,288: the fp64 radix sort's capacity
        d = int(rng.choice([8, 64, 128, 384, 768]))
        nb = int(rng.choice([max(2, n // 2), n, n + 50]))
        n_src = int(rng.integers(1, 9))
        e = rng.standard_normal((n, d), dtype=np.float32)
        kind = rng.choice(["normal", "dups", "constant"])
        if kind == "dups":
            e[rng.integers(0, n, size=max(1, n // 3))] = e[0]
        elif kind == "constant":
            e[:] = e[0]  # every score equal: the order is the row order, every cut is inside a tie
        sid = rng.integers(0, n_src, size=n).astype(np.uint16)
        sidb = rng.integers(0, n_src, size=nb).astype(np.uint16)
        vocab = int(rng.choice([4, 60, 3000]))
        lens = rng.integers(0, 12, size=nb)
        zipf = rng.zipf(1.3, size=int(lens.sum())) % vocab
        corpus, at = [], 0
        for L in lens:
            corpus.append([str(t) for t in zipf[at: at + L]])
            at += L
        if not any(corpus):
            corpus[0] = ["0"]
        ref = BM25Okapi(corpus, k1=1.7, b=0.83, epsilon=0.05)
        bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
        space = max(n, nb) + 7
        map_d = rng.permutation(space)[:n].astype(np.int64)
        map_b = rng.permutation(space)[:nb].astype(np.int64)
        log("rank corpus", rounds, "n", n, "d", d, "nb", nb, kind, "vocab", vocab)
        with Index(0) as di, Index(0) as bx:
            di.dense_load(e, source_id=sid)
            bx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sidb)
            lib = nat.load_library()
            for trial in range(3):
                nq = int(rng.choice([1, 3, 8, 9, 21]))
                k = int(rng.choice([1, 7, 64, 65, 300, 12000, min(cap64, nb), min(cap32, n)]))
                if min(k, nb) > cap64 or min(k, n) > cap32:
                    continue
                top_n = int(rng.choice([1, 15, k, 2 * k + 5]))
                allow_d = None if rng.random() < 0.5 else (rng.random(n_src) < rng.choice([0.1, 0.6])).astype(np.uint8)
                allow_b = None if allow_d is None else allow_d
                q = e[rng.integers(0, n, size=nq)] + 0.1 * rng.standard_normal((nq, d)).astype(np.float32)
                q = np.ascontiguousarray(q, dtype=np.float32)
                tls = [bi.term_ids([str(t) for t in rng.integers(0, vocab + 2, size=int(rng.integers(0, 7)))]) for _ in range(nq)]
                log("  trial", trial, "nq", nq, "k", k, "top_n", top_n, "filter", allow_d is not None)
                # per-query lists (the library-sort path for k > 64, the register top-k below)
                dl, bl = [], []
                for i in range(nq):
                    doc, sc_, cnt = di.dense_search(q[i], k, allow_d)
                    dl.append((doc[0, :int(cnt[0])], sc_[0, :int(cnt[0])]))
                    if len(tls[i]):
                        bdoc, bsc, bc = bx.bm25_search(tls[i], k, allow_b)
                        bl.append((bdoc[:bc], bsc[:bc]))
                    else:
                        bl.append((np.zeros(0, np.int64), np.zeros(0)))
                # single legs
                ids, sc, cnt = rank_batch([dict(index=di, weight=1.0, allow=allow_d, queries=q)], nq, k, 40, top_n, want_scores=True)
                for i in range(nq):
                    m = min(top_n, len(dl[i][0]))
                    assert cnt[i] == m and ids[i, :m].tolist() == dl[i][0][:m].tolist(), ("rank dense ids", n, d, k, top_n, i)
                    assert np.array_equal(sc[i, :m], dl[i][1][:m].astype(np.float64)), ("rank dense scores", n, d, k)
                ids, sc, cnt = rank_batch([dict(index=bx, weight=1.0, allow=allow_b, term_lists=tls, doc_of_row=map_b)], nq, k, 40,
                                          top_n, want_scores=True)
                for i in range(nq):
                    m = min(top_n, len(bl[i][0]))
                    assert cnt[i] == m and ids[i, :m].tolist() == map_b[bl[i][0][:m]].tolist(), ("rank bm25 ids", nb, k, top_n, i)
                    assert np.array_equal(sc[i, :m], bl[i][1][:m]), ("rank bm25 scores", nb, k)
                # fusion of the two legs (+ the dense leg again under another weight and map: three lists)
                three = rng.random() < 0.5
                legs = [dict(index=di, weight=5.0, allow=allow_d, queries=q, doc_of_row=map_d),
                        dict(index=bx, weight=1.0, allow=allow_b, term_lists=tls, doc_of_row=map_b)]
                names = {"a": 5.0, "b": 1.0}
                if three:
                    map_c = rng.permutation(space)[:n].astype(np.int64)
                    legs.insert(1, dict(index=di, weight=2.0, allow=allow_d, queries=q, doc_of_row=map_c))
                    names["c"] = 2.0
                if min(top_n, space, len(legs) * k) > cap64:
                    top_n = cap64  # the fused order is cut to what one workgroup can sort (the call refuses beyond)
                ids, sc, cnt = rank_batch(legs, nq, k, 40, top_n, id_space=space, want_scores=True)
                # rank of an expected document, with the lists and without them (counts instead of sorts): a document
                # of the answer, one past its cut, one nowhere -- for the fusion and for each leg alone
                for use, kw in ((legs, dict(id_space=space)), (legs[:1], {}), (legs[-1:], {})):
                    full, _, cfull = rank_batch(use, nq, k, 40, top_n, **kw)
                    expect = np.array([full[i, rng.integers(0, cfull[i])] if cfull[i] and rng.random() < 0.7
                                       else rng.integers(0, space + 3) for i in range(nq)], dtype=np.int64)
                    _, _, c1, r1 = rank_batch(use, nq, k, 40, top_n, expect=expect, **kw)
                    none, _, c2, r2 = rank_batch(use, nq, k, 40, top_n, expect=expect, want_ids=False, **kw)
                    assert none is None and np.array_equal(cfull, c1) and np.array_equal(cfull, c2), ("rank counts", n, nb, k, top_n)
                    for i in range(nq):
                        pos = np.nonzero(full[i, :cfull[i]] == expect[i])[0]
                        want = int(pos[0]) + 1 if len(pos) else -1
                        assert r1[i] == want and r2[i] == want, ("rank of expected", n, nb, k, top_n, i, len(use), r1[i], r2[i], want)
                        checks += 1
                for i in range(nq):
                    lists = [(map_d[dl[i][0]].tolist(), "a")]
                    if three:
                        lists.append((map_c[dl[i][0]].tolist(), "c"))
                    lists.append((map_b[bl[i][0]].tolist(), "b"))
                    lists = [l for l in lists if l[0]]
                    if len(lists) > 1:
                        fused = ref_search.weighted_reciprocal_rank_fusion(lists, names, 40)[:top_n]
                        assert cnt[i] == len(fused) and ids[i, :cnt[i]].tolist() == [x for x, _ in fused], ("rank fused ids", n, nb, k, top_n, i)
                        assert sc[i, :cnt[i]].tolist() == [s_ for _, s_ in fused], ("rank fused scores", n, nb, k, i)
                    elif len(lists) == 1:
                        m = min(top_n, len(lists[0][0]))
                        assert cnt[i] == m and ids[i, :m].tolist() == lists[0][0][:m], ("rank one list", n, nb, k, i)
                    else:
                        assert cnt[i] == 0
                    checks += 1
            # single dense queries on the scan lanes, interleaved with a group call and syncs
            if d % 64 == 0:
                import torch

                dev = torch.device("cuda", 0)
                nl = int(rng.integers(3, 40))
                ql = np.ascontiguousarray(e[rng.integers(0, n, size=nl)] + 0.1 * rng.standard_normal((nl, d)).astype(np.float32))
                Q = torch.from_numpy(ql).to(dev)
                kk = int(rng.choice([1, 10, 64]))
                out = torch.zeros((nl, kk, 2), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                i = 0
                while i < nl:
                    g = 1 if rng.random() < 0.8 else int(min(nl - i, rng.integers(2, 12)))
                    nat.check(lib.anrag_dense_search_device(di.handle, Q[i].data_ptr(), g, kk, None, out[i].data_ptr()))
                    if rng.random() < 0.15:
                        di.sync()
                    i += g
                di.sync()
                got = out.cpu().numpy()
                for i in range(nl):
                    doc, sc_, c = di.dense_search(ql[i], kk)
                    m = int(c[0])
                    assert got[i, :m, 1].tolist() == doc[0, :m].tolist(), ("lane ids", n, d, kk, i)
                    assert np.array_equal(got[i, :m, 0].copy().view(np.float64), sc_[0, :m].astype(np.float64)), ("lane scores", n, d, kk, i)
                    checks += 1
    print(f"rank fuzz ok: {rounds} corpora, {checks} checks in {budget:.0f} s (seed {seed})")
    return checks


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "big":
        run_big(budget, seed)
    elif len(sys.argv) > 3 and sys.argv[3] == "rank":
        run_rank(budget, seed)
    elif len(sys.argv) > 3 and sys.argv[3] == "batched":
        print("K2 fuzz ok:", run_batched(budget, seed), "query checks")
    else:
        run(budget, seed)
