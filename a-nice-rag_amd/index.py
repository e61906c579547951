"""`Index`: one GPU's shard (corpus rows + postings) resident in HBM, over the C ABI.

Thin, numpy-in / numpy-out; no torch types here.  The reference-shaped classes
(`database_manager.DatabaseManager`, `search_engine.SearchEngine`) sit on top.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _allow_bytes(allow) -> Tuple[Optional[np.ndarray], int]:
    if allow is None:
        return None, 0
    a = np.ascontiguousarray(allow, dtype=np.uint8)
    return a, int(a.size)


def rank_caps() -> Tuple[int, int]:
    """Longest list `rank_batch` returns for a dense leg / for a BM25 leg or a fused ranking."""
    a, b = C.c_int32(), C.c_int32()
    nat.check(nat.load_library().anrag_rank_caps(C.byref(a), C.byref(b)))
    return a.value, b.value


def rank_batch(legs: Sequence[dict], n_queries: int, similarity_k: int, wrrf_k: float, top_n: int,
               id_space: int = 0, want_scores: bool = False, expect=None, want_ids: bool = True):
    """`anrag_rank_batch`: full-ranking retrieval (large k) for a LIST of queries, every step on the device.
    `legs`: dicts in the reference's list order (dense models first, BM25 last) with
        index (Index), weight, allow (uint8 per source or None), doc_of_row (int64 per row or None), and either
        queries (float32 [nq, dim]) or term_lists (one int32 sequence per query; an empty one = no BM25 for that query).
    `expect` (int64 per query): also return the 1-based rank of that document in each answer (-1: absent);
    `want_ids=False` then skips the id lists altogether.
    -> (ids [nq, top_n] int64 (-1 padded) or None, scores [nq, top_n] float64 or None, counts [nq] int32)
       -- plus ranks [nq] int32 as a fourth element when `expect` is given."""
    lib = nat.load_library()
    arr = (nat.RankLeg * len(legs))()
    keep = []
    for i, leg in enumerate(legs):
        ix = leg["index"]
        arr[i].idx = ix.handle
        arr[i].weight = float(leg["weight"])
        allow, ns = _allow_bytes(leg.get("allow"))
        keep.append(allow)
        arr[i].allow_source = nat.ptr(allow)
        arr[i].n_sources = ns
        dmap = leg.get("doc_of_row")
        if dmap is not None:
            dmap = np.ascontiguousarray(dmap, dtype=np.int64)
            keep.append(dmap)
        arr[i].doc_of_row = nat.ptr(dmap)
        if leg.get("queries") is not None:
            q = _f32(leg["queries"])
            if q.ndim == 1:
                q = q[None, :]
            assert q.shape == (n_queries, ix.dim), f"leg {i}: queries {q.shape} != ({n_queries}, {ix.dim})"
            keep.append(q)
            arr[i].kind = nat.LEG_DENSE
            arr[i].queries = q.ctypes.data
        else:
            term_lists = leg["term_lists"]
            assert len(term_lists) == n_queries
            offsets = np.zeros(n_queries + 1, dtype=np.int64)
            np.cumsum(np.fromiter(map(len, term_lists), np.int64, n_queries), out=offsets[1:])
            if not offsets[-1]:
                terms = np.zeros(1, np.int32)
            elif all(type(t) is np.ndarray and t.dtype == np.int32 and t.ndim == 1 for t in term_lists):
                terms = np.concatenate(term_lists)  # (what Bm25Index.term_ids returns: no per-list conversion)
            else:
                terms = np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in term_lists]).astype(np.int32)
            terms = np.ascontiguousarray(terms)
            keep += [offsets, terms]
            arr[i].kind = nat.LEG_BM25
            arr[i].term_ids = terms.ctypes.data
            arr[i].term_offsets = offsets.ctypes.data
    out_id = np.empty((n_queries, top_n), np.int64) if want_ids else None
    out_score = np.empty((n_queries, top_n), np.float64) if want_scores and want_ids else None
    count = np.zeros(n_queries, np.int32)
    exp = ranks = None
    if expect is not None:
        exp = np.ascontiguousarray(expect, dtype=np.int64)
        assert exp.shape == (n_queries,)
        ranks = np.empty(n_queries, np.int32)
    nat.check(lib.anrag_rank_batch(C.cast(arr, C.c_void_p), len(legs), int(n_queries), int(similarity_k), float(wrrf_k),
                                   int(top_n), int(id_space), nat.ptr(out_id), nat.ptr(out_score),
                                   count.ctypes.data, nat.ptr(exp), nat.ptr(ranks)))
    if expect is not None:
        return out_id, out_score, count, ranks
    return out_id, out_score, count


class Index:
    """Owns an `anrag_index*`.  Not copyable; `close()` (or GC) frees the HBM."""

    def __init__(self, device: int = 0):
        self._lib = nat.load_library()
        h = C.c_void_p()
        nat.check(self._lib.anrag_index_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.n_rows = 0
        self.dim = 0
        self.n_docs = 0

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.anrag_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise nat.AnragError(-3, "index is closed")
        return self._h

    def set_streams(self, primary: int = 0, secondary: int = 0, fusion: int = 0) -> None:
        """Run on caller-owned HIP streams (e.g. torch.cuda.current_stream().cuda_stream); 0 = the index's own.
        Roles: include/anrag.h, anrag_index_set_streams."""
        nat.check(self._lib.anrag_index_set_streams(self.handle, primary or None, secondary or None, fusion or None))

    def sync(self) -> None:
        nat.check(self._lib.anrag_index_sync(self.handle))

    def wait_stream(self, stream: int = 0) -> None:
        """Order the index's streams after the work already enqueued on `stream` (e.g.
        torch.cuda.current_stream().cuda_stream): inputs produced and outputs allocated / filled there are safe to hand
        to the *_device entry points, no host sync."""
        nat.check(self._lib.anrag_index_wait_stream(self.handle, stream or None))

    def signal_stream(self, stream: int = 0) -> None:
        """Order `stream` after everything the index has enqueued: the caller's framework may read the results on
        that stream without a host sync."""
        nat.check(self._lib.anrag_index_signal_stream(self.handle, stream or None))

    # ------------------------------------------------------------------ dense
    def dense_load(self, embeddings, source_id=None, doc_id=None, doc_id_base: int = 0) -> None:
        """Upload the row-major fp32 corpus matrix (numpy array, or an (address, n, d) tuple of a
        device buffer such as a torch tensor's data_ptr()).  A device buffer must be COMPLETE when this is
        called (synchronise the stream that produced it): the library copies on its own stream."""
        if isinstance(embeddings, tuple):
            addr, n, d = embeddings
            keep = None
        else:
            keep = _f32(embeddings)
            assert keep.ndim == 2
            n, d = keep.shape
            addr = keep.ctypes.data
        src = None if source_id is None else np.ascontiguousarray(source_id, dtype=np.uint16)
        doc = None if doc_id is None else np.ascontiguousarray(doc_id, dtype=np.int64)
        assert src is None or src.shape == (n,)
        assert doc is None or doc.shape == (n,)
        nat.check(self._lib.anrag_dense_load(self.handle, addr, n, d, nat.ptr(src), nat.ptr(doc), int(doc_id_base)))
        self.n_rows, self.dim = int(n), int(d)

    def dense_search(self, queries, k: int, allow_source=None):
        """-> (doc [nq,k] int64, score [nq,k] float32, count [nq] int32); rank order, tail = -1/-inf."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        if self.n_rows == 0:
            raise nat.AnragError(-3, "dense search before dense_load")
        assert q.shape[1] == self.dim, f"query dim {q.shape[1]} != index dim {self.dim}"
        nq = q.shape[0]
        allow, ns = _allow_bytes(allow_source)
        doc = np.empty((nq, k), np.int64)
        score = np.empty((nq, k), np.float32)
        count = np.empty(nq, np.int32)
        nat.check(self._lib.anrag_dense_search(self.handle, q.ctypes.data, nq, int(k), nat.ptr(allow), ns,
                                               doc.ctypes.data, score.ctypes.data, count.ctypes.data))
        return doc, score, count

    def dense_search_f64(self, query, k: int, allow_source=None):
        """One float64 query, scored in fp64 like numpy scores it (`anrag_dense_search_f64`).
        -> (doc [k] int64, score [k] float64, count)."""
        q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
        assert q.size == self.dim, f"query dim {q.size} != index dim {self.dim}"
        allow, ns = _allow_bytes(allow_source)
        doc = np.empty(k, np.int64)
        score = np.empty(k, np.float64)
        count = np.zeros(1, np.int32)
        nat.check(self._lib.anrag_dense_search_f64(self.handle, q.ctypes.data, int(k), nat.ptr(allow), ns,
                                                   doc.ctypes.data, score.ctypes.data, count.ctypes.data))
        return doc, score, int(count[0])

    def set_batched_precision(self, mode: str = "f32") -> None:
        """"f32": exact f32 MFMA (default).  "bf16x3": split-precision products on the bf16 matrix cores (scores within
        ~3e-5 of f32 for unit-norm vectors)."""
        nat.check(self._lib.anrag_set_batched_precision(self.handle, {"f32": 0, "bf16x3": 1}[mode]))

    def dense_scores(self, query) -> np.ndarray:
        q = _f32(query).reshape(-1)
        assert q.size == self.dim
        out = np.empty(self.n_rows, np.float32)
        nat.check(self._lib.anrag_dense_scores(self.handle, q.ctypes.data, out.ctypes.data))
        return out

    def dense_search_device(self, d_queries: int, n_queries: int, k: int, d_allow_bits: int, d_out: int) -> None:
        nat.check(self._lib.anrag_dense_search_device(self.handle, d_queries, n_queries, k, d_allow_bits or None, d_out))

    # ------------------------------------------------------------------ BM25
    def bm25_load(self, indptr, post_doc, post_tf, idf, doc_len, avgdl: float, k1: float, b: float,
                  source_id=None, doc_id=None, doc_id_base: int = 0) -> None:
        """Postings may be numpy arrays or (device address, count) tuples of int32 buffers already in HBM
        (e.g. torch tensors); indptr / idf / doc_len are host arrays."""
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        idf = np.ascontiguousarray(idf, dtype=np.float64)
        doc_len = np.ascontiguousarray(doc_len, dtype=np.int32)
        n_terms, n_docs = indptr.size - 1, doc_len.size

        def as_ptr(a):
            if isinstance(a, tuple):
                addr, count = a
                return None, int(addr), int(count)
            arr = np.ascontiguousarray(a, dtype=np.int32)
            return arr, arr.ctypes.data, int(arr.size)

        keep_d, pd_addr, pd_n = as_ptr(post_doc)
        keep_t, pt_addr, pt_n = as_ptr(post_tf)
        assert idf.size == n_terms and pd_n == pt_n == int(indptr[-1])
        src = None if source_id is None else np.ascontiguousarray(source_id, dtype=np.uint16)
        doc = None if doc_id is None else np.ascontiguousarray(doc_id, dtype=np.int64)
        nat.check(self._lib.anrag_bm25_load(self.handle, indptr.ctypes.data, n_terms, pd_addr or None,
                                            pt_addr or None, idf.ctypes.data, doc_len.ctypes.data, n_docs,
                                            float(avgdl), float(k1), float(b), nat.ptr(src), nat.ptr(doc),
                                            int(doc_id_base)))
        self.n_docs = int(n_docs)

    def bm25_search(self, term_ids: Sequence[int], k: int, allow_source=None):
        """-> (doc [k] int64, score [k] float64, count)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        allow, ns = _allow_bytes(allow_source)
        doc = np.empty(k, np.int64)
        score = np.empty(k, np.float64)
        count = np.zeros(1, np.int32)
        nat.check(self._lib.anrag_bm25_search(self.handle, nat.ptr(t) if t.size else None, int(t.size), int(k),
                                              nat.ptr(allow), ns, doc.ctypes.data, score.ctypes.data,
                                              count.ctypes.data))
        return doc, score, int(count[0])

    def bm25_scores(self, term_ids: Sequence[int]) -> np.ndarray:
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        out = np.empty(self.n_docs, np.float64)
        nat.check(self._lib.anrag_bm25_scores(self.handle, nat.ptr(t) if t.size else None, int(t.size),
                                              out.ctypes.data))
        return out

    # ------------------------------------------------------------------ fusion
    def wrrf(self, lists: Sequence[Sequence[int]], weights: Sequence[float], k: float, top_n: int):
        """-> (id [m] int64, score [m] float64), m = min(top_n, distinct ids)."""
        ids = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int64) for l in lists])
                                   if len(lists) else np.empty(0, np.int64))
        lens = np.ascontiguousarray([len(l) for l in lists], dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        top_n = int(min(top_n, max(ids.size, 1)))
        out_id = np.empty(top_n, np.int64)
        out_score = np.empty(top_n, np.float64)
        count = np.zeros(1, np.int32)
        nat.check(self._lib.anrag_wrrf(self.handle, nat.ptr(ids) if ids.size else None, lens.ctypes.data,
                                       w.ctypes.data, len(lists), float(k), top_n, out_id.ctypes.data,
                                       out_score.ctypes.data, count.ctypes.data))
        m = int(count[0])
        return out_id[:m], out_score[:m]

    def hybrid_search(self, query, term_ids: Sequence[int], similarity_k: int, w_dense: float, w_bm25: float,
                      wrrf_k: float, top_n: int, allow_dense=None, allow_bm25=None):
        q = _f32(query).reshape(-1)
        assert q.size == self.dim
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        ad, nd = _allow_bytes(allow_dense)
        ab, nb = _allow_bytes(allow_bm25)
        out_id = np.empty(top_n, np.int64)
        out_score = np.empty(top_n, np.float64)
        count = np.zeros(1, np.int32)
        nat.check(self._lib.anrag_hybrid_search(self.handle, q.ctypes.data, nat.ptr(t) if t.size else None,
                                                int(t.size), int(similarity_k), float(w_dense), float(w_bm25),
                                                float(wrrf_k), int(top_n), nat.ptr(ad), nd, nat.ptr(ab), nb,
                                                out_id.ctypes.data, out_score.ctypes.data, count.ctypes.data))
        m = int(count[0])
        return out_id[:m], out_score[:m]

    def hybrid_search_batch(self, queries, term_lists: Sequence[Sequence[int]], similarity_k: int, w_dense: float,
                            w_bm25: float, wrrf_k: float, top_n: int, allow_dense=None, allow_bm25=None):
        """Many hybrid queries in one call through the device pipeline (`anrag_hybrid_search_batch`).
        -> (ids [nq, top_n] int64 (-1 padded), fused scores [nq, top_n] float64, counts [nq] int32); row i equals
        `hybrid_search(queries[i], term_lists[i], ...)`."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        assert q.shape[1] == self.dim and len(term_lists) == nq
        offsets = np.zeros(nq + 1, dtype=np.int64)
        np.cumsum([len(t) for t in term_lists], out=offsets[1:])
        terms = (np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in term_lists]).astype(np.int32)
                 if offsets[-1] else np.zeros(0, np.int32))
        terms = np.ascontiguousarray(terms)
        ad, nd = _allow_bytes(allow_dense)
        ab, nb = _allow_bytes(allow_bm25)
        out_id = np.empty((nq, top_n), np.int64)
        out_score = np.empty((nq, top_n), np.float64)
        count = np.zeros(nq, np.int32)
        nat.check(self._lib.anrag_hybrid_search_batch(
            self.handle, q.ctypes.data, nat.ptr(terms) if terms.size else None, offsets.ctypes.data, nq,
            int(similarity_k), float(w_dense), float(w_bm25), float(wrrf_k), int(top_n), nat.ptr(ad), nd, nat.ptr(ab), nb,
            out_id.ctypes.data, out_score.ctypes.data, count.ctypes.data))
        return out_id, out_score, count

    # ------------------------------------------------------------------ measurement
    def profile(self, on, kernels=None, every: int = 1) -> None:
        """Bracket launches with HIP events: on=True times every kernel id, `kernels=[ids]` only those,
        `every=n` only every n-th launch (a bracket costs ~10 us of stream time)."""
        mask = 0
        if on:
            mask = 0xFFFFFFFF if kernels is None else sum(1 << int(k) for k in kernels)
        nat.check(self._lib.anrag_profile_set_sampling(self.handle, max(1, int(every))))
        nat.check(self._lib.anrag_profile_enable(self.handle, mask))

    def profile_reset(self) -> None:
        nat.check(self._lib.anrag_profile_reset(self.handle))

    def profile_read(self, kernel_id: int) -> Tuple[float, int]:
        ms = C.c_double(0)
        n = C.c_int64(0)
        nat.check(self._lib.anrag_profile_read(self.handle, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_units(self, kernel_id: int) -> int:
        """Queries the timed launches of a kernel carried (K1 launches carry up to 8 for query groups)."""
        n = C.c_int64(0)
        nat.check(self._lib.anrag_profile_read_units(self.handle, int(kernel_id), C.byref(n)))
        return n.value

    def info(self) -> dict:
        a, d, b, p, h = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        nat.check(self._lib.anrag_index_info(self.handle, C.byref(a), C.byref(d), C.byref(b), C.byref(p), C.byref(h)))
        return dict(dense_rows=a.value, dense_dim=d.value, bm25_docs=b.value, bm25_postings=p.value,
                    hbm_bytes=h.value)

    # raw device memory (for callers without torch)
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        nat.check(self._lib.anrag_device_alloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, addr: int) -> None:
        nat.check(self._lib.anrag_device_free(self.handle, addr))

    def to_device(self, addr: int, arr: np.ndarray) -> None:
        a = np.ascontiguousarray(arr)
        nat.check(self._lib.anrag_copy_to_device(self.handle, addr, a.ctypes.data, a.nbytes))

    def to_host(self, arr: np.ndarray, addr: int) -> None:
        assert arr.flags["C_CONTIGUOUS"]
        nat.check(self._lib.anrag_copy_to_host(self.handle, arr.ctypes.data, addr, arr.nbytes))
