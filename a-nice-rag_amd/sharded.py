"""Row-sharded hybrid search over up to 8 MI355X: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI) for the one exchange step the path has.

The reference has no distributed code at all; this is the north-star's
"shard the corpus matrix and postings row-wise ... RCCL all-gather of per-shard top-k".

Per query, on every rank:
    1. local legs          K1 dense scan + K3 BM25 on the rank's rows, tail kernel -> 2k candidate records
                           (`anrag_hybrid_candidates_group_device`: scans on the compute stream -- the queries of
                           an exchange group share scan launches, 8 per launch, each still its own pass --, BM25
                           on the index's second stream, tail + copy-out on the communication stream)
    2. exchange            all-gather of the candidate records -- 2k x 16 B per query and rank (k=25: 800 B):
                           latency-bound, nowhere near the 7 x 153 GB/s xGMI links.  It runs on a separate
                           communication stream under the following scans, and `group` in-flight queries
                           share ONE all-gather so that the collective's fixed cost is amortised
    3. replicated merge    G sorted lists -> global top-k per modality  } one launch per group:
    4. fusion              weighted RRF + top-n on the GLOBAL ranks     } `anrag_merge_fuse_device`
Every query is still scanned on its own (batch = 1 kernels); only the exchange is grouped.
BM25 statistics (idf, avgdl, N) are GLOBAL and replicated at index build, otherwise shard-local
scores would differ from the single-index reference.

The shard engine is duck-typed so that the collective plumbing can be exercised on CPU with gloo
(tests/test_sharded_gloo.py plugs the oracle in); the product engine is `HipShardEngine` and it
raises if the HIP library or a GPU is missing.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

RECORD_WORDS = 2  # anrag_candidate = {double score, int64 doc} carried as 2 x int64


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [lo, hi).  Blocks differ by at most one row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipShardEngine:
    """The rank's shard in HBM (`anrag.index.Index`) plus a data-less helper index bound to the
    communication stream for the global-tail kernel."""

    def __init__(self, index, device: torch.device):
        from . import _native as nat
        from .index import Index

        self.index = index
        self.device = device
        self.compute_stream = torch.cuda.current_stream(device)
        self.comm_stream = torch.cuda.Stream(device)
        # scans on torch's current stream, BM25 on the index's own second stream, and everything that
        # produces or consumes candidate records on the communication stream, in order with the collective
        index.set_streams(self.compute_stream.cuda_stream, 0, self.comm_stream.cuda_stream)
        self.aux = Index(index.device)
        self.aux.set_streams(0, 0, self.comm_stream.cuda_stream)
        self._lib = index._lib
        self._nat = nat

    def prepare_allow(self, allow_source) -> Optional[torch.Tensor]:
        """Allow list (one byte per interned source id, as `Index.dense_search` takes it) -> the 2,048-word device
        bitmap the device entry points take (bit s % 32 of word s / 32 = source s passes).  Source ids must mean
        the same on every rank: intern the `source` strings globally when the shards are built."""
        if allow_source is None:
            return None
        bits = np.zeros(65536, dtype=np.uint8)
        a = np.asarray(allow_source, dtype=np.uint8)
        bits[: a.size] = a != 0
        t = torch.from_numpy(np.packbits(bits, bitorder="little").view(np.int32).copy()).to(self.device)
        if t.is_cuda:
            torch.cuda.synchronize(t.device)  # the copy runs on torch's stream, the kernels that read it on the index's
        return t

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return t.data_ptr() if t is not None else None

    def legs(self, d_query: torch.Tensor, d_terms: torch.Tensor, n_terms: int, k: int, out: torch.Tensor,
             allow_dense: Optional[torch.Tensor] = None, allow_bm25: Optional[torch.Tensor] = None) -> None:
        self._nat.check(self._lib.anrag_hybrid_candidates_device(
            self.index.handle, d_query.data_ptr(), d_terms.data_ptr() if n_terms else None, n_terms, k,
            self._ptr(allow_dense), self._ptr(allow_bm25), out.data_ptr()))

    def legs_group(self, queries, terms, n_terms, k: int, outs, allow_dense: Optional[torch.Tensor] = None,
                   allow_bm25: Optional[torch.Tensor] = None) -> None:
        """`legs` for the queries of one exchange group in ONE library call: the shard is scanned once per query,
        but in a single launch per group of up to 8 (`anrag_hybrid_candidates_group_device`).  A query with
        n_terms == 0 skips BM25 (its BM25 half of the payload is padding), as the reference does for a query
        without tokens (search_engine.py:216-217)."""
        import ctypes as C

        n = len(queries)
        ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._nat.check(self._lib.anrag_hybrid_candidates_group_device(
            self.index.handle, ptrs(queries), ptrs(terms), (C.c_int32 * n)(*[int(x) for x in n_terms]), n, k,
            self._ptr(allow_dense), self._ptr(allow_bm25), ptrs(outs)))

    def merge_fuse(self, lists: torch.Tensor, n_lists: int, k: int, stride: int, w_dense: float, w_bm25: float,
                   wrrf_k: float, top_n: int, n_queries: int, out: torch.Tensor, count: torch.Tensor) -> None:
        self._nat.check(self._lib.anrag_merge_fuse_device(
            self.aux.handle, lists.data_ptr(), n_lists, k, stride, w_dense, w_bm25, wrrf_k, top_n, n_queries,
            out.data_ptr(), count.data_ptr()))


class ShardedSearcher:
    """Pipelined hybrid search over a row-sharded corpus.

    `submit()` enqueues one query (no host sync) and returns a ticket; `drain()` waits; `result(ticket)` reads.
    Up to `depth` exchange groups of `group` queries each are in flight: the all-gather and the global tail
    of a group run on the communication stream while later scans run on the compute stream.
    """

    def __init__(self, engine, k: int = 25, top_n: int = 10, w_dense: float = 5.0, w_bm25: float = 1.0,
                 wrrf_k: float = 40.0, depth: int = 4, group: int = 1, pg=None, device: Optional[torch.device] = None):
        self.engine = engine
        self.k, self.top_n = int(k), int(top_n)
        self.w_dense, self.w_bm25, self.wrrf_k = float(w_dense), float(w_bm25), float(wrrf_k)
        self.pg = pg
        self.distributed = dist.is_initialized()
        self.world = dist.get_world_size(pg) if self.distributed else 1
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.depth, self.group = int(depth), int(group)
        mk = lambda *shape: torch.zeros(*shape, RECORD_WORDS, dtype=torch.int64, device=self.device)
        self.send = [mk(self.group, 2 * self.k) for _ in range(self.depth)]
        self.recv = [mk(self.world, self.group, 2 * self.k) for _ in range(self.depth)]
        self.out = [mk(self.group, self.top_n) for _ in range(self.depth)]
        self.count = [torch.zeros(self.group, dtype=torch.int32, device=self.device) for _ in range(self.depth)]
        if self.cuda:
            # torch zero-fills these on ITS stream; the engine writes them on its own streams: the fills must be over
            # before the first query (a late fill would wipe an answer)
            torch.cuda.synchronize(self.device)
        self._slot = 0
        self._filled = 0
        self._grouped = self.group > 1 and hasattr(engine, "legs_group")
        self._pending: List[tuple] = []
        self._allow = (None, None)  # engine-prepared source filters (dense rows, BM25 sections)
        self._filtered = False

    def set_filter(self, allow_dense=None, allow_bm25=None) -> None:
        """Source filter for the queries submitted from now on (the reference passes `filename_type_filter` per
        call, search_engine.py:36-55 / :221-231): allow lists over the interned source ids, one byte per id, None =
        no filter.  Every rank must set the same filter at the same point; queries submitted before are flushed."""
        self.flush()
        self._allow = (self.engine.prepare_allow(allow_dense), self.engine.prepare_allow(allow_bm25))
        self._filtered = allow_dense is not None or allow_bm25 is not None

    def submit(self, d_query: torch.Tensor, d_terms: torch.Tensor, n_terms: int) -> Tuple[int, int]:
        slot, g = self._slot, self._filled
        # the engine writes send[slot][g] in communication-stream order, so the slot's previous all-gather
        # (same stream) has finished with it: no events needed
        if self._grouped:
            # the local legs of a whole exchange group go to the engine together at flush(): one scan launch per
            # group (the caller keeps the query tensors alive until then -- they are rows of its own buffers)
            self._pending.append((d_query, d_terms, int(n_terms), self.send[slot][g]))
        elif not self._filtered:
            self.engine.legs(d_query, d_terms, n_terms, self.k, self.send[slot][g])
        else:
            self.engine.legs(d_query, d_terms, n_terms, self.k, self.send[slot][g], *self._allow)
        self._filled += 1
        if self._filled == self.group:
            self.flush()
        return slot, g

    def flush(self) -> None:
        """Exchange and fuse the queries submitted since the last flush (every rank must flush alike)."""
        n = self._filled
        if n == 0:
            return
        slot, eng, k = self._slot, self.engine, self.k
        pending, self._pending = self._pending, []
        try:
            if pending:
                qs, ts, nts, outs = zip(*pending)
                if not self._filtered:
                    eng.legs_group(qs, ts, nts, k, outs)
                else:
                    eng.legs_group(qs, ts, nts, k, outs, *self._allow)
            ctx = torch.cuda.stream(eng.comm_stream) if self.cuda else _NullCtx()
            with ctx:
                if self.distributed:
                    dist.all_gather_into_tensor(self.recv[slot].view(-1), self.send[slot].view(-1), group=self.pg)
                else:
                    self.recv[slot].view(-1).copy_(self.send[slot].view(-1))
                eng.merge_fuse(self.recv[slot], self.world, k, self.group * 2 * k, self.w_dense, self.w_bm25,
                               self.wrrf_k, self.top_n, n, self.out[slot], self.count[slot])
        finally:  # a failed group is dropped, not retried: the searcher stays usable for the next submit
            self._slot = (self._slot + 1) % self.depth
            self._filled = 0

    def drain(self) -> None:
        """Flush a partial group and wait for everything submitted so far."""
        self.flush()
        if self.cuda:
            self.engine.index.sync()
            self.engine.comm_stream.synchronize()

    def result(self, ticket: Tuple[int, int]) -> Tuple[np.ndarray, np.ndarray]:
        """(doc ids, fused fp64 scores) of a submitted query (call drain() first; a slot is reused after
        `depth` groups)."""
        slot, g = ticket
        n = int(self.count[slot][g].item())
        rec = self.out[slot][g, :n].cpu().numpy()
        return rec[:, 1].copy(), rec[:, 0].copy().view(np.float64)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
