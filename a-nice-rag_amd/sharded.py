"""Row-sharded hybrid search over up to 8 MI355X: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI) for the one exchange step the path has.

The reference has no distributed code at all; this is the north-star's
"shard the corpus matrix and postings row-wise ... RCCL all-gather of per-shard top-k".

Per query, on every rank:
    1. local legs          dense scan + BM25 on the rank's rows -> 2k candidate records
                           (`anrag_hybrid_candidates_device`: scan on the compute stream, BM25 on the
                           index's second stream, list merges + copy-out on the communication stream)
    2. exchange            ONE all-gather of 2k x 16 B per rank (k=25: 800 B) -- latency-bound,
                           nowhere near the 7 x 153 GB/s xGMI links, so it runs on a separate
                           communication stream and overlaps the next query's scan
    3. replicated merge    G sorted lists -> global top-k per modality  } one launch:
    4. fusion              weighted RRF + top-n on the GLOBAL ranks     } `anrag_merge_fuse_device`
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
    communication stream for the merge / fusion kernels."""

    def __init__(self, index, device: torch.device):
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
        from . import _native as nat

        self._nat = nat

    def legs(self, d_query: torch.Tensor, d_terms: torch.Tensor, n_terms: int, k: int, out: torch.Tensor) -> None:
        self._nat.check(self._lib.anrag_hybrid_candidates_device(
            self.index.handle, d_query.data_ptr(), d_terms.data_ptr() if n_terms else None, n_terms, k, None, None,
            out.data_ptr()))

    def merge(self, lists: torch.Tensor, n_lists: int, k: int, stride: int, offset: int, out: torch.Tensor) -> None:
        self._nat.check(self._lib.anrag_merge_candidates_device(
            self.aux.handle, lists.data_ptr() + offset * 16, n_lists, k, stride, out.data_ptr()))

    def merge_fuse(self, lists: torch.Tensor, n_lists: int, k: int, stride: int, w_dense: float, w_bm25: float,
                   wrrf_k: float, top_n: int, out: torch.Tensor, count: torch.Tensor) -> None:
        """Global tail in one launch: per-modality merge of the gathered lists + WRRF + top-n."""
        self._nat.check(self._lib.anrag_merge_fuse_device(
            self.aux.handle, lists.data_ptr(), n_lists, k, stride, w_dense, w_bm25, wrrf_k, top_n, out.data_ptr(),
            count.data_ptr()))

    def fuse(self, dense: torch.Tensor, bm25: torch.Tensor, k: int, w_dense: float, w_bm25: float, wrrf_k: float,
             top_n: int, out: torch.Tensor, count: torch.Tensor) -> None:
        self._nat.check(self._lib.anrag_wrrf_device(
            self.aux.handle, dense.data_ptr(), k, bm25.data_ptr(), k, w_dense, w_bm25, wrrf_k, top_n,
            out.data_ptr(), count.data_ptr()))


class ShardedSearcher:
    """Pipelined hybrid search over a row-sharded corpus.

    `submit()` enqueues a query (no host sync); `collect()` returns finished results in order.
    `depth` result slots are in flight at once: the all-gather / merge / fusion of query i run on
    the communication stream while the scan of query i+1 runs on the compute stream.
    """

    def __init__(self, engine, k: int = 25, top_n: int = 10, w_dense: float = 5.0, w_bm25: float = 1.0,
                 wrrf_k: float = 40.0, depth: int = 4, group=None, device: Optional[torch.device] = None):
        self.engine = engine
        self.k, self.top_n = int(k), int(top_n)
        self.w_dense, self.w_bm25, self.wrrf_k = float(w_dense), float(w_bm25), float(wrrf_k)
        self.group = group
        self.distributed = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.depth = depth
        mk = lambda *shape: torch.zeros(*shape, RECORD_WORDS, dtype=torch.int64, device=self.device)
        self.send = [mk(2 * self.k) for _ in range(depth)]
        self.recv = [mk(self.world, 2 * self.k) for _ in range(depth)]
        self.merged = [mk(2 * self.k) for _ in range(depth)]
        self.out = [mk(self.top_n) for _ in range(depth)]
        self.count = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self._next = 0
        self._pending: List[int] = []

    def submit(self, d_query: torch.Tensor, d_terms: torch.Tensor, n_terms: int) -> int:
        slot = self._next % self.depth
        self._next += 1
        eng, k = self.engine, self.k
        # the engine writes send[slot] in communication-stream order, so the slot's previous all-gather
        # (same stream) has finished with it and the one below sees the new records: no events needed
        eng.legs(d_query, d_terms, n_terms, k, self.send[slot])
        ctx = torch.cuda.stream(eng.comm_stream) if self.cuda else _NullCtx()
        with ctx:
            if self.distributed:
                dist.all_gather_into_tensor(self.recv[slot].view(-1), self.send[slot].view(-1), group=self.group)
            else:
                self.recv[slot].view(-1).copy_(self.send[slot].view(-1))
            if hasattr(eng, "merge_fuse"):
                eng.merge_fuse(self.recv[slot], self.world, k, 2 * k, self.w_dense, self.w_bm25, self.wrrf_k,
                               self.top_n, self.out[slot], self.count[slot])
            else:  # engines that only expose the three primitives
                eng.merge(self.recv[slot], self.world, k, 2 * k, 0, self.merged[slot][:k])
                eng.merge(self.recv[slot], self.world, k, 2 * k, k, self.merged[slot][k:])
                eng.fuse(self.merged[slot][:k], self.merged[slot][k:], k, self.w_dense, self.w_bm25, self.wrrf_k,
                         self.top_n, self.out[slot], self.count[slot])
        self._pending.append(slot)
        return slot

    def drain(self) -> None:
        """Wait for everything submitted so far."""
        if self.cuda:
            self.engine.index.sync()
            self.engine.comm_stream.synchronize()

    def result(self, slot: int) -> Tuple[np.ndarray, np.ndarray]:
        """(doc ids, fused fp64 scores) of the query last submitted into `slot` (call drain() first)."""
        n = int(self.count[slot].item())
        rec = self.out[slot][:n].cpu().numpy()
        return rec[:, 1].copy(), rec[:, 0].copy().view(np.float64)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
