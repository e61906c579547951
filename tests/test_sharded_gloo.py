"""CPU, world_size 2, gloo: the sharded path's host logic (row partition, one all-gather of per-shard
candidate records, replicated merge, fusion on global ranks).  The HIP shard engine cannot run here, so
the ORACLE stands in for it (test infrastructure: the product `HipShardEngine` has no such fallback);
what is under test is `anrag.sharded.ShardedSearcher` + torch.distributed plumbing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_search
from oracle.make_golden import synth_chunks, synth_dense, synth_query
from oracle.ref_bm25 import CsrBM25, csr_get_scores


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleShardEngine:
    """Duck-types anrag.sharded.HipShardEngine on CPU tensors."""

    def __init__(self, e_local, lo, glob: CsrBM25, corpus_local, sid_local=None):
        self.e, self.lo = e_local, lo
        self.sid = sid_local  # interned source id per local row (the same numbering on every rank)
        self.glob = glob
        # shard-local postings over the GLOBAL vocabulary, global idf / avgdl
        n_terms = len(glob.vocab)
        plists = [[] for _ in range(n_terms)]
        tfs = [[] for _ in range(n_terms)]
        for d, doc in enumerate(corpus_local):
            counts = {}
            for w in doc:
                counts[glob.term_id[w]] = counts.get(glob.term_id[w], 0) + 1
            for t, c in counts.items():
                plists[t].append(d)
                tfs[t].append(c)
        df = np.array([len(p) for p in plists], dtype=np.int64)
        self.indptr = np.zeros(n_terms + 1, dtype=np.int64)
        np.cumsum(df, out=self.indptr[1:])
        self.post_doc = np.array([d for p in plists for d in p], dtype=np.int32)
        self.post_tf = np.array([c for p in tfs for c in p], dtype=np.int32)
        self.doc_len = np.array([len(d) for d in corpus_local], dtype=np.int32)

    @staticmethod
    def _write(out, docs, scores, k):
        rec = np.zeros((k, 2), dtype=np.int64)
        rec[:, 0] = np.array([-np.inf] * k).view(np.int64)
        rec[:, 1] = -1
        rec[: len(docs), 0] = np.asarray(scores, dtype=np.float64).view(np.int64)
        rec[: len(docs), 1] = docs
        out.copy_(torch.from_numpy(rec))

    def prepare_allow(self, allow_source):
        return None if allow_source is None else np.asarray(allow_source, dtype=bool)

    def legs(self, d_query, d_terms, n_terms, k, out, allow_dense=None, allow_bm25=None):
        q = d_query.numpy()
        sims = ref_search.dense_scores(q, self.e)
        top = ref_search.canonical_topk(sims, k, None if allow_dense is None else allow_dense[self.sid])
        self._write(out[:k], top + self.lo, sims[top].astype(np.float64), k)
        if n_terms == 0:  # no tokens: the BM25 leg is skipped (search_engine.py:216-217), its half is padding
            self._write(out[k:], [], [], k)
            return
        terms = d_terms.numpy()[:n_terms].tolist()
        sc = csr_get_scores(self.indptr, self.post_doc, self.post_tf, self.glob.idf, self.doc_len, self.glob.avgdl,
                            self.glob.k1, self.glob.b, terms)
        top = ref_search.canonical_topk(sc, k, None if allow_bm25 is None else allow_bm25[self.sid])
        self._write(out[k:], top + self.lo, sc[top], k)

    def merge_fuse(self, lists, n_lists, k, stride, w_dense, w_bm25, wrrf_k, top_n, n_queries, out, count):
        rec = lists.numpy().reshape(-1, 2)
        for q in range(n_queries):
            merged = []
            for leg in range(2):
                cands = []
                for l in range(n_lists):
                    base = l * stride + q * 2 * k + leg * k
                    for r in rec[base: base + k]:
                        if r[1] >= 0:
                            cands.append((float(np.int64(r[0]).view(np.float64)), int(r[1])))
                cands.sort(key=lambda c: (-c[0], c[1]))
                merged.append([c[1] for c in cands[:k]])
            fused = ref_search.weighted_reciprocal_rank_fusion(
                [(merged[0], "d"), (merged[1], "b")], {"d": w_dense, "b": w_bm25}, int(wrrf_k))[:top_n]
            self._write(out[q], [i for i, _ in fused], [s for _, s in fused], top_n)
            count[q] = len(fused)


class GroupedOracleShardEngine(OracleShardEngine):
    """The same with the product engine's group call, so that the searcher takes its deferred-legs route."""

    def legs_group(self, queries, terms, n_terms, k, outs, allow_dense=None, allow_bm25=None):
        for q, t, nt, out in zip(queries, terms, n_terms, outs):
            self.legs(q, t, nt, k, out, allow_dense, allow_bm25)


def _worker(rank, world, port, ret, grouped=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anrag.sharded import ShardedSearcher, shard_bounds

    n, d, k, top_n = 301, 32, 25, 10
    chunks = [c for c in synth_chunks(n + 20, 5) if c["tokens"]][:n]
    corpus = [c["tokens"] for c in chunks]
    e = synth_dense(n, d, 6)
    glob = CsrBM25(corpus, k1=1.7, b=0.83, epsilon=0.05)
    lo, hi = shard_bounds(n, world, rank)
    sid = (np.arange(n) * 7 % 5).astype(np.int64)  # 5 "sources", the same numbering on every rank
    eng = (GroupedOracleShardEngine if grouped else OracleShardEngine)(e[lo:hi], lo, glob, corpus[lo:hi], sid[lo:hi])
    rng = np.random.default_rng(11)
    ok = True
    # one query per all-gather, grouped exchanges with a partial last group, and a source filter on both legs
    for group, allow in ((1, None), (3, None), (3, np.array([1, 0, 1, 1, 0], dtype=np.uint8))):
        searcher = ShardedSearcher(eng, k=k, top_n=top_n, w_dense=5.0, w_bm25=1.0, wrrf_k=40, depth=2, group=group)
        mask = None
        if allow is not None:
            searcher.set_filter(allow, allow)
            mask = allow.astype(bool)[sid]
        tickets, wants = [], []
        for trial in range(5):
            target = int(rng.integers(n))
            q = synth_query(e, 100 + trial, target)
            # trial 2 asks without tokens (a query of stopwords only): dense-only, as the reference answers it
            toks = [str(t) for t in rng.choice(corpus[target], size=4)] if trial != 2 else []
            terms = np.array(glob.term_ids(toks), dtype=np.int32)
            tickets.append(searcher.submit(torch.from_numpy(q), torch.from_numpy(np.resize(terms, max(len(terms), 1))),
                                           len(terms)))
            sims = ref_search.dense_scores(q, e)  # single-index oracle
            dl = ref_search.canonical_topk(sims, k, mask).tolist()
            bl = ref_search.canonical_topk(glob.get_scores(toks), k, mask).tolist() if toks else []
            wants.append(ref_search.weighted_reciprocal_rank_fusion([(dl, "d"), (bl, "b")], {"d": 5.0, "b": 1.0},
                                                                    40)[:top_n])
            if group == 1 or trial == 4 or len(tickets) % (2 * group) == 0:
                searcher.drain()  # within depth=2 groups: read before the slots are reused
                for t, want in zip(tickets, wants):
                    ids, scores = searcher.result(t)
                    ok = ok and ids.tolist() == [i for i, _ in want] and scores.tolist() == [s for _, s in want]
                tickets, wants = [], []
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(int(t.item()))
    dist.destroy_process_group()


def test_shard_bounds_cover_rows():
    from anrag.sharded import shard_bounds

    for n in (1, 7, 8, 1_000_000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,grouped", [(2, False), (3, True)])
def test_sharded_search_gloo(world, grouped):
    """world 2: one engine call per query; world 3 (uneven shards: 101/100/100 rows): the exchange group's legs in
    one engine call, as the HIP engine takes them."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, grouped)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert ret.get(timeout=5) == 1
