"""GPU: the sharded pipeline with the real HIP shard engine and a real RCCL all-gather (world_size 1 -- one GPU
per box here; world_size 2 of the same host logic runs on gloo in test_sharded_gloo.py).  Queries are submitted
back to back so that the three-stream pipeline and the slot rotation are exercised, then checked against the
single-call fused path and the oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_pipeline_world1_rccl():
    import torch
    import torch.distributed as dist
    from oracle import ref_search
    from oracle.make_golden import synth_chunks, synth_dense, synth_query
    from oracle.ref_bm25 import BM25Okapi
    from anrag.bm25_index import Bm25Index
    from anrag.index import Index
    from anrag.sharded import HipShardEngine, ShardedSearcher

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        n, d, k, top_n = 20000, 256, 25, 10
        chunks = [c for c in synth_chunks(n + 700, 9) if c["tokens"]][:n]
        corpus = [c["tokens"] for c in chunks]
        e = synth_dense(n, d, 10)
        bi = Bm25Index(corpus, k1=1.7, b=0.83, epsilon=0.05)
        ref = BM25Okapi(corpus, k1=1.7, b=0.83, epsilon=0.05)
        idx = Index(0)
        sid = (np.arange(n) * 7 % 5).astype(np.uint16)  # 5 interned sources
        idx.dense_load(e, source_id=sid, doc_id_base=1000)
        idx.bm25_load(bi.indptr, bi.post_doc, bi.post_tf, bi.idf, bi.doc_len, bi.avgdl, bi.k1, bi.b, source_id=sid,
                      doc_id_base=1000)
        engine = HipShardEngine(idx, device)
        rng = np.random.default_rng(3)
        queries, toks_all = [], []
        for i in range(4):
            target = int(rng.integers(n))
            queries.append(synth_query(e, 50 + i, target))
            # query 2 has no tokens (stopwords only): the reference skips BM25 and answers dense-only
            # (search_engine.py:216-217); the rank's payload keeps its 2k shape, BM25 half padding
            toks_all.append([str(t) for t in rng.choice(corpus[target], size=5)] if i != 2 else [])
        Q = torch.from_numpy(np.stack(queries)).to(device)
        T = torch.full((4, 8), -1, dtype=torch.int32, device=device)
        nt = []
        for i, toks in enumerate(toks_all):
            t = bi.term_ids(toks)
            T[i, : len(t)] = torch.from_numpy(t).to(device)
            nt.append(len(t))
        torch.cuda.synchronize()  # the uploads ran on torch's stream; the engine reads Q / T on its own streams
        # group=3: the local legs of an exchange group go to the library in one call (grouped scan launches, a
        # partial group at drain); group=1: one call per query
        allow = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
        # rounds 4 and 5: the same with a source filter on both legs (grouped call, then one call per query)
        for rounds, group in ((0, 3), (1, 3), (2, 1), (3, 4), (4, 3), (5, 1)):  # drained once per round
            if rounds in (0, 2, 3, 4, 5):
                searcher = ShardedSearcher(engine, k=k, top_n=top_n, w_dense=5.0, w_bm25=1.0, wrrf_k=40, depth=4,
                                           group=group, device=device)
            mask = None
            if rounds >= 4:
                searcher.set_filter(allow, allow)
                mask = allow.astype(bool)[sid]
            slots = [searcher.submit(Q[i], T[i], nt[i]) for i in range(4)]
            searcher.drain()
            for i, slot in enumerate(slots):
                ids, scores = searcher.result(slot)
                sims = ref_search.dense_scores(queries[i], e)
                dl = (ref_search.canonical_topk(sims, k, mask) + 1000).tolist()
                bl = (ref_search.canonical_topk(ref.get_scores(toks_all[i]), k, mask) + 1000).tolist() if toks_all[i] else []
                want = ref_search.weighted_reciprocal_rank_fusion([(dl, "d"), (bl, "b")], {"d": 5.0, "b": 1.0}, 40)[:top_n]
                assert ids.tolist() == [j for j, _ in want], (rounds, i)
                assert scores.tolist() == [s for _, s in want]
        # the single-call fused path gives the same answer (the last round ran filtered)
        fid, fs = idx.hybrid_search(queries[0], bi.term_ids(toks_all[0]), k, 5.0, 1.0, 40, top_n, allow, allow)
        ids, scores = searcher.result(slots[0])
        assert fid.tolist() == ids.tolist() and fs.tolist() == scores.tolist()
    finally:
        dist.destroy_process_group()
