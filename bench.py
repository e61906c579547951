#!/usr/bin/env python3
"""Headline benchmark: hybrid retrieval queries/sec on a 1M x 768 corpus (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = ONE query at batch=1 through the whole hot path: dense dot-product scan + top-25, BM25
term-at-a-time + top-25, weighted RRF (5:1, k=40), top-10 -- the body of the reference's
`retrieve_documents` for one dense model + BM25 (src/query_rag_retrieval.py:197-220, :304-378).
Inputs (corpus matrix, postings, queries, term ids) are resident in HBM before the timed region;
queries are enqueued back to back with no host synchronisation in between.

N > 1: the SAME 1M-row corpus row-sharded N ways (strong scaling, the metric's "1M x 768 corpus,
1/2/4/8 MI355X"): per-rank legs, one RCCL all-gather of 2 x 25 candidate records per rank on a
communication stream, replicated merge + fusion (a-nice-rag_amd/sharded.py).

Prints ONE JSON line on rank 0 (fields: README of the build contract + `roofline` + `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1_000_000, help="corpus rows (whole job)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--similarity-k", type=int, default=25)
    ap.add_argument("--top-n", type=int, default=10)
    ap.add_argument("--queries", type=int, default=64, help="distinct synthetic queries cycled through")
    ap.add_argument("--workload", choices=["hybrid", "dense", "batched"], default="hybrid",
                    help="hybrid = headline (C3); dense = K1 only (C2-shaped); batched = 256-query MFMA passes (C4)")
    ap.add_argument("--batch", type=int, default=256, help="queries per pass of --workload batched")
    ap.add_argument("--batch-precision", choices=["f32", "bf16x3"], default="f32",
                    help="--workload batched: exact f32 MFMA (default) or opt-in split-precision bf16 x 3 products")
    ap.add_argument("--exchange-group", type=int, default=8,
                    help="N > 1: in-flight queries that share one all-gather (each is still scanned alone)")
    ap.add_argument("--filter", action="store_true",
                    help="diagnostic: source-prefix filter as retrieval_eval.py:280 passes it ('CG,NG'); rows "
                         "carry one of 300 source ids, 15 %% of them outside the filter (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=12, help="queries of the bounded CPU sample")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libanrag has no CPU path")
    if local_rank >= torch.cuda.device_count():
        local_rank = 0  # the launcher narrowed this rank's visibility to one device
    # ANRAG_FORCE_SHARDED=1: take the sharded (all-gather) route even at world size 1 -- a rehearsal of the
    # N > 1 code path on a one-GPU box
    sharded = world > 1 or os.environ.get("ANRAG_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from anrag import _native as nat
    from anrag import synth
    from anrag.index import Index
    from anrag.sharded import HipShardEngine, ShardedSearcher, shard_bounds

    K, TOPN = args.similarity_k, args.top_n
    W_DENSE, W_BM25, WRRF_K = 5.0, 1.0, 40.0  # src/config.py:30-36, retrieval_eval.py:279
    hybrid = args.workload == "hybrid"
    batched = args.workload == "batched"
    if batched:
        args.queries = args.batch
    lo, hi = shard_bounds(args.rows, world, rank)
    n_local = hi - lo

    # ------------------------------------------------------------------ synthetic shard, resident in HBM
    t_build = time.time()
    E = synth.dense_corpus(n_local, args.dim, 1234 + rank, device)
    idx = Index(local_rank)
    torch.cuda.synchronize()  # device-pointer operands must be complete: the library copies on its own stream
    src = allowed_rows = d_allow = None
    if args.filter:
        if batched:
            raise SystemExit("--filter is a batch=1 diagnostic")
        src = ((np.arange(n_local, dtype=np.int64) + lo) % 300).astype(np.uint16)  # 300 guideline codes, cyclic
        allow = np.zeros(65536, np.uint8)
        allow[45:300] = 1  # ids 0..44 = the ~15 % QS/TA/PH sources the 'CG,NG' filter drops
        allowed_rows = allow[src].astype(bool)
        # 2048 words, bit (s % 32) of word s / 32 = source s allowed
        d_allow = torch.from_numpy(np.packbits(allow, bitorder="little").view(np.int32).copy()).to(device)
    idx.dense_load((E.data_ptr(), n_local, args.dim), source_id=src, doc_id_base=lo)
    if batched:
        idx.set_batched_precision(args.batch_precision)
    # queries: planted next to rows of rank 0's shard, identical on every rank
    Q, planted = synth.dense_queries(E, args.queries, 4321)
    if world > 1:
        dist.broadcast(Q, 0)
    post = None
    term_lists = [np.zeros(0, np.int32)] * args.queries
    if hybrid:
        post = synth.bm25_postings(n_local, args.vocab, 777 + rank, device)
        df = post["df"].clone()
        tot = torch.tensor([post["total_len"]], device=device, dtype=torch.int64)
        if world > 1:  # GLOBAL statistics, replicated (sharded.py docstring)
            dist.all_reduce(df)
            dist.all_reduce(tot)
        avgdl = int(tot.item()) / args.rows
        idf = synth.bm25_idf(df.cpu().numpy(), args.rows)
        torch.cuda.synchronize()
        idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                      (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], avgdl,
                      synth.BM25_K1, synth.BM25_B, source_id=src, doc_id_base=lo)
        term_lists = synth.bm25_queries(post, args.queries, 99) if rank == 0 else []
    # query term ids as one padded device tensor; rank 0's are broadcast so that every rank asks the same query
    MAX_TERMS = 16
    T = torch.full((args.queries, MAX_TERMS), -1, dtype=torch.int32, device=device)
    NT = torch.zeros(args.queries, dtype=torch.int32, device=device)
    if rank == 0:
        for i, t in enumerate(term_lists):
            t = np.asarray(t, dtype=np.int32)[:MAX_TERMS]
            if len(t):
                T[i, : len(t)] = torch.from_numpy(t).to(device)
            NT[i] = len(t)
    if world > 1:
        dist.broadcast(T, 0)
        dist.broadcast(NT, 0)
    n_terms = [int(x) for x in NT.cpu().tolist()]
    term_lists = [T[i, : n_terms[i]].cpu().numpy() for i in range(args.queries)]
    torch.cuda.synchronize()
    build_s = time.time() - t_build

    # ------------------------------------------------------------------ the step
    lib = nat.load_library()
    if not sharded:
        out = torch.zeros((args.queries, TOPN, 2), dtype=torch.int64, device=device)
        cnt = torch.zeros(args.queries, dtype=torch.int32, device=device)
        allow_ptr = d_allow.data_ptr() if d_allow is not None else None

        def step(i):
            qi = i % args.queries
            if hybrid:
                nat.check(lib.anrag_hybrid_search_device(
                    idx.handle, Q[qi].data_ptr(), T[qi].data_ptr(), n_terms[qi], K, W_DENSE, W_BM25, WRRF_K, TOPN,
                    allow_ptr, allow_ptr, out[qi].data_ptr(), cnt[qi:].data_ptr()))
            elif batched:  # one step = one pass of args.batch queries
                nat.check(lib.anrag_dense_search_batch_device(idx.handle, Q.data_ptr(), args.batch, TOPN, None,
                                                              out.data_ptr(), cnt.data_ptr()))
            else:
                nat.check(lib.anrag_dense_search_device(idx.handle, Q[qi].data_ptr(), 1, TOPN, allow_ptr,
                                                        out[qi].data_ptr()))

        def finish():
            idx.sync()
    else:
        if not hybrid:
            raise SystemExit("--workload dense is a single-GPU diagnostic")
        engine = HipShardEngine(idx, device)
        searcher = ShardedSearcher(engine, k=K, top_n=TOPN, w_dense=W_DENSE, w_bm25=W_BM25, wrrf_k=WRRF_K,
                                   depth=4, group=args.exchange_group, device=device)
        if args.filter:
            searcher.set_filter(allow[:300], allow[:300])  # source ids are global: (row % 300) on every rank

        def step(i):
            qi = i % args.queries
            searcher.submit(Q[qi], T[qi], n_terms[qi])

        def finish():
            searcher.drain()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if sharded and args.warmup == 0:
        step(0)  # the first all-gather sets the collective up: never inside the timed region
    finish()
    # events around the dominant kernel only: bracketing every launch would perturb the pipeline
    main_kernel = nat.KERNEL_DENSE_BATCHED if batched else nat.KERNEL_DENSE_SCAN
    # HIP events around the dominant kernel, on every 8th launch: a bracket holds its stream for ~10 us, which
    # at batch=1 would be charged to every query (measured: 463 -> 452 us/step at 1M rows)
    # (a sharded K1 launch carries a whole exchange group: every 2nd launch is sample enough and cheap enough)
    idx.profile(True, kernels=[main_kernel], every=1 if batched else (2 if sharded else 8))
    idx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish()
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_n = idx.profile_read(main_kernel)
    scan_units = idx.profile_units(main_kernel)  # queries the timed launches carried (sharded: one exchange group per K1 launch)
    idx.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # single-query latency (SURVEY.md 8d asks for p50/p99 next to the throughput): one query in flight,
    # enqueue -> all streams idle, host clock.  Outside the timed region.
    latency = None
    if not sharded and not batched:
        lat = []
        for i in range(200):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(i)
            finish()
            lat.append((time.perf_counter() - t1) * 1e6)
        lat = np.sort(np.asarray(lat[20:]))
        latency = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[min(len(lat) - 1, int(len(lat) * 0.99))]),
                   "n": int(len(lat)), "what": "one query in flight: enqueue + wait for its result, host clock"}

    # ------------------------------------------------------------------ report (rank 0)
    if rank == 0:
        scan_avg_ms = scan_ms / max(scan_n, 1)
        # N*D*4 per query (this rank's rows), + N*2 with a filter
        per_query_bytes = n_local * args.dim * 4 + (n_local * 2 if args.filter else 0)
        queries_per_launch = (scan_units / scan_n) if (scan_n and not batched) else 1.0
        alg_bytes = per_query_bytes * queries_per_launch  # a K1 launch of the sharded path scans a GROUP of queries,
        achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_n else 0.0  # each its own pass over the shard
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_dense_scan.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    rec = json.load(f)
                if rec.get("rows") == n_local and rec.get("dim") == args.dim:
                    traffic = rec.get("hbm_bytes_per_launch")  # measured on one-query launches
                    if traffic is not None:
                        traffic = traffic * queries_per_launch
            except Exception:
                traffic = None
        per_step = args.batch if batched else 1
        line = {
            "metric": "queries/sec, hybrid (dense + BM25) RRF top-10 at batch=1, 1M x 768 corpus" if hybrid else
                      ("queries/sec, dense top-%d, batch=%d (MFMA)" % (TOPN, args.batch) if batched else
                       "queries/sec, dense top-%d at batch=1" % TOPN),
            "value": args.steps * per_step / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("C3: %d x %d hybrid (dense + BM25 CSR postings) RRF top-%d, batch=1" % (args.rows, args.dim, TOPN))
                if hybrid else (("C4: %d x %d dense, batch=%d queries per pass (fp32 MFMA), top-%d"
                                 % (args.rows, args.dim, args.batch, TOPN)) if batched else
                                ("dense-only brute-force top-%d, %d x %d, batch=1" % (TOPN, args.rows, args.dim))),
                "rows": args.rows, "dim": args.dim, "rows_per_gpu": n_local,
                "postings_per_gpu": (int(post["post_doc"].numel()) if post else 0), "vocab": args.vocab if hybrid else 0,
                "similarity_k": K, "top_n": TOPN, "wrrf_k": WRRF_K, "weights": [W_DENSE, W_BM25],
                "sharding": ("rows/%d + RCCL all-gather of per-shard top-k, %d queries per all-gather"
                             % (world, args.exchange_group)) if sharded else "none",
                "bm25_arith": "f64", "source_filter": "CG,NG-shaped: 255 of 300 source ids allowed" if args.filter else None,
            },
            "roofline": {
                "kernel": "dense_scan_kernel (K1)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n,
                "queries_per_launch": queries_per_launch,
            },
            "index_build_s": build_s,
        }
        if latency:
            line["latency_us"] = latency
        if batched:
            flop = 2.0 * args.batch * n_local * args.dim  # 2*Q*N*D per pass
            tf = flop / (scan_avg_ms * 1e-3) / 1e12 if scan_n else 0.0
            split = args.batch_precision == "bf16x3"
            peak = 2500.0 / 3.0 if split else 157.3  # three bf16 MFMAs per product against the ~2.5 PF dense bf16 peak
            line["roofline"] = {
                "kernel": ("dense_batched_split_kernel" if split else "dense_batched_kernel")
                          + " (K2: sample pass + threshold + filter pass)", "bound": "mfma",
                "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                "algorithmic_flop_per_launch": flop, "avg_launch_ms": scan_avg_ms, "launches": scan_n,
                "note": ("bf16 x 3 split products (hi.hi + hi.lo + lo.hi), f32 accumulate; peak = 2.5 PF / 3; scores "
                         "within ~1e-6 of f32 (bound 3e-5), not bit-equal" if split else
                         "f32-in/f32-acc v_mfma_f32_32x32x2_f32") + "; the timed span includes the sampled-threshold "
                        "pre-pass, the flop count does not"}
            line["dtype"] = "bf16x3" if split else "f32"
        if not sharded and not args.no_cpu_baseline and not batched:
            line["cpu_baseline"] = cpu_baseline(args, E, Q, post, idf if hybrid else None,
                                                avgdl if hybrid else None, term_lists, out, cnt, hybrid, K, TOPN,
                                                (W_DENSE, W_BM25, WRRF_K), allowed_rows)
            line["recall_at_10"] = line["cpu_baseline"].pop("niceqa_recall_at_10")
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, E, Q, post, idf, avgdl, term_lists, gpu_out, gpu_cnt, hybrid, K, TOPN, fusion, allowed=None):
    """The reference-shaped CPU path (oracle = port of src/search_engine.py) timed on this box's host
    cores over a bounded sample of the same queries, and used as the parity check of the GPU results."""
    import torch
    from oracle import ref_bm25, ref_search

    w_dense, w_bm25, wrrf_k = fusion
    nq = max(1, min(args.cpu_queries, args.queries))
    e_host = E.cpu().numpy()
    q_host = Q[:nq].cpu().numpy()
    if hybrid:
        post_doc = post["post_doc"].cpu().numpy()
        post_tf = post["post_tf"].cpu().numpy()
    rows = list(e_host)  # the reference keeps one ndarray per DataFrame row (database_manager.py:49)
    t_ref = t_pre = 0.0
    parity_ok = True
    max_dscore = 0.0
    for qi in range(nq):
        t0 = time.perf_counter()
        emb = np.stack(rows)  # search_engine.py:80 -- re-materialised on every query
        sims = np.dot(q_host[qi].reshape(1, -1), emb.T).flatten()
        if allowed is not None:  # the reference filters the DataFrame first (pandas str ops, not timed here)
            sims = np.where(allowed, sims, -np.inf)
        top = ref_search.numpy_topk_idiom(sims, K)
        dense_list = top.tolist()
        t_stack_path = time.perf_counter() - t0
        t1 = time.perf_counter()
        sims2 = np.dot(q_host[qi].reshape(1, -1), e_host.T).flatten()
        ref_search.numpy_topk_idiom(sims2, K)
        t_dense_pre = time.perf_counter() - t1
        t2 = time.perf_counter()
        if hybrid:
            scores = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl,
                                             1.7, 0.83, term_lists[qi].tolist())
            bm_list = ref_search.canonical_topk(scores, K, allowed).tolist()
            fused = ref_search.weighted_reciprocal_rank_fusion(
                [(dense_list, "dense"), (bm_list, "BM25")], {"dense": w_dense, "BM25": w_bm25}, int(wrrf_k))[:TOPN]
            want_ids = [i for i, _ in fused]
        else:
            want_ids = dense_list[:TOPN]
        t_rest = time.perf_counter() - t2
        t_ref += t_stack_path + t_rest
        t_pre += t_dense_pre + t_rest
        # parity of the GPU's answer for this query (full-size check, same inputs)
        n = int(gpu_cnt[qi].item()) if hybrid else TOPN
        rec = gpu_out[qi, :n].cpu().numpy()
        got_ids = rec[:, 1].tolist()
        if got_ids != want_ids:
            # dense near-ties may legitimately reorder within 1e-4; compare as sets then
            parity_ok = parity_ok and (set(got_ids) == set(want_ids))
        if hybrid:
            got_s = rec[:, 0].copy().view(np.float64)
            max_dscore = max(max_dscore, float(np.max(np.abs(got_s - np.array([s for _, s in fused])))) if n else 0.0)
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    recall = niceqa_recall()
    return {
        "niceqa_recall_at_10": recall,
        "value": nq / t_ref, "unit": "queries/s", "cores": blas_threads, "kind": "port",
        "sample": "%d of the %d benchmark queries, full %d x %d corpus on the host: reference-shaped path "
                  "(np.stack of per-row arrays per query as search_engine.py:80 does, np.dot, argpartition; CSR "
                  "restatement of BM25Okapi.get_scores; Python WRRF)" % (nq, args.queries, args.rows, args.dim),
        "value_prestacked": nq / t_pre,
        "note_prestacked": "same sample with the corpus matrix stacked once up front (removes the reference's "
                           "per-query np.stack)",
        "gpu_results_match_cpu": bool(parity_ok), "max_abs_fused_score_diff": max_dscore,
    }


def niceqa_recall():
    """Recall@10 on data/NICEQA.csv over the stand-in corpus (anrag/niceqa.py): GPU path vs CPU reference path.
    The acceptance criterion is equality; the absolute value says little (hashed-BoW stand-in embeddings)."""
    from anrag import niceqa
    from oracle.niceqa_ref import cpu_ranked_ids

    gold = os.path.join(REPO, "tests", "golden")
    try:
        data = niceqa.load_standin(os.path.join(gold, "suggested_queries_bm25_preprocessed.json.gz"),
                                   os.path.join(gold, "NICEQA.csv"))
    except OSError as e:
        return {"error": str(e)}
    gpu = niceqa.gpu_ranked_ids(data)
    cpu = cpu_ranked_ids(data, *niceqa.encode_questions(data))
    rg, rc = niceqa.recall_at_10(data, gpu), niceqa.recall_at_10(data, cpu)
    return {"gpu": rg["recall_at_10"], "cpu_reference": rc["recall_at_10"], "equal": rg == rc,
            "questions": rg["questions"], "with_gold_chunk": rg["with_gold_chunk"],
            "identical_top10_lists": sum(a == b for a, b in zip(gpu, cpu)),
            "corpus": "stand-in: 9,609 shipped chunk ids x 384-d hashed-BoW embeddings (no encoder weights offline)"}


if __name__ == "__main__":
    main()
